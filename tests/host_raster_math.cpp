// TEST INFRASTRUCTURE.  Host build (g++ -ffp-contract=off) of the product's per-(pixel, face) arithmetic and per-pixel list
// (differentiable-blocksworld_amd/csrc/raster_math.h, the same header the HIP kernels compile) driven by a plain per-pixel loop, so
// that `pytest -m "not gpu"` can hold FaceRec / div_fast / tile_culled / TopK bit-exact to oracle/raster_ref.c without a GPU.
// Not a fallback: nothing in dbw_amd/ loads this.
#include <stdint.h>
#include <stdlib.h>
#include <vector>

#include "../differentiable-blocksworld_amd/csrc/raster_math.h"

using namespace dbw;

namespace {

struct Stats { long long evals, unsafe, culled, staged; };

template <int KMAX>
void run(const float *fv, const int64_t *first, const int64_t *num, const int64_t *nbr, int N, int H, int W, int K, float blur, int persp,
         int clipb, int cull, int fastdiv, int tile, int bounded, int64_t *p2f, float *zbuf, float *bary, float *dists, Stats &st) {
    const float margin = (float)sqrt((double)blur);
    for (int n = 0; n < N; ++n) {
        const int64_t f0 = first[n], nf = num[n];
        std::vector<FaceRec> recs((size_t)nf);
        for (int64_t j = 0; j < nf; ++j) {
            float box[4];
            make_face_rec(fv + (f0 + j) * 9, margin, cull, nbr ? (int)nbr[f0 + j] : -1, recs[(size_t)j], box);
        }
        for (int yi = 0; yi < H; ++yi)
            for (int xi = 0; xi < W; ++xi) {
                f2 p;
                p.x = pix_to_ndc(W - 1 - xi, W, H);
                p.y = pix_to_ndc(H - 1 - yi, H, W);
                // extents of the pixel centres of the tile this pixel lies in (raster_common.h)
                float txmin = 0, txmax = 0, tymin = 0, tymax = 0;
                if (tile > 0) {
                    const int x0 = (xi / tile) * tile, y0 = (yi / tile) * tile;
                    const int x1 = x0 + tile - 1 < W - 1 ? x0 + tile - 1 : W - 1, y1 = y0 + tile - 1 < H - 1 ? y0 + tile - 1 : H - 1;
                    txmax = pix_to_ndc(W - 1 - x0, W, H); txmin = pix_to_ndc(W - 1 - x1, W, H);
                    tymax = pix_to_ndc(H - 1 - y0, H, W); tymin = pix_to_ndc(H - 1 - y1, H, W);
                }
                TopK<KMAX> q;
                pay4 home[KMAX];
                q.init();
                for (int64_t j = 0; j < nf; ++j) {
                    const FaceRec &r = recs[(size_t)j];
                    if (tile > 0) {
                        if (txmax < r.xlo || txmin > r.xhi || tymax < r.ylo || tymin > r.yhi) continue;
                        if (tile_culled(r, txmin, txmax, tymin, tymax)) { ++st.culled; continue; }
                        ++st.staged;
                    }
                    if (p.x < r.xlo || p.x > r.xhi || p.y < r.ylo || p.y > r.yhi) continue;
                    float pz = 0, sd = 0;
                    f3 bc{0, 0, 0};
                    bool keep, unsafe = false;
                    ++st.evals;
                    if (fastdiv && (r.flags & REC_FAST)) {
                        keep = eval_pair<true>(r, p, blur, persp, clipb, pz, sd, bc, unsafe);
                        if (unsafe) { ++st.unsafe; keep = eval_pair<false>(r, p, blur, persp, clipb, pz, sd, bc, unsafe); }
                    } else keep = eval_pair<false>(r, p, blur, persp, clipb, pz, sd, bc, unsafe);
                    if (!keep) continue;
                    const pay4 v{sd, bc.x, bc.y, bc.z};
                    bool done = false;
                    if (r.nb != -1) done = q.sibling(K, true, r.nb, sd < 0.f ? -sd : sd, pz, (int)(f0 + j), v, home, 1, 0);
                    if (bounded) q.insert_ordered(K, !done, pz, (int)(f0 + j), v, home, 1, 0);
                    else q.insert(K, !done, pz, (int)(f0 + j), v, home, 1, 0);
                }
                const int64_t o = (((int64_t)n * H + yi) * W + xi) * K;
                for (int k = 0; k < K; ++k) {
                    float pz = -1.f;
                    int fi = -1;
                    pay4 v{-1.f, -1.f, -1.f, -1.f};
                    bool ok = false;
                    // `k` must be a compile-time constant on the device; on the host a switch-free loop over KMAX does the same
                    for (int kk = 0; kk < KMAX; ++kk) if (kk == k) ok = q.get(kk, home, 1, 0, pz, fi, v);
                    (void)ok;
                    p2f[o + k] = fi; zbuf[o + k] = pz; dists[o + k] = v.x;
                    bary[(o + k) * 3] = v.y; bary[(o + k) * 3 + 1] = v.z; bary[(o + k) * 3 + 2] = v.w;
                }
            }
    }
}

}  // namespace

extern "C" int host_rasterize(const float *fv, const int64_t *first, const int64_t *num, const int64_t *nbr, int N, int H, int W, int K,
                              float blur, int persp, int clipb, int cull, int fastdiv, int tile, int rcp_perturb, int exact_k, int bounded,
                              int64_t *p2f, float *zbuf, float *bary, float *dists, long long *stats4) {
    g_host_rcp_perturb = rcp_perturb;
    Stats st{0, 0, 0, 0};
#define RUN(KM) run<KM>(fv, first, num, nbr, N, H, W, K, blur, persp, clipb, cull, fastdiv, tile, bounded, p2f, zbuf, bary, dists, st)
    if (exact_k && K == 1) RUN(1);
    else if (exact_k && K == 4) RUN(4);
    else if (exact_k && K == 10) RUN(10);
    else if (K <= 25) RUN(25);
    else return -1;
#undef RUN
    stats4[0] = st.evals; stats4[1] = st.unsafe; stats4[2] = st.culled; stats4[3] = st.staged;
    return 0;
}

// pix_to_ndc_fast against pix_to_ndc for every pixel index of an axis (and a few beyond it); returns the number of mismatches
extern "C" long long host_ndccheck(int S1, int S2, int rcp_perturb) {
    g_host_rcp_perturb = rcp_perturb;
    const NdcAxis a = ndc_axis(S1, S2);
    long long bad = 0;
    for (int i = -16; i < S1 + 16; ++i) bad += f2u(pix_to_ndc_fast(i, a)) != f2u(pix_to_ndc(i, S1, S2));
    return bad;
}

// div_fast against the IEEE quotient on caller-supplied operands; returns the number of mismatches (bit compare, NaN == NaN)
extern "C" long long host_divcheck(const float *n, const float *d, long long count, int rcp_perturb) {
    g_host_rcp_perturb = rcp_perturb;
    long long bad = 0;
    for (long long i = 0; i < count; ++i) {
        const float q = div_fast(n[i], d[i], rcp_refined(d[i])), e = n[i] / d[i];
        if (f2u(q) != f2u(e) && !(q != q && e != e)) ++bad;
    }
    return bad;
}

// the launch positions of an XCD segment of `len` tiles of which O are occupied (work_position, what work_scatter_kernel computes per tile):
// pos[0 .. O) = the occupied tiles by rank, pos[O .. len) = the empty ones by rank
extern "C" void host_work_positions(int len, int O, int32_t *pos) {
    for (int r = 0; r < O; ++r) pos[r] = (int32_t)work_position(true, r, O, len);
    for (int e = 0; e < len - O; ++e) pos[O + e] = (int32_t)work_position(false, e, O, len);
}
