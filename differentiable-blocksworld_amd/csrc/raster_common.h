// Tile rasterisation shared by raster_fwd_kernel (raster.hip) and the fused render kernel (render_fused.hip).
#pragma once
#include "dbw_common.h"

namespace dbw {

constexpr int LIST_CAP = 384;      // 16x16 tiles: 24 KB of face records per block -> 5 resident blocks/CU (the VGPR limit) instead of 4
// Two-level binning: coarse_bin_kernel (raster.hip) first compacts, per view and per COARSE x COARSE pixel bin, the indices of
// the faces whose blur-expanded box touches the bin (face order preserved); a tile then only scans the list of the bin it
// lies in instead of every face of the view.  list == nullptr: single-level scan.
#ifdef DBW_PROFILE_FWD
// cycle accounting of the fused forward (tools/fwd_cycles.py only): per-wave s_memtime deltas, summed in g_fprof
// 0 binning (list walk + LDS fill), 1 per-pixel evaluation of the staged faces, 2 shading + stores, 3 whole kernel, 4 staged faces,
// 5 (pixel, face) evaluations that passed the box test, 6 accepted inserts
__device__ unsigned long long g_fprof[8];
#define FPROF_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define FPROF_ADD(i, v) if (KMAX > 1 && (threadIdx.x & 63) == 0) atomicAdd(&g_fprof[i], (unsigned long long)(v))      // soft passes only
#define FPROF_CNT(i, pred) { const unsigned long long m_ = __ballot(pred); if (KMAX > 1 && (threadIdx.x & 63) == 0 && m_) atomicAdd(&g_fprof[i], (unsigned long long)__popcll(m_)); }
#else
#define FPROF_T(x)
#define FPROF_ADD(i, v)
#define FPROF_CNT(i, pred)
#endif
constexpr int COARSE = 64;
struct CoarseBins {
    const int *list;    // view n, bin b: entries [first_idx[n] * nb + b * num_faces[n], +count[n * nb + b]), indices relative to first_idx[n]
    const int *count;   // (N, nb)
    int nx, ny;         // bins per row / column, nb = nx * ny
};
// minimum waves per SIMD the raster kernels are compiled for (caps the VGPR budget: the top-K list lives in registers)
#define DBW_RASTER_WAVES(KMAX) ((KMAX) <= 4 ? 4 : (KMAX) <= 10 ? 2 : 1)

struct __attribute__((aligned(16))) FaceRec {
    float v[9];
    float xlo, xhi, ylo, yhi;
    int nb;
    int id;
    int pad;
};
static_assert(sizeof(FaceRec) == 64, "FaceRec must be 64 B");

template <int KMAX>
struct TopK {
    float pz[KMAX], ds[KMAX], b0[KMAX], b1[KMAX], b2[KMAX];
    int fi[KMAX];

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < KMAX; ++i) { pz[i] = INFINITY; fi[i] = 0x7fffffff; ds[i] = b0[i] = b1[i] = b2[i] = -1.f; }
    }
    __device__ __forceinline__ static bool less(float pa, int fa, float pb, int fb) {
        return (pa < pb) || (!(pb < pa) && fa < fb);
    }
    __device__ __forceinline__ void swap_with(int i, float &cp, int &cf, float &cd, float &c0, float &c1, float &c2) {
        float t;
        int ti;
        t = pz[i]; pz[i] = cp; cp = t;
        ti = fi[i]; fi[i] = cf; cf = ti;
        t = ds[i]; ds[i] = cd; cd = t;
        t = b0[i]; b0[i] = c0; c0 = t;
        t = b1[i]; b1[i] = c1; c1 = t;
        t = b2[i]; b2[i] = c2; c2 = t;
    }
    // sorted insert; the displaced largest entry falls off the end (== emplace_back, sort, pop_back if size > K)
    __device__ __forceinline__ void insert(int K, float cp, int cf, float cd, float c0, float c1, float c2) {
#pragma unroll
        for (int i = 0; i < KMAX; ++i)
            if (i < K && less(cp, cf, pz[i], fi[i])) swap_with(i, cp, cf, cd, c0, c1, c2);
    }
    __device__ __forceinline__ void cswap(int i) {  // order entries i, i+1
        if (less(pz[i + 1], fi[i + 1], pz[i], fi[i])) swap_with(i, pz[i + 1], fi[i + 1], ds[i + 1], b0[i + 1], b1[i + 1], b2[i + 1]);
    }
    // sibling rule: returns true if `nb` was found (entry possibly replaced, list re-sorted)
    __device__ __forceinline__ bool sibling(int K, int nb, float dist, float cp, int cf, float cd, float c0, float c1, float c2) {
        bool found = false;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
            if (i < K && !found && fi[i] == nb) {
                found = true;
                const float nd = ds[i] < 0.f ? -ds[i] : ds[i];
                if (dist < nd) { pz[i] = cp; fi[i] = cf; ds[i] = cd; b0[i] = c0; b1[i] = c1; b2[i] = c2; }
            }
        }
        if (found) {  // one entry may be out of place: one forward + one backward adjacent pass restores the order
#pragma unroll
            for (int i = 0; i < KMAX - 1; ++i) if (i + 1 < K) cswap(i);
#pragma unroll
            for (int i = KMAX - 2; i >= 0; --i) if (i + 1 < K) cswap(i);
        }
        return found;
    }
};

// Rasterises the tile of this workgroup: on return every thread holds the sorted top-K list of its pixel (xi, yi) of view n.
// Returns false for the padding blocks of the XCD-aware grid.  All threads of the block must call it.
template <int KMAX, int TW, int TH, int GROUP = 2>
__device__ __forceinline__ bool raster_tile(const float *__restrict__ fv, const float4 *__restrict__ bbox,
                                            const int *__restrict__ first_idx, const int *__restrict__ num_faces,
                                            const int *__restrict__ neighbor, int H, int W, int K, float blur, int persp,
                                            int clipb, long long total_blocks, const CoarseBins &cb, int &n, int &xi, int &yi,
                                            TopK<KMAX> &q) {
    static_assert(COARSE % TW == 0 && COARSE % TH == 0, "a tile must lie inside one coarse bin");
    constexpr int NT = TW * TH, NW = NT / DBW_WAVE, CAP = NT >= 256 ? LIST_CAP : 4 * NT;
    __shared__ FaceRec s_face[CAP];
    __shared__ int s_wcnt[NW];

    const long long logical = xcd_remap(blockIdx.x, total_blocks);
    if (logical < 0) return false;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    n = (int)(logical / (tiles_x * tiles_y));
    const int t = (int)(logical % (tiles_x * tiles_y));
    const int ty = t / tiles_x, tx = t % tiles_x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // a wave always owns an 8-aligned compact footprint: lanes 0..63 -> 8x8 (TW == 8) or 16x4 (TW == 16) pixels
    xi = tx * TW + (tid % TW);
    yi = ty * TH + (tid / TW);
    const bool in_img = xi < W && yi < H;
    f2 p;
    p.x = pix_to_ndc(W - 1 - xi, W, H);
    p.y = pix_to_ndc(H - 1 - yi, H, W);
    const int x0 = tx * TW, y0 = ty * TH;
    const int x1 = min(x0 + TW - 1, W - 1), y1 = min(y0 + TH - 1, H - 1);
    const float txmax = pix_to_ndc(W - 1 - x0, W, H), txmin = pix_to_ndc(W - 1 - x1, W, H);
    const float tymax = pix_to_ndc(H - 1 - y0, H, W), tymin = pix_to_ndc(H - 1 - y1, H, W);

    q.init();
#ifdef DBW_PROFILE_FWD
    unsigned long long t_bin = __builtin_readcyclecounter();
    int n_ins_ = 0;
#endif

    const int f_begin = first_idx[n];
    int nf = num_faces[n];
    const int *lst = nullptr;
    if (cb.list) {
        const int nb = cb.nx * cb.ny, bin = (y0 / COARSE) * cb.nx + (x0 / COARSE);
        lst = cb.list + (long long)f_begin * nb + (long long)bin * nf;
        nf = cb.count[n * nb + bin];
    }
    int cnt = 0;
    // The face scan is latency bound (every tile walks the whole per-view bbox table): fetch the boxes of GROUP chunks with
    // independent loads before consuming them, so a tile pays nf / (GROUP * NT) memory round trips instead of nf / NT.
#pragma unroll 1
    for (int base0 = 0; base0 < nf; base0 += GROUP * NT) {
        float4 bbs[GROUP];
        bool hits[GROUP];
        int fjs[GROUP];
#pragma unroll
        for (int g = 0; g < GROUP; ++g) {
            const int j = base0 + g * NT + tid;
            hits[g] = false;
            fjs[g] = j;
            if (j < nf && lst) fjs[g] = lst[j];
        }
#pragma unroll
        for (int g = 0; g < GROUP; ++g) {
            const int j = base0 + g * NT + tid;
            if (j < nf) {
                bbs[g] = bbox[f_begin + fjs[g]];
                hits[g] = !(txmax < bbs[g].x || txmin > bbs[g].y || tymax < bbs[g].z || tymin > bbs[g].w);
            }
        }
#pragma unroll
        for (int g = 0; g < GROUP; ++g) {
        const int base = base0 + g * NT;
        if (base >= nf) break;
        const int j = fjs[g];
        const bool hit = hits[g];
        const float4 bb = bbs[g];
        const unsigned long long m = __ballot(hit);
        const int prefix = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wcnt[wv] = __popcll(m);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const int c = s_wcnt[w]; if (w < wv) woff += c; tot += c; }
        if (hit) {
            FaceRec &r = s_face[cnt + woff + prefix];
            const float *src = fv + (long long)(f_begin + j) * 9;
#pragma unroll
            for (int i = 0; i < 9; ++i) r.v[i] = src[i];
            r.xlo = bb.x; r.xhi = bb.y; r.ylo = bb.z; r.yhi = bb.w;
            r.nb = neighbor ? neighbor[f_begin + j] : -1;
            r.id = f_begin + j;
        }
        cnt += tot;
        __syncthreads();
        if (cnt > CAP - NT || base + NT >= nf) {
            FPROF_T(t_ev0);
            FPROF_ADD(0, t_ev0 - t_bin);
            FPROF_ADD(4, cnt);
            FaceRec nxt = s_face[0];          // software pipeline: the next record's LDS read overlaps this face's arithmetic
#pragma unroll 1
            for (int i = 0; i < cnt; ++i) {
                const FaceRec r = nxt;
                nxt = s_face[i + 1 < cnt ? i + 1 : i];
                const bool inbox_ = in_img && !(p.x < r.xlo || p.x > r.xhi || p.y < r.ylo || p.y > r.yhi);
                FPROF_CNT(5, inbox_);
                if (inbox_) {
                    const f2 a{r.v[0], r.v[1]}, b{r.v[3], r.v[4]}, c{r.v[6], r.v[7]};
                    const float z0 = r.v[2], z1 = r.v[5], z2 = r.v[8];
                    // bary_fwd, opened up: for a hard pass (blur == 0) a pixel outside the triangle can never be kept, and
                    // "outside" is decided exactly by the signs of the edge functions (b_i = e_i / area <= 0 for some i),
                    // before paying the twelve IEEE divisions of the full evaluation
                    const float area = edge_fn(c, a, b) + DBW_EPS;
                    const float e0 = edge_fn(p, b, c), e1 = edge_fn(p, c, a), e2 = edge_fn(p, a, b);
                    if (blur == 0.f) {
                        const bool pos = area > 0.f;
                        if (e0 == 0.f || e1 == 0.f || e2 == 0.f || (e0 > 0.f) != pos || (e1 > 0.f) != pos || (e2 > 0.f) != pos) continue;
                    }
                    f3 bary0;
                    bary0.x = e0 / area; bary0.y = e1 / area; bary0.z = e2 / area;
                    const f3 bp = persp ? persp_fwd(bary0, z0, z1, z2) : bary0;
                    const f3 bc = clipb ? clip_fwd(bp) : bp;
                    const float pzv = bc.x * z0 + bc.y * z1 + bc.z * z2;
                    if (!(pzv < 0.f)) {
                        const float dist = point_tri_dist(p, a, b, c);
                        const bool inside = bp.x > 0.f && bp.y > 0.f && bp.z > 0.f;
                        if (inside || !(dist >= blur)) {
                            const float sd = inside ? -dist : dist;
                            bool done = false;
                            if (r.nb != -1) done = q.sibling(K, r.nb, dist, pzv, r.id, sd, bc.x, bc.y, bc.z);
                            if (!done) q.insert(K, pzv, r.id, sd, bc.x, bc.y, bc.z);
#ifdef DBW_PROFILE_FWD
                            ++n_ins_;
#endif
                        }
                    }
                }
            }
            cnt = 0;
            __syncthreads();
#ifdef DBW_PROFILE_FWD
            { const unsigned long long t_ev1 = __builtin_readcyclecounter(); FPROF_ADD(1, t_ev1 - t_ev0); t_bin = t_ev1; }
#endif
        }
        }
    }
#ifdef DBW_PROFILE_FWD
    if (KMAX > 1 && n_ins_) atomicAdd(&g_fprof[6], (unsigned long long)n_ins_);
#endif
    return true;
}

}  // namespace dbw
