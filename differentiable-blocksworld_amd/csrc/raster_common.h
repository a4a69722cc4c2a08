// Tile rasterisation shared by raster_fwd_kernel (raster.hip) and the fused render kernel (render_fused.hip).
//
// Round-2 design (the round-1 kernel spent ~70 % of its issued VALU lane-slots on masked-off work and on moving six-register list
// entries around; profiles/r01_*):
//  * face_setup_kernel evaluates everything pixel-independent once per (view, face) into a 128 B FaceRec (raster_math.h); a tile
//    stages only the INDICES of the faces that touch it (4 B each, wave-ballot ordered compaction as before) and its waves then
//    read each staged record with wave-uniform SCALAR loads (s_load_dwordx16: the record lives in SGPRs, costs no VGPRs and no
//    LDS, and every VALU instruction takes its face operand straight from an SGPR);
//  * besides the box test, binning applies a conservative tile-vs-edge-line test (tile_culled) so that faces whose blur-expanded box
//    touches the tile but whose blur-expanded triangle does not are never staged;
//  * the per-pair arithmetic shares one reciprocal per division stage (div_fast, bit-identical to the IEEE quotient inside its
//    guarded operand range, IEEE `/` outside);
//  * the per-pixel top-K list is a register array of 64-bit keys updated by a branch-free compare-exchange chain (5 VALU per
//    slot); payloads are written once to a fixed slot of an LDS home array and fetched back only when the tile is shaded / stored.
#pragma once
#include "dbw_common.h"

namespace dbw {

// Two-level binning: coarse_bin_kernel (raster.hip) first compacts, per view and per COARSE x COARSE pixel bin, the indices of
// the faces whose blur-expanded box touches the bin (face order preserved); a tile then only scans the list of the bin it
// lies in instead of every face of the view.  list == nullptr: single-level scan.
constexpr int COARSE = 64;
struct CoarseBins {
    const int *list;    // view n, bin b: entries [first_idx[n] * nb + b * num_faces[n], +count[n * nb + b]); an entry = face index relative
                        // to first_idx[n] (20 bits) | first / last cell column (3 + 3 bits) | first / last cell row (3 + 3 bits) the
                        // face's box reaches, cells = the 8x8-pixel squares of the bin
    const int *count;   // (N, nb)
    const unsigned *mask;   // (N, nb, 2): bit (8 * row + column) = some entry reaches that cell
    int nx, ny;         // bins per row / column, nb = nx * ny
    // fine level (cell_bin_kernel, raster.hip): the ordered face list of every cell = 8x8-pixel tile.  nullptr: coarse level only
    const int2 *cell;   // (N, tiles): {first entry in `pool`, count}; count < 0: the pool was full, the tile walks its coarse bin
    const int *pool;    // view-local face indices
    const int2 *work;   // (N * tiles): launch order of the tiles (work_scatter_kernel): position in the XCD-aware grid -> {view, tile row << 16 |
                        // tile column} -- resolved once per tile by the thread that places it, not by two integer divisions in every render wave
    float ndc[4];       // pixel -> NDC constants of the pass, computed on the host (ndc_axis_given): range and offset of x, of y
};
#ifndef DBW_CELL_LISTS
#define DBW_CELL_LISTS 1
#endif
#ifndef DBW_CELL_POOL_PER_TILE
#define DBW_CELL_POOL_PER_TILE 128
#endif
// minimum waves per SIMD the raster kernels are compiled for (the LDS home array of the payloads bounds the residency anyway:
// KMAX * 16 B per pixel)
#define DBW_RASTER_WAVES(KMAX) ((KMAX) <= 4 ? 4 : (KMAX) <= 10 ? 3 : (KMAX) <= 16 ? 2 : 1)

// One staged record -> SGPRs: two s_load_dwordx16 issued together at the top of the iteration (field-by-field scalar loads would
// each be sunk next to their first use and put a scalar-cache round trip in front of every stage of the evaluation).
typedef float v16f __attribute__((ext_vector_type(16)));
__device__ __forceinline__ FaceRec load_rec_uniform(const FaceRec *__restrict__ rp) {
    const v16f *vp = (const v16f *)rp;
    union { v16f v[2]; FaceRec r; } u;
    u.v[0] = vp[0];
    u.v[1] = vp[1];
    // both halves are requested before anything looks at the box in the second one (otherwise the first half is only asked for
    // after the box test: two scalar-cache round trips per face instead of one)
    float pin = u.v[0][0];
    asm volatile("" : "+s"(pin));
    u.v[0][0] = pin;
    return u.r;
}

#ifdef DBW_PROFILE_FWD
// cycle / event accounting of the soft (K > 1) forward passes (tools/fwd_cycles.py only): per-wave s_memtime deltas and counts, kept
// per workgroup (no cross-workgroup contention: a shared counter would serialise 10^6 atomics and distort what it measures) and
// summed on the host: 0 binning, 1 staged-face loop, 2 shading + stores, 3 whole kernel, 4 staged (tile, face) pairs, 5 of those with
// a pixel in the box, 6 (pixel, face) evaluations, 7 kept, 8 wave re-evaluations with IEEE divisions, 9 tiles, 10 tiles with staged
// faces, 11 candidates culled by the tile-vs-edge test, 12 prologue (index arithmetic + the scalar loads of the view's face range)
constexpr int FPROF_BLOCKS = 1 << 17;
__device__ unsigned long long g_fprof[FPROF_BLOCKS * 16];
#define FPROF_T(x) const unsigned long long x = __builtin_readcyclecounter()
#ifdef DBW_PROFILE_FWD_K1
#define FPROF_SEL (KMAX == 1)
#else
#define FPROF_SEL (KMAX > 1)
#endif
#define FPROF_ADD(i, v) { if (FPROF_SEL && (threadIdx.x & 63) == 0 && blockIdx.x < FPROF_BLOCKS) atomicAdd(&g_fprof[(size_t)blockIdx.x * 16 + (i)], (unsigned long long)(v)); }
#define FPROF_CNT(i, pred) { const unsigned long long m_ = __ballot(pred); if (m_) FPROF_ADD(i, __popcll(m_)); }
#else
#define FPROF_T(x)
#define FPROF_ADD(i, v)
#define FPROF_CNT(i, pred)
#endif

#ifndef DBW_RASTER_FASTDIV
#define DBW_RASTER_FASTDIV 1
#endif
#ifndef DBW_RASTER_TILECULL
#define DBW_RASTER_TILECULL 1
#endif

// One chunk of up to 64 staged faces (lane i of `jl` = view-local index of the i-th one) evaluated for the pixels of this wave:
// records arrive in SGPRs (two s_load_dwordx16 per face), lanes are pixels.  Shared by the cell-list path and the legacy staging path.
// (Requesting the record of face i + 1 while face i is evaluated was measured slower at every batch size: profiles/r04_experiments.md.)
// The same record through plain loads, for a site whose address is uniform but whose result need not be pinned to SGPRs:
__device__ __forceinline__ FaceRec load_rec_nowait(const FaceRec *__restrict__ rp) {
    const v16f *vp = (const v16f *)rp;
    union { v16f v[2]; FaceRec r; } u;
    u.v[0] = vp[0];
    u.v[1] = vp[1];
    return u.r;
}
template <int KMAX, bool PAY3>
__device__ __forceinline__ void eval_staged_chunk(const FaceRec *__restrict__ recs, int f_begin, int jl, int mcnt, bool in_img, f2 p, int K, float blur,
                                                  int persp, int clipb, bool fastdiv, bool sign_only, TopK<KMAX, PAY3> &q, pay4 *home, int NT, int tid, bool no_insert = false, bool no_eval = false,
                                                  unsigned long long mask = ~0ull) {
    // (mask: the lanes of `jl` that hold faces of the chunk, in order)
    auto face = [&](const FaceRec &r, int j) {
        const bool inbox = in_img && !(p.x < r.xlo || p.x > r.xhi || p.y < r.ylo || p.y > r.yhi);
        if (__ballot(inbox) == 0ull || no_eval) return;           // (no_eval: ablation switch of tools/diag)
        FPROF_ADD(5, 1);
        FPROF_CNT(6, inbox);
        float pz = 0.f, sd = 0.f;
        f3 bc{0.f, 0.f, 0.f};
        bool keep = false;
        bool redo = !(fastdiv && (r.flags & REC_FAST));
        if (!redo) {
            bool unsafe = false;
            if (inbox) keep = eval_pair<true>(r, p, blur, persp, clipb, pz, sd, bc, unsafe, sign_only);
            redo = __ballot(inbox && unsafe) != 0ull;
        }
        if (redo) {
            FPROF_ADD(8, 1);
            bool unused;
            keep = false;
            if (inbox) keep = eval_pair<false>(r, p, blur, persp, clipb, pz, sd, bc, unused, sign_only);
        }
        if (__ballot(keep) == 0ull || no_insert) return;        // (no_insert: ablation switch of tools/diag, dbw_debug_set_flags 32768)
        FPROF_CNT(7, keep);
        const pay4 v{sd, bc.x, bc.y, bc.z};
        bool done = false;
        if (r.nb != -1) done = q.sibling(K, keep, r.nb, sd < 0.f ? -sd : sd, pz, f_begin + j, v, home, NT, tid);
#if DBW_TOPK_ORDERED
        q.insert_ordered(K, keep && !done, pz, f_begin + j, v, home, NT, tid);
#else
        q.insert(K, keep && !done, pz, f_begin + j, v, home, NT, tid);
#endif
    };
#pragma unroll 1
    for (int i = 0; i < mcnt; ++i) {
        if (!((mask >> i) & 1ull)) continue;
        const int j = __builtin_amdgcn_readlane(jl, i);
        const FaceRec r = load_rec_uniform(recs + f_begin + j);
        face(r, j);
    }
}

// Rasterises the tile of this workgroup: on return every thread holds the sorted top-K list of its pixel (xi, yi) of view n
// (keys in `q`, payloads in the LDS array `home`, stride TW * TH, lane threadIdx.x).  Returns false for the padding blocks of the
// XCD-aware grid.  All threads of the block must call it.  dbg: bit 0 = plain IEEE divisions, bit 1 = no tile culling (parity tests
// run every variant against the oracle), bit 3 = hard pass whose distances are only read for their sign (eval_pair's sign_only).
// (PRE: a callable run once the tile's view and pixel are known and BEFORE the tile's own list is built -- the fused forward evaluates the
// hard env layer of its pixel there, while nothing of the soft pass is live in registers yet)
struct NoPre { __device__ __forceinline__ void operator()(int, int, int, bool) const {} };
template <int KMAX, int TW, int TH, int GROUP = 2, bool PAY3 = false, class PRE = NoPre>
__device__ __forceinline__ bool raster_tile(const FaceRec *__restrict__ recs, const float4 *__restrict__ bbox,
                                            const int *__restrict__ first_idx, const int *__restrict__ num_faces, int H, int W, int K,
                                            float blur, int persp, int clipb, long long total_blocks, const CoarseBins &cb, int dbg,
                                            int &n, int &xi, int &yi, TopK<KMAX, PAY3> &q, pay4 *&home, bool *known_empty = nullptr, PRE pre = PRE()) {
    static_assert(COARSE % TW == 0 && COARSE % TH == 0, "a tile must lie inside one coarse bin");
    // 8x8 tiles read the cell lists of cell_bin_kernel; walking the coarse bin (below) is their fallback -- a bin whose cell lists did
    // not fit the pool, or a caller without a binned workspace -- and gets by with the smallest staging area
    constexpr bool CELLS = DBW_CELL_LISTS && TW == 8 && TH == 8;
    constexpr int G = CELLS ? 1 : GROUP;
    constexpr int NT = TW * TH, NW = NT / DBW_WAVE, CAP = (CELLS ? 1 : 2) * G * NT;      // staged face indices: flushed when more than half full
    __shared__ int s_list[CAP];
    __shared__ int s_wcnt[NW];
    __shared__ pay4 s_home[KMAX == 1 ? 1 : (PAY3 ? (KMAX * NT * 3 + 3) / 4 : KMAX * NT)];
    home = s_home;

    FPROF_T(t_pro);
    const long long logical = xcd_remap(blockIdx.x, total_blocks);
    if (logical < 0) return false;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const unsigned per_view = (unsigned)(tiles_x * tiles_y), lg = (unsigned)logical;      // total_blocks < 2^31 (checked by the host side)
    bool cell_mode = false, empty_tile = false;
    int coff = 0, ccnt = 0;
    unsigned lt = lg;
    int ty, tx;
    if (CELLS && cb.cell) {
        // the tile this position of the grid was given (heavy tiles first, the empty ones spread between them) and its face list
        const int2 w = cb.work[lg];
        n = w.x; ty = (int)((unsigned)w.y >> 16); tx = w.y & 0xffff;
        lt = (unsigned)n * per_view + (unsigned)(ty * tiles_x + tx);
        const int2 c = cb.cell[lt];
        if (c.y == 0) empty_tile = true;
        else if (c.y > 0) { cell_mode = true; coff = c.x; ccnt = c.y; }
    } else {
        n = (int)(lt / per_view);
        const int t = (int)(lt - (unsigned)n * per_view);
        ty = t / tiles_x; tx = t - ty * tiles_x;
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // a wave always owns an 8-aligned compact footprint: lanes 0..63 -> 8x8 (TW == 8) or 16x4 (TW == 16) pixels
    xi = tx * TW + (tid % TW);
    yi = ty * TH + (tid / TW);
    const bool in_img = xi < W && yi < H;
    const int x0 = tx * TW, y0 = ty * TH;
    const int x1 = min(x0 + TW - 1, W - 1), y1 = min(y0 + TH - 1, H - 1);

    pre(n, xi, yi, in_img);
    if (known_empty) *known_empty = empty_tile;
    if (known_empty && empty_tile) return true;        // (the caller has a path of its own for tiles without faces: no list to set up)
    q.init();

    int f_begin = 0, nf;
    const int *lst = nullptr;
    // cells (8x8-pixel squares of the coarse bin) this tile covers
    const int ccx0 = (x0 & (COARSE - 1)) >> 3, ccx1 = (x1 & (COARSE - 1)) >> 3, ccy0 = (y0 & (COARSE - 1)) >> 3, ccy1 = (y1 & (COARSE - 1)) >> 3;
    if (cell_mode || empty_tile) nf = 0;
    else if (cb.list) {
        const int nb = cb.nx * cb.ny, bin = (y0 / COARSE) * cb.nx + (x0 / COARSE);
        nf = cb.count[n * nb + bin];
        // occupied cells of the bin: a tile that covers none of them has nothing to rasterise (two of three tiles of a sparse soft
        // pass) and skips the whole prologue below
        const unsigned mlo = cb.mask[(n * nb + bin) * 2], mhi = cb.mask[(n * nb + bin) * 2 + 1];
        const unsigned long long occ = ((unsigned long long)mhi << 32) | mlo;
        unsigned long long mine = 0ull;
#pragma unroll
        for (int cy = 0; cy < TH / 8; ++cy)
#pragma unroll
            for (int cx = 0; cx < TW / 8; ++cx) mine |= 1ull << (8 * min(ccy0 + cy, 7) + min(ccx0 + cx, 7));
        if (!(occ & mine)) nf = 0;
        if (nf > 0) {
            f_begin = first_idx[n];
            lst = cb.list + (long long)f_begin * nb + (long long)bin * num_faces[n];
        }
    } else {
        f_begin = first_idx[n];
        nf = num_faces[n];
    }
    f2 p{0.f, 0.f};
    float txmax = 0.f, txmin = 0.f, tymax = 0.f, tymin = 0.f;
    if (nf > 0 || ccnt > 0) {
        const NdcAxis ax = ndc_axis_given(W, cb.ndc[0], cb.ndc[1]), ay = ndc_axis_given(H, cb.ndc[2], cb.ndc[3]);
        p.x = pix_to_ndc_fast(W - 1 - xi, ax);
        p.y = pix_to_ndc_fast(H - 1 - yi, ay);
        txmax = pix_to_ndc_fast(W - 1 - x0, ax); txmin = pix_to_ndc_fast(W - 1 - x1, ax);
        tymax = pix_to_ndc_fast(H - 1 - y0, ay); tymin = pix_to_ndc_fast(H - 1 - y1, ay);
    }
    const bool fastdiv = DBW_RASTER_FASTDIV && !(dbg & 1), tilecull = DBW_RASTER_TILECULL && !(dbg & 2), sign_only = (dbg & 8) != 0, no_insert = (dbg & 128) != 0, no_eval = (dbg & 256) != 0, no_faces = (dbg & 512) != 0;
    int cnt = 0;
    FPROF_T(t_begin);
    FPROF_ADD(9, 1);
    FPROF_ADD(12, t_begin - t_pro);
#ifdef DBW_PROFILE_FWD
    unsigned long long t_evsum = 0;
    bool any_staged = false;
#endif
    if (no_faces) { ccnt = 0; nf = 0; }                       // (ablation: every tile treated as empty)
    if (cell_mode && ccnt > 0) {
        // the tile's own face list: nothing to test, nothing to stage
        FPROF_T(t_ev0);
        FPROF_ADD(4, ccnt);
        const int fb = first_idx[n];
        const int *__restrict__ cl = cb.pool + coff;
#pragma unroll 1
        for (int cb0 = 0; cb0 < ccnt; cb0 += DBW_WAVE) {
            const int jl = cb0 + lane < ccnt ? cl[cb0 + lane] : 0;
            eval_staged_chunk<KMAX, PAY3>(recs, fb, jl, min(DBW_WAVE, ccnt - cb0), in_img, p, K, blur, persp, clipb, fastdiv, sign_only, q, home, NT, tid, no_insert, no_eval);
        }
#ifdef DBW_PROFILE_FWD
        t_evsum += __builtin_readcyclecounter() - t_ev0;
        any_staged = true;
#endif
    }
    // The face scan is latency bound: fetch the boxes of G chunks with independent loads before consuming them, so a tile
    // pays nf / (G * NT) memory round trips instead of nf / NT.
#pragma unroll 1
    for (int base0 = 0; base0 < nf; base0 += G * NT) {
        bool hits[G];
        int fjs[G];
        if (lst) {      // coarse-bin entries carry the cell range of the face's box: one load decides
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int j = base0 + g * NT + tid;
                const int e = j < nf ? lst[j] : 0;
                fjs[g] = e & 0xfffff;
                hits[g] = j < nf && !(ccx1 < ((e >> 20) & 7) || ccx0 > ((e >> 23) & 7) || ccy1 < ((e >> 26) & 7) || ccy0 > ((e >> 29) & 7));
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int j = base0 + g * NT + tid;
                fjs[g] = j;
                hits[g] = false;
                if (j < nf) {
                    const float4 bb = bbox[f_begin + j];
                    hits[g] = !(txmax < bb.x || txmin > bb.y || tymax < bb.z || tymin > bb.w);
                }
            }
        }
        if (tilecull) {
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (hits[g] && tile_culled(recs[f_begin + fjs[g]], txmin, txmax, tymin, tymax)) { hits[g] = false; FPROF_CNT(11, true); }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int base = base0 + g * NT;
            if (base >= nf) break;
            const bool hit = hits[g];
            const unsigned long long m = __ballot(hit);
            const int prefix = __popcll(m & ((1ull << lane) - 1ull));
            int woff = 0, tot = __popcll(m);
            if (NW > 1) {
                if (lane == 0) s_wcnt[wv] = tot;
                __syncthreads();
                tot = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) { const int c = s_wcnt[w]; if (w < wv) woff += c; tot += c; }
            }
            if (hit) s_list[cnt + woff + prefix] = fjs[g];
            cnt += tot;
            if (NW > 1) __syncthreads();         // s_wcnt is rewritten by the next chunk
        }
        __syncthreads();
        if (cnt > CAP - G * NT || base0 + G * NT >= nf) {
            // every wave walks the staged faces: records arrive in SGPRs, lanes are pixels
            FPROF_T(t_ev0);
            FPROF_ADD(4, cnt);
#ifdef DBW_PROFILE_FWD
            any_staged = any_staged || cnt > 0;
#endif
#pragma unroll 1
            for (int cb0 = 0; cb0 < cnt; cb0 += DBW_WAVE) {
                const int jl = cb0 + lane < cnt ? s_list[cb0 + lane] : 0;
                const int mcnt = min(DBW_WAVE, cnt - cb0);
eval_staged_chunk<KMAX, PAY3>(recs, f_begin, jl, mcnt, in_img, p, K, blur, persp, clipb, fastdiv, sign_only, q, home, NT, tid, no_insert, no_eval);
            }
            cnt = 0;
            if (NW > 1) __syncthreads();
#ifdef DBW_PROFILE_FWD
            t_evsum += __builtin_readcyclecounter() - t_ev0;
#endif
        }
    }
#ifdef DBW_PROFILE_FWD
    { const unsigned long long t_end = __builtin_readcyclecounter(); FPROF_ADD(1, t_evsum); FPROF_ADD(0, t_end - t_begin - t_evsum); if (any_staged) FPROF_ADD(10, 1); }
#endif
    return true;
}

}  // namespace dbw
