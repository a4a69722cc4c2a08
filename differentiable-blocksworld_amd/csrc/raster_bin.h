// Two-level face binning of a render pass, as device functions of ONE workgroup = one (view, 64x64-pixel bin): shared by the stand-alone
// kernels of raster.hip (coarse_bin_kernel, cell_bin_kernel: the operator-level entry points and stage 1 of the fused passes) and by the
// fused set-up kernel of the training step (train_step.hip: both levels of both scenes in one launch).
#pragma once
#include "raster_common.h"

namespace dbw {

// Launch order of the tiles of a render pass: the tile count is fixed by the image, the work per tile is not -- a tile of a soft pass
// costs about as many microseconds as it has faces, two thirds of the tiles have none, and the tiles that were started last with
// dozens of faces used to keep a handful of waves busy for 80 us after everything else had drained.  Inside every XCD segment of the
// XCD-aware grid (the tiles xcd_remap gives that XCD: same tiles, same L2 locality) the tiles are ordered by face-count class,
// heaviest first, and the empty tiles -- whose composite + loss epilogue is pure memory traffic -- are spread evenly between the
// occupied ones, so that the streaming work hides behind the arithmetic instead of piling up at the end.
//   cell_bin_block: class + rank inside (segment, class) of every tile (returning atomics on hdr[1 + segment * 16 + class])
//   work_scatter_kernel (raster.hip): thread = tile: work[position] = {view, tile row << 16 | tile column}
constexpr int WORK_CLASSES = 10, WORK_RANK_BITS = 27;      // (rank < tiles of the pass < 2^27: checked where the workspace is laid out)
__device__ __forceinline__ int work_class(int c) {
    return c < 0 ? 0 : c == 0 ? 9 : c >= 64 ? 0 : c >= 48 ? 1 : c >= 32 ? 2 : c >= 24 ? 3 : c >= 16 ? 4 : c >= 12 ? 5 : c >= 8 ? 6 : c >= 4 ? 7 : 8;
}

constexpr int CELL_ENTRY_CAP = 1024, CELL_CHUNKS = CELL_ENTRY_CAP / 64;
constexpr int CELL_HDR_INTS = 1 + 8 * 16;      // cell-list header: pool cursor, 8 x 16 class cursors

// LDS of one binning workgroup (256 threads)
// (12.7 KB: twelve workgroups per CU -- the per-(chunk, cell) offsets of the fill used to sit here too, 4 KB more, nine per CU; they are a
// popcount prefix over `col`, recomputed where they are used)
struct BinShared {
    int wcnt[2][4];
    float cmin[2][8], cmax[2][8];              // NDC extents of the pixel centres of cell column / row c (empty beyond the image)
    unsigned mask[2];
    int base;
    int ent[CELL_ENTRY_CAP];                   // the first CELL_ENTRY_CAP entries of the bin's list
    unsigned long long col[CELL_CHUNKS][64];
    int celloff[64];                           // first pool entry of every cell's list
};

// all 256 threads; followed by a barrier of the caller
__device__ __forceinline__ void bin_cell_extents(BinShared &S, int H, int W, int x0, int y0) {
    if (threadIdx.x < 16) {
        const int axis = threadIdx.x >> 3, c = threadIdx.x & 7;
        const int S1 = axis ? H : W, S2 = axis ? W : H, p0 = (axis ? y0 : x0) + 8 * c, p1 = min(p0 + 7, S1 - 1);
        S.cmax[axis][c] = p0 < S1 ? pix_to_ndc(S1 - 1 - p0, S1, S2) : -INFINITY;
        S.cmin[axis][c] = p0 < S1 ? pix_to_ndc(S1 - 1 - p1, S1, S2) : INFINITY;
    }
    if (threadIdx.x < 2) S.mask[threadIdx.x] = 0u;
}

// Coarse level: the workgroup compacts the faces of view n whose box touches the bin, in face order (wave ballots), so that tiles see the
// same candidate sequence as a full scan.  An entry packs the face index with the range of 8x8-pixel cells of the bin the box reaches
// (pixel-centre extents, the same comparisons a tile would make), so that a tile decides from the entry alone -- no second, dependent
// load of the box -- and the bin also gets a 64-bit mask of its occupied cells: a tile none of whose cells is occupied exits before its
// prologue.  Needs bin_cell_extents + barrier before it.  -> the length of the list (every thread); its first CELL_ENTRY_CAP entries
// stay in S.ent for a cell_bin_block that follows in the same workgroup.
__device__ __forceinline__ int coarse_bin_block(const float4 *__restrict__ bbox, const int *__restrict__ first_idx, const int *__restrict__ num_faces,
                                                int H, int W, int nx, int ny, int *__restrict__ list, int *__restrict__ count,
                                                unsigned *__restrict__ mask, int n, int bin, BinShared &S) {
    const int nb = nx * ny;
    const int x0 = (bin % nx) * COARSE, y0 = (bin / nx) * COARSE;
    const int x1 = min(x0 + COARSE - 1, W - 1), y1 = min(y0 + COARSE - 1, H - 1);
    const float bxmax = pix_to_ndc(W - 1 - x0, W, H), bxmin = pix_to_ndc(W - 1 - x1, W, H);
    const float bymax = pix_to_ndc(H - 1 - y0, H, W), bymin = pix_to_ndc(H - 1 - y1, H, W);
    const int f_begin = first_idx[n], nf = num_faces[n];
    int *out = list + (long long)f_begin * nb + (long long)bin * nf;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int cnt = 0;
    // The rounds of a bin are a chain -- the workgroup's latency, not its work, is what the step pays for at small batches --, so: the
    // boxes of 1024 faces are requested together (a round used to start with a load every thread then waited for), and a round covers
    // 512 faces, two per thread, behind ONE pair of barriers (face order: first half waves 0..3, then second half waves 0..3).
    for (int base0 = 0; base0 < nf; base0 += 1024) {
        float4 bbs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = base0 + r * 256 + threadIdx.x;
            bbs[r] = j < nf ? bbox[f_begin + j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const int base = base0 + r2 * 512;
            if (base >= nf) break;
            bool hit[2] = {false, false};
            int entry[2] = {0, 0};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = base + h * 256 + threadIdx.x;
                if (j < nf) {
                    const float4 bb = bbs[r2 * 2 + h];
                    if (!(bxmax < bb.x || bxmin > bb.y || bymax < bb.z || bymin > bb.w)) {
                        int cx0 = 8, cx1 = -1, cy0 = 8, cy1 = -1;
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            if (!(S.cmax[0][c] < bb.x || S.cmin[0][c] > bb.y)) { cx0 = min(cx0, c); cx1 = c; }
                            if (!(S.cmax[1][c] < bb.z || S.cmin[1][c] > bb.w)) { cy0 = min(cy0, c); cy1 = c; }
                        }
                        hit[h] = cx1 >= 0 && cy1 >= 0;        // a box that slips between the pixel centres of two cells touches no pixel at all
                        if (hit[h]) {
                            entry[h] = j | (cx0 << 20) | (cx1 << 23) | (cy0 << 26) | (cy1 << 29);
                            const unsigned row = ((1u << (cx1 - cx0 + 1)) - 1u) << cx0;
                            unsigned lo = 0u, hi = 0u;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                if (q >= cy0 && q <= cy1) lo |= row << (8 * q);
                                if (q + 4 >= cy0 && q + 4 <= cy1) hi |= row << (8 * q);
                            }
                            if (lo) atomicOr(&S.mask[0], lo);
                            if (hi) atomicOr(&S.mask[1], hi);
                        }
                    }
                }
            }
            const unsigned long long m0 = __ballot(hit[0]), m1 = __ballot(hit[1]);
            if (lane == 0) { S.wcnt[0][wv] = __popcll(m0); S.wcnt[1][wv] = __popcll(m1); }
            __syncthreads();
            int off0 = 0, tot0 = 0, off1 = 0, tot1 = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int c0 = S.wcnt[0][w], c1 = S.wcnt[1][w];
                if (w < wv) { off0 += c0; off1 += c1; }
                tot0 += c0; tot1 += c1;
            }
            const unsigned long long below = (1ull << lane) - 1ull;
            if (hit[0]) {
                const int pos = cnt + off0 + __popcll(m0 & below);
                out[pos] = entry[0];
                if (pos < CELL_ENTRY_CAP) S.ent[pos] = entry[0];
            }
            if (hit[1]) {
                const int pos = cnt + tot0 + off1 + __popcll(m1 & below);
                out[pos] = entry[1];
                if (pos < CELL_ENTRY_CAP) S.ent[pos] = entry[1];
            }
            cnt += tot0 + tot1;
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) count[n * nb + bin] = cnt;
    if (threadIdx.x < 2) mask[(n * nb + bin) * 2 + threadIdx.x] = S.mask[threadIdx.x];
    return cnt;
}

// Fine level: the workgroup splits the bin's ordered list into the ordered lists of its 64 cells (8x8 pixels = the tile of one wave of the
// soft passes), so that a render wave reads exactly the faces it has to evaluate -- no list walk, no compaction, no staging in LDS, no
// tile-vs-edge test per (tile, face) at render time -- and so that the number of faces of every tile is known before the render kernel
// starts (work_scatter_kernel).
//   masks:     thread = list entry: the 64-bit mask of the cells of the entry's box range that the blur-expanded triangle can touch
//              (tile_culled, conservative)
//   transpose: 64 ballots per chunk of 64 entries: the column of a cell = bit i set where entry i of the chunk touches the cell
//   reserve:   one atomic on the pool cursor per bin; a bin whose lists do not fit (or with more than CELL_ENTRY_CAP entries) marks
//              its cells "walk the coarse list" (count -1)
//   fill:      thread = (cell, chunk): the set bits of its column, in order, behind the entries of the chunks before it -- every
//              loop runs over entries that exist, not over the whole list
// cnt_all = the length of the bin's list.  STASHED: its entries are already in S.ent (coarse_bin_block ran in this workgroup, barrier
// in between); otherwise they are read from `lst`.  Needs bin_cell_extents + barrier before it.
template <bool STASHED>
__device__ __forceinline__ void cell_bin_block(const FaceRec *__restrict__ recs, const int *__restrict__ first_idx, int N, int H, int W, int nx, int ny,
                                               const int *__restrict__ lst, int cnt_all, int2 *__restrict__ cell, int *__restrict__ pool,
                                               int pool_cap, int *__restrict__ hdr, int *__restrict__ rank, int n, int bin, BinShared &S,
                                               int *__restrict__ dom = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int x0 = (bin % nx) * COARSE, y0 = (bin / nx) * COARSE;
    const int tiles_x = (W + 7) >> 3, tiles_y = (H + 7) >> 3, tiles = tiles_x * tiles_y;
    const int f_begin = first_idx[n];
    const bool too_long = cnt_all > CELL_ENTRY_CAP;
    const int cnt = too_long ? 0 : cnt_all, chunks = (cnt + 63) >> 6;
    for (int ch = wv; ch < chunks; ch += 4) {
        const int idx = ch * 64 + lane;
        unsigned mlo = 0u, mhi = 0u;
        if (idx < cnt) {
            const int e = STASHED ? S.ent[idx] : lst[idx];
            const int j = e & 0xfffff, ex0 = (e >> 20) & 7, ex1 = (e >> 23) & 7, ey0 = (e >> 26) & 7, ey1 = (e >> 29) & 7;
            if (!STASHED) S.ent[idx] = e;
            // (by value, through 128-bit loads: every field the tile test reads sits in registers before the loop over the cells starts --
            // a reference left a dependent global load per edge test inside it)
            FaceRec r;
            {
                const uint4 *src = (const uint4 *)(recs + f_begin + j);
                uint4 *dst = (uint4 *)&r;
#pragma unroll
                for (int w = 0; w < 8; ++w) dst[w] = src[w];
            }
            for (int cy = ey0; cy <= ey1; ++cy)
                for (int cx = ex0; cx <= ex1; ++cx)
                    if (!tile_culled(r, S.cmin[0][cx], S.cmax[0][cx], S.cmin[1][cy], S.cmax[1][cy])) {
                        if (cy < 4) mlo |= 1u << (8 * cy + cx); else mhi |= 1u << (8 * (cy - 4) + cx);
                    }
        }
        unsigned long long col = 0ull;          // lane c: the entries of this chunk that touch cell c
#pragma unroll 2
        for (int c = 0; c < 32; ++c) {
            const unsigned long long b0 = __ballot((mlo >> c) & 1u), b1 = __ballot((mhi >> c) & 1u);
            if (lane == c) col = b0;
            if (lane == c + 32) col = b1;
        }
        S.col[ch][lane] = col;
    }
    __syncthreads();
    // per cell (wave 0, lane = cell): entries per chunk -> exclusive prefix over the chunks, total; then over the cells
    const int px = x0 + 8 * (lane & 7), py = y0 + 8 * (lane >> 3);
    const bool in_img = px < W && py < H;
    const int tile = (py >> 3) * tiles_x + (px >> 3);
    if (wv == 0) {
        int all = 0;
        for (int ch = 0; ch < chunks; ++ch) all += __popcll(S.col[ch][lane]);
        int incl = all;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        const int total = __shfl(incl, 63, 64);
        // The bin's returning atomics -- its share of the pool, and one per distinct (XCD segment, face-count class) of its tiles for their
        // ranks -- are ISSUED TOGETHER and read afterwards: one round trip instead of up to eleven in a row (they were a third of the
        // workgroup's latency).  Whether the pool overflowed is only known with the pool atomic's answer, so the tiles of a bin that does
        // overflow (and then walk the coarse list) keep the class of their face count: the class only decides WHERE in the launch order a
        // tile goes, any assignment of ranks that is a permutation is a correct one.  (Handing the ranks back to take new ones for class 0
        // is not: another bin may have taken ranks above them in between -- two tiles on one position, a position without a tile.)
        int base_off = 0;
        if (lane == 0 && total > 0) base_off = atomicAdd(&hdr[0], total);
        const long long per = ((long long)N * tiles + 7) / 8;
        const long long L = (long long)n * tiles + tile;
        const int seg16 = in_img ? (int)(L / per) * 16 : 0;
        auto ranks = [&](int key, int &r_out, int &leader_out, int &gcount_out) {
            unsigned long long rem = __ballot(key >= 0);
            int leader = lane, gcount = 0, grank = 0;
            while (rem) {
                const int l0 = __ffsll((long long)rem) - 1;
                const int k0 = __builtin_amdgcn_readlane(key, l0);
                const unsigned long long m = __ballot(key == k0);
                if (key == k0) { leader = l0; gcount = __popcll(m); grank = __popcll(m & ((1ull << lane) - 1ull)); }
                rem &= ~m;
            }
            int b0 = 0;
            if (key >= 0 && lane == leader) b0 = atomicAdd(&hdr[1 + key], gcount);
            r_out = __shfl(b0, leader, 64) + grank;
            leader_out = leader; gcount_out = gcount;
        };
        const int key = in_img ? seg16 + work_class(too_long ? -1 : all) : -1;
        int r = 0, leader = lane, gcount = 0;
        ranks(key, r, leader, gcount);
        base_off = __shfl(base_off, 0, 64);
        const bool overflow = too_long || (total > 0 && (long long)base_off + total > (long long)pool_cap);
        const int off = base_off + (incl - all);
        S.celloff[lane] = off;
        if (lane == 0) S.base = (overflow || total == 0) ? -1 : 0;
        const int count = overflow ? -1 : all;
        if (in_img) cell[L] = make_int2(overflow ? 0 : off, count);
        // (class, and rank of the tile inside its (XCD segment, class): work_scatter_kernel turns them into the tile's place in the launch
        // order.  The class travels with the rank -- for a bin that overflowed the pool it is not the class of the stored count, -1)
        if (in_img) rank[L] = r | ((key - seg16) << WORK_RANK_BITS);
    }
    __syncthreads();
    // dom (the env scene of the training step: a hard pass of a few huge faces): the face of a cell's list that is IN FRONT OF all the
    // others over its whole extent -- its farthest vertex nearer than every other listed face's nearest (depths are convex combinations of
    // the vertex depths, so it wins at every pixel it covers) -- and is no half of a split quad and inside the guarded range of the
    // shared-reciprocal divisions; -1 where there is none (or the list has more than four entries).  The ground in front of the sky dome, a
    // sky face alone: the fg pass's folded env layer then evaluates ONLY that face for the tile (render_fused.hip: env_fold_pixel) -- decided
    // here once per tile by one lane, next to the fill, instead of by every render wave through a chain of dependent scalar loads.
    if (dom && wv == 1) {
        int jb = -1, all = 0;
        for (int ch = 0; ch < chunks; ++ch) all += __popcll(S.col[ch][lane]);
        if (S.base == 0 && all >= 1 && all <= 4) {
            int jbest = 0;
            float near_b = 0.f, far_b = INFINITY, near_others = INFINITY;
            for (int ch = 0; ch < chunks; ++ch) {
                unsigned long long bits = S.col[ch][lane];
                while (bits) {
                    const int i = __ffsll((long long)bits) - 1;
                    bits &= bits - 1ull;
                    const int j = S.ent[ch * 64 + i] & 0xfffff;
                    const float *zp = (const float *)(recs + f_begin + j) + 6;          // FaceRec::z0, z1, z2
                    const float z0 = zp[0], z1 = zp[1], z2 = zp[2];
                    const float zn = fminf(z0, fminf(z1, z2)), zf = fmaxf(z0, fmaxf(z1, z2));
                    if (zf < far_b) { near_others = fminf(near_others, near_b > 0.f ? near_b : INFINITY); jbest = j; near_b = zn; far_b = zf; }
                    else near_others = fminf(near_others, zn);
                }
            }
            if (far_b * 1.00001f < near_others) {
                const FaceRec *rb = recs + f_begin + jbest;
                if ((rb->flags & REC_FAST) && rb->nb == -1) jb = jbest;
            }
        }
        if (in_img) dom[(long long)n * tiles + tile] = jb;
    }
    if (S.base < 0) return;
    for (int ch = wv; ch < chunks; ch += 4) {
        unsigned long long bits = S.col[ch][lane];
        int o = S.celloff[lane];
        for (int c = 0; c < ch; ++c) o += __popcll(S.col[c][lane]);
        while (bits) {
            const int i = __ffsll((long long)bits) - 1;
            pool[o++] = S.ent[ch * 64 + i] & 0xfffff;
            bits &= bits - 1ull;
        }
    }
}

// Where the pieces of a rasteriser workspace live (raster.hip: dbw_raster_workspace_layout).  boxes = the workspace itself.
struct RasterWorkspace {
    float4 *bbox;
    FaceRec *recs;
    void *shade_recs;                  // F_total x 64 B (ShadeRec of the fused soft forward)
    bool binned, cells;
    int nx, ny;
    int *count; unsigned *mask; int *list;       // coarse level
    int *hdr; int2 *cell; int2 *work; int *rank; int *pool; int pool_cap;      // fine level
    int *dom;                          // (N * tiles) dominant face of every cell or -1 (cell_bin_block; filled for the env scene of the training step only)
};

}  // namespace dbw

// (raster.hip) layout of `workspace` for a pass of N views with F_total packed faces; binned / cells say what fits and is switched on
int dbw_raster_workspace_layout(void *workspace, size_t workspace_bytes, long long F_total, long long max_faces_per_view, int N, int H, int W,
                                bool want_cells, dbw::RasterWorkspace &L);
