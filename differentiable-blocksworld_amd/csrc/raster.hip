// Tiled screen-space rasteriser for gfx950 (MI355X): forward (per-pixel top-K nearest faces) and backward.
//
// Replaces pytorch3d 0.7.1 `_C.rasterize_meshes` / `_C.rasterize_meshes_backward`, reached by the reference through
// src/model/renderer.py:53-54,92-94 (MeshRasterizer).  Semantics: SURVEY.md A.5 / A.6, canonical = CPU naive path
// (list kept sorted by (z, face id); split-quad siblings de-duplicated).  Results are bit-exact to
// oracle/raster_ref.c given the same fp32 face_verts.
//
// Design (MI355X-first, not a translation of PyTorch3D's coarse/fine CUDA pair):
//  * face_setup_kernel turns every face into a 128 B record of its pixel-independent arithmetic (raster_math.h: FaceRec) and a
//    blur-expanded box; coarse_bin_kernel compacts, per 64x64-pixel bin, the ordered list of faces touching it;
//  * one workgroup = one pixel tile of one view (8x8 = one wave64 for the soft K-layer passes, 16x16 for the hard 1-layer pass
//    whose faces are huge); it bins the faces of its coarse bin against the tile with a wave-ballot ORDERED compaction (face
//    order must be preserved: the sibling de-duplication rule is order dependent) and stages their indices in LDS; its waves
//    then read each staged record with wave-uniform scalar loads and evaluate it for their 64 pixels;
//  * each lane keeps the sorted 64-bit keys of its top-K list in VGPRs and the payloads in an LDS home array (raster_common.h);
//  * blockIdx -> tile mapping is XCD-aware: all tiles of a view run on one XCD so its face table and its output
//    rows stay in that XCD's L2;
//  * measured (profiles/): the kernel is instruction-issue bound, not HBM bound -- see DESIGN.md section 4.
#include "raster_common.h"
#include "raster_bin.h"
#include "step_kernels.h"
#include "texture_body.h"
#include "../../include/dbw_hip.h"

#include <math.h>

using namespace dbw;

namespace {

constexpr int TILE = 16;
constexpr int NT = 256;

// Per-face record + screen box expanded by sqrt(blur_radius); faces that can never be hit (touching/behind the camera plane,
// zero area, culled) get an empty box.  One rounding per value, same as the oracle's per-pixel expression.
__global__ void face_setup_kernel(const float *__restrict__ fv, const int *__restrict__ neighbor, long long F, float margin, int cull,
                                  float4 *__restrict__ bbox, FaceRec *__restrict__ recs, int *__restrict__ zero, int nzero) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // (the cursors of the cell lists, cell_bin_kernel: cleared here, two launches ahead of their first use)
    for (long long z = i; z < nzero; z += (long long)gridDim.x * blockDim.x) zero[z] = 0;
    if (i >= F) return;
    FaceRec r;
    float box[4];
    make_face_rec(fv + i * 9, margin, cull, neighbor ? neighbor[i] : -1, r, box);
    bbox[i] = make_float4(box[0], box[1], box[2], box[3]);
    recs[i] = r;
}

// The two levels of the binning (raster_bin.h) as stand-alone kernels: one workgroup per (view, COARSE x COARSE pixel bin).
__global__ __launch_bounds__(256) void coarse_bin_kernel(const float4 *__restrict__ bbox, const int *__restrict__ first_idx,
                                                         const int *__restrict__ num_faces, int H, int W, int nx, int ny,
                                                         int *__restrict__ list, int *__restrict__ count, unsigned *__restrict__ mask) {
    __shared__ BinShared S;
    const int nb = nx * ny, n = blockIdx.x / nb, bin = blockIdx.x % nb;
    bin_cell_extents(S, H, W, (bin % nx) * COARSE, (bin / nx) * COARSE);
    __syncthreads();
    coarse_bin_block(bbox, first_idx, num_faces, H, W, nx, ny, list, count, mask, n, bin, S);
}

__global__ __launch_bounds__(256) void work_scatter_kernel(const int2 *__restrict__ cell, const int *__restrict__ rank, const int *__restrict__ hdr,
                                                           long long total, int tiles_x, int per_view, int2 *__restrict__ work, unsigned *sync_flag,
                                                           unsigned sync_val) {
    if (sync_flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(sync_flag, sync_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const long long L = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= total) return;
    const long long per = (total + 7) / 8;
    const int x = (int)(L / per);
    const long long seg0 = (long long)x * per;
    const int len = (int)min(per, total - seg0);
    const int kr = rank[L], k = kr >> WORK_RANK_BITS;          // (the class the rank was taken in, cell_bin_block)
    int r = kr & ((1 << WORK_RANK_BITS) - 1), O = 0;
#pragma unroll
    for (int c = 0; c < WORK_CLASSES - 1; ++c) { const int h = hdr[1 + x * 16 + c]; if (c < k) r += h; O += h; }
    // (raster_math.h: work_position -- the occupied tiles spread evenly between the empty ones, then scrambled inside windows of 64: an even
    // comb lines up with the hardware's round robin over the shader engines)
    const unsigned p = k < WORK_CLASSES - 1 ? work_position(true, r, O, len) : work_position(false, r - O, O, len);
    const int n = (int)(L / per_view), t = (int)(L - (long long)n * per_view), ty = t / tiles_x;
    work[seg0 + p] = make_int2(n, (ty << 16) | (t - ty * tiles_x));
}

__global__ __launch_bounds__(256) void cell_bin_kernel(const FaceRec *__restrict__ recs, const int *__restrict__ first_idx,
                                                       const int *__restrict__ num_faces, int N, int H, int W, int nx, int ny,
                                                       const int *__restrict__ clist, const int *__restrict__ ccount, int2 *__restrict__ cell,
                                                       int *__restrict__ pool, int pool_cap, int *__restrict__ hdr, int *__restrict__ rank) {
    __shared__ BinShared S;
    const int nb = nx * ny, n = blockIdx.x / nb, bin = blockIdx.x % nb;
    bin_cell_extents(S, H, W, (bin % nx) * COARSE, (bin / nx) * COARSE);
    __syncthreads();
    const int *lst = clist + (long long)first_idx[n] * nb + (long long)bin * num_faces[n];
    cell_bin_block<false>(recs, first_idx, N, H, W, nx, ny, lst, ccount[n * nb + bin], cell, pool, pool_cap, hdr, rank, n, bin, S);
}

// Training step: both levels of the binning of both scenes in one launch (step_kernels.h): one workgroup per (bin, view, scene) runs the
// coarse level and, where the scene's pass reads per-tile lists, goes straight on to the fine level with the bin's list still in LDS.
__global__ __launch_bounds__(256) void scene_bins_kernel(const SceneBinsArgs A) {
    __shared__ BinShared S;
    if ((int)blockIdx.z >= A.nscenes) {        // the step's texture preparation, in the shadow of the bins (SceneBinsArgs::tex)
        const int zi = (int)blockIdx.z - A.nscenes, set = zi / A.tex_z, slice = zi - set * A.tex_z;
        const long long per = (long long)gridDim.x * gridDim.y;
        const dbw_texture_set &t = A.tex.s[set];
        texture_prep_fwd_body(t.texture, t.n, t.h, t.w, t.decim, t.maps, t.sig, (long long)slice * per + (long long)blockIdx.y * gridDim.x + blockIdx.x,
                              per * A.tex_z);
        return;
    }
    // (workgroups start in grid order and the launch is more than one round of them at 49 views: the last scene -- the blocks, whose bins
    // hold the long lists -- goes first, the env scene's short workgroups fill the tail)
    const SceneBinsArgs::One &G = A.sc[A.scene0 + (A.nscenes - 1 - (int)blockIdx.z)];
    const int bin = blockIdx.x, n = blockIdx.y;
    bin_cell_extents(S, A.H, A.W, (bin % A.nx) * COARSE, (bin / A.nx) * COARSE);
    __syncthreads();
    const int cnt = coarse_bin_block(G.bbox, G.first_idx, G.num_faces, A.H, A.W, A.nx, A.ny, G.list, G.count, G.mask, n, bin, S);
    if (!G.cells) return;
    // (coarse_bin_block ends its last round with a barrier: S.ent is complete)
    cell_bin_block<true>((const FaceRec *)G.recs, G.first_idx, A.B, A.H, A.W, A.nx, A.ny, nullptr, cnt, G.cell, G.pool, G.pool_cap, G.hdr, G.rank, n, bin, S, G.dom);
}

template <int KMAX, int TW, int TH>
__global__ __launch_bounds__(TW * TH, DBW_RASTER_WAVES(KMAX)) void raster_fwd_kernel(
    const FaceRec *__restrict__ recs, const float4 *__restrict__ bbox, const int *__restrict__ first_idx,
    const int *__restrict__ num_faces, int N, int H, int W, int K, float blur,
    int persp, int clipb, long long total_blocks, CoarseBins cb, int *__restrict__ p2f, float *__restrict__ zbuf,
    float *__restrict__ bary, float *__restrict__ dists, int dbg) {
    int n, xi, yi;
    TopK<KMAX> q;
    pay4 *home;
    if (!raster_tile<KMAX, TW, TH>(recs, bbox, first_idx, num_faces, H, W, K, blur, persp, clipb, total_blocks, cb, dbg >> 8, n, xi, yi, q, home)) return;
    const bool in_img = xi < W && yi < H;
    if (!in_img) return;
    const long long o = (((long long)n * H + yi) * W + xi) * K;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K && !(dbg & 16)) {     // dbg 16: ablate the stores (tools/ablate.py)
            float pz = -1.f;
            int fi = -1;
            pay4 v{-1.f, -1.f, -1.f, -1.f};
            q.get(k, home, TW * TH, threadIdx.x, pz, fi, v);
            p2f[o + k] = fi;
            if (zbuf) zbuf[o + k] = pz;
            dists[o + k] = v.x;
            bary[(o + k) * 3 + 0] = v.y;
            bary[(o + k) * 3 + 1] = v.z;
            bary[(o + k) * 3 + 2] = v.w;
        }
    }
}

__global__ __launch_bounds__(NT) void raster_bwd_kernel(
    const float *__restrict__ fv, const int *__restrict__ p2f, const float *__restrict__ gzb,
    const float *__restrict__ gbary, const float *__restrict__ gdist, int N, int H, int W, int K, int persp,
    int clipb, long long total_blocks, float *__restrict__ gfv) {
    const long long logical = xcd_remap(blockIdx.x, total_blocks);
    if (logical < 0) return;
    const int tiles_x = (W + TILE - 1) / TILE, tiles_y = (H + TILE - 1) / TILE;
    const int n = (int)(logical / (tiles_x * tiles_y));
    const int t = (int)(logical % (tiles_x * tiles_y));
    const int ty = t / tiles_x, tx = t % tiles_x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int xi = tx * TILE + (tid & (TILE - 1)), yi = ty * TILE + (tid >> 4);
    const bool in_img = xi < W && yi < H;
    f2 p;
    p.x = pix_to_ndc(W - 1 - xi, W, H);
    p.y = pix_to_ndc(H - 1 - yi, H, W);
    const long long o = (((long long)n * H + yi) * W + xi) * K;
    for (int k = 0; k < K; ++k) {
        int f = -1;
        if (in_img) f = p2f[o + k];
        const bool active = f >= 0;
        if (__ballot(active) == 0ull) continue;
        float g[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) g[i] = 0.f;
        if (active) {
            const float *q = fv + (long long)f * 9;
            const f2 a{q[0], q[1]}, b{q[3], q[4]}, c{q[6], q[7]};
            const float z0 = q[2], z1 = q[5], z2 = q[8];
            const float gd = gdist ? gdist[o + k] : 0.f;
            const float gz = gzb ? gzb[o + k] : 0.f;
            f3 gb{0.f, 0.f, 0.f};
            if (gbary) { gb.x = gbary[(o + k) * 3]; gb.y = gbary[(o + k) * 3 + 1]; gb.z = gbary[(o + k) * 3 + 2]; }
            const f3 bary0 = bary_fwd(p, a, b, c);
            const f3 bp = persp ? persp_fwd(bary0, z0, z1, z2) : bary0;
            const f3 bc = clipb ? clip_fwd(bp) : bp;
            const bool inside = bp.x > 0.f && bp.y > 0.f && bp.z > 0.f;
            const float sign = inside ? -1.f : 1.f;
            f2 d0, d1, d2;
            point_tri_dist_bwd(p, a, b, c, sign * gd, d0, d1, d2);
            f3 gg{gb.x + gz * z0, gb.y + gz * z1, gb.z + gz * z2};
            if (clipb) gg = clip_bwd(bp, gg);
            float pz0 = 0.f, pz1 = 0.f, pz2 = 0.f;
            if (persp) gg = persp_bwd(bary0, z0, z1, z2, gg, pz0, pz1, pz2);
            f2 e0, e1, e2;
            bary_bwd(p, a, b, c, gg, e0, e1, e2);
            g[0] = e0.x + d0.x; g[1] = e0.y + d0.y; g[2] = gz * bc.x + pz0;
            g[3] = e1.x + d1.x; g[4] = e1.y + d1.y; g[5] = gz * bc.y + pz1;
            g[6] = e2.x + d2.x; g[7] = e2.y + d2.y; g[8] = gz * bc.z + pz2;
        }
        wave_agg_atomic<9>(gfv, (long long)f, active, g, lane);
    }
}

thread_local int g_raster_dbg = 0;      // (dbw_debug_set_flags: per host thread)

template <int KMAX, int TW, int TH>
int launch_fwd_t(const FaceRec *recs, const float4 *bbox, const int *first_idx, const int *num_faces,
                 int N, int H, int W, int K, float blur, int persp, int clipb, const CoarseBins &cb, int *p2f, float *zbuf,
                 float *bary, float *dists, hipStream_t s) {
    const long long total = (long long)N * ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
    DBW_REQUIRE(total < (1LL << 31) - 8, "more than 2^31 tiles in one pass");
    hipLaunchKernelGGL((raster_fwd_kernel<KMAX, TW, TH>), dim3(dbw_xcd_grid(total)), dim3(TW * TH), 0, s, recs, bbox, first_idx,
                       num_faces, N, H, W, K, blur, persp, clipb, total, cb, p2f, zbuf, bary, dists, g_raster_dbg);
    return dbw_check_launch("raster_fwd_kernel");
}

template <int KMAX>
int launch_fwd(const FaceRec *recs, const float4 *bbox, const int *first_idx, const int *num_faces,
               int N, int H, int W, int K, float blur, int persp, int clipb, const CoarseBins &cb, int *p2f, float *zbuf,
               float *bary, float *dists, hipStream_t s) {
    // hard single-layer passes rasterise few, huge faces: 16x16 tiles (fewer tiles re-scan the face list, the payload stays in
    // registers); soft K-layer passes: one wave64 per 8x8 tile (the LDS home array is KMAX * 16 B per pixel)
    if constexpr (KMAX == 1) return launch_fwd_t<KMAX, 16, 16>(recs, bbox, first_idx, num_faces, N, H, W, K, blur, persp, clipb, cb, p2f, zbuf, bary, dists, s);
    else return launch_fwd_t<KMAX, 8, 8>(recs, bbox, first_idx, num_faces, N, H, W, K, blur, persp, clipb, cb, p2f, zbuf, bary, dists, s);
}

}  // namespace

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// workspace = [boxes: F x 16 B][records: F x 128 B][shading records of the fused forward: F x 64 B][coarse-bin counts][coarse-bin lists]
extern "C" size_t dbw_rasterize_workspace_bytes(int64_t F_total) {
    const size_t F = (size_t)(F_total > 0 ? F_total : 1);
    return align256(F * sizeof(float4)) + align256(F * sizeof(FaceRec)) + align256(F * 64);
}

// ... [coarse-bin lists][cell-list header: pool cursor, 8 x 16 class cursors][cell table (N, tiles) int2][work list (N * tiles)]
//     [tile ranks (N * tiles)][dominant faces (N * tiles)][cell-list pool]     (a work-list entry is an int2)
static size_t coarse_bytes(int64_t F_total, int N, int H, int W) {
    const size_t nb = (size_t)((W + COARSE - 1) / COARSE) * ((H + COARSE - 1) / COARSE);
    return align256((size_t)(N > 0 ? N : 1) * nb * 3 * sizeof(int)) + align256((size_t)(F_total > 0 ? F_total : 1) * nb * sizeof(int));
}
static size_t cell_tiles(int H, int W) { return (size_t)((W + 7) / 8) * ((H + 7) / 8); }
static size_t cell_pool_entries(int64_t F_total, int N, int H, int W) {
    // room for DBW_CELL_POOL_PER_TILE faces per cell on average (config 2 needs 5 of them; 50 overlapping blocks at config 5 several
    // dozen); a bin that does not fit falls back to walking its coarse list, so this is a performance knob, not a limit
    const size_t want = (size_t)(N > 0 ? N : 1) * cell_tiles(H, W) * DBW_CELL_POOL_PER_TILE;
    return want < ((size_t)1 << 30) ? want : ((size_t)1 << 30);
}
extern "C" size_t dbw_rasterize_workspace_bytes_binned(int64_t F_total, int N, int H, int W) {
    const size_t n = (size_t)(N > 0 ? N : 1), t = cell_tiles(H, W);
    return dbw_rasterize_workspace_bytes(F_total) + coarse_bytes(F_total, N, H, W) + align256(CELL_HDR_INTS * sizeof(int)) +
           2 * align256(n * t * sizeof(int2)) + 2 * align256(n * t * sizeof(int)) + align256(cell_pool_entries(F_total, N, H, W) * sizeof(int));
}

const FaceRec *dbw_workspace_recs(const void *workspace, long long F_total) {
    return (const FaceRec *)((const char *)workspace + align256((size_t)(F_total > 0 ? F_total : 1) * sizeof(float4)));
}

void *dbw_workspace_shade_recs(void *workspace, long long F_total) {
    const size_t F = (size_t)(F_total > 0 ? F_total : 1);
    return (char *)workspace + align256(F * sizeof(float4)) + align256(F * sizeof(FaceRec));
}

int dbw_raster_workspace_layout(void *workspace, size_t workspace_bytes, long long F_total, long long max_faces_per_view, int N, int H, int W,
                                bool want_cells, dbw::RasterWorkspace &L) {
    L = dbw::RasterWorkspace{};
    if (F_total <= 0) return DBW_OK;
    if (F_total >= (1LL << TOPK_ID_BITS) - 1) {
        dbw_set_error("rasteriser: %lld packed faces, the per-pixel list keys hold face ids below 2^%d - 1", F_total, TOPK_ID_BITS);
        return DBW_ERR_UNSUPPORTED;
    }
    if (((uintptr_t)workspace & 127) != 0) {
        dbw_set_error("rasteriser: the workspace must be 128-byte aligned");
        return DBW_ERR_INVALID;
    }
    L.bbox = (float4 *)workspace;
    L.recs = (FaceRec *)dbw_workspace_recs(workspace, F_total);
    L.shade_recs = dbw_workspace_shade_recs(workspace, F_total);
    // (a coarse-bin entry packs the view-local face index into 20 bits: views of a million faces and more scan without bins)
    L.binned = workspace_bytes >= dbw_rasterize_workspace_bytes_binned(F_total, N, H, W) && !(g_raster_dbg & 128) && max_faces_per_view <= (1 << 20);
    L.cells = L.binned && want_cells && !(g_raster_dbg & 4096) && DBW_CELL_LISTS && (long long)N * (long long)cell_tiles(H, W) < (1LL << WORK_RANK_BITS);
    char *p = (char *)workspace + dbw_rasterize_workspace_bytes(F_total);
    L.hdr = (int *)(p + coarse_bytes(F_total, N, H, W));
    if (L.binned) {
        L.nx = (W + COARSE - 1) / COARSE; L.ny = (H + COARSE - 1) / COARSE;
        L.count = (int *)p;
        L.mask = (unsigned *)(L.count + (size_t)N * L.nx * L.ny);
        L.list = (int *)(p + align256((size_t)N * L.nx * L.ny * 3 * sizeof(int)));
        if (L.cells) {
            const size_t t = cell_tiles(H, W);
            L.cell = (int2 *)((char *)L.hdr + align256(CELL_HDR_INTS * sizeof(int)));
            L.work = (int2 *)((char *)L.cell + align256((size_t)N * t * sizeof(int2)));
            L.rank = (int *)((char *)L.work + align256((size_t)N * t * sizeof(int2)));
            L.dom = (int *)((char *)L.rank + align256((size_t)N * t * sizeof(int)));
            L.pool = (int *)((char *)L.dom + align256((size_t)N * t * sizeof(int)));
            L.pool_cap = (int)cell_pool_entries(F_total, N, H, W);
        }
    }
    return DBW_OK;
}

// Face boxes + records, then (when the workspace has room for it) the coarse bins.  boxes = workspace.
int dbw_prepare_raster(const float *face_verts, const int *first_idx, const int *num_faces, const int *neighbor, int N, long long F_total,
                       long long max_faces_per_view, int H, int W, float margin, int cull, void *workspace, size_t workspace_bytes,
                       dbw::CoarseBins &cb, hipStream_t s, bool launch, bool want_cells) {
    // launch == false: the workspace was filled by an earlier call with the same arguments (a staged render pass, or the fused set-up
    // kernels of the training step); only `cb` is rebuilt
    cb.list = nullptr; cb.count = nullptr; cb.mask = nullptr; cb.nx = cb.ny = 0;
    cb.cell = nullptr; cb.pool = nullptr; cb.work = nullptr;
    {       // pixel -> NDC constants (SURVEY A.1 NonSquarePixToNdc): IEEE single divisions, the same bits as the device's
        float rx = 2.0f, ry = 2.0f;
        if (W > H) rx = ((float)W * rx) / (float)H;
        if (H > W) ry = ((float)H * ry) / (float)W;
        cb.ndc[0] = rx; cb.ndc[1] = rx / 2.0f; cb.ndc[2] = ry; cb.ndc[3] = ry / 2.0f;
    }
    if (F_total <= 0) return DBW_OK;
    dbw::RasterWorkspace L;
    int rc = dbw_raster_workspace_layout(workspace, workspace_bytes, F_total, max_faces_per_view, N, H, W, want_cells, L);
    if (rc) return rc;
    if (launch) {
        hipLaunchKernelGGL(face_setup_kernel, dim3((unsigned)((F_total + 255) / 256)), dim3(256), 0, s, face_verts, neighbor, F_total, margin, cull,
                           L.bbox, L.recs, L.cells ? L.hdr : nullptr, L.cells ? CELL_HDR_INTS : 0);
        rc = dbw_check_launch("face_setup_kernel");
        if (rc) return rc;
    }
    if (L.binned) {
        if (launch) {
            hipLaunchKernelGGL(coarse_bin_kernel, dim3((unsigned)(N * L.nx * L.ny)), dim3(256), 0, s, (const float4 *)L.bbox, first_idx,
                               num_faces, H, W, L.nx, L.ny, L.list, L.count, L.mask);
            rc = dbw_check_launch("coarse_bin_kernel");
            if (rc) return rc;
        }
        cb.list = L.list; cb.count = L.count; cb.mask = L.mask; cb.nx = L.nx; cb.ny = L.ny;
        if (L.cells) {
            if (launch) {
                hipLaunchKernelGGL(cell_bin_kernel, dim3((unsigned)(N * L.nx * L.ny)), dim3(256), 0, s, (const FaceRec *)L.recs, first_idx,
                                   num_faces, N, H, W, L.nx, L.ny, L.list, L.count, L.cell, L.pool, L.pool_cap, L.hdr, L.rank);
                rc = dbw_check_launch("cell_bin_kernel");
                if (rc) return rc;
                const long long total = (long long)N * (long long)cell_tiles(H, W);
                hipLaunchKernelGGL(work_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, L.cell, L.rank, L.hdr, total, (W + 7) / 8,
                                   (int)cell_tiles(H, W), L.work, (unsigned *)nullptr, 0u);
                rc = dbw_check_launch("work_scatter_kernel");
                if (rc) return rc;
            }
            cb.cell = L.cell; cb.pool = L.pool; cb.work = L.work;
        }
    }
    return DBW_OK;
}

int dbw::launch_scene_bins(const SceneBinsArgs &A, hipStream_t s) {
    DBW_REQUIRE(A.B > 0 && A.H > 0 && A.W > 0 && A.nx == (A.W + COARSE - 1) / COARSE && A.ny == (A.H + COARSE - 1) / COARSE, "bad size");
    DBW_REQUIRE(A.scene0 >= 0 && A.nscenes >= 1 && A.scene0 + A.nscenes <= 2, "bad scene range");
    for (int i = A.scene0; i < A.scene0 + A.nscenes; ++i) {
        const SceneBinsArgs::One &G = A.sc[i];
        DBW_REQUIRE(G.bbox && G.recs && G.first_idx && G.num_faces && G.list && G.count && G.mask, "null pointer");
        DBW_REQUIRE(!G.cells || (G.cell && G.pool && G.hdr && G.rank && G.pool_cap > 0), "null pointer (cell lists)");
    }
    DBW_REQUIRE(A.tex_sets >= 0 && A.tex_sets <= STEP_MAX_SETS && (A.tex_sets == 0 || A.tex_z >= 1), "bad texture slices");
    for (int i = 0; i < A.tex_sets; ++i) {
        const dbw_texture_set &t = A.tex.s[i];
        DBW_REQUIRE(t.texture && t.maps && t.n > 0 && t.h > 1 && t.w > 1 && t.decim >= 1 && (t.decim == 1 || (t.sig && t.h % t.decim == 0 && t.w % t.decim == 0)),
                    "bad texture set");
    }
    hipLaunchKernelGGL(scene_bins_kernel, dim3((unsigned)(A.nx * A.ny), (unsigned)A.B, (unsigned)(A.nscenes + A.tex_sets * A.tex_z)), dim3(256), 0, s, A);
    return dbw_check_launch("scene_bins_kernel");
}

// the launch-order kernel alone (the training step's fused set-up fills cell / rank / hdr itself)
int dbw::dbw_launch_work_scatter(const dbw::RasterWorkspace &L, int N, int H, int W, hipStream_t s, unsigned *sync_flag, unsigned sync_val) {
    const long long total = (long long)N * (long long)cell_tiles(H, W);
    hipLaunchKernelGGL(work_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, L.cell, L.rank, L.hdr, total, (W + 7) / 8, (int)cell_tiles(H, W),
                       L.work, sync_flag, sync_val);
    return dbw_check_launch("work_scatter_kernel");
}

extern "C" int dbw_rasterize_fwd(const float *face_verts, const int32_t *first_idx, const int32_t *num_faces,
                                 const int32_t *neighbor, int N, int64_t F_total, int H, int W, int K,
                                 float blur_radius, int perspective_correct, int clip_barycentric_coords,
                                 int cull_backfaces, int32_t *pix_to_face, float *zbuf, float *bary, float *dists,
                                 void *workspace, size_t workspace_bytes, dbw_stream_t stream) {
    DBW_REQUIRE(face_verts && first_idx && num_faces && pix_to_face && bary && dists && workspace, "null pointer");
    DBW_REQUIRE(N >= 0 && H > 0 && W > 0 && K > 0 && F_total >= 0, "bad size");
    DBW_REQUIRE(workspace_bytes >= dbw_rasterize_workspace_bytes(F_total), "workspace too small");
    DBW_REQUIRE(blur_radius >= 0.f, "blur_radius < 0");
    if (K > DBW_MAX_FACES_PER_PIXEL) {
        dbw_set_error("dbw_rasterize_fwd: faces_per_pixel=%d > %d", K, DBW_MAX_FACES_PER_PIXEL);
        return DBW_ERR_UNSUPPORTED;
    }
    if (N == 0) return DBW_OK;
    hipStream_t s = (hipStream_t)stream;
    const float margin = (float)sqrt((double)blur_radius);
    float4 *bbox = (float4 *)workspace;
    const FaceRec *recs = dbw_workspace_recs(workspace, F_total);
    CoarseBins cb;
    int rc = dbw_prepare_raster(face_verts, first_idx, num_faces, neighbor, N, F_total, F_total, H, W, margin, cull_backfaces, workspace, workspace_bytes, cb, s, true, /*want_cells=*/K > 1);
    if (rc) return rc;
#define DBW_FWD(KM) launch_fwd<KM>(recs, bbox, first_idx, num_faces, N, H, W, K, blur_radius, \
                                   perspective_correct, clip_barycentric_coords, cb, pix_to_face, zbuf, bary, dists, s)
    if (K == 1) return DBW_FWD(1);
    if (K <= 4) return DBW_FWD(4);
    if (K <= 10) return DBW_FWD(10);
    if (K <= 16) return DBW_FWD(16);
    return DBW_FWD(25);
#undef DBW_FWD
}

extern "C" int dbw_rasterize_bwd(const float *face_verts, const int32_t *pix_to_face, const float *grad_zbuf,
                                 const float *grad_bary, const float *grad_dists, int N, int64_t F_total, int H, int W,
                                 int K, int perspective_correct, int clip_barycentric_coords, float *grad_face_verts,
                                 dbw_stream_t stream) {
    DBW_REQUIRE(face_verts && pix_to_face && grad_face_verts, "null pointer");
    DBW_REQUIRE(N >= 0 && H > 0 && W > 0 && K > 0 && F_total >= 0, "bad size");
    if (N == 0 || (!grad_zbuf && !grad_bary && !grad_dists)) return DBW_OK;
    const long long total = (long long)N * ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    hipLaunchKernelGGL(raster_bwd_kernel, dim3(dbw_xcd_grid(total)), dim3(NT), 0, (hipStream_t)stream, face_verts,
                       pix_to_face, grad_zbuf, grad_bary, grad_dists, N, H, W, K, perspective_correct,
                       clip_barycentric_coords, total, grad_face_verts);
    return dbw_check_launch("raster_bwd_kernel");
}

void dbw_set_raster_dbg(int flags) { g_raster_dbg = flags; }

#ifdef DBW_DIAG
// tools/diag only (builds with -DDBW_DIAG, tools/variants.sh): byte offsets, inside a binned workspace, of {cell-list header, cell table, work list, tile ranks, pool} and the
// pool capacity in entries -- lets a script read the per-tile face counts of a pass and try launch orders of its own
extern "C" void dbw_debug_cell_layout(int64_t F_total, int N, int H, int W, unsigned long long *out6) {
    const size_t n = (size_t)(N > 0 ? N : 1), t = cell_tiles(H, W);
    size_t o = dbw_rasterize_workspace_bytes(F_total) + coarse_bytes(F_total, N, H, W);
    out6[0] = o; o += align256(CELL_HDR_INTS * sizeof(int));
    out6[1] = o; o += align256(n * t * sizeof(int2));
    out6[2] = o; o += align256(n * t * sizeof(int2));       // (work-list entries are int2 {view, tile row << 16 | tile column})
    out6[3] = o; o += 2 * align256(n * t * sizeof(int));      // (ranks, then the dominant faces)
    out6[4] = o;
    out6[5] = cell_pool_entries(F_total, N, H, W);
}
#endif
