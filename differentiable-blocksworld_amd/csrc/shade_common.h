// Shading helpers shared by the stand-alone shade/blend kernels (shade_blend.hip) and the fused render kernels
// (render_fused.hip): fragment decoding (clipped -> original face, barycentric back-conversion, geometric + learned alpha)
// and the bilinear texture footprint (grid_sample semantics, v flip, circular u wrap, decimation shift).
#pragma once
#include "dbw_common.h"
#include "shade_math.h"

namespace dbw {

#ifndef DBW_TEX_LOG2
#define DBW_TEX_LOG2 8       // 128 / 256 / 512 / 1024 slots: uv backward 0.49 / 0.323 / 0.353 / 0.54 ms at config 2 (round 3, after the lane merge:
#endif                       // 19 instead of 26 KB of LDS per workgroup = 8 instead of 5-6 waves per SIMD; 128 slots overflow into memory atomics)
typedef LdsAgg<3, DBW_TEX_LOG2> TexAgg;      // 256 texels x (key + fp64 rgb) = 7 KB
typedef LdsAgg<1, 8> AlphaAgg;    // 256 faces x (key + fp64)       =  3 KB
typedef LdsAgg<9, 7> FaceAgg;     // 128 faces x (key + fp64 3x3)   = 9.5 KB

struct ShadeArgs {
    const int *p2f; const float *bary; const float *dists;
    const int *c2o; const int *code; const float *cw; int Fc_stride;
    const float *face_uvs; const int *face_map; const int *map_desc; const float *maps;
    const float *faces_alpha; int alpha_len;
    int N, H, W, K, F; float sigma, inv_sigma; float bg[3];     // sigma > 0: exp(-max(d, 0) / sigma); == 0: hard; < 0: sigmoid(-d / |sigma|) (clip_inside = False); inv_sigma = 1 / |sigma|
    int tiled; // fragment layout: 0 = (N,H,W,K[,3]) as PyTorch3D returns them; 1, 2 = internal 8x8-tile planar layout of the
               // fused path: [n][tile_y][tile_x][k][64 lanes] (bary: [..][k][3][64]) -> every wave access is one 256 B line pair;
               // 2 = same, with the bary planes holding (u, v, bitcast(face | map << 20)) for detach_bary passes
    int agg;   // backward: 0 = wave-aggregated global atomics, 1 = LDS hash pre-aggregation
    // texture-space binning of texel gradients (backward, agg == 0): fragments are appended as 32 B records to the bin of the
    // 32x32-texel tile their bilinear footprint starts in; dbw_texbin_reduce then accumulates every bin in LDS.  NULL = off.
    const int *bin_base;   // (M) first bin of each map
    float ndc[4];          // pixel -> NDC constants computed on the host: range and offset of x, of y (raster_math.h: ndc_axis_given)
    int *bin_cursor;       // (nbins) append cursors
    int4 *bin_records;     // (nbins, bin_cap, 2)
    int bin_cap;
    const unsigned *bin_layout;   // optional (nbins * DBW_BIN_SUBCURSORS, 2): {first record, capacity} of every sub-range; NULL = bin_cap / SUB each
    int dbg;   // ablation switches (dbw_debug_set_flags): 1 = no texel atomics, 2 = no opacity atomics, 4 = no wave aggregation
    // backward: device scalar every incoming image gradient is multiplied by (the upstream gradient of the loss node), NULL = 1
    const float *gscale;
    int geom_begin;             // backward: faces with an original index below it have constant vertices (no geometry gradient)
    // forward, decoupled composite + MSE epilogue (dbw_render_fwd_fused_mse; all NULL otherwise): instead of storing its image the
    // pass composites it over env_img, compares with target and stores d(mse_scale * sum of squares) / d(fg image), d / d(env image)
    // and the tile's sum of squares
    const float *env_img, *target;
    float mse_scale;
    float *loss_part, *g_fg, *g_env;
    // layout of the image-shaped buffers of a pass (its image, the gradient image of its backward, the env image / target / gradient
    // images of the loss epilogue): 0 = (N, C, H, W) planes as torch holds them; 1 = 8x8-tile planar [n][tile_y][tile_x][C][64], the
    // layout of the fragments -- a wave's access to one plane of its tile is then ONE 256 B line instead of eight 32 B row pieces
    int img_tiled;
    // forward with the loss epilogue, training step: write only what the two backward kernels read of the gradient images -- no alpha plane of
    // g_env (identically zero; the hard backward reads the colour planes), no g_fg for a tile without a single fragment (the soft backward
    // only loads the gradient of pixels that hold fragments): 90 of the 376 MB the epilogue wrote per step at config 2
    int lean_grads;
    // the perceptual term of the training step (dbw.py:369-371), which a network outside this path evaluates on the composite: a first run of
    // the pass stores `rec` (N, 3, H, W planes, what the network takes), a second run adds d perceptual / d rec (same layout) to the MSE's
    // gradient in front of the chain rule through the composite.  NULL: off
    float *rec_out;
    const float *grad_rec;
    // training step: the first workgroup of the backward stores sync_val to *sync_flag when it starts -- a kernel that has started says that
    // everything in front of it on its stream is complete, so the step's other streams poll this word instead of waiting for an event
    // recorded between two kernels of the critical chain (train_step.hip).  NULL: off
    unsigned *sync_flag; unsigned sync_val;
};

// The map descriptors of a pass in LDS.  A fragment's footprint starts with its map's six descriptor ints; read from memory that is
// a dependent round trip per fragment (face | map comes out of the fragment itself) in front of every table update of the backward.
// Row 0 of map_desc carries the number of rows in its 7th int (include/dbw_hip.h; 0 = not given): tables of up to MD_CACHE_MAPS rows are
// copied to LDS once per workgroup, larger or uncounted ones are read from memory as before.  Used by the binned uv backward only: the
// instantiation for decimated maps is bound by the LDS atomic unit, and the extra LDS reads cost it 0.39 -> 0.41 ms.
#ifndef DBW_MD_LDS
#define DBW_MD_LDS 1
#endif
constexpr int MD_CACHE_MAPS = 64;
// (the LDS array is passed to every call by name: a pointer member would make the compiler merge it with the memory pointer into a
// generic one -- flat loads, which wait for the vector-memory AND the LDS counters)
struct MapDescCache {
    bool in_lds;
    // all threads of the workgroup; the caller's next barrier publishes the copy.  `extra` (optional, one int per map) goes to the
    // 8th int of the rows: the first texture bin of each map for the binned backward
    __device__ __forceinline__ void load(const ShadeArgs &A, int *s_md, const int *extra, int tid, int nthreads, bool enable) {
        const int rows = (DBW_MD_LDS && enable) ? __builtin_amdgcn_readfirstlane(A.map_desc[6]) : 0;
        in_lds = rows > 0 && rows <= MD_CACHE_MAPS;
        if (in_lds)
            for (int i = tid; i < rows * 8; i += nthreads) s_md[i] = ((i & 7) == 7 && extra) ? extra[i >> 3] : A.map_desc[i];
    }
    __device__ __forceinline__ void get(const ShadeArgs &A, const int *s_md, int map, int (&d)[6]) const {
        if (in_lds) {
            const int4 a = *(const int4 *)(s_md + map * 8);
            const int2 b = *(const int2 *)(s_md + map * 8 + 4);
            d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y;
        } else {
            const int *md = A.map_desc + map * 8;
#pragma unroll
            for (int i = 0; i < 6; ++i) d[i] = md[i];
            // (consumed INSIDE this branch: the wait for these loads would otherwise sit where the two paths meet, i.e. the LDS path
            // too would drain the vector-memory counter -- every load issued ahead -- at each descriptor look-up)
            asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]));
        }
    }
    __device__ __forceinline__ int get_extra(const int *s_md, const int *extra, int map) const {
        int v;
        if (in_lds) v = s_md[map * 8 + 7];
        else {
            v = extra[map];
            asm volatile("" : "+v"(v));      // (as above; it also keeps the two loads from being merged into one flat load)
        }
        return v;
    }
};

// element (n, channel c of C, yi, xi) of an image-shaped buffer
struct ImgAddr { long long base; long long cstride; };
__device__ __forceinline__ ImgAddr img_addr(const ShadeArgs &A, int n, int yi, int xi, int C) {
    ImgAddr a;
    if (A.img_tiled) {
        const int tx = (A.W + 7) >> 3, ty = (A.H + 7) >> 3;
        const long long tile = ((long long)n * ty + (yi >> 3)) * tx + (xi >> 3);
        a.base = tile * (C * 64) + (((yi & 7) << 3) | (xi & 7));
        a.cstride = 64;
    } else {
        const long long plane = (long long)A.H * A.W;
        a.base = (long long)n * C * plane + (long long)yi * A.W + xi;
        a.cstride = plane;
    }
    return a;
}

struct Frag {
    int fc;           // clipped face id of the slot (index into face_verts_c)
    int j;            // local original face id
    int cd;           // clip code
    float w2, w3;
    float bo[3];      // barycentrics w.r.t. the original face
    float e;          // geometric alpha exp(-max(d,0)/sigma) or hard indicator
    float fa;         // learned face opacity (1 if none)
    float a;          // blend opacity e * fa
    float col[3];     // frag_layout 2 only: the texture colour the forward sampled for this fragment
    float T;          // frag_layout 2 only: transmittance in front of this fragment, as blended by the forward
    long long aidx;   // index into faces_alpha
    float d;
    float u, v;       // texture coordinates
    int map;          // row of map_desc
};

// index of a fragment's learned opacity: per face shared by the views (alpha_len == F), per face and view (N * F), or -- alpha_len < 0
// -- per texture map (one opacity per mesh of the scene: dbw.py:219 repeats each block's opacity over its faces)
__device__ __forceinline__ long long alpha_index(const ShadeArgs &A, int n, int j, int map) {
    return A.alpha_len < 0 ? (long long)map : ((A.alpha_len == A.F) ? (long long)j : (long long)n * A.F + j);
}

// ... and of its gradient: per-map opacities spread their gradient over DBW_ALPHA_SPREAD partial sums (by face), which the caller
// adds up -- every fragment of a mesh in every view otherwise lands on ONE address (measured: the fused backward of the bench
// config went from 0.37 to 1.78 ms on ten addresses)
#define DBW_ALPHA_SPREAD 64
__device__ __forceinline__ long long alpha_grad_index(const ShadeArgs &A, int n, int j, int map) {
    return A.alpha_len < 0 ? (long long)map * DBW_ALPHA_SPREAD + (j & (DBW_ALPHA_SPREAD - 1)) : alpha_index(A, n, j, map);
}

// geometric alpha from the signed distance + learned per-face opacity (renderer.py:252-260)
// FAST (fused backward only, gradients are compared at 1e-4): v_exp_f32 and a multiplication by 1/sigma
template <bool FAST = false>
__device__ __forceinline__ void frag_alpha(const ShadeArgs &A, int n, Frag &fr) {
    if (A.sigma == 0.f) fr.e = fr.d <= 0.f ? 1.f : 0.f;
    else if (A.sigma < 0.f) fr.e = FAST ? __builtin_amdgcn_rcpf(1.f + __expf(fr.d * A.inv_sigma)) : 1.f / (1.f + expf(fr.d / -A.sigma));      // clip_inside = False
    else if (FAST) fr.e = __expf(-(fr.d > 0.f ? fr.d : 0.f) * A.inv_sigma);
    else fr.e = expf(-(fr.d > 0.f ? fr.d : 0.f) / A.sigma);
    fr.fa = 1.f;
    fr.aidx = 0;
    if (A.faces_alpha) {
        fr.aidx = alpha_index(A, n, fr.j, fr.map);
        fr.fa = A.faces_alpha[fr.aidx];
    }
    fr.a = fr.e * fr.fa;
}

// decode one fragment (clipped face id fc >= 0, clipped barycentrics b, signed distance d)
template <bool FAST = false>
__device__ __forceinline__ void decode_frag(const ShadeArgs &A, int n, int fc, const float b[3], float d, Frag &fr) {
    if (A.c2o) {
        fr.j = A.c2o[fc];
        fr.cd = A.code[fc];
        fr.w2 = A.cw[(long long)fc * 2];
        fr.w3 = A.cw[(long long)fc * 2 + 1];
    } else {
        fr.j = fc - n * A.F;
        fr.cd = -1;
        fr.w2 = fr.w3 = 0.f;
    }
    convert_bary(fr.cd, fr.w2, fr.w3, b, fr.bo);
    const float *uv = A.face_uvs + (long long)fr.j * 6;
    interp_uv(fr.bo, uv, fr.u, fr.v);
    fr.map = A.face_map[fr.j];
    fr.d = d;
    frag_alpha<FAST>(A, n, fr);
}

// Addressing of fragment slot k of pixel (n, yi, xi): `s` indexes pix_to_face / dists, `b + c * bstride` the barycentric c.
// frag_layout 2: pix_to_face of a pixel's FIRST layer = clipped face id | fragment count << 26 (ids < 2^26, K <= 25)
constexpr int FRAG_COUNT_SHIFT = 26, FRAG_FACE_MASK = (1 << FRAG_COUNT_SHIFT) - 1;

struct FragAddr {
    long long s, b;
    int bstride;
};

__device__ __forceinline__ FragAddr frag_addr(const ShadeArgs &A, int n, int yi, int xi, int k) {
    FragAddr a;
    if (A.tiled) {       // 1: barycentrics, 2: (u, v, face|map) -- same addressing
        const int tx = (A.W + 7) >> 3, ty = (A.H + 7) >> 3;
        const long long tile = ((long long)n * ty + (yi >> 3)) * tx + (xi >> 3);
        const int lane = ((yi & 7) << 3) | (xi & 7);
        a.s = ((tile * A.K + k) << 6) + lane;
        a.b = (((tile * A.K + k) * (A.tiled == 2 ? 8 : 3)) << 6) + lane;      // layout 2: u, v, face|map, blend opacity, r, g, b, T
        a.bstride = 64;
    } else {
        a.s = (((long long)n * A.H + yi) * A.W + xi) * A.K + k;
        a.b = a.s * 3;
        a.bstride = 1;
    }
    return a;
}

// fetch + decode one fragment slot from memory; returns false for empty slots
template <bool FAST = false>
__device__ __forceinline__ bool load_frag(const ShadeArgs &A, int n, const FragAddr &o, Frag &fr) {
    const int fc = A.p2f[o.s];
    if (fc < 0) return false;
    fr.fc = A.tiled == 2 ? (fc & FRAG_FACE_MASK) : fc;       // layout 2: the first layer's id also carries the pixel's fragment count
    if (A.tiled == 2) {   // shading inputs, blend opacity and sampled colour were resolved by the forward pass: one hop of
                          // coalesced loads, no table gathers, no texel fetch (the footprint is only needed for the scatter)
        fr.u = A.bary[o.b];
        fr.v = A.bary[o.b + o.bstride];
        const int jm = __float_as_int(A.bary[o.b + 2 * o.bstride]);
        fr.a = A.bary[o.b + 3 * o.bstride];
        fr.col[0] = A.bary[o.b + 4 * o.bstride]; fr.col[1] = A.bary[o.b + 5 * o.bstride]; fr.col[2] = A.bary[o.b + 6 * o.bstride];
        fr.T = A.bary[o.b + 7 * o.bstride];
        fr.j = jm & 0xfffff;
        fr.map = jm >> 20;
        fr.cd = -1; fr.w2 = fr.w3 = 0.f; fr.bo[0] = fr.bo[1] = fr.bo[2] = 0.f;
        fr.d = A.dists[o.s];
        if (A.sigma == 0.f) fr.e = fr.d <= 0.f ? 1.f : 0.f;
        else if (A.sigma < 0.f) fr.e = FAST ? __builtin_amdgcn_rcpf(1.f + __expf(fr.d * A.inv_sigma)) : 1.f / (1.f + expf(fr.d / -A.sigma));  // clip_inside = False
        else if (FAST) fr.e = __expf(-(fr.d > 0.f ? fr.d : 0.f) * A.inv_sigma);
        else fr.e = expf(-(fr.d > 0.f ? fr.d : 0.f) / A.sigma);
        fr.fa = 1.f;      // not needed: a = e * fa is stored
        fr.aidx = A.faces_alpha ? alpha_index(A, n, fr.j, fr.map) : 0;
        return true;
    }
    const float b[3] = {A.bary[o.b], A.bary[o.b + o.bstride], A.bary[o.b + 2 * o.bstride]};
    decode_frag<FAST>(A, n, fc, b, A.dists[o.s], fr);
    return true;
}

// frag_layout 2: the ten raw words of one slot, requested without looking at them, so that the backward can ask for the next layer
// while it works on the current one (`ok` = the slot exists)
struct RawUV { int fc; float u, v, jm, a, c0, c1, c2, T, d; bool ok; };

__device__ __forceinline__ RawUV load_raw_uv(const ShadeArgs &A, const FragAddr &o, bool ok) {
    RawUV r;
    r.ok = ok;
    r.fc = -1; r.u = r.v = r.jm = r.a = r.c0 = r.c1 = r.c2 = r.d = 0.f; r.T = 1.f;
    if (ok) {
        r.fc = A.p2f[o.s] & FRAG_FACE_MASK;
        r.u = A.bary[o.b]; r.v = A.bary[o.b + o.bstride]; r.jm = A.bary[o.b + 2 * o.bstride]; r.a = A.bary[o.b + 3 * o.bstride];
        r.c0 = A.bary[o.b + 4 * o.bstride]; r.c1 = A.bary[o.b + 5 * o.bstride]; r.c2 = A.bary[o.b + 6 * o.bstride];
        r.T = A.bary[o.b + 7 * o.bstride];
        r.d = A.dists[o.s];
    }
    return r;
}

template <bool FAST>
__device__ __forceinline__ void frag_from_raw_uv(const ShadeArgs &A, int n, const RawUV &r, Frag &fr) {
    fr.fc = r.fc;
    fr.u = r.u; fr.v = r.v;
    const int jm = __float_as_int(r.jm);
    fr.a = r.a;
    fr.col[0] = r.c0; fr.col[1] = r.c1; fr.col[2] = r.c2;
    fr.T = r.T;
    fr.j = jm & 0xfffff;
    fr.map = jm >> 20;
    fr.cd = -1; fr.w2 = fr.w3 = 0.f; fr.bo[0] = fr.bo[1] = fr.bo[2] = 0.f;
    fr.d = r.d;
    if (A.sigma == 0.f) fr.e = fr.d <= 0.f ? 1.f : 0.f;
    else if (A.sigma < 0.f) fr.e = FAST ? __builtin_amdgcn_rcpf(1.f + __expf(fr.d * A.inv_sigma)) : 1.f / (1.f + expf(fr.d / -A.sigma));      // clip_inside = False
    else if (FAST) fr.e = __expf(-(fr.d > 0.f ? fr.d : 0.f) * A.inv_sigma);
    else fr.e = expf(-(fr.d > 0.f ? fr.d : 0.f) / A.sigma);
    fr.fa = 1.f;
    fr.aidx = A.faces_alpha ? alpha_index(A, n, fr.j, fr.map) : 0;
}

__device__ __forceinline__ void footprint(const ShadeArgs &A, const Frag &fr, Sample &s) {
    const int *md = A.map_desc + fr.map * 8;
    footprint_desc(fr.u, fr.v, md[0], md[1], md[2], md[3], md[4], md[5], s);
}

// Per-clipped-face shading record of the fused forward (built once per pass by shade_setup_kernel, render_fused.hip): everything
// decode_frag + footprint gather through the c2o -> {uv, map, opacity} -> map descriptor table chain, in ONE 64 B record, so that a
// fragment is two dependent loads (record, texels) away from its colour instead of four.
struct __attribute__((aligned(16))) ShadeRec {
    float uv[6];        // texture coordinates of the ORIGINAL face's vertices
    int j, cd;          // local original face id, clip code
    float w2, w3;       // clip interpolation weights
    int map;            // row of map_desc
    float fa;           // learned face opacity (1 if none)
    int off, hw, pads, sh;   // map descriptor: float offset, h << 16 | w, pad_left << 16 | pad_right, decimation shift
};
static_assert(sizeof(ShadeRec) == 64, "ShadeRec must be 64 B");


}  // namespace dbw
