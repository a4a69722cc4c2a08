// Fused forward of one render pass: tile rasterisation -> per-pixel top-K (registers) -> shading + layered blend, in ONE
// kernel.  The image is produced straight from the register-resident fragment lists; the fragments are also stored (once)
// because the backward pass consumes them, but they are never read back in the forward direction -- the reference path
// (PyTorch3D rasterize -> interpolate -> grid_sample -> ~12 blend kernels, renderer.py:92-94,219-273) re-reads them ~15x.
#include "raster_common.h"
#include "shade_common.h"
#include "loss_math.h"
#include "raster_bin.h"
#include "step_kernels.h"
#include "../../include/dbw_hip.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

using namespace dbw;

// implemented in raster.hip / shade_blend.hip
int dbw_prepare_raster(const float *face_verts, const int *first_idx, const int *num_faces, const int *neighbor, int N, long long F_total,
                       long long max_faces_per_view, int H, int W, float margin, int cull, void *workspace, size_t workspace_bytes,
                       dbw::CoarseBins &cb, hipStream_t s, bool launch, bool want_cells);
const dbw::FaceRec *dbw_workspace_recs(const void *workspace, long long F_total);
void *dbw_workspace_shade_recs(void *workspace, long long F_total);
int dbw_fill_shade_args(ShadeArgs &A, const int32_t *pix_to_face, const float *bary, const float *dists, const int32_t *c2o,
                        const int32_t *clip_code, const float *clip_w, int Fc_stride, const float *face_uvs,
                        const int32_t *face_map, const int32_t *map_desc, const float *maps, const float *faces_alpha,
                        int alpha_len, int N, int H, int W, int K, int F, float sigma, const float *background3);

#ifdef DBW_TILE_CLOCK
// tools-only (tools/diag/r06_spike4.py): when every workgroup of the fused forward finished, and on which tile -- {view, tile row << 16 |
// tile column, low 32 bits of the 100 MHz wall clock at its end, the same at its start} per workgroup of the last launch (the start stamp
// goes to memory at once: kept in registers across the kernel it costs the scalar registers the env layer's record loads need, and the
// backend fails on them)
namespace dbw { __device__ unsigned g_tile_clock[1 << 18][4]; }
extern "C" void dbw_debug_read_tile_clock(unsigned *out, int nblocks) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(dbw::g_tile_clock), (size_t)(nblocks < (1 << 18) ? nblocks : (1 << 18)) * 16);
}
#endif
#ifdef DBW_PROFILE_FWD
// sums the per-workgroup records into out16 (host) and optionally clears them
extern "C" void dbw_debug_read_fwd_profile(unsigned long long *out16, int reset) {
    static unsigned long long *host = nullptr;
    const size_t bytes = (size_t)dbw::FPROF_BLOCKS * 16 * sizeof(unsigned long long);
    if (!host) host = (unsigned long long *)malloc(bytes);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(dbw::g_fprof), bytes);
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    for (size_t b = 0; b < (size_t)dbw::FPROF_BLOCKS; ++b)
        for (int i = 0; i < 16; ++i) out16[i] += host[b * 16 + i];
    if (reset) { memset(host, 0, bytes); (void)hipMemcpyToSymbol(HIP_SYMBOL(dbw::g_fprof), host, bytes); }
}
#endif
// Fragments and gradient images are written once and read once by a kernel that starts half a millisecond (and a gigabyte of traffic)
// later: stored non-temporally they stream past the L2 instead of allocating lines in it and evicting the face records, shading
// records and texels the kernel keeps coming back to (measured: fused forward 0.38-0.40 -> 0.345 ms)
#ifndef DBW_NT_STORES
#define DBW_NT_STORES 1
#endif
#ifndef DBW_FWD_KEXACT
#define DBW_FWD_KEXACT 1
#endif
#ifndef DBW_FWD_FAST_EXP
#define DBW_FWD_FAST_EXP 1
#endif
#ifdef DBW_PROFILE_FWD
// the raw per-workgroup records (tools/fwd_timeline.py): out = nblocks x 16
extern "C" void dbw_debug_read_fwd_profile_raw(unsigned long long *out, int nblocks, int reset) {
    (void)hipDeviceSynchronize();
    const size_t bytes = (size_t)(nblocks < dbw::FPROF_BLOCKS ? nblocks : dbw::FPROF_BLOCKS) * 16 * sizeof(unsigned long long);
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(dbw::g_fprof), bytes);
    if (reset) {
        unsigned long long *z = (unsigned long long *)calloc((size_t)dbw::FPROF_BLOCKS * 16, sizeof(unsigned long long));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(dbw::g_fprof), z, (size_t)dbw::FPROF_BLOCKS * 16 * sizeof(unsigned long long));
        free(z);
    }
}
#endif
thread_local int g_render_variant = 0;
thread_local int g_render_dbg = 0;       // bit 0: plain IEEE divisions in the rasteriser, bit 1: no tile culling, bit 2: generic shading, bit 3: hard passes
                            // compute their (unused) distances too (dbw_debug_set_flags >> 8)
#ifdef DBW_DIAG
extern "C" void dbw_debug_set_render_variant(int v) { g_render_variant = v; }      // (tile shapes of the hard pass: tools/ builds only)
#endif
void dbw_set_render_dbg(int v) { g_render_dbg = v; }

namespace {

// One record per clipped face slot of the pass (ShadeRec, shade_common.h).  Slots beyond a view's face count get a benign record
// (1x1 map at offset 0, opacity 0), so that the branch-free shading loop may fetch through ANY record without a validity test.
__global__ void shade_setup_kernel(ShadeArgs A, const int *__restrict__ first_idx, const int *__restrict__ num_faces, long long F_total,
                                   ShadeRec *__restrict__ out) {
    const long long fc = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (fc >= F_total) return;
    ShadeRec r;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.uv[i] = 0.f;
    r.j = 0; r.cd = -1; r.w2 = r.w3 = 0.f; r.map = 0; r.fa = 0.f; r.off = 0; r.hw = (1 << 16) | 1; r.pads = 0; r.sh = 0;
    const int stride = A.c2o ? A.Fc_stride : A.F;
    const int n = (int)(fc / stride);
    const long long local = fc - first_idx[n];
    if (n < A.N && local >= 0 && local < num_faces[n]) {
        if (A.c2o) { r.j = A.c2o[fc]; r.cd = A.code[fc]; r.w2 = A.cw[fc * 2]; r.w3 = A.cw[fc * 2 + 1]; }
        else r.j = (int)(fc - (long long)n * A.F);
        const float *uv = A.face_uvs + (long long)r.j * 6;
#pragma unroll
        for (int i = 0; i < 6; ++i) r.uv[i] = uv[i];
        r.map = A.face_map[r.j];
        r.fa = A.faces_alpha ? A.faces_alpha[alpha_index(A, n, r.j, r.map)] : 1.f;
        const int *md = A.map_desc + r.map * 8;
        r.off = md[0]; r.hw = (md[1] << 16) | md[2]; r.pads = (md[3] << 16) | md[4]; r.sh = md[5];
    }
    out[fc] = r;
}

// ---- shading + blend + fragment stores: generic form (any tile shape, any fragment layout) -------------------------------------------
template <class T>
__device__ __forceinline__ void st_stream(T *p, T v) {
#if DBW_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

template <int KMAX, int NT>
__device__ __forceinline__ void shade_generic(const ShadeArgs &A, const TopK<KMAX> &q, const pay4 *home, int n, int xi, int yi,
                                              int *__restrict__ p2f, float *__restrict__ bary, float *__restrict__ dists,
                                              float *__restrict__ image) {
    BlendFront bl;
    blend_front_init(bl);
    int cnt = 0;                             // fragments of this pixel (the list is filled front to back)
#pragma unroll
    for (int k = 0; k < KMAX; ++k) cnt += (k < A.K && q.valid(k)) ? 1 : 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < A.K) {
            float pzk = -1.f;
            int fik = -1;
            pay4 v{-1.f, -1.f, -1.f, -1.f};
            const bool valid = q.get(k, home, NT, threadIdx.x, pzk, fik, v);
            const FragAddr o = frag_addr(A, n, yi, xi, k);
            // layout 2: the first layer's id carries the fragment count, so that the backward knows how deep to go from one load
            // (and which slots exist at all: the -1 of an empty slot below the first layer is not even stored)
            if (valid || k == 0 || A.tiled != 2)
                st_stream(p2f + o.s, valid ? ((A.tiled == 2 && k == 0) ? (fik | (cnt << FRAG_COUNT_SHIFT)) : fik) : -1);
            // internal layouts: empty slots carry only the -1 face id (the backward never reads the rest; a wave whose 64
            // pixels are all empty at this depth issues no store at all); the PyTorch3D-shaped layout 0 is filled with -1
            if (valid || A.tiled == 0) {
                if (A.tiled != 3) st_stream(dists + o.s, v.x);           // (layout 3: a kept pixel of a hard pass lies inside its face, that is all)
                if (A.tiled != 2 && A.tiled != 3) {
                    st_stream(bary + o.b, v.y);
                    st_stream(bary + o.b + o.bstride, v.z);
                    st_stream(bary + o.b + 2 * o.bstride, v.w);
                }
            }
            if (valid) {
                Frag fr;
                const float bc[3] = {v.y, v.z, v.w};
                decode_frag(A, n, fik, bc, v.x, fr);
                if (A.tiled == 2 || A.tiled == 3) {       // hand the resolved shading inputs to the backward pass
                    st_stream(bary + o.b, fr.u);
                    st_stream(bary + o.b + o.bstride, fr.v);
                    st_stream(bary + o.b + 2 * o.bstride, __int_as_float(fr.j | (fr.map << 20)));
                }
                const float a = fr.e * fr.fa;
                float c[3] = {0.f, 0.f, 0.f};
                const float T_front = bl.T;
                if (a != 0.f) {
                    Sample s;
                    footprint(A, fr, s);
                    fetch(A.maps, s, c);
                    blend_front_step(bl, a, c);
                }
                if (A.tiled == 2) {       // ... together with the blend opacity, the sampled colour (0 where the opacity is 0) and
                                          // the transmittance in front of the fragment
                    bary[o.b + 7 * o.bstride] = T_front;
                    bary[o.b + 3 * o.bstride] = a;
                    bary[o.b + 4 * o.bstride] = c[0]; bary[o.b + 5 * o.bstride] = c[1]; bary[o.b + 6 * o.bstride] = c[2];
                }
            }
        }
    }
    const ImgAddr ia = img_addr(A, n, yi, xi, 4);
    float *out = image + ia.base;
    float px[4];
    blend_front_finish(bl, A.bg, px);
    out[0] = px[0]; out[ia.cstride] = px[1]; out[2 * ia.cstride] = px[2]; out[3 * ia.cstride] = px[3];
}

// ---- the same for the training path's soft pass: 8x8 tile = one wave, uv-fragments (layout 2) --------------------------------------------
// Same arithmetic, organised for throughput: the tile index is wave-uniform, so every fragment plane is addressed as a scalar base +
// a constant + lane * 4; each layer's inputs come from the payload home (LDS) and ONE ShadeRec gather, requested a layer ahead; the
// per-layer arithmetic is branch-free (texels are fetched through the record of an empty slot too -- slot validity only masks the
// result and the stores) and the unrolled layer loop ends at the deepest layer any pixel of the wave holds.
struct UvSlot { bool valid; int fik; pay4 v; ShadeRec sr; };

template <int KMAX>
__device__ __forceinline__ UvSlot uv_slot(const TopK<KMAX, true> &q, const pay4 *home, const ShadeRec *__restrict__ srec, int k, bool in_img) {
    UvSlot s;
    float pz = 0.f;
    s.fik = 0;
    s.v = pay4{0.f, 0.f, 0.f, 0.f};
    s.valid = q.get(k, home, 64, threadIdx.x, pz, s.fik, s.v) && in_img;
    if (!s.valid) s.fik = 0;
    s.sr = srec[s.fik];
    return s;
}

// ---- the env layer folded into the soft pass (training step) ---------------------------------------------------------------------------
// The decoupled render (dbw.py:213-223) draws sky + ground in a pass of their own -- hard, one face per pixel -- whose image the fg pass
// composites over.  As a kernel of its own that pass sat on the step's critical path in front of the fg pass (0.16 ms of 1.03 at config 2)
// and its image made a round trip through memory.  Folded: every 8x8 tile of the fg pass first rasterises ITS pixel of the env scene from
// the env scene's per-tile face list (2-3 large faces: the same eval_pair / TopK<1> / sibling rule the env pass runs, so the same face, the
// same barycentrics), shades it through the face's ShadeRec and keeps the colour in three registers for the composite; what the env
// BACKWARD needs -- the hard uv-fragment (frag_layout 3: clipped face, u, v, face | map) -- is stored exactly as the env pass stores it.
// a shading record through scalar loads (wave-uniform address): one s_load_dwordx16 instead of four vector loads per lane
__device__ __forceinline__ ShadeRec load_srec_uniform(const ShadeRec *__restrict__ sp) {
    union { v16f v; ShadeRec r; } u;
    u.v = *(const v16f *)sp;
    return u.r;
}

struct EnvFold {
    const FaceRec *recs;            // nullptr: not folded (the epilogue reads env_img)
    const int *first_idx;
    const int2 *cell; const int *pool;                  // per-tile face lists of the env scene (cell_bin_block)
    const int *dom;                                     // ... and the face that is in front of all others of a tile's list, or -1
    const int *clist, *ccount; int nx, ny;              // its coarse bins: what a tile whose list did not fit the pool walks
    const int *num_faces;
    const ShadeRec *srec;
    const float *maps;
    float bg[3];
    int *p2f; float *uvj;                               // hard uv-fragments out: [tile][64], [tile][3][64]
    int persp, dbg;
    float ndc[4];                                       // CoarseBins::ndc of the pass
};

__device__ __forceinline__ void env_fold_pixel(const EnvFold &E, int H, int W, int n_, int xi, int yi, bool in_img, float (&rgb)[3]) {
    const int lane = threadIdx.x;
    const int n = __builtin_amdgcn_readfirstlane(n_);          // (the tile's view: wave-uniform, and the record loads below need to know)
    const int tiles_x = (W + 7) >> 3, tiles_y = (H + 7) >> 3;
    const int L = __builtin_amdgcn_readfirstlane((n * tiles_y + (yi >> 3)) * tiles_x + (xi >> 3));
    const NdcAxis ax = ndc_axis_given(W, E.ndc[0], E.ndc[1]), ay = ndc_axis_given(H, E.ndc[2], E.ndc[3]);
    f2 p;
    p.x = pix_to_ndc_fast(W - 1 - xi, ax);
    p.y = pix_to_ndc_fast(H - 1 - yi, ay);
    const bool fastdiv = DBW_RASTER_FASTDIV && !(E.dbg & 1);
    const int fb = __builtin_amdgcn_readfirstlane(E.first_idx[n]);
    // Most tiles of the env layer lie INSIDE one large face that is in front of everything else on their list -- the ground in front of
    // the sky dome, a sky face alone: the binning has looked (raster_bin.h: `dom`, the listed face whose farthest vertex is nearer than
    // every other listed face's nearest, no half of a split quad, inside the guarded range of the shared-reciprocal divisions).  Uniform
    // fast path: ONLY that face is evaluated -- the same eval_pair, the same barycentrics, bit for bit -- and if every pixel of the tile
    // passes its box and inside tests the tile is done: no list, no top-1 list, no evaluation of the faces behind, shading record through
    // scalar loads, no validity masks.  Anything else -- a pixel outside the face, overlapping depth ranges, an operand outside the guarded
    // range -- takes the general path below.
    const int jb = __builtin_amdgcn_readfirstlane(E.dom[L]);
    if (jb >= 0 && fastdiv && !(E.dbg & 4096)) {          // (dbw_debug_set_flags 1 << 20: the general path everywhere)
        const FaceRec r = load_rec_nowait(E.recs + fb + jb);          // (uniform address: scalar loads)
        float pz1 = 0.f, sd1 = 0.f;
        f3 bc1{0.f, 0.f, 0.f};
        bool unsafe = false;
        const bool inbox = in_img && !(p.x < r.xlo || p.x > r.xhi || p.y < r.ylo || p.y > r.yhi);
        const bool keep = inbox && eval_pair<true>(r, p, 0.f, E.persp, 1, pz1, sd1, bc1, unsafe, true);
        if (__ballot(in_img && (!keep || unsafe)) == 0ull) {
            const ShadeRec sr = load_srec_uniform(E.srec + fb + jb);
            const float bc[3] = {bc1.x, bc1.y, bc1.z};
            float bo[3], u, vv;
            convert_bary(sr.cd, sr.w2, sr.w3, bc, bo);
            interp_uv(bo, sr.uv, u, vv);
            SampleFwd s;
            footprint_fwd(u, vv, sr.off, sr.hw >> 16, sr.hw & 0xffff, sr.pads >> 16, sr.pads & 0xffff, sr.sh, s);
            fetch_fwd(E.maps, s, rgb);
            if (in_img) {
                const long long o = ((long long)L << 6) + lane;
                float *bp = E.uvj + ((long long)L * 3 << 6) + lane;
#if DBW_NT_STORES
                __builtin_nontemporal_store(fb + jb, E.p2f + o);
                __builtin_nontemporal_store(u, bp); __builtin_nontemporal_store(vv, bp + 64);
                __builtin_nontemporal_store(__int_as_float(sr.j | (sr.map << 20)), bp + 128);
#else
                E.p2f[o] = fb + jb;
                bp[0] = u; bp[64] = vv; bp[128] = __int_as_float(sr.j | (sr.map << 20));
#endif
            }
            return;
        }
    }
    TopK<1, false> q;
    q.init();
    const int2 c_ = E.cell[L];
    const int2 c = make_int2(__builtin_amdgcn_readfirstlane(c_.x), __builtin_amdgcn_readfirstlane(c_.y));
    // the tile's own list; or, where the bin's lists did not fit the pool (count < 0), the bin's coarse list: its entries in order, those
    // whose cell range covers this tile.  ONE evaluation site for both (everything wave-uniform, and said so: the record loads inside
    // want scalar addresses under uniform control flow)
    const bool walk = c.y < 0;
    int total = c.y > 0 ? c.y : 0, cx = 0, cy = 0;
    const int *__restrict__ lst = E.pool + c.x;
    if (walk) {
        const int x0 = __builtin_amdgcn_readfirstlane(xi & ~7), y0 = __builtin_amdgcn_readfirstlane(yi & ~7);
        const int nb = E.nx * E.ny, bin = (y0 / COARSE) * E.nx + (x0 / COARSE);
        total = __builtin_amdgcn_readfirstlane(E.ccount[n * nb + bin]);
        lst = E.clist + (long long)fb * nb + (long long)bin * __builtin_amdgcn_readfirstlane(E.num_faces[n]);
        cx = (x0 & (COARSE - 1)) >> 3; cy = (y0 & (COARSE - 1)) >> 3;
    }
#pragma unroll 1
    for (int cb0 = 0; cb0 < total; cb0 += DBW_WAVE) {
        const bool have = cb0 + lane < total;
        const int e = have ? lst[cb0 + lane] : 0;
        const bool hit = have && (!walk || !(cx < ((e >> 20) & 7) || cx > ((e >> 23) & 7) || cy < ((e >> 26) & 7) || cy > ((e >> 29) & 7)));
        const unsigned long long m = __ballot(hit);
        eval_staged_chunk<1, false>(E.recs, fb, e & 0xfffff, min(DBW_WAVE, total - cb0), in_img, p, 1, 0.f, E.persp, 1, fastdiv, true, q, nullptr, DBW_WAVE,
                                    lane, false, false, m);
    }
    float pz = 0.f;
    int fi = 0;
    pay4 v{0.f, 0.f, 0.f, 0.f};
    const bool valid = q.get(0, nullptr, DBW_WAVE, lane, pz, fi, v) && in_img;
    if (!valid) fi = 0;
    const ShadeRec sr = E.srec[fi];
    const float bc[3] = {v.y, v.z, v.w};
    float bo[3], u, vv;
    convert_bary(sr.cd, sr.w2, sr.w3, bc, bo);
    interp_uv(bo, sr.uv, u, vv);
    SampleFwd s;
    footprint_fwd(u, vv, sr.off, sr.hw >> 16, sr.hw & 0xffff, sr.pads >> 16, sr.pads & 0xffff, sr.sh, s);
    float col[3];
    fetch_fwd(E.maps, s, col);
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb[k] = valid ? col[k] : E.bg[k];
    if (in_img) {
        const long long o = ((long long)L << 6) + lane;
#if DBW_NT_STORES
        __builtin_nontemporal_store(valid ? fi : -1, E.p2f + o);
#else
        E.p2f[o] = valid ? fi : -1;
#endif
        if (valid) {
            float *bp = E.uvj + ((long long)L * 3 << 6) + lane;
#if DBW_NT_STORES
            __builtin_nontemporal_store(u, bp); __builtin_nontemporal_store(vv, bp + 64);
            __builtin_nontemporal_store(__int_as_float(sr.j | (sr.map << 20)), bp + 128);
#else
            bp[0] = u; bp[64] = vv; bp[128] = __int_as_float(sr.j | (sr.map << 20));
#endif
        }
    }
}

// the pixel's blended colour -> the image, or (training) the composite + MSE partials and the two image gradients
// (env_rgb: the env layer's colour of this pixel when the env pass is folded into this one (`fold`), else read from A.env_img; three values
// and a flag instead of a nullable pointer: selecting between a local array and nullptr at run time kept the array in scratch memory --
// 44 B per lane, a scratch store behind the env layer and a scratch load in the epilogue of every tile)
template <bool TILED>
__device__ __forceinline__ void uv8_epilogue_t(const ShadeArgs &A, int n, int xi, int yi, bool in_img, int tile, int lane, const float (&px)[4],
                                               float *__restrict__ image, const float (&env_rgb)[3], bool fold, bool no_fragments) {
    // image-shaped buffers of this pixel: in the 8x8-tile planar layout (TILED: the training step) every plane of the tile is a wave-uniform
    // base + the lane -- scalar address arithmetic, one store instruction per plane with the plane as its immediate offset; otherwise
    // (N, C, H, W)
    const long long plane = (long long)A.H * A.W, pix = (long long)yi * A.W + xi;
    const long long cs = TILED ? 64 : plane;
    const long long o4 = TILED ? (long long)tile * 256 : (long long)n * 4 * plane, o3 = TILED ? (long long)tile * 192 : (long long)n * 3 * plane;      // (wave-uniform)
    const unsigned ul = (unsigned)lane;
    const long long ol = TILED ? (long long)ul : pix;
    const float f0 = px[0], f1 = px[1], f2 = px[2], m = px[3];
    if (A.target) {
        // decoupled composite + MSE on registers (dbw.py:223,366-367): rec = fg_rgb * mask + (1 - mask) * env_rgb (the fg colour is
        // premultiplied AND multiplied by the mask again); the loss gradient is local to the pixel, so the pass hands
        // d loss / d fg and d loss / d env straight to the two backward passes and never stores its image
        float sq = 0.f;
        if (in_img) {
            const float *tg = A.target + o3 + ol;
            float ec3[3];
            if (fold) { ec3[0] = env_rgb[0]; ec3[1] = env_rgb[1]; ec3[2] = env_rgb[2]; }
            else { const float *ev = A.env_img + o4 + ol; ec3[0] = ev[0]; ec3[1] = ev[cs]; ec3[2] = ev[2 * cs]; }
            const float fc3[3] = {f0, f1, f2}, t3[3] = {tg[0], tg[cs], tg[2 * cs]};
            float rec3[3], gf3[3], ge3[3], gmask, gp3[3] = {0.f, 0.f, 0.f};
            // rec_out / grad_rec (the perceptual term's two phases): always (N, 3, H, W)
            if (A.grad_rec) { const long long po = (long long)n * 3 * plane + pix; gp3[0] = A.grad_rec[po]; gp3[1] = A.grad_rec[po + plane]; gp3[2] = A.grad_rec[po + 2 * plane]; }
            sq = composite_mse_pixel(fc3, m, ec3, t3, true, 2.f * A.mse_scale, rec3, gf3, ge3, gmask, gp3, A.grad_rec != nullptr);      // loss_math.h
            if (A.rec_out) { const long long po = (long long)n * 3 * plane + pix; A.rec_out[po] = rec3[0]; A.rec_out[po + plane] = rec3[1]; A.rec_out[po + 2 * plane] = rec3[2]; }
            float *gf = A.g_fg + o4 + ol, *ge = A.g_env + o4 + ol;
#if DBW_NT_STORES
            if (!(A.lean_grads && no_fragments)) {
                __builtin_nontemporal_store(gf3[0], gf); __builtin_nontemporal_store(gf3[1], gf + cs); __builtin_nontemporal_store(gf3[2], gf + 2 * cs);
                __builtin_nontemporal_store(gmask, gf + 3 * cs);
            }
            __builtin_nontemporal_store(ge3[0], ge); __builtin_nontemporal_store(ge3[1], ge + cs); __builtin_nontemporal_store(ge3[2], ge + 2 * cs);
            if (!A.lean_grads) __builtin_nontemporal_store(0.f, ge + 3 * cs);
#else
            gf[0] = gf3[0]; gf[cs] = gf3[1]; gf[2 * cs] = gf3[2];
            gf[3 * cs] = gmask;
            ge[0] = ge3[0]; ge[cs] = ge3[1]; ge[2 * cs] = ge3[2]; ge[3 * cs] = 0.f;
#endif
        }
        const float tot = wave_sum_dpp(sq);
        if (lane == 0) A.loss_part[tile] = tot;
    } else if (in_img) {
        float *out = image + o4 + ol;
        out[0] = f0;
        out[cs] = f1;
        out[2 * cs] = f2;
        out[3 * cs] = m;
    }
}
__device__ __forceinline__ void uv8_epilogue(const ShadeArgs &A, int n, int xi, int yi, bool in_img, int tile, int lane, const float (&px)[4],
                                             float *__restrict__ image, const float (&env_rgb)[3], bool fold, bool no_fragments = false) {
    if (A.img_tiled) uv8_epilogue_t<true>(A, n, xi, yi, in_img, tile, lane, px, image, env_rgb, fold, no_fragments);
    else uv8_epilogue_t<false>(A, n, xi, yi, in_img, tile, lane, px, image, env_rgb, fold, no_fragments);
}

// a tile no face reaches (cell list of length 0): every pixel is the background; the fragment record is the count 0
template <int KMAX>
__device__ __forceinline__ void shade_uv8_empty(const ShadeArgs &A, int K, int n, int xi, int yi, int *__restrict__ p2f, float *__restrict__ image,
                                                const float (&env_rgb)[3], bool fold) {
    const int lane = threadIdx.x;
    const bool in_img = xi < A.W && yi < A.H;
    const int tiles_x = (A.W + 7) >> 3, tiles_y = (A.H + 7) >> 3;
    const int tile = __builtin_amdgcn_readfirstlane((n * tiles_y + (yi >> 3)) * tiles_x + (xi >> 3));
    if (in_img) p2f[(((long long)tile * K) << 6) + lane] = -1;
    BlendFront bl;
    blend_front_init(bl);
    float px[4];
    blend_front_finish(bl, A.bg, px);
    uv8_epilogue(A, n, xi, yi, in_img, tile, lane, px, image, env_rgb, fold, true);
}

template <int KMAX>
__device__ __forceinline__ void shade_uv8(const ShadeArgs &A, int K, const ShadeRec *__restrict__ srec, const TopK<KMAX, true> &q, const pay4 *home, int n,
                                          int xi, int yi, int *__restrict__ p2f, float *__restrict__ bary, float *__restrict__ dists,
                                          float *__restrict__ image, int dbg, const float (&env_rgb)[3], bool fold) {
    // dbg (tools/diag ablations, dbw_debug_set_flags): 32 = no fragment stores (flags 8192), 64 = no layer loop at all (16384)
    const int lane = threadIdx.x;            // == ((yi & 7) << 3) | (xi & 7): the fragment lane of the 8x8-tile planar layout
    const bool in_img = xi < A.W && yi < A.H;
    const int tiles_x = (A.W + 7) >> 3, tiles_y = (A.H + 7) >> 3;
    const int tile = __builtin_amdgcn_readfirstlane((n * tiles_y + (yi >> 3)) * tiles_x + (xi >> 3));
    const long long tb = ((long long)tile * K) << 6;
    int *__restrict__ p2f_t = p2f + tb;
    float *__restrict__ dists_t = dists + tb;
    float *__restrict__ bary_t = bary + tb * 8;
#if DBW_TOPK_ORDERED
    const int cnt = in_img ? q.cnt : 0;              // (insert_ordered keeps the fill of the list)
#else
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) cnt += (in_img && k < K && q.valid(k)) ? 1 : 0;
#endif
    if (in_img && cnt == 0) p2f_t[lane] = -1;         // an empty pixel still tells the backward its fragment count (0)
    BlendFront bl;
    blend_front_init(bl);
    bool more = __ballot(cnt > 0) != 0ull && !(dbg & 64);       // (a tile without fragments goes straight to the epilogue)
    UvSlot cur;
    cur.valid = false; cur.fik = 0; cur.v = pay4{0.f, 0.f, 0.f, 0.f};
    if (more) cur = uv_slot(q, home, srec, 0, in_img);
    UvSlot nxt = cur;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        more = more && k < K && __ballot(cnt > k) != 0ull;               // wave-uniform: lists are filled front to back
        if (!more) break;          // (break, not continue: nothing of the pipeline's state flows back into the skipped iterations)
        if (k + 1 < KMAX) nxt = uv_slot(q, home, srec, k + 1, in_img);
        const ShadeRec &sr = cur.sr;
        const float bc[3] = {cur.v.y, cur.v.z, cur.v.w};
        float bo[3];
        convert_bary(sr.cd, sr.w2, sr.w3, bc, bo);
        float u, v;
        interp_uv(bo, sr.uv, u, v);
        const float d = cur.v.x;
        float e;
        if (A.sigma == 0.f) e = d <= 0.f ? 1.f : 0.f;
#if DBW_FWD_FAST_EXP
        // v_exp_f32 on d * (1 / sigma): |d / sigma| <= 9.3 inside the blur radius, so the two roundings of the argument move the
        // opacity by < 1e-6 relative -- far inside the 1e-4 bar of the rendered colours; libm's expf and an IEEE division cost ~25
        // instructions per layer here
        else e = __expf(-(d > 0.f ? d : 0.f) * A.inv_sigma);
#else
        else e = expf(-(d > 0.f ? d : 0.f) / A.sigma);
#endif
        const float a = cur.valid ? e * sr.fa : 0.f;
        SampleFwd s;
        footprint_fwd(u, v, sr.off, sr.hw >> 16, sr.hw & 0xffff, sr.pads >> 16, sr.pads & 0xffff, sr.sh, s);
        float c[3];
        fetch_fwd(A.maps, s, c);
        if (!(a != 0.f)) c[0] = c[1] = c[2] = 0.f;
        const float T = bl.T;                  // transmittance in front of this layer
        blend_front_step(bl, a, c);
        if (cur.valid && !(dbg & 32)) {
            const int o = (k << 6) + lane;
#if DBW_NT_STORES
            __builtin_nontemporal_store(k == 0 ? (cur.fik | (cnt << FRAG_COUNT_SHIFT)) : cur.fik, p2f_t + o);
            __builtin_nontemporal_store(d, dists_t + o);
            float *bp = bary_t + (k << 9) + lane;                          // 8 planes of 64 lanes per layer
            __builtin_nontemporal_store(u, bp);
            __builtin_nontemporal_store(v, bp + 64);
            __builtin_nontemporal_store(__int_as_float(sr.j | (sr.map << 20)), bp + 128);
            __builtin_nontemporal_store(a, bp + 192);
            __builtin_nontemporal_store(c[0], bp + 256); __builtin_nontemporal_store(c[1], bp + 320); __builtin_nontemporal_store(c[2], bp + 384);
            __builtin_nontemporal_store(T, bp + 448);
#else
            p2f_t[o] = k == 0 ? (cur.fik | (cnt << FRAG_COUNT_SHIFT)) : cur.fik;
            dists_t[o] = d;
            float *bp = bary_t + (k << 9) + lane;                          // 8 planes of 64 lanes per layer
            bp[0] = u;
            bp[64] = v;
            bp[128] = __int_as_float(sr.j | (sr.map << 20));
            bp[192] = a;
            bp[256] = c[0]; bp[320] = c[1]; bp[384] = c[2];
            bp[448] = T;
#endif
        }
        cur = nxt;
    }
    float px[4];
    blend_front_finish(bl, A.bg, px);
    uv8_epilogue(A, n, xi, yi, in_img, tile, lane, px, image, env_rgb, fold);
}

// UV: the specialised shading of uv-fragments on 8x8 tiles (shade_uv8) with 12 B payloads; otherwise the generic form
#define DBW_RENDER_WAVES(KMAX, UV) ((UV) ? ((KMAX) <= 4 ? 6 : (KMAX) <= 10 ? 5 : (KMAX) <= 16 ? 3 : 2) : DBW_RASTER_WAVES(KMAX))
// KEX: faces_per_pixel == KMAX (every shipped configuration: 4, 10, 16, 25) -- the list length is then a compile-time constant: no run-time
// selection of the list's last entry per inserted face, no `k < K` tests in the unrolled loops
template <int KMAX, int TW, int TH, int GROUP, bool UV, bool KEX = false>
__global__ __launch_bounds__(TW * TH, DBW_RENDER_WAVES(KMAX, UV)) void render_fwd_kernel(const FaceRec *__restrict__ recs, const float4 *__restrict__ bbox,
                                                             const int *__restrict__ first_idx, const int *__restrict__ num_faces,
                                                             float blur, int persp, int dbg,
                                                             long long total_blocks, ShadeArgs A, CoarseBins cb, const ShadeRec *__restrict__ srec,
                                                             int *__restrict__ p2f, float *__restrict__ bary, float *__restrict__ dists,
                                                             float *__restrict__ image, const EnvFold E) {
    static_assert(!UV || (TW == 8 && TH == 8), "shade_uv8 needs one wave per 8x8 tile");
    int n, xi, yi;
    TopK<KMAX, UV> q;
    pay4 *home;
    FPROF_T(t_k0);
    FPROF_ADD(13, wall_clock64());            // (100 MHz, common to the XCDs: the wave's place on the kernel's time line)
#ifdef DBW_TILE_CLOCK
    if (KMAX > 1 && blockIdx.x < (1u << 18)) g_tile_clock[blockIdx.x][3] = (unsigned)wall_clock64();       // (start stamp: to memory at once, by every lane -- no divergent branch in front of the record loads)
#endif
    bool empty = false;
    float env_rgb[3] = {0.f, 0.f, 0.f};
#if defined(DBW_PROFILE_FWD) || defined(DBW_TILE_CLOCK)
    constexpr bool fold = false;        // (the cycle-accounting build measures the pass without the folded env layer: with the counters'
    (void)E;                            // extra control flow around the record loads the backend fails on the second evaluation site)
#else
    const bool fold = UV && E.recs != nullptr;
#endif
    auto env_layer = [&](int n_, int xi_, int yi_, bool in_img_) {
        if constexpr (UV) { if (fold) env_fold_pixel(E, A.H, A.W, n_, xi_, yi_, in_img_, env_rgb); }
    };
    const int K = KEX ? KMAX : A.K;
    if (!raster_tile<KMAX, TW, TH, GROUP, UV>(recs, bbox, first_idx, num_faces, A.H, A.W, K, blur, persp, 1, total_blocks, cb,
                                              (dbg & (3 | 128)) | ((KMAX == 1 && A.tiled != 0 && !(dbg & 8)) ? 8 : 0), n, xi, yi, q, home,
                                              UV ? &empty : nullptr, env_layer)) return;
    FPROF_T(t_k1);
    if constexpr (UV) {
        if (empty) shade_uv8_empty<KMAX>(A, K, n, xi, yi, p2f, image, env_rgb, fold);
        else shade_uv8<KMAX>(A, K, srec, q, home, n, xi, yi, p2f, bary, dists, image, dbg, env_rgb, fold);
    }
    else {
        if (xi >= A.W || yi >= A.H) return;
        shade_generic<KMAX, TW * TH>(A, q, home, n, xi, yi, p2f, bary, dists, image);
    }
    FPROF_T(t_k2);
    FPROF_ADD(2, t_k2 - t_k1);
    FPROF_ADD(3, t_k2 - t_k0);
    FPROF_ADD(14, wall_clock64());
#ifdef DBW_TILE_CLOCK
    if (KMAX > 1 && threadIdx.x == 0 && blockIdx.x < (1u << 18)) {
        unsigned *o = g_tile_clock[blockIdx.x];
        // (HW_REG_HW_ID of gfx9: wave [3:0], SIMD [5:4], pipe [7:6], CU [11:8], shader array [12], shader engine [15:13])
        o[0] = (unsigned)n | ((__builtin_amdgcn_s_getreg((16 - 1) << 11 | 4) & 0xffffu) << 16); o[1] = ((unsigned)(yi >> 3) << 16) | (unsigned)(xi >> 3); o[2] = (unsigned)wall_clock64();
    }
#endif
}

template <int KMAX, int TW, int TH, int GROUP, bool UV, bool KEX = false>
int launch_v(const FaceRec *recs, const float4 *bbox, const int *first_idx, const int *num_faces, float blur,
             int persp, ShadeArgs &A, const CoarseBins &cb, const ShadeRec *srec, int *p2f, float *bary, float *dists, float *image, const EnvFold &E,
             hipStream_t s) {
    const long long total = (long long)A.N * ((A.W + TW - 1) / TW) * ((A.H + TH - 1) / TH);
    DBW_REQUIRE(total < (1LL << 31) - 8, "more than 2^31 tiles in one pass");
    hipLaunchKernelGGL((render_fwd_kernel<KMAX, TW, TH, GROUP, UV, KEX>), dim3(dbw_xcd_grid(total)), dim3(TW * TH), 0, s, recs, bbox, first_idx,
                       num_faces, blur, persp, g_render_dbg, total, A, cb, srec, p2f, bary, dists, image, E);
    return dbw_check_launch("render_fwd_kernel");
}

template <int KMAX>
int launch(const FaceRec *recs, const float4 *bbox, const int *first_idx, const int *num_faces, float blur,
           int persp, ShadeArgs &A, const CoarseBins &cb, const ShadeRec *srec, int *p2f, float *bary, float *dists, float *image, const EnvFold &E,
           hipStream_t s) {
#define DBW_V(TW, TH, G, UV) launch_v<KMAX, TW, TH, G, UV>(recs, bbox, first_idx, num_faces, blur, persp, A, cb, srec, p2f, bary, dists, image, E, s)
    if constexpr (KMAX == 1) {                                 // hard K=1 pass: large faces (sky dome, ground); the single payload stays in registers
        if (g_render_variant == 1) return DBW_V(8, 8, 2, false);
        if (g_render_variant == 2) return DBW_V(16, 8, 2, false);
        return DBW_V(16, 16, 1, false);        // (one list chunk in flight: a bin of the hard pass holds a few dozen faces, 256 lanes test them at once)
    }
    else {
        // soft K-layer passes: one wave64 per 8x8 tile; uv-fragments (the training path) take the specialised shading
#ifndef DBW_FWD_GROUP
#define DBW_FWD_GROUP 2
#endif
        if (A.tiled == 2 && (A.target || !(g_render_dbg & 4)) && A.sigma >= 0.f) {      // (sigma < 0, the sigmoid opacity: generic shading)
#if DBW_FWD_KEXACT
            if (A.K == KMAX) return launch_v<KMAX, 8, 8, DBW_FWD_GROUP, true, true>(recs, bbox, first_idx, num_faces, blur, persp, A, cb, srec, p2f, bary, dists, image, E, s);
#endif
            return DBW_V(8, 8, DBW_FWD_GROUP, true);
        }
        return DBW_V(8, 8, DBW_FWD_GROUP, false);
    }
#undef DBW_V
}

}  // namespace

struct MseArgs { const float *env_img, *target; float scale; float *loss_part, *g_fg, *g_env; float *rec_out = nullptr; const float *grad_rec = nullptr; };

static int render_fwd_impl(const float *face_verts_c, const int32_t *first_idx, const int32_t *num_faces,
                                    const int32_t *neighbor, const int32_t *c2o, const int32_t *clip_code,
                                    const float *clip_w, int Fc_stride, const float *face_uvs, const int32_t *face_map,
                                    const int32_t *map_desc, const float *maps, const float *faces_alpha, int alpha_len,
                                    int N, int64_t F_total, int H, int W, int K, int F, float sigma, float blur_radius,
                                    int perspective_correct, const float *background3, int32_t *pix_to_face, float *bary,
                                    float *dists, float *image, void *workspace, size_t workspace_bytes,
                                    int frag_layout, const MseArgs *mse, int stage, int image_layout, dbw_stream_t stream,
                                    const dbw::EnvFoldHost *fold = nullptr) {
    DBW_REQUIRE(stage >= 0 && stage <= 2, "stage must be 0 (whole pass), 1 (workspace only) or 2 (workspace already prepared)");
    DBW_REQUIRE(face_verts_c && first_idx && num_faces && pix_to_face && bary && dists && (image || mse) && workspace, "null pointer");
    DBW_REQUIRE(workspace_bytes >= dbw_rasterize_workspace_bytes(F_total), "workspace too small");
    DBW_REQUIRE(blur_radius >= 0.f && F_total >= 0, "bad blur_radius / F_total");
    ShadeArgs A;
    int rc = dbw_fill_shade_args(A, pix_to_face, bary, dists, c2o, clip_code, clip_w, Fc_stride, face_uvs, face_map, map_desc, maps,
                                 faces_alpha, alpha_len, N, H, W, K, F, sigma, background3);
    if (rc) return rc;
    DBW_REQUIRE(frag_layout >= 0 && frag_layout <= 3, "frag_layout must be 0 (N,H,W,K), 1 (8x8-tile planar), 2 (planar, uv) or 3 (planar, hard uv)");
    DBW_REQUIRE(frag_layout < 2 || F < (1 << 20), "frag_layouts 2 and 3 pack the face id in 20 bits");
    DBW_REQUIRE(frag_layout != 3 || (K == 1 && sigma == 0.f && blur_radius == 0.f && !faces_alpha),
                "frag_layout 3 is the hard single-layer pass: K == 1, sigma == 0, no faces_alpha");
    DBW_REQUIRE(frag_layout != 2 || F_total < (1LL << FRAG_COUNT_SHIFT), "frag_layout 2 packs the clipped face id in 26 bits");
    DBW_REQUIRE(image_layout == 0 || image_layout == 1, "image_layout must be 0 (N,C,H,W) or 1 (8x8-tile planar)");
    A.tiled = frag_layout;
    A.img_tiled = image_layout;
    if (mse) {
        DBW_REQUIRE(frag_layout == 2 && K > 1 && sigma >= 0.f, "the composite + MSE epilogue belongs to the uv-fragment soft pass (frag_layout 2, K > 1, sigma >= 0)");
        DBW_REQUIRE(stage == 1 || ((mse->env_img || fold) && mse->target && mse->loss_part && mse->g_fg && mse->g_env), "null pointer");
        A.env_img = mse->env_img; A.target = mse->target; A.mse_scale = mse->scale; A.loss_part = mse->loss_part; A.g_fg = mse->g_fg; A.g_env = mse->g_env;
        A.rec_out = mse->rec_out; A.grad_rec = mse->grad_rec;
    }
    if (K > DBW_MAX_FACES_PER_PIXEL) {
        dbw_set_error("dbw_render_fwd_fused: faces_per_pixel=%d > %d", K, DBW_MAX_FACES_PER_PIXEL);
        return DBW_ERR_UNSUPPORTED;
    }
    if (N == 0) return DBW_OK;
    hipStream_t s = (hipStream_t)stream;
    const float margin = (float)sqrt((double)blur_radius);
    CoarseBins cb;
    rc = dbw_prepare_raster(face_verts_c, first_idx, num_faces, neighbor, N, F_total, c2o ? (long long)Fc_stride : F_total, H, W, margin, 0, workspace, workspace_bytes, cb, s,
                            /*launch=*/stage != 2, /*want_cells: the 8x8-tile kernels*/ K > 1 || g_render_variant == 1);
    if (rc) return rc;
    const float4 *bbox = (const float4 *)workspace;
    const FaceRec *recs = dbw_workspace_recs(workspace, F_total);
    ShadeRec *srec = nullptr;
    if (frag_layout == 2 && K > 1 && F_total > 0) {
        srec = (ShadeRec *)dbw_workspace_shade_recs(workspace, F_total);
        if (stage != 2) {
            hipLaunchKernelGGL(shade_setup_kernel, dim3((unsigned)((F_total + 255) / 256)), dim3(256), 0, s, A, first_idx, num_faces, (long long)F_total, srec);
            rc = dbw_check_launch("shade_setup_kernel");
            if (rc) return rc;
        }
    }
    if (stage == 1) return DBW_OK;
    EnvFold E;
    memset(&E, 0, sizeof(E));
    if (fold) {
        DBW_REQUIRE(mse && frag_layout == 2 && K > 1, "the env layer folds into the soft pass with the loss epilogue");
        DBW_REQUIRE(fold->ws && fold->ws->cells && fold->ws->dom && fold->ws->shade_recs && fold->first_idx && fold->num_faces && fold->maps && fold->p2f && fold->uvj, "null pointer (env fold)");
        const dbw::RasterWorkspace &w = *fold->ws;
        E.recs = w.recs; E.first_idx = fold->first_idx; E.cell = w.cell; E.pool = w.pool; E.dom = w.dom; E.clist = w.list; E.ccount = w.count; E.nx = w.nx; E.ny = w.ny;
        E.num_faces = fold->num_faces; E.srec = (const ShadeRec *)w.shade_recs; E.maps = fold->maps;
        for (int i = 0; i < 3; ++i) E.bg[i] = fold->bg[i];
        E.p2f = fold->p2f; E.uvj = fold->uvj; E.persp = perspective_correct; E.dbg = g_render_dbg;
        for (int i = 0; i < 4; ++i) E.ndc[i] = cb.ndc[i];
        A.lean_grads = 1;          // (the training step: its two backward kernels are the only readers of the gradient images)
    }
#define DBW_RF(KM) launch<KM>(recs, bbox, first_idx, num_faces, blur_radius, perspective_correct, A, cb, srec, pix_to_face, bary, dists, image, E, s)
    if (K == 1) return DBW_RF(1);
    if (K <= 4) return DBW_RF(4);
    if (K <= 10) return DBW_RF(10);
    if (K <= 16) return DBW_RF(16);
    return DBW_RF(25);
#undef DBW_RF
}

extern "C" int dbw_render_fwd_fused(const float *face_verts_c, const int32_t *first_idx, const int32_t *num_faces,
                                    const int32_t *neighbor, const int32_t *c2o, const int32_t *clip_code,
                                    const float *clip_w, int Fc_stride, const float *face_uvs, const int32_t *face_map,
                                    const int32_t *map_desc, const float *maps, const float *faces_alpha, int alpha_len,
                                    int N, int64_t F_total, int H, int W, int K, int F, float sigma, float blur_radius,
                                    int perspective_correct, const float *background3, int32_t *pix_to_face, float *bary,
                                    float *dists, float *image, void *workspace, size_t workspace_bytes,
                                    int frag_layout, int stage, int image_layout, dbw_stream_t stream) {
    return render_fwd_impl(face_verts_c, first_idx, num_faces, neighbor, c2o, clip_code, clip_w, Fc_stride, face_uvs, face_map, map_desc, maps,
                           faces_alpha, alpha_len, N, F_total, H, W, K, F, sigma, blur_radius, perspective_correct, background3, pix_to_face,
                           bary, dists, image, workspace, workspace_bytes, frag_layout, nullptr, stage, image_layout, stream);
}

extern "C" int dbw_render_fwd_fused_mse(const float *face_verts_c, const int32_t *first_idx, const int32_t *num_faces,
                                        const int32_t *neighbor, const int32_t *c2o, const int32_t *clip_code,
                                        const float *clip_w, int Fc_stride, const float *face_uvs, const int32_t *face_map,
                                        const int32_t *map_desc, const float *maps, const float *faces_alpha, int alpha_len,
                                        int N, int64_t F_total, int H, int W, int K, int F, float sigma, float blur_radius,
                                        int perspective_correct, const float *background3, int32_t *pix_to_face, float *bary,
                                        float *dists, void *workspace, size_t workspace_bytes, const float *env_image,
                                        const float *target, float mse_scale, float *loss_partial, float *grad_fg,
                                        float *grad_env, int stage, int image_layout, dbw_stream_t stream) {
    const MseArgs mse{env_image, target, mse_scale, loss_partial, grad_fg, grad_env, nullptr, nullptr};
    return render_fwd_impl(face_verts_c, first_idx, num_faces, neighbor, c2o, clip_code, clip_w, Fc_stride, face_uvs, face_map, map_desc, maps,
                           faces_alpha, alpha_len, N, F_total, H, W, K, F, sigma, blur_radius, perspective_correct, background3, pix_to_face,
                           bary, dists, nullptr, workspace, workspace_bytes, 2, &mse, stage, image_layout, stream);
}

// the fg pass of the training step with the env layer folded in (step_kernels.h): stage 2 only -- the step's fused set-up kernels have
// filled both workspaces
int dbw::render_fwd_fused_mse_fold(const float *face_verts_c, const int32_t *first_idx, const int32_t *num_faces, const int32_t *neighbor,
                                   const int32_t *c2o, const int32_t *clip_code, const float *clip_w, int Fc_stride, const float *face_uvs,
                                   const int32_t *face_map, const int32_t *map_desc, const float *maps, const float *faces_alpha, int alpha_len, int N,
                                   int64_t F_total, int H, int W, int K, int F, float sigma, float blur_radius, int perspective_correct,
                                   const float *background3, int32_t *pix_to_face, float *bary, float *dists, void *workspace, size_t workspace_bytes,
                                   const float *target, float mse_scale, float *loss_partial, float *grad_fg, float *grad_env, const EnvFoldHost &fold,
                                   float *rec_out, const float *grad_rec, hipStream_t stream) {
    const MseArgs mse{nullptr, target, mse_scale, loss_partial, grad_fg, grad_env, rec_out, grad_rec};
    return render_fwd_impl(face_verts_c, first_idx, num_faces, neighbor, c2o, clip_code, clip_w, Fc_stride, face_uvs, face_map, map_desc, maps,
                           faces_alpha, alpha_len, N, F_total, H, W, K, F, sigma, blur_radius, perspective_correct, background3, pix_to_face,
                           bary, dists, nullptr, workspace, workspace_bytes, 2, &mse, 2, 1, (dbw_stream_t)stream, &fold);
}
