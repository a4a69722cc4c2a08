// Fused shading + layered alpha compositing for gfx950, forward and hand-derived backward.
//
// One kernel replaces, per fragment, what the reference spreads over PyTorch3D + ~15 torch kernels:
//   convert_clipped_rasterization_to_original_faces (barycentric back-conversion),
//   TexturesUV.sample_textures = interpolate_face_attributes + flip + F.grid_sample(bilinear, align_corners=True,
//   border) (renderer.py:226), the circular u-padding of dbw.py:339-341 (resolved here by index
//   arithmetic on the UNPADDED map, so no padded / flipped / per-view texture copy ever exists), and
//   layered_rgb_blend (renderer.py:241-273, clip_inside=True) with the per-face learned opacity gather.
// HBM traffic is the fragment stream (20 B per fragment slot in, 16 B per pixel out); texels come from L2/MALL.
//
// Backward (reference: torch autograd through the same ops): one pass recomputes alpha/colour per layer, a reverse
// pass accumulates the "colour behind" U_k and "transmittance behind" V_k so that
//   d rgb / d a_k = T_k (c_k - U_k),  d A / d a_k = T_k V_k        (no division by (1 - a_k))
// and scatters: texel gradients (wave-aggregated atomics: under magnification -- sky dome, ground -- most lanes of a
// wave share one bilinear footprint), per-face opacity gradients, d/d dists, and optionally d/d barycentrics.
#include "shade_common.h"
#include "env_bwd.h"
#include "../../include/dbw_hip.h"
#include "step_kernels.h"

#include <stdlib.h>
#include <string.h>

using namespace dbw;

// Fragments, gradient images and texel-gradient records are produced once and consumed once, a gigabyte of traffic later: they are
// loaded / stored non-temporally so that they stream past the L2 instead of evicting the tables the kernels keep coming back to
#ifndef DBW_NT_LOADS
#define DBW_NT_LOADS 1
#endif
template <class T>
__device__ __forceinline__ T ld_stream(const T *p) {
#if DBW_NT_LOADS
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st_stream4(int4 *p, int4 v) {
#if DBW_NT_LOADS
    int *q = (int *)p;
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    __builtin_nontemporal_store(w, (v4i *)q);
#else
    *p = v;
#endif
}

// One texel-gradient record of the texture bins: 24 B = three 8 B words {packed footprint, wx1} {wy1, g0} {g1, g2} at index `idx` of the
// record array (round 6: 32 B before, two of its eight words padding -- the bin reduction runs at the rate its records stream in)
__device__ __forceinline__ void st_record(int4 *records, unsigned long long idx, int packed, float wx1, float wy1, float g0, float g1, float g2) {
    typedef int v2i __attribute__((ext_vector_type(2)));
    v2i *q = (v2i *)records + idx * 3ull;
    v2i a, b, c;
    a.x = packed; a.y = __float_as_int(wx1); b.x = __float_as_int(wy1); b.y = __float_as_int(g0); c.x = __float_as_int(g1); c.y = __float_as_int(g2);
#if DBW_NT_LOADS
    __builtin_nontemporal_store(a, q); __builtin_nontemporal_store(b, q + 1); __builtin_nontemporal_store(c, q + 2);
#else
    q[0] = a; q[1] = b; q[2] = c;
#endif
}

namespace {

#ifndef DBW_BWD_UNROLL
#define DBW_BWD_UNROLL 2
#endif
constexpr int NT = 256;
constexpr int TILE = 16;

// The uv backward of the training step takes its blocks INTERLEAVED over the XCDs in chunks of BWD_CHUNK (every XCD works on a share of
// every view) instead of one contiguous range of views per XCD: a backward workgroup streams its own fragments, nothing in an XCD's L2 is
// worth keeping between workgroups, and with contiguous ranges the XCDs finished 10 % apart (the views differ in coverage; the kernel
// ends with its slowest XCD).  Measured, alternating on one box: the kernel next to the env backward 0.431 -> 0.421 ms, the step 0.878-0.883
// -> 0.865-0.866 ms at epoch 0 (chunks of 1, 4, 8, 32 alike, 128 no gain); nothing either way at epoch 800 or for the env backward, whose
// blocks all cost the same.  Grid: bwd_interleave_grid(total) workgroups (the padding ones leave at once)
#ifndef DBW_BWD_CHUNK
#define DBW_BWD_CHUNK 8
#endif
constexpr int BWD_CHUNK = DBW_BWD_CHUNK;
static inline unsigned bwd_interleave_grid(long long total) {
    if (BWD_CHUNK <= 0) return dbw_xcd_grid(total);
    const long long g = 8LL * BWD_CHUNK;
    return (unsigned)(g * ((total + g - 1) / g));
}
template <bool INTERLEAVED = false>
__device__ __forceinline__ bool pixel_of_block(const ShadeArgs &A, long long total_blocks, int &n, int &xi, int &yi) {
    long long logical;
    if (INTERLEAVED && BWD_CHUNK > 0) {
        const long long j = blockIdx.x >> 3;
        logical = ((j / BWD_CHUNK) * 8 + (blockIdx.x & 7)) * BWD_CHUNK + j % BWD_CHUNK;
        if (logical >= total_blocks) return false;
    } else {
        logical = xcd_remap(blockIdx.x, total_blocks);
        if (logical < 0) return false;
    }
    const int tiles_x = (A.W + TILE - 1) / TILE, tiles_y = (A.H + TILE - 1) / TILE;
    n = (int)(logical / (tiles_x * tiles_y));
    const int t = (int)(logical % (tiles_x * tiles_y));
    // wave w owns the 8x8 quadrant (w & 1, w >> 1) of the 16x16 tile, lane l the pixel (l & 7, l >> 3) of it: matches the
    // 8x8-tile planar fragment layout, so fragment loads are fully coalesced
    const int wq = threadIdx.x >> 6, l = threadIdx.x & 63;
    xi = (t % tiles_x) * TILE + ((wq & 1) << 3) + (l & 7);
    yi = (t / tiles_x) * TILE + ((wq >> 1) << 3) + (l >> 3);
    if (A.dbg & 32) {      // profiling switch: 16x4 strip per wave (tools/sweep_bwd.py)
        xi = (t % tiles_x) * TILE + (threadIdx.x & 15);
        yi = (t / tiles_x) * TILE + (threadIdx.x >> 4);
    }
    return true;
}

__global__ __launch_bounds__(NT) void shade_blend_fwd_kernel(ShadeArgs A, long long total_blocks, float *__restrict__ image) {
    int n, xi, yi;
    if (!pixel_of_block(A, total_blocks, n, xi, yi)) return;
    if (xi >= A.W || yi >= A.H) return;
    const long long pix = ((long long)n * A.H + yi) * A.W + xi;
    BlendFront bl;
    blend_front_init(bl);
    for (int k = 0; k < A.K; ++k) {
        Frag fr;
        if (!load_frag(A, n, frag_addr(A, n, yi, xi, k), fr)) continue;
        const float a = fr.e * fr.fa;
        if (a != 0.f) {             // (a == 0 leaves colour and transmittance as they are: no texel fetch)
            Sample s;
            footprint(A, fr, s);
            float c[3];
            fetch(A.maps, s, c);
            blend_front_step(bl, a, c);
        }
    }
    const long long plane = (long long)A.H * A.W;
    float *o = image + (long long)n * 4 * plane + (long long)yi * A.W + xi;
    float px[4];
    blend_front_finish(bl, A.bg, px);
    o[0] = px[0]; o[plane] = px[1]; o[2 * plane] = px[2]; o[3 * plane] = px[3];
}

#ifdef DBW_STATS_BWD
// table-traffic statistics of the uv backward (tools/diag/bwd_stats.py only; -DDBW_STATS_BWD implies -DDBW_PROFILE_BWD's buffer and read
// hook, with the cycle stamps off): per wave and layer, how many lanes update the texel / face tables and how they share their keys
#define DBW_PROFILE_BWD 1
#endif
#ifdef DBW_PROFILE_BWD
// cycle accounting of the fused backward (tools/bwd_cycles.py only): per-wave s_memtime deltas of the phases, kept per workgroup (a
// shared counter would serialise the atomics of 10^5 waves and distort what it measures) and summed on the host
constexpr int PROF_BLOCKS = 1 << 16;
__device__ unsigned long long g_prof[PROF_BLOCKS * 8];
#define PROF_T(x) const unsigned long long x = __builtin_readcyclecounter()
#ifdef DBW_PROFILE_BWD_K1
#define PROF_SEL SINGLE
#else
#define PROF_SEL !SINGLE
#endif
#ifdef DBW_STATS_BWD
#define PROF_ADD(i, a, b)
#define STAT_ADD(i, v) if ((threadIdx.x & 63) == 0 && blockIdx.x < PROF_BLOCKS) atomicAdd(&g_prof[(size_t)blockIdx.x * 8 + (i)], (unsigned long long)(v))
// lanes of `on` grouped by key: number of distinct keys, and the sum over the four 16-lane groups of the largest number of lanes that
// share one key (an LDS atomic replays once per lane of the most contended address of each group, profiles/r01_lds_atomic_ubench.txt)
__device__ __forceinline__ void key_stats(int key, bool on, int &distinct, int &replays) {
    unsigned long long rem = __ballot(on);
    int mx[4] = {0, 0, 0, 0};
    distinct = 0;
    while (rem) {
        const int L = __ffsll((long long)rem) - 1;
        const int k0 = __shfl(key, L, 64);
        const unsigned long long mm = __ballot(on && key == k0);
        for (int g = 0; g < 4; ++g) mx[g] = max(mx[g], __popc((unsigned)((mm >> (16 * g)) & 0xffffull)));
        ++distinct;
        rem &= ~mm;
    }
    replays = mx[0] + mx[1] + mx[2] + mx[3];
}
#else
#define PROF_ADD(i, a, b) if (PROF_SEL && (threadIdx.x & 63) == 0 && blockIdx.x < PROF_BLOCKS) atomicAdd(&g_prof[(size_t)blockIdx.x * 8 + (i)], (b) - (a))
#endif
#else
#define PROF_T(x)
#define PROF_ADD(i, a, b)
#endif
constexpr int BIN_LOG2 = 7, BIN_SLOTS = 1 << BIN_LOG2;   // per-block hash table of touched texture bins
constexpr int BIN_SUB = DBW_BIN_SUBCURSORS;                 // sub-ranges (each with its own cursor) of a bin's record range: a hot bin takes
                                                          // ~2000 reservations per launch, and returning atomics on ONE address serialise at
                                                          // ~0.2 us each (0.43 ms of a 0.86 ms kernel); 16 addresses per bin make that 16 chains
#ifndef DBW_BIN_SUB_PER_WG
#define DBW_BIN_SUB_PER_WG 2
#endif
constexpr int BIN_SUB_PER_WG = DBW_BIN_SUB_PER_WG;                         // sub-ranges one texbin_reduce workgroup accumulates

// the record sub-range (bin, sub): its first record in the record array and its capacity, from the caller's layout table (equal shares
// or capacities that follow the demand of the previous launch, ops.py).  ONE load and no alternative: a second way to the same two
// values would put the wait for the load where the two meet, i.e. right here, and drain every load the layer loop has in flight
// (0.467 -> 0.51 ms when the table was optional)
struct SubRange { unsigned first; int cap; };
__device__ __forceinline__ SubRange sub_range(const ShadeArgs &A, int bin, int sub) {
    const uint2 l = *(const uint2 *)(A.bin_layout + ((long long)bin * BIN_SUB + sub) * 2);
    SubRange r;
    r.first = l.x; r.cap = (int)l.y;
    return r;
}

// a footprint the bin's 33x33 LDS tile can hold: at most one row up and one column right of (r0, c0), no wrap
__device__ __forceinline__ bool bin_regular(const Sample &s) {
    return (s.r1 == s.r0 || s.r1 == s.r0 - 1) && (s.c1 == s.c0 || s.c1 == s.c0 + 1) &&
           ((s.r0 & 31) != 0 || s.r1 == s.r0 || s.r0 > 0) && (s.c1 >> 5) - (s.c0 >> 5) <= 1;
}

// FUSED = true additionally runs the rasteriser backward on the fly: d/d dists and d/d barycentrics are
// consumed in registers and only d/d face_verts leaves the kernel (pre-aggregated per face in LDS).
// SINGLE = true: a hard single-layer pass (K == 1; the env pass) -- its own instantiation, so that the layer loops fold away and a
// kernel trace tells the two passes of an iteration apart.
template <bool FUSED, bool BINNED, bool SINGLE>
__global__ __launch_bounds__(NT, (SINGLE || !FUSED ? 1 : 4)) void shade_blend_bwd_kernel(ShadeArgs A, long long total_blocks,
                                                             const float *__restrict__ gimg, float *__restrict__ gmaps,
                                                             float *__restrict__ galpha, float *__restrict__ gdists,
                                                             float *__restrict__ gbary, const float *__restrict__ fv,
                                                             float *__restrict__ gfv, int want_bary, int persp) {
    const int KK = SINGLE ? 1 : A.K;
    extern __shared__ __attribute__((aligned(16))) float s_layers[];   // [K][NT]: transmittance T_k in front of layer k
    float *s_T = s_layers + threadIdx.x;
    TexAgg tex_agg;
    AlphaAgg alpha_agg;
    // A.agg bit 0: texel gradients through the LDS hash (magnified / decimated maps); bit 1: per-face opacity gradients through
    // the LDS hash (always in the fused kernel: a wave sees one to three distinct opacities)
    const bool use_lds = (A.agg & 1) != 0, lds_alpha = (A.agg & 2) != 0;
    FaceAgg face_agg;
    // BINNED: block-level slot reservation.  Pass 1 counts the block's records per bin in a small LDS hash table, one global
    // cursor atomic per (block, bin) reserves the range, pass 2 writes the records -- a single atomic round trip per block.
    int *s_key = nullptr, *s_cnt = nullptr, *s_base = nullptr, *s_room = nullptr, *s_ent = nullptr;
    {
        char *nxt = (char *)(s_layers + (long long)KK * NT);
        if (use_lds) { tex_agg.bind(nxt); nxt += TexAgg::BYTES; tex_agg.clear(threadIdx.x, NT); }          // block-uniform
        if (lds_alpha) { alpha_agg.bind(nxt); nxt += AlphaAgg::BYTES; alpha_agg.clear(threadIdx.x, NT); }
        if (FUSED) { face_agg.bind(nxt); face_agg.clear(threadIdx.x, NT); nxt += FaceAgg::BYTES; }
        if (BINNED) {
            s_key = (int *)nxt; s_cnt = s_key + BIN_SLOTS; s_base = s_cnt + BIN_SLOTS; s_room = s_base + BIN_SLOTS; s_ent = s_room + BIN_SLOTS + threadIdx.x;
            if (threadIdx.x < BIN_SLOTS) { s_key[threadIdx.x] = -1; s_cnt[threadIdx.x] = 0; }
        }
    }
    int n, xi, yi;
    if (!pixel_of_block(A, total_blocks, n, xi, yi)) return;
    if (use_lds || lds_alpha || FUSED) __syncthreads();
    const bool in_img = xi < A.W && yi < A.H;
    const int lane = threadIdx.x & 63;
    f2 pndc;
    pndc.x = pix_to_ndc(A.W - 1 - xi, A.W, A.H);
    pndc.y = pix_to_ndc(A.H - 1 - yi, A.H, A.W);
    const long long pix = ((long long)n * A.H + yi) * A.W + xi;
    const long long plane = (long long)A.H * A.W;
    float gr = 0.f, gg = 0.f, gbl = 0.f, gA = 0.f;
    if (in_img) {
        const ImgAddr ia = img_addr(A, n, yi, xi, 4);
        const float *gi = gimg + ia.base;
        const float gs = A.gscale ? *A.gscale : 1.f;
        gr = ld_stream(gi) * gs; gg = ld_stream(gi + ia.cstride) * gs; gbl = ld_stream(gi + 2 * ia.cstride) * gs; gA = ld_stream(gi + 3 * ia.cstride) * gs;
    }
    // The deepest layer in which any pixel of this wave holds a fragment: only ~20 % of the slots of a soft render are occupied
    // and most waves see few layers, so both passes stop there instead of walking all K layers.  Fragments written by
    // dbw_render_fwd_fused (tiled layouts) fill the slots of a pixel front to back, so pass 1 finds the bound on its way: it stops
    // after the first batch of 5 layers whose last layer is empty everywhere (measured: pass 0 was 17 % of the wave time; fg
    // backward 0.73 -> 0.67 ms).  For fragments of unknown origin (layout 0) a pass 0 looks at every slot first -- exact without
    // that assumption; the single-layer instantiation keeps pass 0 as well (one layer has nothing to skip).
    PROF_T(t_begin);
    const bool prefix = FUSED && !SINGLE && A.tiled != 0;
    // uv-fragments (layout 2): the first layer's id carries the pixel's fragment count and every fragment its transmittance, so
    // the unbinned kernel needs no pass 1 at all: one load gives the bound, pass 2 reads T_k with the rest of the payload
    const bool stored_T = FUSED && !BINNED && A.tiled == 2;
    int kmax = 0;
    int cnt = KK;                            // layout 2: slots [0, cnt) of this pixel exist, the rest was never written
    if (FUSED && A.tiled == 2) {
        const int raw0 = in_img ? A.p2f[frag_addr(A, n, yi, xi, 0).s] : -1;
        cnt = raw0 < 0 ? 0 : (raw0 >> FRAG_COUNT_SHIFT);
    }
    if (stored_T) {
        for (int k = 0; k < KK; ++k) {
            if (__ballot(cnt > k) == 0ull) break;
            kmax = k + 1;
        }
    } else if (!prefix) {
#pragma unroll 5
        for (int k = 0; k < KK; ++k) {
            const bool occ = in_img && A.p2f[frag_addr(A, n, yi, xi, k).s] >= 0;
            if (__ballot(occ) != 0ull) kmax = k + 1;
        }
        if (!FUSED) kmax = KK;             // the unfused kernel writes d/d dists and d/d barycentrics of every slot
    }
    PROF_T(t_p0);
    PROF_ADD(0, t_begin, t_p0);
    // pass 1 (front to back): alpha and transmittance per layer; no texture access
    if (!stored_T) {
        float T = 1.f;
        const int klimit = prefix ? KK : kmax;
#pragma unroll 5
        for (int k = 0; k < klimit; ++k) {  // unrolled: the fragment loads of several layers are in flight together
            float ak = 0.f;
            Frag fr;
            bool valid;
            if (FUSED && !BINNED && A.tiled == 2) {
                // uv-fragments carry the blend opacity the forward used: two coalesced loads per layer instead of the whole
                // payload + the opacity gather + an exponential (pass 1 was 20 % of the wave time)
                const FragAddr o = frag_addr(A, n, yi, xi, k);
                valid = in_img && k < cnt && A.p2f[o.s] >= 0;
                if (valid) fr.a = A.bary[o.b + 3 * o.bstride];
            } else {
                valid = in_img && k < cnt && load_frag<FUSED>(A, n, frag_addr(A, n, yi, xi, k), fr);
            }
            if (prefix) {
                const bool anyv = __ballot(valid) != 0ull;
                if (anyv) kmax = k + 1;
                else if (k % 5 == 4) break;     // (the stores below of an all-empty layer are never read: pass 2 stops at kmax)
            }
            if (valid) ak = fr.a;
            s_T[k * NT] = T;
            if (BINNED) {
                int ent = -1;
                const float wgt = T * ak;
                if (valid && (wgt * gr != 0.f || wgt * gg != 0.f || wgt * gbl != 0.f)) {
                    Sample s;
                    footprint(A, fr, s);
                    if (bin_regular(s)) {
                        const int bin = A.bin_base[fr.map] + (s.r0 >> 5) * ((s.ws + 31) >> 5) + (s.c0 >> 5);
                        unsigned h = ((unsigned)bin * 2654435761u) >> (32 - BIN_LOG2);
                        for (int probe = 0; probe < 8; ++probe) {
                            const int prev = atomicCAS(&s_key[h], -1, bin);
                            if (prev == -1 || prev == bin) { ent = (int)(h << 16) | atomicAdd(&s_cnt[h], 1); break; }
                            h = (h + 1) & (BIN_SLOTS - 1);
                        }
                    }
                }
                s_ent[k * NT] = ent;
            }
            T *= (1.f - ak);
        }
    }
    if (BINNED) {
        __syncthreads();
        if (threadIdx.x < BIN_SLOTS && s_key[threadIdx.x] >= 0) {
            // -> first record of the block's reservation in the record array, and the room left behind it in the sub-range
            const int sub = blockIdx.x & (BIN_SUB - 1), ci = s_key[threadIdx.x] * BIN_SUB + sub;
            const int b0 = atomicAdd(A.bin_cursor + ci, s_cnt[threadIdx.x]);
            const SubRange sr = sub_range(A, s_key[threadIdx.x], sub);
            s_base[threadIdx.x] = (int)(sr.first + (unsigned)b0);
            s_room[threadIdx.x] = sr.cap - b0;
        }
        __syncthreads();
    }
    PROF_T(t_p1);
    PROF_ADD(1, t_p0, t_p1);
    // pass 2 (back to front)
    BlendBack bk;
    blend_back_init(bk, A.bg);
    // uv-fragments without pass 1: the payload of layer k - 1 is requested while layer k is being processed (the only unhidden
    // latency of the loop is the single hop of coalesced loads at the top of each iteration)
    RawUV nxt;
    nxt.ok = false;
    if (stored_T && kmax > 0) nxt = load_raw_uv(A, frag_addr(A, n, yi, xi, kmax - 1), in_img && kmax - 1 < cnt);
#pragma unroll DBW_BWD_UNROLL
    for (int k = kmax - 1; k >= 0; --k) {    // unrolled by 2: the gather chains of two layers overlap
        PROF_T(t_it);
        Frag fr;
        bool valid = false;
        const FragAddr fo = frag_addr(A, n, yi, xi, k);
        if (stored_T) {
            const RawUV cur = nxt;
            if (k > 0) nxt = load_raw_uv(A, frag_addr(A, n, yi, xi, k - 1), in_img && k - 1 < cnt);
            valid = cur.ok;
            if (valid) frag_from_raw_uv<FUSED>(A, n, cur, fr);
        } else if (in_img && k < cnt) valid = load_frag<FUSED>(A, n, fo, fr);
        const float ak = valid ? fr.a : 0.f, Tk = stored_T ? (valid ? fr.T : 1.f) : s_T[k * NT];
        Sample s;
        s.a00 = s.a01 = s.a10 = s.a11 = 0;
        float c[3] = {0.f, 0.f, 0.f};
        if (valid) {
            footprint(A, fr, s);
            if (A.tiled == 2) { c[0] = fr.col[0]; c[1] = fr.col[1]; c[2] = fr.col[2]; }     // sampled by the forward (0 where ak == 0)
            else if (ak != 0.f) fetch(A.maps, s, c);
        }
        const float ga_ = blend_back_step(bk, Tk, ak, c[0], c[1], c[2], gr, gg, gbl, gA);      // (ak == 0 for an empty slot: state unchanged)
        const float ga = valid ? ga_ : 0.f;
        const float wgt = valid ? Tk * ak : 0.f;
        PROF_T(t_a);
        PROF_ADD(2, t_it, t_a);
        // geometric alpha -> dists ; learned opacity
        float gd = 0.f;
        if (valid && A.sigma > 0.f && fr.d >= 0.f) gd = ga * fr.a * (FUSED ? -A.inv_sigma : -1.f / A.sigma);
        else if (valid && A.sigma < 0.f) gd = ga * fr.a * (1.f - fr.e) * (FUSED ? -A.inv_sigma : 1.f / A.sigma);      // d sigmoid(-d / s) / dd = -e (1 - e) / s, every d
        if (!FUSED && gdists && in_img) gdists[pix * KK + k] = gd;
        if (galpha && !(A.dbg & 2)) {
            const float gfa[1] = {valid ? ga * fr.e : 0.f};
            const long long gidx = valid ? alpha_grad_index(A, n, fr.j, fr.map) : 0;
            if (lds_alpha) alpha_agg.add_wave(galpha, (int)gidx, gfa, valid && gfa[0] != 0.f);
            else wave_agg_atomic<1>(galpha, gidx, valid && gfa[0] != 0.f, gfa, lane);
        }
        PROF_T(t_b);
        PROF_ADD(3, t_a, t_b);
        // colour -> texels (and -> uv -> barycentrics)
        const float gc[3] = {wgt * gr, wgt * gg, wgt * gbl};
        const bool tex = valid && (gc[0] != 0.f || gc[1] != 0.f || gc[2] != 0.f);
        if (use_lds && !(A.dbg & 1)) {
            // merge the footprint's texels that fall into the same stored cell, then one LDS insert per cell
            float w00 = s.w00, w01 = s.w01, w10 = s.w10, w11 = s.w11;
            if (s.a01 == s.a00) { w00 += w01; w01 = 0.f; }
            if (s.a10 == s.a00) { w00 += w10; w10 = 0.f; }
            if (s.a11 == s.a00) { w00 += w11; w11 = 0.f; }
            else if (s.a11 == s.a01) { w01 += w11; w11 = 0.f; }
            else if (s.a11 == s.a10) { w10 += w11; w11 = 0.f; }
            const int ad[4] = {s.a00, s.a01, s.a10, s.a11};
            const float wt[4] = {w00, w01, w10, w11};
            // hard single-layer passes over magnified maps: 4 horizontally adjacent pixels usually share their footprint -- sum their
            // contributions in registers (DPP) and let the first lane of the four update the table: 4x fewer lanes on one slot
            bool lead = true;
            int same4 = 0;
            if (SINGLE) {
                same4 = quad_and((tex && s.a00 == quad_first(s.a00) && s.a11 == quad_first(s.a11)) ? 1 : 0);
                lead = !same4 || (lane & 3) == 0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[3] = {gc[0] * wt[q], gc[1] * wt[q], gc[2] * wt[q]};
                if (SINGLE) {
#pragma unroll
                    for (int c3 = 0; c3 < 3; ++c3) { const float qs = quad_sum(tex ? v[c3] : 0.f); v[c3] = same4 ? qs : v[c3]; }
                }
                const bool on = tex && lead && (same4 ? (v[0] != 0.f || v[1] != 0.f || v[2] != 0.f) : wt[q] != 0.f);
                // ... and over decimated maps a whole wave usually sits inside one cell
                if (SINGLE) tex_agg.add_wave(gmaps, (int)((unsigned)ad[q] / 3u), v, on);
                else if (on) tex_agg.add(gmaps, (int)((unsigned)ad[q] / 3u), v);
            }
        } else if (__ballot(tex) != 0ull && !(A.dbg & 1)) {
            bool pending = tex;
            if (BINNED) {
                // texture-space binning: append a 32 B record to the bin of the 32x32-texel tile the footprint starts in (the
                // bin's LDS tile has a 1-texel halo: row r0-1 and column c0+1); irregular footprints (circular wrap, clamped
                // borders leaving the halo), hash-table misses and bin overflow fall through to the atomic path below
                const int ent = s_ent[k * NT];
                if (tex && ent >= 0) {
                    const int h = ent >> 16, rank = ent & 0xffff;
                    if (rank < s_room[h]) {
                        const unsigned packed = (unsigned)(s.r0 & 31) | ((unsigned)(s.c0 & 31) << 5) | ((unsigned)(s.r0 - s.r1) << 10) |
                                                ((unsigned)(s.c1 - s.c0) << 11);
                        st_record(A.bin_records, (unsigned long long)((unsigned)s_base[h] + (unsigned)rank), (int)packed, s.wx1, s.wy1, gc[0], gc[1], gc[2]);
                        pending = false;
                    }
                }
            }
            // lanes sharing the same top-left texel share all four addresses
            unsigned long long rem = __ballot(pending);
            int iter = 0;
            while (rem) {
                if (iter >= 8 || (A.dbg & 4)) {
                    if (pending && ((rem >> lane) & 1ull)) {
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            unsafeAtomicAdd(gmaps + s.a00 + ch, gc[ch] * s.w00);
                            unsafeAtomicAdd(gmaps + s.a01 + ch, gc[ch] * s.w01);
                            unsafeAtomicAdd(gmaps + s.a10 + ch, gc[ch] * s.w10);
                            unsafeAtomicAdd(gmaps + s.a11 + ch, gc[ch] * s.w11);
                        }
                    }
                    break;
                }
                const int leader = __ffsll((long long)rem) - 1;
                const int k00 = __shfl(s.a00, leader, 64), k11 = __shfl(s.a11, leader, 64);
                const bool match = pending && s.a00 == k00 && s.a11 == k11;
                const unsigned long long mm = __ballot(match);
                if (__popcll(mm) > 1) {
                    const int k01 = __shfl(s.a01, leader, 64), k10 = __shfl(s.a10, leader, 64);
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const float v00 = wave_sum(match ? gc[ch] * s.w00 : 0.f);
                        const float v01 = wave_sum(match ? gc[ch] * s.w01 : 0.f);
                        const float v10 = wave_sum(match ? gc[ch] * s.w10 : 0.f);
                        const float v11 = wave_sum(match ? gc[ch] * s.w11 : 0.f);
                        if (lane == leader) {
                            unsafeAtomicAdd(gmaps + k00 + ch, v00);
                            unsafeAtomicAdd(gmaps + k01 + ch, v01);
                            unsafeAtomicAdd(gmaps + k10 + ch, v10);
                            unsafeAtomicAdd(gmaps + k11 + ch, v11);
                        }
                    }
                } else if (match) {
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        unsafeAtomicAdd(gmaps + s.a00 + ch, gc[ch] * s.w00);
                        unsafeAtomicAdd(gmaps + s.a01 + ch, gc[ch] * s.w01);
                        unsafeAtomicAdd(gmaps + s.a10 + ch, gc[ch] * s.w10);
                        unsafeAtomicAdd(gmaps + s.a11 + ch, gc[ch] * s.w11);
                    }
                }
                rem &= ~mm;
                ++iter;
            }
        }
        PROF_T(t_c);
        PROF_ADD(4, t_b, t_c);
        float g9[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int fc = 0;
        bool has_g9 = false;
        if ((FUSED ? (want_bary != 0) : (gbary != nullptr)) && in_img || (FUSED && in_img)) {
            float gb[3] = {0.f, 0.f, 0.f};
            // (faces below geom_begin have constant vertices -- the sky dome: nothing flows through their barycentrics)
            if (FUSED ? (want_bary != 0 && fr.j >= A.geom_begin) : true)
            if (tex) {
                float gu, gv;
                sample_grad_uv(A.maps, s, gc, gu, gv);
                const float *uv = A.face_uvs + (long long)fr.j * 6;
                const float go[3] = {gu * uv[0] + gv * uv[1], gu * uv[2] + gv * uv[3], gu * uv[4] + gv * uv[5]};
                convert_bary_bwd(fr.cd, fr.w2, fr.w3, go, gb);
            }
            if (!FUSED) {
                const long long o = (pix * KK + k) * 3;
                gbary[o] = gb[0]; gbary[o + 1] = gb[1]; gbary[o + 2] = gb[2];
            } else if (valid && (gd != 0.f || gb[0] != 0.f || gb[1] != 0.f || gb[2] != 0.f)) {
                // rasteriser backward for this fragment (same math as raster_bwd_kernel, grad_zbuf = 0; divisions by v_rcp_f32).
                // With the exponential opacity d/d dist is non-zero only outside the triangle (fr.d >= 0: sign +1); the sigmoid opacity of
                // clip_inside = False (sigma < 0) also differentiates inside, where the stored distance is MINUS the distance to the nearest
                // edge.  The barycentrics are only recomputed when a barycentric gradient has to be propagated.
                fc = BINNED ? (A.tiled == 2 ? (A.p2f[fo.s] & FRAG_FACE_MASK) : A.p2f[fo.s]) : fr.fc;      // (the binned instantiation re-reads the id: one live register less, it sits at the 128-VGPR edge)
                has_g9 = true;
                const float *q = fv + (long long)fc * 9;
                const f2 a{q[0], q[1]}, b{q[3], q[4]}, c{q[6], q[7]};
                if (gd != 0.f) {
                    f2 d0, d1, d2;
                    point_tri_dist_bwd<true>(pndc, a, b, c, fr.d < 0.f ? -gd : gd, d0, d1, d2);
                    g9[0] = d0.x; g9[1] = d0.y; g9[3] = d1.x; g9[4] = d1.y; g9[6] = d2.x; g9[7] = d2.y;
                }
                if (gb[0] != 0.f || gb[1] != 0.f || gb[2] != 0.f) {
                    const float z0 = q[2], z1 = q[5], z2 = q[8];
                    const f3 bary0 = bary_fwd(pndc, a, b, c);
                    const f3 bp = persp ? persp_fwd(bary0, z0, z1, z2) : bary0;
                    f3 gg{gb[0], gb[1], gb[2]};
                    gg = clip_bwd(bp, gg);
                    float pz0 = 0.f, pz1 = 0.f, pz2 = 0.f;
                    if (persp) gg = persp_bwd(bary0, z0, z1, z2, gg, pz0, pz1, pz2);
                    f2 e0, e1, e2;
                    bary_bwd(pndc, a, b, c, gg, e0, e1, e2);
                    g9[0] += e0.x; g9[1] += e0.y; g9[2] += pz0;
                    g9[3] += e1.x; g9[4] += e1.y; g9[5] += pz1;
                    g9[6] += e2.x; g9[7] += e2.y; g9[8] += pz2;
                }
            }
        }
        PROF_T(t_d);
        PROF_ADD(5, t_c, t_d);
        if (FUSED && !(A.dbg & 64)) {           // dbg 64: ablate the aggregation (tools/ablate.py)
            // hard single-layer passes rasterise few, large faces (a wave usually sits inside one): sum across the wave first;
            // soft multi-layer passes see several small faces per wave and layer, where the uniformity test does not pay
            if (SINGLE) face_agg.add_wave(gfv, fc, g9, has_g9);
            else if (has_g9) face_agg.add(gfv, fc, g9);
        }
        PROF_T(t_e);
        PROF_ADD(6, t_d, t_e);
    }
    PROF_T(t_end);
    PROF_ADD(7, t_begin, t_end);
    if (use_lds || lds_alpha || FUSED) __syncthreads();
    if (use_lds) tex_agg.flush(gmaps, threadIdx.x, NT);
    if (lds_alpha && galpha) alpha_agg.flush(galpha, threadIdx.x, NT);
    if (FUSED) face_agg.flush(gfv, threadIdx.x, NT);
}

// ---- the training path's soft pass, specialised: uv-fragments (layout 2), detached barycentrics, texel gradients through the LDS
// hash (decimated maps) ----------------------------------------------------------------------------------------------------------
// Same mathematics as shade_blend_bwd_kernel<true, false, false> on those inputs, organised for residency and issue rate (the
// generic kernel sits at 127 VGPRs / 36.5 KB of LDS = 4 waves per SIMD and spends 60 % of its wave time waiting):
//  * every run-time switch of the generic kernel is resolved, the per-layer transmittance array in LDS is gone (the fragments
//    carry T), the opacity and face-vertex tables are ONE table keyed by the clipped face (one claiming ds_cmpst per fragment
//    instead of two) -> 22 KB of LDS per 256 threads;
//  * the tile index is wave-uniform: fragment planes are addressed as scalar base + lane;
//  * the distance backward picks the closest edge with selects and differentiates that one edge (the generic form runs the three
//    candidate branches under divergence).
#ifndef DBW_FACE_LOG2
#define DBW_FACE_LOG2 7
#endif
struct FaceAlphaAgg {      // key = clipped face id -> 6 vertex xy-gradients + 1 opacity gradient (destination index in `aux`)
    static constexpr int LOG2 = DBW_FACE_LOG2, NSLOT = 1 << LOG2, NV = 7;
    static constexpr size_t BYTES = (size_t)NSLOT * (NV * 8 + 8);
    int *keys, *aux;
    double *vals;
    __device__ __forceinline__ void bind(void *lds) { keys = (int *)lds; aux = keys + NSLOT; vals = (double *)(aux + NSLOT); }
    __device__ __forceinline__ void clear(int tid, int nthreads) {
        for (int i = tid; i < NSLOT; i += nthreads) keys[i] = -1;
        for (int i = tid; i < NSLOT * NV; i += nthreads) vals[i] = 0.0;
    }
    __device__ __forceinline__ void add(float *__restrict__ gfv, float *__restrict__ galpha, int key, int aidx, const float (&v)[NV]) {
        unsigned h = ((unsigned)key * 2654435761u) >> (32 - LOG2);
#pragma unroll 1
        for (int p = 0; p < 8; ++p) {
            const int old = atomicCAS(&keys[h], -1, key);
            if (old == -1 || old == key) {
                aux[h] = aidx;                      // every lane of a face writes the same opacity index
#pragma unroll
                for (int c = 0; c < NV; ++c)
                    if (v[c] != 0.f) atomicAdd(&vals[h * NV + c], (double)v[c]);
                return;
            }
            h = (h + 1) & (NSLOT - 1);
        }
#pragma unroll
        for (int c = 0; c < 6; ++c)
            if (v[c] != 0.f) unsafeAtomicAdd(gfv + (long long)key * 9 + (c >> 1) * 3 + (c & 1), v[c]);
        if (v[6] != 0.f && galpha) unsafeAtomicAdd(galpha + aidx, v[6]);
    }
    __device__ __forceinline__ void flush(float *__restrict__ gfv, float *__restrict__ galpha, int tid, int nthreads) {
        for (int i = tid; i < NSLOT; i += nthreads) {
            const int k = keys[i];
            if (k >= 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const float x = (float)vals[i * NV + c];
                    if (x != 0.f) unsafeAtomicAdd(gfv + (long long)k * 9 + (c >> 1) * 3 + (c & 1), x);
                }
                const float x = (float)vals[i * NV + 6];
                if (x != 0.f && galpha) unsafeAtomicAdd(galpha + aux[i], x);
            }
        }
    }
};

// squared distance to segment a -> b and the clamped parameter of the closest point (point_line_dist_t<true> + its t)
__device__ __forceinline__ float seg_dist_t(f2 p, f2 a, f2 b, float &tt) {
    const float dx = b.x - a.x, dy = b.y - a.y;
    const float l2 = dx * dx + dy * dy;
    tt = 1.f;
    if (l2 <= DBW_EPS) return (p.x - b.x) * (p.x - b.x) + (p.y - b.y) * (p.y - b.y);
    const float t = (dx * (p.x - a.x) + dy * (p.y - a.y)) * __builtin_amdgcn_rcpf(l2);
    tt = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
    const float qx = a.x + tt * dx, qy = a.y + tt * dy;
    return (p.x - qx) * (p.x - qx) + (p.y - qy) * (p.y - qy);
}

// BINNED = true: the texel gradients of FULL-RESOLUTION maps leave as 32 B records in texture-space bins (see shade_blend_bwd_kernel's
// BINNED instantiation and texbin_reduce_kernel) instead of going through the LDS texel table.  The generic kernel reserves the
// record slots per workgroup, which costs it a counting pass over all layers; here a WAVE reserves the slots of one layer with one
// returning atomic per distinct bin (an 8x8-pixel patch of one layer touches one to four 32x32-texel bins), software-pipelined one
// layer ahead of its use: the fragments are loaded two layers ahead, the reservation for layer k - 1 is issued at the top of
// iteration k and its result is first looked at in iteration k - 1.  A fragment carries T and its blend opacity, so whether it emits a
// record at all (weight * pixel gradient != 0) is known without the back-to-front recurrences.  Where a sub-range lies in the record
// array and how much it holds comes from the caller's layout table (include/dbw_hip.h: bin_layout -- equal shares or by demand).
struct BinRes {            // reservation of one fragment's record: bin, rank among the wave's records of that bin, the lane that holds
    int bin, rank, leader, base, packed;   // `base` (it issued the atomic), and the footprint in the record's packed form
    float wx1, wy1;
    uint2 lay;                             // (leader only) {first record, capacity} of the sub-range the reservation was made in
};

constexpr int ALPHA_DIRECT_MAPS = 64, ALPHA_DIRECT_SPREAD = 8;
constexpr size_t ALPHA_DIRECT_BYTES = (size_t)ALPHA_DIRECT_MAPS * ALPHA_DIRECT_SPREAD * sizeof(double);
// merging steps (lane_merge, dbw_common.h) in front of the texel table and the face table of the uv backward: 0 / 1 / 2 / 3 / 4 steps on
// both: 0.390 / 0.356 / 0.354 / 0.377 / 0.390 ms (decimated maps); one step costs half the VALU instructions of two (+17 % instead of
// +34 % of the kernel) for the same time, alone and in the step
#ifndef DBW_TEX_MERGE
#define DBW_TEX_MERGE 1
#endif
#ifndef DBW_FACE_MERGE
#define DBW_FACE_MERGE 1
#endif
#ifndef DBW_FACE_MERGE_BINNED
#define DBW_FACE_MERGE_BINNED 1
#endif
constexpr int TEX_MERGE = DBW_TEX_MERGE, FACE_MERGE = DBW_FACE_MERGE;
// (round 4, off: the table claims of a layer issued together / ahead of the work in front of their use -- measured slower, see the layer loop)
#ifndef DBW_UVB_WAVES
#define DBW_UVB_WAVES 4      // (the binned instantiation keeps two layers of fragments + one of vertices in flight: 4 waves of 128 VGPRs, no spills --
                             // a spill reload in the layer loop is a vmcnt(0); 4 / 5 / 6 waves per SIMD measured alike before)
#endif
#ifdef DBW_TILE_CLOCK
// tools-only (tools/diag/r06_bwd_clock.py): {view, layers of the workgroup's first wave (-1: left at the first barrier), end stamp, start
// stamp (low 32 bits of the 100 MHz wall clock)} per workgroup of the last launch of the uv backward
__device__ unsigned g_bwd_clock[1 << 16][4];
#define BWD_CLOCK_END(n_, layers_) { if (threadIdx.x == 0 && blockIdx.x < (1u << 16)) { unsigned *o_ = g_bwd_clock[blockIdx.x]; o_[0] = (unsigned)(n_); o_[1] = (unsigned)(layers_); o_[2] = (unsigned)wall_clock64(); } }
#else
#define BWD_CLOCK_END(n_, layers_)
#endif
template <bool BINNED>
__global__ __launch_bounds__(NT, (BINNED ? DBW_UVB_WAVES : 5)) void render_bwd_uv_kernel(ShadeArgs A, long long total_blocks, const float *__restrict__ gimg,
                                                              float *__restrict__ gmaps, float *__restrict__ galpha,
                                                              const float *__restrict__ fv, float *__restrict__ gfv) {
    extern __shared__ __attribute__((aligned(16))) float s_uvbwd[];
    constexpr bool SINGLE = false;      // (cycle accounting macros)
    (void)SINGLE;
    TexAgg tex_agg;
    FaceAlphaAgg fa_agg;
    if (A.sync_flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(A.sync_flag, A.sync_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#ifdef DBW_TILE_CLOCK
    if (blockIdx.x < (1u << 16)) { g_bwd_clock[blockIdx.x][3] = (unsigned)wall_clock64(); g_bwd_clock[blockIdx.x][2] = 0u; }
#endif
    int n, xi, yi;
    if (!pixel_of_block<true>(A, total_blocks, n, xi, yi)) return;
    const bool in_img = xi < A.W && yi < A.H;
    const int lane = threadIdx.x & 63;
    // wave-uniform tile of the 8x8-tile planar fragment layout
    const int tiles_x = (A.W + 7) >> 3, tiles_y = (A.H + 7) >> 3;
    // (a 16x16 workgroup at the lower / right image border can hold 8x8 quadrants that lie outside the tile grid -- H or W not a multiple of
    // 16: such a wave holds no pixel of the image; it is pointed at the view's first tile so that the unconditional loads of the binned
    // instantiation stay inside the fragment buffers)
    const bool phantom = (yi >> 3) >= tiles_y || (xi >> 3) >= tiles_x;
    const int tile = __builtin_amdgcn_readfirstlane(phantom ? n * tiles_y * tiles_x : (n * tiles_y + (yi >> 3)) * tiles_x + (xi >> 3));
    const long long tb = ((long long)tile * A.K) << 6;
    const int *__restrict__ p2f_t = A.p2f + tb + lane;
    const float *__restrict__ dists_t = A.dists + tb + lane;
    const float *__restrict__ bary_t = A.bary + tb * 8 + lane;
    // the fragment counts of the block's pixels are requested first; the tables are cleared while they travel; a block none of whose
    // pixels holds a fragment -- six of ten at config 2 -- leaves at the barrier that ends the clearing: no gradient image loads, no
    // second barrier, no flush scans
    int cnt = 0;
    if (in_img && (yi >> 3) < tiles_y && (xi >> 3) < tiles_x) {
        const int raw0 = ld_stream(p2f_t);
        cnt = raw0 < 0 ? 0 : (raw0 >> FRAG_COUNT_SHIFT);
    }
    if (!BINNED) {
        tex_agg.bind(s_uvbwd);
        tex_agg.clear(threadIdx.x, NT);
    }
    fa_agg.bind((char *)s_uvbwd + (BINNED ? 0 : TexAgg::BYTES));
    fa_agg.clear(threadIdx.x, NT);
#ifndef DBW_MD_LDS_UNBINNED
#define DBW_MD_LDS_UNBINNED 0
#endif
    constexpr bool MD_LDS = BINNED || DBW_MD_LDS_UNBINNED;
    __shared__ __attribute__((aligned(16))) int s_md[MD_LDS ? MD_CACHE_MAPS * 8 : 8];
#ifndef DBW_LAST_WAVE_FLUSH
#define DBW_LAST_WAVE_FLUSH 1
#endif
    constexpr bool LAST_WAVE_FLUSH = BINNED && DBW_LAST_WAVE_FLUSH;
    __shared__ int s_done;                     // waves of the workgroup that have finished their layers
    if (threadIdx.x == 0) s_done = 0;
    MapDescCache mdc;
    mdc.load(A, s_md, BINNED ? A.bin_base : nullptr, threadIdx.x, NT, MD_LDS);
    // one opacity per texture map (the training path: one per block, alpha_len = -M) with few maps: the opacity gradient goes to a small
    // DIRECT-mapped fp64 array, map * 8 + a lane-derived spread -- a fire-and-forget ds_add_f64 per fragment, no slot look-up.  The
    // face table then only sees the fragments whose distance carries a gradient (outside their face, inside the blur band): the
    // fragments inside their face, four of ten, used to claim a slot (ds_cmpst, a round trip the wave waits for) for that one value
    double *alpha_dir = (double *)((char *)s_uvbwd + (BINNED ? 0 : TexAgg::BYTES) + FaceAlphaAgg::BYTES);
    const int n_maps = -A.alpha_len;
    const bool alpha_direct = galpha && A.faces_alpha && A.alpha_len < 0 && n_maps <= ALPHA_DIRECT_MAPS;
    if (alpha_direct)
        for (int i = threadIdx.x; i < n_maps * ALPHA_DIRECT_SPREAD; i += NT) alpha_dir[i] = 0.0;
    if (!__syncthreads_or(cnt > 0)) { BWD_CLOCK_END(n, -1); return; }
    f2 pndc;
    pndc.x = pix_to_ndc(A.W - 1 - xi, A.W, A.H);
    pndc.y = pix_to_ndc(A.H - 1 - yi, A.H, A.W);
    const long long plane = (long long)A.H * A.W;
    float gr = 0.f, gg = 0.f, gbl = 0.f, gA = 0.f;
    if (in_img && cnt > 0) {          // (a pixel without fragments passes nothing on: its gradient is not read -- the forward of the training
                                      // step does not even write it for a tile without fragments, ShadeArgs::lean_grads)
        const ImgAddr ia = img_addr(A, n, yi, xi, 4);
        const float *gi = gimg + ia.base;
        const float gs = A.gscale ? *A.gscale : 1.f;
        gr = ld_stream(gi) * gs; gg = ld_stream(gi + ia.cstride) * gs; gbl = ld_stream(gi + 2 * ia.cstride) * gs; gA = ld_stream(gi + 3 * ia.cstride) * gs;
    }
    PROF_T(t_begin);
    int kmax = 0;
    for (int k = 0; k < A.K; ++k) {
        if (__ballot(cnt > k) == 0ull) break;
        kmax = k + 1;
    }
    PROF_T(t_p0);
    PROF_ADD(0, t_begin, t_p0);
    BlendBack bk;
    blend_back_init(bk, A.bg);
    struct Raw { int fc; float u, v, jm, a, c0, c1, c2, T, d; };
    auto load = [&](int k, bool ok) {
        Raw r;
        r.fc = 0; r.u = r.v = r.jm = r.a = r.c0 = r.c1 = r.c2 = r.d = 0.f; r.T = 1.f;
        if (ok) {
            r.fc = ld_stream(p2f_t + (k << 6)) & FRAG_FACE_MASK;
            const float *b = bary_t + (k << 9);
            r.u = ld_stream(b); r.v = ld_stream(b + 64); r.jm = ld_stream(b + 128); r.a = ld_stream(b + 192);
            r.c0 = ld_stream(b + 256); r.c1 = ld_stream(b + 320); r.c2 = ld_stream(b + 384); r.T = ld_stream(b + 448);
            r.d = ld_stream(dists_t + (k << 6));
        }
        return r;
    };
    // BINNED: the memory schedule of the layer loop.  vmcnt counts loads, stores and returning atomics in ONE in-order counter, so
    // waiting for anything drains whatever was issued before it; the loop therefore issues, at its top and unconditionally (a known
    // number of operations, in one basic block), the fragment loads of layer k - 2 and the vertex loads of layer k - 1's faces --
    // everything consumed in an iteration was requested at least one iteration earlier, and the record stores / the cursor atomic of
    // an iteration are only waited for in the next one.  (Before: the distance backward loaded its face's vertices where it needed
    // them, and that wait drained the loads issued ahead, the record stores and the atomic of the same iteration -- 69 % of the
    // wave time was waiting, whatever the occupancy.)  Layer indices are clamped; every lane of a launched tile owns a slot in every
    // plane, so the unmasked loads stay inside the fragment buffers and are masked when they are consumed.
    auto load_u = [&](int k) {
        Raw r;
        r.fc = ld_stream(p2f_t + (k << 6));
        const float *b = bary_t + (k << 9);
        r.u = ld_stream(b); r.v = ld_stream(b + 64); r.jm = ld_stream(b + 128); r.a = ld_stream(b + 192);
        r.c0 = ld_stream(b + 256); r.c1 = ld_stream(b + 320); r.c2 = ld_stream(b + 384); r.T = ld_stream(b + 448);
        r.d = ld_stream(dists_t + (k << 6));
        return r;
    };
    auto masked = [&](const Raw &x, bool ok) {
        Raw r;
        r.fc = ok ? (x.fc & FRAG_FACE_MASK) : 0;
        r.u = ok ? x.u : 0.f; r.v = ok ? x.v : 0.f; r.jm = ok ? x.jm : 0.f; r.a = ok ? x.a : 0.f;
        r.c0 = ok ? x.c0 : 0.f; r.c1 = ok ? x.c1 : 0.f; r.c2 = ok ? x.c2 : 0.f; r.T = ok ? x.T : 1.f; r.d = ok ? x.d : 0.f;
        return r;
    };
    struct FaceXY { float2 v0, v1, v2; };
    auto load_xy = [&](int fc) {
        FaceXY q;
        const float *p = fv + (long long)fc * 9;
        q.v0 = make_float2(p[0], p[1]); q.v1 = make_float2(p[3], p[4]); q.v2 = make_float2(p[6], p[7]);
        return q;
    };
    // BINNED: footprint of a fragment in record form + the slot reservation of the wave's records of its layer (in the sub-range
    // `sub` of the bin: neighbouring tiles, which hit the same bins at the same time, use different cursors)
    const int sub = tile & (BIN_SUB - 1);
    auto reserve = [&](const Raw &r, bool ok) {
        BinRes R;
        R.bin = -1; R.rank = 0; R.leader = 0; R.base = 0; R.packed = 0; R.wx1 = R.wy1 = 0.f; R.lay = make_uint2(0u, 0u);
        const float wgt = r.T * r.a;
        const bool tex = ok && (wgt * gr != 0.f || wgt * gg != 0.f || wgt * gbl != 0.f) && !(A.dbg & (1 << 19));      // (1 << 19: ablation of the whole record path)
        if (__ballot(tex) == 0ull) return R;
        const int map = __float_as_int(r.jm) >> 20;
        int md[6];
        mdc.get(A, s_md, tex ? map : 0, md);
        Sample s;
        footprint_desc(r.u, r.v, md[0], md[1], md[2], md[3], md[4], md[5], s);
        const bool on = tex && bin_regular(s);
        if (on) {
            R.bin = mdc.get_extra(s_md, A.bin_base, map) + (s.r0 >> 5) * ((s.ws + 31) >> 5) + (s.c0 >> 5);
            R.packed = (int)((unsigned)(s.r0 & 31) | ((unsigned)(s.c0 & 31) << 5) | ((unsigned)(s.r0 - s.r1) << 10) | ((unsigned)(s.c1 - s.c0) << 11));
            R.wx1 = s.wx1; R.wy1 = s.wy1;
        }
        unsigned long long rem = __ballot(on);
        const unsigned long long below = (1ull << lane) - 1ull;
        while (rem) {
            const int L = __ffsll((long long)rem) - 1;
            const int b = __shfl(R.bin, L, 64);          // (v_readlane instead of this ds_bpermute: 0.46 -> 0.55 ms -- the shorter loop issues the
                                                         // returning atomics of the hot bins closer together, profiles/r03_experiments.md)
            const bool mine = on && R.bin == b;
            const unsigned long long mm = __ballot(mine);
            if (mine) { R.rank = __popcll(mm & below); R.leader = L; }
            if (lane == L && !(A.dbg & (1 << 18))) R.base = atomicAdd(A.bin_cursor + b * BIN_SUB + sub, __popcll(mm));     // (1 << 18: ablation, tools/diag)
            rem &= ~mm;
        }
        // the leaders' sub-ranges: ONE masked load behind the loop, straight into the reservation's registers (inside the loop the
        // compiler reuses those registers for the atomic's address and waits for the pending load first: vmcnt(0) per bin)
        if (on && R.leader == lane) R.lay = *(const uint2 *)(A.bin_layout + ((long long)R.bin * BIN_SUB + sub) * 2);
        return R;
    };
    Raw nxt = load(kmax > 0 ? kmax - 1 : 0, kmax > 0 && kmax - 1 < cnt);
    Raw nxt2 = nxt;
    BinRes nres;
    nres.bin = -1; nres.rank = nres.leader = nres.base = nres.packed = 0; nres.wx1 = nres.wy1 = 0.f; nres.lay = make_uint2(0u, 0u);
    FaceXY nxtq;
    nxtq.v0 = nxtq.v1 = nxtq.v2 = make_float2(0.f, 0.f);
    constexpr bool PIPE = BINNED;      // the two-deep memory schedule (for decimated maps it measures 0.366 against 0.355 ms: not used there)
    if (PIPE) {
        nxt2 = load_u(kmax > 1 ? kmax - 2 : 0);
        nxtq = load_xy(nxt.fc);
        if (BINNED && kmax > 0) nres = reserve(nxt, kmax - 1 < cnt);
    }
#pragma unroll 1
    for (int k = kmax - 1; k >= 0; --k) {
        PROF_T(t_it);
        const Raw cur = nxt;
        const BinRes cres = nres;
        const FaceXY curq = nxtq;
        const bool valid = k < cnt;
        if (PIPE) {
            nxt = masked(nxt2, k > 0 && k - 1 < cnt);
            nxt2 = load_u(k > 1 ? k - 2 : 0);
            nxtq = load_xy(nxt.fc);
            if (BINNED && k > 0) nres = reserve(nxt, k - 1 < cnt);
        } else if (k > 0) nxt = load(k - 1, k - 1 < cnt);
        const float ak = valid ? cur.a : 0.f, Tk = valid ? cur.T : 1.f;
        const float ga_ = blend_back_step(bk, Tk, ak, cur.c0, cur.c1, cur.c2, gr, gg, gbl, gA);   // (ak == 0 for an empty slot: state unchanged)
        const float ga = valid ? ga_ : 0.f;
        const float wgt = Tk * ak;
        const int jm = __float_as_int(cur.jm);
        const int j = jm & 0xfffff, map = jm >> 20;
        // geometric alpha e = exp(-max(d, 0) / sigma) (the opacity gradient is ga * e), d/d dist of the blend opacity for d >= 0
        const float e = A.sigma == 0.f ? (cur.d <= 0.f ? 1.f : 0.f) : __expf(-(cur.d > 0.f ? cur.d : 0.f) * A.inv_sigma);
        const float gd = (valid && A.sigma != 0.f && cur.d >= 0.f) ? ga * ak * -A.inv_sigma : 0.f;
        const float gc[3] = {wgt * gr, wgt * gg, wgt * gbl};
        const bool tex = valid && (gc[0] != 0.f || gc[1] != 0.f || gc[2] != 0.f) && !(BINNED && (A.dbg & (1 << 19)));
        float g7[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, (galpha && valid) ? ga * e : 0.f};
        if (alpha_direct) {
            if (g7[6] != 0.f) atomicAdd(&alpha_dir[map * ALPHA_DIRECT_SPREAD + (lane & (ALPHA_DIRECT_SPREAD - 1))], (double)g7[6]);
            g7[6] = 0.f;
        }
        // (claiming the tables' slots ahead of their use -- one LDS round trip per layer instead of three -- was measured slower: returning LDS
        // atomics that go out back to back queue behind each other's same-address replays, profiles/r04_experiments.md)
        const bool f_pre = valid && (gd != 0.f || g7[6] != 0.f) && !(A.dbg & 2);
        PROF_T(t_a);
        PROF_ADD(2, t_it, t_a);
        if (BINNED) {
            // colour -> one record in the fragment's texture bin; irregular footprints (circular wrap, clamped borders leaving the
            // bin's halo) and bin overflow take the atomic path, so the result is exact either way
            if (__ballot(tex) != 0ull) {
                // the leader's reservation: its first record in the record array, and the room left behind it in the sub-range
                const unsigned start = (unsigned)__shfl((int)(cres.lay.x + (unsigned)cres.base), cres.leader, 64);
                const int room = __shfl((int)cres.lay.y - cres.base, cres.leader, 64);
                bool pending = tex;
                if (tex && cres.bin >= 0) {
                    if (cres.rank < room) {
                        if (!(A.dbg & (1 << 17)))                      // (1 << 17: ablation of the record stores, tools/diag)
                            st_record(A.bin_records, (unsigned long long)(start + (unsigned)cres.rank), cres.packed, cres.wx1, cres.wy1, gc[0], gc[1], gc[2]);
                        pending = false;
                    }
                }
                if (pending) {
                    int md[6];
                    mdc.get(A, s_md, map, md);
                    Sample s;
                    footprint_desc(cur.u, cur.v, md[0], md[1], md[2], md[3], md[4], md[5], s);
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        unsafeAtomicAdd(gmaps + s.a00 + ch, gc[ch] * s.w00);
                        unsafeAtomicAdd(gmaps + s.a01 + ch, gc[ch] * s.w01);
                        unsafeAtomicAdd(gmaps + s.a10 + ch, gc[ch] * s.w10);
                        unsafeAtomicAdd(gmaps + s.a11 + ch, gc[ch] * s.w11);
                    }
                }
            }
        } else if (__ballot(tex) != 0ull) {
            // colour -> texels of the decimated map: the bilinear footprint's texels that fall into the same stored cell are merged
            int md[6];
            mdc.get(A, s_md, valid ? map : 0, md);
            Sample s;
            footprint_desc(cur.u, cur.v, md[0], md[1], md[2], md[3], md[4], md[5], s);
            float w00 = s.w00, w01 = s.w01, w10 = s.w10, w11 = s.w11;
            if (s.a01 == s.a00) { w00 += w01; w01 = 0.f; }
            if (s.a10 == s.a00) { w00 += w10; w10 = 0.f; }
            if (s.a11 == s.a00) { w00 += w11; w11 = 0.f; }
            else if (s.a11 == s.a01) { w01 += w11; w11 = 0.f; }
            else if (s.a11 == s.a10) { w10 += w11; w11 = 0.f; }
            const int ad[4] = {s.a00, s.a01, s.a10, s.a11};
            const float wt[4] = {w00, w01, w10, w11};
#ifdef DBW_STATS_BWD
            {
                int dk, rp, taps = 0, tap_lanes = 0, tap_replays = 0;
                for (int q = 0; q < 4; ++q) {
                    const bool onq = tex && wt[q] != 0.f;
                    key_stats((int)((unsigned)ad[q] / 3u), onq, dk, rp);
                    taps += __ballot(onq) != 0ull ? 1 : 0;
                    tap_lanes += __popcll(__ballot(onq));
                    tap_replays += rp;
                    if (q == 0) STAT_ADD(2, dk);
                }
                STAT_ADD(1, tap_lanes); STAT_ADD(3, taps); STAT_ADD(7, tap_replays);
            }
#endif
            // Tap 0, then the lane's first other tap that carries weight, as wave-wide passes (merge + table); what is left after those, by
            // the lanes that have it.  On a decimated map a footprint has a second texel only where it crosses a cell border (one pixel in
            // four at decimation 8) and four only where it crosses one in x AND in y (1.6 %) -- but some lane of the 64 does, nearly always,
            // so four wave-wide passes per layer used to run for 60 updates of which 45 belong to tap 0: a third of this kernel's
            // instructions, and four table round trips in a row where two (and half a third) do.
            const int f = wt[1] != 0.f ? 1 : (wt[2] != 0.f ? 2 : 3);
            const int a2[2] = {ad[0], f == 1 ? ad[1] : (f == 2 ? ad[2] : ad[3])};
            const float w2[2] = {wt[0], f == 1 ? wt[1] : (f == 2 ? wt[2] : wt[3])};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float v3[3] = {gc[0] * w2[q], gc[1] * w2[q], gc[2] * w2[q]};
                bool on = tex && w2[q] != 0.f && !(A.dbg & 1);
                if (TEX_MERGE > 0 && __ballot(on) != 0ull) lane_merge<3, TEX_MERGE>((int)((unsigned)a2[q] / 3u), on, v3);
                if (on) tex_agg.add(gmaps, (int)((unsigned)a2[q] / 3u), v3);      // (dbg 1, 2, 16, 128: ablations, tools/diag)
            }
            const bool rest = tex && !(A.dbg & 1) && ((f == 1 && (wt[2] != 0.f || wt[3] != 0.f)) || (f == 2 && wt[3] != 0.f));
            if (__ballot(rest) != 0ull) {
#pragma unroll
                for (int q = 2; q < 4; ++q)
                    if (rest && q > f && wt[q] != 0.f) {
                        const float v3[3] = {gc[0] * wt[q], gc[1] * wt[q], gc[2] * wt[q]};
                        tex_agg.add(gmaps, (int)((unsigned)ad[q] / 3u), v3);
                    }
            }
        }
        PROF_T(t_b);
        PROF_ADD(4, t_a, t_b);
        // distance -> the two vertices of the closest edge; opacity; one table update per fragment
        if (__ballot(gd != 0.f) != 0ull && !(A.dbg & 16)) {
            f2 v0, v1, v2;
            if (PIPE) { v0 = f2{curq.v0.x, curq.v0.y}; v1 = f2{curq.v1.x, curq.v1.y}; v2 = f2{curq.v2.x, curq.v2.y}; }
            else {
                // (round 4, by ablation: this round trip per layer is 44 of the kernel's 361 us, the texel table 124, the face table 85.
                // Requesting the vertices in front of the texel path -- so that its table updates run while they travel -- takes 13 more
                // registers, 8 -> 6 waves per SIMD: 0.361 -> 0.361 ms; capped at 64 registers it spills: 0.40 ms.  Even the ablation switch
                // that measured it cost 5 registers and a wave per SIMD, and is gone again)
                const float *q = fv + (long long)(valid ? cur.fc : 0) * 9;
                v0 = f2{q[0], q[1]}; v1 = f2{q[3], q[4]}; v2 = f2{q[6], q[7]};
            }
            float t01, t02, t12;
            const float e01 = seg_dist_t(pndc, v0, v1, t01), e02 = seg_dist_t(pndc, v0, v2, t02), e12 = seg_dist_t(pndc, v1, v2, t12);
            const int sel = (e01 <= e02 && e01 <= e12) ? 0 : ((e02 <= e01 && e02 <= e12) ? 1 : ((e12 <= e01 && e12 <= e02) ? 2 : 3));
            const f2 ea = sel == 2 ? v1 : v0, eb = sel == 0 ? v1 : v2;
            const float tt = sel == 0 ? t01 : (sel == 1 ? t02 : t12);
            const float qx = (1.f - tt) * ea.x + tt * eb.x, qy = (1.f - tt) * ea.y + tt * eb.y;
            const float g = sel == 3 ? 0.f : gd;
            const float gax = g * (1.f - tt) * 2.f * (qx - pndc.x), gay = g * (1.f - tt) * 2.f * (qy - pndc.y);
            const float gbx = g * tt * 2.f * (qx - pndc.x), gby = g * tt * 2.f * (qy - pndc.y);
            // sel 0: (v0, v1), 1: (v0, v2), 2: (v1, v2)
            g7[0] = sel <= 1 ? gax : 0.f; g7[1] = sel <= 1 ? gay : 0.f;
            g7[2] = sel == 0 ? gbx : (sel == 2 ? gax : 0.f); g7[3] = sel == 0 ? gby : (sel == 2 ? gay : 0.f);
            g7[4] = sel >= 1 ? gbx : 0.f; g7[5] = sel >= 1 ? gby : 0.f;
        }
        PROF_T(t_c);
        PROF_ADD(5, t_b, t_c);
#ifdef DBW_STATS_BWD
        {
            int dk, rp;
            const bool f_on = valid && (gd != 0.f || g7[6] != 0.f);
            key_stats(cur.fc, f_on, dk, rp);
            STAT_ADD(0, 1); STAT_ADD(4, __popcll(__ballot(f_on))); STAT_ADD(5, dk); STAT_ADD(6, rp);
        }
#endif
        bool f_on = f_pre;
        constexpr int FM = BINNED ? DBW_FACE_MERGE_BINNED : FACE_MERGE;
        if (FM > 0 && __ballot(f_on) != 0ull) lane_merge<7, FM>(cur.fc, f_on, g7);
        if (f_on) {
            const int aidx = A.faces_alpha ? (int)alpha_grad_index(A, n, j, map) : 0;
            fa_agg.add(gfv, galpha, cur.fc, aidx, g7);
        }
        PROF_T(t_d);
        PROF_ADD(6, t_c, t_d);
    }
    PROF_T(t_end);
    // The workgroup's tables are flushed by whichever of its waves finishes LAST; the others leave at once.  The four tiles of a
    // workgroup hold very different numbers of layers, and behind a closing barrier the waves of the light tiles kept their
    // registers and wave slots while waiting for the heaviest one -- 19 % (decimated maps), 20 % and 33 % (full-resolution phases) of
    // the wave time (tools/bwd_cycles.py).  A wave's LDS instructions execute in the order it issued them, so its table updates
    // precede its increment of the counter, and the wave that reads NW - 1 there sees the updates of all the others.
    // (decimated maps: the kernel is bound by the LDS atomic unit and by the workgroups' LDS allocations, waiting waves cost nothing
    // there, and four waves scan the 512 + 128 slots faster than one: the closing barrier stays, 0.39 vs 0.42 ms)
    int last = 0;
    if (LAST_WAVE_FLUSH) {
        if (lane == 0) last = atomicAdd(&s_done, 1) == NT / 64 - 1 ? 1 : 0;
        last = __builtin_amdgcn_readfirstlane(last);
    } else __syncthreads();
    PROF_T(t_sync);
    PROF_ADD(3, t_end, t_sync);
    if ((LAST_WAVE_FLUSH && !last) || (A.dbg & 128)) return;
    const int f_tid = LAST_WAVE_FLUSH ? lane : (int)threadIdx.x, f_n = LAST_WAVE_FLUSH ? 64 : NT;
    if (!BINNED) tex_agg.flush(gmaps, f_tid, f_n);
    fa_agg.flush(gfv, galpha, f_tid, f_n);
    if (alpha_direct)            // -> the DBW_ALPHA_SPREAD partial sums of every map (alpha_grad_index), spread by workgroup
        for (int i = f_tid; i < n_maps * ALPHA_DIRECT_SPREAD; i += f_n) {
            const float x = (float)alpha_dir[i];
            if (x != 0.f) unsafeAtomicAdd(galpha + (long long)(i / ALPHA_DIRECT_SPREAD) * DBW_ALPHA_SPREAD + ((blockIdx.x * ALPHA_DIRECT_SPREAD + i) & (DBW_ALPHA_SPREAD - 1)), x);
        }
    PROF_T(t_fl);
    PROF_ADD(1, t_sync, t_fl);                 // (flushes)
    PROF_ADD(7, t_begin, t_fl);
    BWD_CLOCK_END(n, kmax);
}

// One workgroup per (texture bin, BIN_SUB_PER_WG of its record sub-ranges): accumulate the records into a (32+1)x(32+1) texel LDS tile
// (1-texel halo: row -1, column +32), then add the tile to the gradient map.  bin_info (nbins,4) = {offset of the map in floats,
// stored width ws, stored height hs, tile_y << 16 | tile_x}.
// The tile is accumulated in 32-bit FIXED POINT with a block exponent (round 6; fp64 before).  What the LDS atomic unit of gfx950 does
// (profiles/r01_lds_atomic_ubench.txt, clk per wave-instruction): ds_add_f32 193 whatever the addresses (one lane per ~3 clk), ds_add_f64
// 8 with distinct addresses but 20 / 44 with two / four lanes per address, ds_add_u32 4 / 6 / 14 -- and the records of a bin arrive as
// 8x8 pixel patches whose neighbours share texels: measured 34 clk per fp64 atomic here, 61 % of this kernel (218 -> 91 us without them;
// tools/diag/r06_trace.sh).  Integer adds are native to the LDS banks.  Exactness: the tile holds value * 2^(SH - e) as int32, e = the
// block exponent: |g| < 2^e for every record accumulated so far -- taken per staged batch (a wave maximum next to the staging loads), and
// when a batch raises it the tile is shifted down first.  Bilinear weights sum to one, so a texel gains less than 2^SH units per record
// whatever the records hold: the workgroup keeps a BUDGET of records it may still add blindly -- (2^31 - 1 - B) >> SH with B a bound on
// |tile|, 2047 records for an empty tile -- and when the budget is used up it looks (one pass over the tile: B = its largest entry, in
// practice a few dozen addends' worth) and goes on, or -- never seen -- flushes and clears first.  No overflow, ever; every addend is
// rounded to 2^(e - SH - 1) = 2^-21 of the bin's largest pixel gradient (measured against the fp64 tile: 2e-6 of a texture gradient's
// largest entry; SH = 17 with a fixed 8192-record budget read 1.4e-5).  A non-finite gradient poisons the bin's first texel instead of
// being quantised away.  Integer addition is associative: between two rescalings the tile does not depend on the order of the records.
// Records arrive in runs of up to 64 written by one wave of the backward, so each batch of BIN_STAGE records is staged in LDS with
// coalesced loads and re-read transposed: the 64 lanes of an instruction then hold records BIN_STAGE/64 apart.
#ifndef DBW_BIN_STAGE
#define DBW_BIN_STAGE 512       // (round 3: 1024 -> 512 records per batch and 4 -> 2 sub-ranges per workgroup: three workgroups per CU instead of
#endif                          // two, eight per bin: 0.268 -> 0.236 ms at config 2; 256 records: 0.234.  Round 6: 30 KB of LDS with the int32 tile)
constexpr int BIN_STAGE = DBW_BIN_STAGE, BIN_PER_THREAD = BIN_STAGE / 256, BIN_LANE_STRIDE = BIN_STAGE / 64;
typedef int bin_fix_t;          // (a 64-bit tile -- ds_add_u64, no budget needed -- measured 196 us against this one's 138 and fp64's 233)
constexpr int BIN_FIX_SH = 20, BIN_FIX_EMIN = -100;
__global__ __launch_bounds__(256) void texbin_reduce_kernel(const int *__restrict__ bin_info, const int *__restrict__ cursor,
                                                            const int4 *__restrict__ records, int cap, const unsigned *__restrict__ layout,
                                                            float *__restrict__ gmaps) {
    __shared__ bin_fix_t tile[33 * 33 * 3];
    __shared__ int2 stage[BIN_STAGE * 3 + BIN_STAGE / BIN_LANE_STRIDE];   // 24 B records; one int2 of padding per lane stride: conflict-free reads
    __shared__ float s_wmax[4];
#ifdef DBW_REDUCE_SCRAMBLE        // (experiment: consecutive workgroups are dealt round robin to the shader engines -- do the hot bins form a comb?)
    const unsigned bx = blockIdx.x, nb_ = gridDim.x;
    const int bin = bx < (nb_ & ~63u) ? (int)((bx & ~63u) | ((__brev(bx & 63u) >> 26) ^ (((bx >> 6) * 2654435761u) >> 26))) : (int)bx;
    const int sub0 = blockIdx.y * BIN_SUB_PER_WG;
#else
    // (the bins in scrambled order, raster_math.h: window_scramble -- consecutive workgroups are dealt to the XCD's shader engines round robin,
    // a bin holds anything between no record and tens of thousands, and hot bins at a regular stride of the bin table piled up on one
    // engine: 128 -> 114 us at config 2, config 5's full-resolution step 22.05 -> 20.75 ms; profiles/r06_experiments.md)
    const int bin = (int)window_scramble(blockIdx.x, gridDim.x), sub0 = blockIdx.y * BIN_SUB_PER_WG;
#endif
    int n_sub[BIN_SUB_PER_WG], total = 0;
    unsigned first[BIN_SUB_PER_WG];
#pragma unroll
    for (int g = 0; g < BIN_SUB_PER_WG; ++g) {
        const int ci = bin * BIN_SUB + sub0 + g;
        first[g] = layout[ci * 2];
        n_sub[g] = min(cursor[ci], (int)layout[ci * 2 + 1]);
        total += n_sub[g];
    }
    if (total == 0) return;
    for (int i = threadIdx.x; i < 33 * 33 * 3; i += 256) tile[i] = 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long off = bin_info[bin * 4];
    const int ws = bin_info[bin * 4 + 1], hs = bin_info[bin * 4 + 2];
    const int ty = bin_info[bin * 4 + 3] >> 16, tx = bin_info[bin * 4 + 3] & 0xffff;
    int e_tile = -1000;            // block exponent of the tile (none yet); every thread holds the same value
    int budget = (int)(0x7fffffffu >> BIN_FIX_SH);          // records that may still be added without looking at the tile (see above)
    // tile -> gradient map (+ clear): at the end, and whenever CHUNK records have gone in
    auto flush = [&](bool clear) {
        const float unit = __int_as_float((unsigned)(e_tile - BIN_FIX_SH + 127) << 23);      // 2^(e - SH): e >= EMIN keeps it a normal float
        for (int i = threadIdx.x; i < 33 * 33 * 3; i += 256) {
            const bin_fix_t q = tile[i];
            if (clear) tile[i] = 0;
            if (q == 0) continue;
            const float v = (float)q * unit;
            const int ch = i % 3, lc = (i / 3) % 33, lr = i / 99;
            const int r = ty * 32 - 1 + lr, c = tx * 32 + lc;
#ifndef DBW_REDUCE_NOFLUSH
            if (r >= 0 && r < hs && c < ws) unsafeAtomicAdd(gmaps + off + ((long long)r * ws + c) * 3 + ch, v);
#endif
        }
    };
    auto accumulate = [&](int m) {
        const float scale = __int_as_float((unsigned)(BIN_FIX_SH - e_tile + 127) << 23);     // 2^(SH - e)
#pragma unroll 2
        for (int j = 0; j < BIN_PER_THREAD; ++j) {
            const int r = lane * BIN_LANE_STRIDE + wv * BIN_PER_THREAD + j;
            if (r >= m) continue;
            const int so = r * 3 + r / BIN_LANE_STRIDE;
            const int2 a = stage[so], b = stage[so + 1], c = stage[so + 2];
            const unsigned p = (unsigned)a.x;
            const int lr0 = (int)(p & 31) + 1, lc0 = (int)((p >> 5) & 31), dr = (int)((p >> 10) & 1), dc = (int)((p >> 11) & 1);
            const float wx1 = __int_as_float(a.y), wy1 = __int_as_float(b.x);
            const float g0 = __int_as_float(b.y), g1 = __int_as_float(c.x), g2 = __int_as_float(c.y);
            const float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
            const float w[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
            const int idx[4] = {(lr0 * 33 + lc0) * 3, (lr0 * 33 + lc0 + dc) * 3, ((lr0 - dr) * 33 + lc0) * 3, ((lr0 - dr) * 33 + lc0 + dc) * 3};
#ifndef DBW_REDUCE_NOATOMIC
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int q0 = __float2int_rn(g0 * w[q] * scale), q1 = __float2int_rn(g1 * w[q] * scale), q2 = __float2int_rn(g2 * w[q] * scale);
                if (q0) atomicAdd(&tile[idx[q]], q0);
                if (q1) atomicAdd(&tile[idx[q] + 1], q1);
                if (q2) atomicAdd(&tile[idx[q] + 2], q2);
            }
#else
            if (w[0] * g0 + w[1] * g1 + w[2] * g2 + w[3] == 123.456f) tile[idx[0]] = 1;        // (ablation build: the arithmetic stays, the LDS atomics go)
#endif
        }
    };
    bool poisoned = false;
#pragma unroll 1
    for (int g = 0; g < BIN_SUB_PER_WG; ++g) {
    const int n = n_sub[g];
    const int2 *rec = (const int2 *)records + (long long)first[g] * 3;
    for (int sb = 0; sb < n; sb += BIN_STAGE) {
        const int m = min(n - sb, BIN_STAGE);
        __syncthreads();                                       // previous batch consumed (first pass: tile cleared)
        if (m > budget) {
            // the blind budget is used up: B = the tile's largest |entry| (wave maxima through s_wmax, which the batch below rewrites
            // behind a barrier of its own)
            int bmax = 0;
            for (int i = threadIdx.x; i < 33 * 33 * 3; i += 256) { const int q = (int)tile[i]; bmax = max(bmax, q < 0 ? -q : q); }
            bmax = max(bmax, __builtin_amdgcn_update_dpp(0, bmax, 0x111, 0xf, 0xf, false));
            bmax = max(bmax, __builtin_amdgcn_update_dpp(0, bmax, 0x112, 0xf, 0xf, false));
            bmax = max(bmax, __builtin_amdgcn_update_dpp(0, bmax, 0x114, 0xf, 0xf, false));
            bmax = max(bmax, __builtin_amdgcn_update_dpp(0, bmax, 0x118, 0xf, 0xf, false));
            bmax = max(bmax, __builtin_amdgcn_update_dpp(0, bmax, 0x142, 0xf, 0xf, false));
            bmax = max(bmax, __builtin_amdgcn_update_dpp(0, bmax, 0x143, 0xf, 0xf, false));
            if (lane == 63) s_wmax[wv] = __int_as_float(bmax);
            __syncthreads();
            const unsigned B = (unsigned)max(max(__float_as_int(s_wmax[0]), __float_as_int(s_wmax[1])), max(__float_as_int(s_wmax[2]), __float_as_int(s_wmax[3])));
            budget = (int)((0x7fffffffu - B) >> BIN_FIX_SH);
            __syncthreads();
            if (m > budget) { flush(true); budget = (int)(0x7fffffffu >> BIN_FIX_SH); __syncthreads(); }
        }
        float gm = 0.f;
#pragma unroll
        for (int it = 0; it < BIN_PER_THREAD; ++it) {
            const int r = it * 256 + threadIdx.x;
            if (r < m) {
                const int2 a = rec[(sb + r) * 3], b = rec[(sb + r) * 3 + 1], c = rec[(sb + r) * 3 + 2];
                const int so = r * 3 + r / BIN_LANE_STRIDE;
                stage[so] = a; stage[so + 1] = b; stage[so + 2] = c;
                // (as unsigned bit patterns: a NaN or an infinity is then the largest of all and ends up in the batch's exponent)
                gm = __uint_as_float(max(__float_as_uint(gm), max((unsigned)b.y & 0x7fffffffu, max((unsigned)c.x & 0x7fffffffu, (unsigned)c.y & 0x7fffffffu))));
            }
        }
        {   // wave maximum of the bit patterns (non-negative ints order like the floats they encode)
            int x = (int)__float_as_uint(gm);
            x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));
            x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));
            x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false));
            x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false));
            x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xf, 0xf, false));
            x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xf, 0xf, false));
            if (lane == 63) s_wmax[wv] = __int_as_float(x);
        }
        __syncthreads();
        const unsigned mb = max(max(__float_as_uint(s_wmax[0]), __float_as_uint(s_wmax[1])), max(__float_as_uint(s_wmax[2]), __float_as_uint(s_wmax[3])));
        const int E = (int)(mb >> 23);                         // biased exponent of the batch's largest |g|: |g| < 2^(E - 126)
        if (E == 255) poisoned = true;                         // infinity / NaN among the gradients
        const int e_b = max(min(E, 254) - 126, BIN_FIX_EMIN);
        if (e_b > e_tile) {                                    // (uniform: every thread read the same four words)
            if (e_tile > -1000) {
                const int sh = min(e_b - e_tile, (int)sizeof(bin_fix_t) * 8 - 1);
                for (int i = threadIdx.x; i < 33 * 33 * 3; i += 256) { const bin_fix_t q = tile[i]; if (q) tile[i] = (q + ((bin_fix_t)1 << (sh - 1))) >> sh; }
                __syncthreads();
            }
            e_tile = e_b;
        }
        if (mb != 0u) { accumulate(m); budget -= m; }          // (a batch of zero gradients adds nothing)
    }
    }
    __syncthreads();
    if (e_tile > -1000) flush(false);
    if (poisoned && threadIdx.x == 0) unsafeAtomicAdd(gmaps + off, __int_as_float(0x7fc00000));
}


thread_local int g_dbg_flags = 0;      // (per host thread, like the error text: dbw_debug_set_flags)
#ifdef DBW_PROFILE_BWD
extern "C" void dbw_debug_read_profile(unsigned long long *out8, int reset) {
    static unsigned long long *host = nullptr;
    const size_t bytes = (size_t)PROF_BLOCKS * 8 * sizeof(unsigned long long);
    if (!host) host = (unsigned long long *)malloc(bytes);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prof), bytes);
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    for (size_t b = 0; b < (size_t)PROF_BLOCKS; ++b)
        for (int i = 0; i < 8; ++i) out8[i] += host[b * 8 + i];
    if (reset) { memset(host, 0, bytes); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), host, bytes); }
}
#endif

int fill_args(ShadeArgs &A, const int32_t *pix_to_face, const float *bary, const float *dists, const int32_t *c2o,
              const int32_t *clip_code, const float *clip_w, int Fc_stride, const float *face_uvs,
              const int32_t *face_map, const int32_t *map_desc, const float *maps, const float *faces_alpha,
              int alpha_len, int N, int H, int W, int K, int F, float sigma, const float *background3) {
    DBW_REQUIRE(pix_to_face && bary && dists && face_uvs && face_map && map_desc && maps, "null pointer");
    DBW_REQUIRE((c2o && clip_code && clip_w) || (!c2o && !clip_code && !clip_w), "c2o/clip_code/clip_w: all or none");
    DBW_REQUIRE(N >= 0 && H > 0 && W > 0 && K > 0 && F > 0, "bad size");
    DBW_REQUIRE(!faces_alpha || alpha_len == F || (long long)alpha_len == (long long)N * F || alpha_len < 0,
                "faces_alpha length must be F, N*F, or -(number of maps) for one opacity per texture map");
    A.p2f = pix_to_face; A.bary = bary; A.dists = dists; A.c2o = c2o; A.code = clip_code; A.cw = clip_w;
    A.Fc_stride = Fc_stride; A.face_uvs = face_uvs; A.face_map = face_map; A.map_desc = map_desc; A.maps = maps;
    A.faces_alpha = faces_alpha; A.alpha_len = alpha_len; A.N = N; A.H = H; A.W = W; A.K = K; A.F = F; A.sigma = sigma; A.inv_sigma = sigma != 0.f ? 1.f / fabsf(sigma) : 0.f;
    for (int i = 0; i < 3; ++i) A.bg[i] = background3 ? background3[i] : 0.f;
    {       // pixel -> NDC constants (SURVEY A.1 NonSquarePixToNdc): IEEE single divisions, the same bits as the device's
        float rx = 2.0f, ry = 2.0f;
        if (W > H) rx = ((float)W * rx) / (float)H;
        if (H > W) ry = ((float)H * ry) / (float)W;
        A.ndc[0] = rx; A.ndc[1] = rx / 2.0f; A.ndc[2] = ry; A.ndc[3] = ry / 2.0f;
    }
    A.dbg = g_dbg_flags;
    A.agg = 0;
    A.tiled = 0;
    A.bin_base = nullptr; A.bin_cursor = nullptr; A.bin_records = nullptr; A.bin_cap = 0; A.bin_layout = nullptr;
    A.gscale = nullptr; A.geom_begin = 0; A.env_img = nullptr; A.target = nullptr; A.mse_scale = 0.f; A.loss_part = nullptr; A.g_fg = nullptr; A.g_env = nullptr;
    A.img_tiled = 0;
    A.lean_grads = 0;
    A.rec_out = nullptr; A.grad_rec = nullptr;
    A.sync_flag = nullptr; A.sync_val = 0;
    return DBW_OK;
}

}  // namespace

extern "C" int dbw_shade_blend_fwd(const int32_t *pix_to_face, const float *bary, const float *dists,
                                   const int32_t *c2o, const int32_t *clip_code, const float *clip_w, int Fc_stride,
                                   const float *face_uvs, const int32_t *face_map, const int32_t *map_desc,
                                   const float *maps, const float *faces_alpha, int alpha_len, int N, int H, int W,
                                   int K, int F, float sigma, const float *background3, float *image,
                                   dbw_stream_t stream) {
    ShadeArgs A;
    int rc = fill_args(A, pix_to_face, bary, dists, c2o, clip_code, clip_w, Fc_stride, face_uvs, face_map, map_desc,
                       maps, faces_alpha, alpha_len, N, H, W, K, F, sigma, background3);
    if (rc) return rc;
    DBW_REQUIRE(image, "null pointer");
    if (N == 0) return DBW_OK;
    const long long total = (long long)N * ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    hipLaunchKernelGGL(shade_blend_fwd_kernel, dim3(dbw_xcd_grid(total)), dim3(NT), 0, (hipStream_t)stream, A, total, image);
    return dbw_check_launch("shade_blend_fwd_kernel");
}

int dbw_fill_shade_args(ShadeArgs &A, const int32_t *pix_to_face, const float *bary, const float *dists, const int32_t *c2o,
                        const int32_t *clip_code, const float *clip_w, int Fc_stride, const float *face_uvs,
                        const int32_t *face_map, const int32_t *map_desc, const float *maps, const float *faces_alpha,
                        int alpha_len, int N, int H, int W, int K, int F, float sigma, const float *background3) {
    return fill_args(A, pix_to_face, bary, dists, c2o, clip_code, clip_w, Fc_stride, face_uvs, face_map, map_desc, maps, faces_alpha,
                     alpha_len, N, H, W, K, F, sigma, background3);
}


// ---- the hard single-layer pass (sky + ground), specialised: hard uv-fragments (layout 3) ---------------------------------------------
// K = 1, sigma = 0, no learned opacity, texel gradients through the LDS hash (magnified / decimated maps).  The forward leaves (clipped
// face, u, v, face | map) per pixel; this kernel loads them with the pixel's image gradient -- four coalesced loads -- and runs the
// per-pixel backward of env_bwd.h (the same function the training step's fused forward runs in its epilogue).
__global__ __launch_bounds__(NT) void render_bwd_hard_kernel(ShadeArgs A, long long total_blocks, const float *__restrict__ gimg,
                                                            float *__restrict__ gmaps, const float *__restrict__ fv,
                                                            float *__restrict__ gfv, int want_bary, int persp) {
    extern __shared__ __attribute__((aligned(16))) float s_hard[];
    HardTexAgg tex_agg;
    HardFaceAgg face_agg;
    tex_agg.bind(s_hard);
    tex_agg.clear(threadIdx.x, NT);
    face_agg.bind((char *)s_hard + HardTexAgg::BYTES);
    face_agg.clear(threadIdx.x, NT);
    int n, xi, yi;
    if (!pixel_of_block(A, total_blocks, n, xi, yi)) return;
    __syncthreads();
    const bool in_img = xi < A.W && yi < A.H;
    int fc = -1, jm = 0;
    float u = 0.f, v = 0.f, gr = 0.f, gg = 0.f, gbl = 0.f;
    const float gs = A.gscale ? *A.gscale : 1.f;
    if (A.tiled && A.img_tiled) {
        // 8x8-tile planar fragments and images (the training step): a wave's quadrant is ONE tile -- scalar base + lane, immediate plane offsets
        const int tiles_x = (A.W + 7) >> 3, tiles_y = (A.H + 7) >> 3;
        const int lane = threadIdx.x & 63;
        const bool real = (yi >> 3) < tiles_y && (xi >> 3) < tiles_x;       // (a 16x16 workgroup at the border can hold quadrants outside the tile grid)
        const int tile = __builtin_amdgcn_readfirstlane(real ? (n * tiles_y + (yi >> 3)) * tiles_x + (xi >> 3) : n * tiles_y * tiles_x);
        if (in_img && real) {
            fc = ld_stream(A.p2f + ((long long)tile << 6) + lane);
            if (fc >= 0) {
                const float *b = A.bary + ((long long)tile * 3 << 6) + lane, *gi = gimg + ((long long)tile * 4 << 6) + lane;
                u = ld_stream(b); v = ld_stream(b + 64); jm = __float_as_int(ld_stream(b + 128));
                gr = ld_stream(gi) * gs; gg = ld_stream(gi + 64) * gs; gbl = ld_stream(gi + 128) * gs;
            }
        }
    } else {
        const FragAddr o = frag_addr(A, n, yi, xi, 0);
        fc = in_img ? ld_stream(A.p2f + o.s) : -1;
        if (fc >= 0) {
            u = ld_stream(A.bary + o.b); v = ld_stream(A.bary + o.b + o.bstride); jm = __float_as_int(ld_stream(A.bary + o.b + 2 * o.bstride));
            const ImgAddr ia = img_addr(A, n, yi, xi, 4);
            const float *gi = gimg + ia.base;
            gr = ld_stream(gi) * gs; gg = ld_stream(gi + ia.cstride) * gs; gbl = ld_stream(gi + 2 * ia.cstride) * gs;
        }
    }
    const bool valid = fc >= 0;
    const float gc[3] = {gr, gg, gbl};               // blend weight of a hard fragment = 1
    EnvBwdArgs E;
    E.map_desc = A.map_desc; E.maps = A.maps; E.face_uvs = A.face_uvs; E.code = A.c2o ? A.code : nullptr; E.cw = A.cw; E.fv = fv; E.gmaps = gmaps; E.gfv = gfv;
    E.H = A.H; E.W = A.W; E.geom_begin = A.geom_begin; E.want_bary = want_bary; E.persp = persp;
    for (int i = 0; i < 4; ++i) E.ndc[i] = A.ndc[i];
    env_bwd_pixel(E, tex_agg, face_agg, valid, fc, u, v, jm, gc, xi, yi);
    __syncthreads();
    tex_agg.flush(gmaps, threadIdx.x, NT);
    face_agg.flush(gfv, threadIdx.x, NT);
}

__global__ void flag_store_kernel(unsigned *flag, unsigned v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

static int launch_bwd(ShadeArgs &A, int N, int H, int W, int K, const float *grad_image, float *grad_maps,
                      float *grad_faces_alpha, float *grad_dists, float *grad_bary, int lds_aggregate, const float *fv,
                      float *gfv, int want_bary, int persp, hipStream_t s) {
    if (K > DBW_MAX_FACES_PER_PIXEL) {
        dbw_set_error("shade/blend backward: faces_per_pixel=%d > %d", K, DBW_MAX_FACES_PER_PIXEL);
        return DBW_ERR_UNSUPPORTED;
    }
    if (N == 0) return DBW_OK;
    const long long total = (long long)N * ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    size_t lds = (size_t)K * NT * sizeof(float);
    const bool fused = gfv != nullptr;
    A.agg = (lds_aggregate && !(g_dbg_flags & 8)) ? 3 : ((fused && grad_faces_alpha && !(g_dbg_flags & 8)) ? 2 : 0);
    if (A.agg & 1) lds += TexAgg::BYTES;
    if (A.agg & 2) lds += AlphaAgg::BYTES;
    if (fused) lds += FaceAgg::BYTES;
    if (fused && A.bin_records) lds += (size_t)(4 * BIN_SLOTS + K * NT) * sizeof(int);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute((const void *)shade_blend_bwd_kernel<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void *)shade_blend_bwd_kernel<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void *)shade_blend_bwd_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void *)shade_blend_bwd_kernel<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            dbw_set_error("shade/blend backward: cannot raise the dynamic LDS limit");
            return DBW_ERR_LAUNCH;
        }
        raised = true;
    }
    // the training path's soft pass: uv-fragments, detached barycentrics; texel gradients through the LDS table (decimated maps) or
    // through texture-space bins (full-resolution maps) -> the specialised kernel
    // (sigma < 0 -- the sigmoid opacity of clip_inside = False, no shipped config -- runs on the general kernel)
    const bool uv_kernel = fused && K > 1 && A.tiled == 2 && !want_bary && ((A.agg & 1) || A.bin_records) && !(g_dbg_flags & (1 << 16)) && A.sigma >= 0.f;
    if (A.sync_flag && !uv_kernel) {       // (only the specialised kernel carries the step's signal: a launch of its own in front of any other)
        hipLaunchKernelGGL(flag_store_kernel, dim3(1), dim3(1), 0, s, A.sync_flag, A.sync_val);
        A.sync_flag = nullptr;
    }
    if (uv_kernel) {
        static bool raised_uv = false;
        if (!raised_uv) {
            if (hipFuncSetAttribute((const void *)render_bwd_uv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess ||
                hipFuncSetAttribute((const void *)render_bwd_uv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess) {
                dbw_set_error("shade/blend backward: cannot raise the dynamic LDS limit");
                return DBW_ERR_LAUNCH;
            }
            raised_uv = true;
        }
        if (A.bin_records)
            hipLaunchKernelGGL(render_bwd_uv_kernel<true>, dim3(bwd_interleave_grid(total)), dim3(NT), FaceAlphaAgg::BYTES + ALPHA_DIRECT_BYTES, s, A, total, grad_image, grad_maps,
                               grad_faces_alpha, fv, gfv);
        else
            hipLaunchKernelGGL(render_bwd_uv_kernel<false>, dim3(bwd_interleave_grid(total)), dim3(NT), TexAgg::BYTES + FaceAlphaAgg::BYTES + ALPHA_DIRECT_BYTES, s, A, total,
                               grad_image, grad_maps, grad_faces_alpha, fv, gfv);
        return dbw_check_launch("render_bwd_uv_kernel");
    }
    if (A.tiled == 3) {       // hard uv-fragments: validated by the caller (K == 1, sigma == 0, no opacities, LDS aggregation)
        hipLaunchKernelGGL(render_bwd_hard_kernel, dim3(dbw_xcd_grid(total)), dim3(NT), HardTexAgg::BYTES + HardFaceAgg::BYTES, s, A, total, grad_image,
                           grad_maps, fv, gfv, want_bary, persp);
        return dbw_check_launch("render_bwd_hard_kernel");
    }
    if (fused && A.bin_records)
        hipLaunchKernelGGL((shade_blend_bwd_kernel<true, true, false>), dim3(dbw_xcd_grid(total)), dim3(NT), lds, s, A, total, grad_image, grad_maps,
                           grad_faces_alpha, nullptr, nullptr, fv, gfv, want_bary, persp);
    else if (fused && K == 1)
        hipLaunchKernelGGL((shade_blend_bwd_kernel<true, false, true>), dim3(dbw_xcd_grid(total)), dim3(NT), lds, s, A, total, grad_image, grad_maps,
                           grad_faces_alpha, nullptr, nullptr, fv, gfv, want_bary, persp);
    else if (fused)
        hipLaunchKernelGGL((shade_blend_bwd_kernel<true, false, false>), dim3(dbw_xcd_grid(total)), dim3(NT), lds, s, A, total, grad_image, grad_maps,
                           grad_faces_alpha, nullptr, nullptr, fv, gfv, want_bary, persp);
    else
        hipLaunchKernelGGL((shade_blend_bwd_kernel<false, false, false>), dim3(dbw_xcd_grid(total)), dim3(NT), lds, s, A, total, grad_image, grad_maps,
                           grad_faces_alpha, grad_dists, grad_bary, nullptr, nullptr, 0, 1);
    return dbw_check_launch("shade_blend_bwd_kernel");
}

extern "C" int dbw_shade_blend_bwd(const int32_t *pix_to_face, const float *bary, const float *dists,
                                   const int32_t *c2o, const int32_t *clip_code, const float *clip_w, int Fc_stride,
                                   const float *face_uvs, const int32_t *face_map, const int32_t *map_desc,
                                   const float *maps, const float *faces_alpha, int alpha_len, int N, int H, int W,
                                   int K, int F, float sigma, const float *background3, const float *grad_image,
                                   float *grad_maps, float *grad_faces_alpha, float *grad_dists, float *grad_bary,
                                   int lds_aggregate, dbw_stream_t stream) {
    ShadeArgs A;
    int rc = fill_args(A, pix_to_face, bary, dists, c2o, clip_code, clip_w, Fc_stride, face_uvs, face_map, map_desc,
                       maps, faces_alpha, alpha_len, N, H, W, K, F, sigma, background3);
    if (rc) return rc;
    DBW_REQUIRE(grad_image && grad_maps, "null pointer");
    DBW_REQUIRE(!grad_faces_alpha || faces_alpha, "grad_faces_alpha without faces_alpha");
    return launch_bwd(A, N, H, W, K, grad_image, grad_maps, grad_faces_alpha, grad_dists, grad_bary, lds_aggregate, nullptr,
                      nullptr, 0, 1, (hipStream_t)stream);
}

// (dbw_render_bwd_fused + the training step's cross-stream signal, see ShadeArgs::sync_flag)
int dbw::render_bwd_fused_signal(const int32_t *pix_to_face, const float *bary, const float *dists,
                                    const int32_t *c2o, const int32_t *clip_code, const float *clip_w, int Fc_stride,
                                    const float *face_uvs, const int32_t *face_map, const int32_t *map_desc,
                                    const float *maps, const float *faces_alpha, int alpha_len, int N, int H, int W,
                                    int K, int F, float sigma, const float *background3, const float *grad_image,
                                    const float *face_verts_c, int perspective_correct, int detach_bary,
                                    float *grad_maps, float *grad_faces_alpha, float *grad_face_verts_c,
                                    int lds_aggregate, int frag_layout, const int32_t *bin_base, int32_t *bin_cursor,
                                    void *bin_records, int bin_cap, const uint32_t *bin_layout, int const_geometry_faces,
                                    const float *grad_scale, int image_layout, dbw_stream_t stream, unsigned *sync_flag, unsigned sync_val) {
    ShadeArgs A;
    int rc = fill_args(A, pix_to_face, bary, dists, c2o, clip_code, clip_w, Fc_stride, face_uvs, face_map, map_desc,
                       maps, faces_alpha, alpha_len, N, H, W, K, F, sigma, background3);
    if (rc) return rc;
    DBW_REQUIRE(grad_image && grad_maps && face_verts_c && grad_face_verts_c, "null pointer");
    DBW_REQUIRE(!grad_faces_alpha || faces_alpha, "grad_faces_alpha without faces_alpha");
    DBW_REQUIRE(frag_layout >= 0 && frag_layout <= 3, "frag_layout must be 0 (N,H,W,K), 1 (8x8-tile planar), 2 (planar, uv) or 3 (planar, hard uv)");
    DBW_REQUIRE(frag_layout != 2 || detach_bary, "frag_layout 2 carries no barycentrics: only valid with detach_bary");
    DBW_REQUIRE(frag_layout != 3 || (K == 1 && sigma == 0.f && !faces_alpha && lds_aggregate && F < (1 << 20)),
                "frag_layout 3 is the hard single-layer pass: K == 1, sigma == 0, no faces_alpha, lds_aggregate, F < 2^20");
    DBW_REQUIRE(const_geometry_faces >= 0 && const_geometry_faces <= F, "const_geometry_faces must lie in [0, F]");
    DBW_REQUIRE(image_layout == 0 || image_layout == 1, "image_layout must be 0 (N,4,H,W) or 1 (8x8-tile planar)");
    A.tiled = frag_layout;
    A.img_tiled = image_layout;
    A.gscale = grad_scale;
    A.geom_begin = const_geometry_faces;
    A.sync_flag = sync_flag; A.sync_val = sync_val;
    DBW_REQUIRE((bin_base && bin_cursor && bin_records && bin_cap >= DBW_BIN_SUBCURSORS) || (!bin_base && !bin_cursor && !bin_records), "texture bins: all or none (bin_cap >= DBW_BIN_SUBCURSORS)");
    if (bin_records && !lds_aggregate) {
        A.bin_base = bin_base; A.bin_cursor = bin_cursor; A.bin_records = (int4 *)bin_records; A.bin_cap = bin_cap; A.bin_layout = bin_layout;
    }
    DBW_REQUIRE(!bin_records || bin_layout, "texture bins need their layout table (bin_layout: dbw_bin_layout; equal shares: all counts 0)");
    return launch_bwd(A, N, H, W, K, grad_image, grad_maps, grad_faces_alpha, nullptr, nullptr, lds_aggregate, face_verts_c,
                      grad_face_verts_c, detach_bary ? 0 : 1, perspective_correct, (hipStream_t)stream);
}

extern "C" int dbw_render_bwd_fused(const int32_t *pix_to_face, const float *bary, const float *dists,
                                    const int32_t *c2o, const int32_t *clip_code, const float *clip_w, int Fc_stride,
                                    const float *face_uvs, const int32_t *face_map, const int32_t *map_desc,
                                    const float *maps, const float *faces_alpha, int alpha_len, int N, int H, int W,
                                    int K, int F, float sigma, const float *background3, const float *grad_image,
                                    const float *face_verts_c, int perspective_correct, int detach_bary,
                                    float *grad_maps, float *grad_faces_alpha, float *grad_face_verts_c,
                                    int lds_aggregate, int frag_layout, const int32_t *bin_base, int32_t *bin_cursor,
                                    void *bin_records, int bin_cap, const uint32_t *bin_layout, int const_geometry_faces,
                                    const float *grad_scale, int image_layout, dbw_stream_t stream) {
    return dbw::render_bwd_fused_signal(pix_to_face, bary, dists, c2o, clip_code, clip_w, Fc_stride, face_uvs, face_map, map_desc, maps, faces_alpha, alpha_len,
                                        N, H, W, K, F, sigma, background3, grad_image, face_verts_c, perspective_correct, detach_bary, grad_maps, grad_faces_alpha,
                                        grad_face_verts_c, lds_aggregate, frag_layout, bin_base, bin_cursor, bin_records, bin_cap, bin_layout, const_geometry_faces,
                                        grad_scale, image_layout, stream, nullptr, 0);
}

#ifdef DBW_TILE_CLOCK
extern "C" void dbw_debug_read_bwd_clock(unsigned *out, int nblocks) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_clock), (size_t)(nblocks < (1 << 16) ? nblocks : (1 << 16)) * 16);
}
#endif

extern "C" int dbw_texbin_reduce(const int32_t *bin_info, const int32_t *bin_cursor, const void *bin_records, int bin_cap,
                                 const uint32_t *bin_layout, int nbins, float *grad_maps, dbw_stream_t stream) {
    DBW_REQUIRE(bin_info && bin_cursor && bin_records && bin_layout && grad_maps, "null pointer");
    DBW_REQUIRE(bin_cap >= DBW_BIN_SUBCURSORS && nbins >= 0, "bad size");
    DBW_REQUIRE((long long)nbins * bin_cap < (1LL << 32), "the record array is indexed with 32 bits: nbins * bin_cap must stay below 2^32 (128 GB of records)");
    if (nbins == 0) return DBW_OK;
    hipLaunchKernelGGL(texbin_reduce_kernel, dim3(nbins, BIN_SUB / BIN_SUB_PER_WG), dim3(256), 0, (hipStream_t)stream, bin_info, bin_cursor,
                       (const int4 *)bin_records, bin_cap, bin_layout, grad_maps);
    return dbw_check_launch("texbin_reduce_kernel");
}

// Record sub-ranges by demand (include/dbw_hip.h: dbw_bin_layout): one workgroup, every thread a contiguous piece of the table --
// wanted = max(asked, min) * 1.25, scaled so that the pieces add up to `total`, prefix sums through LDS.  Deterministic, no atomics.
__global__ __launch_bounds__(1024) void bin_layout_kernel(const int *__restrict__ asked, long long n, double total, int min_records,
                                                          unsigned *__restrict__ layout) {
    __shared__ double s_sum[1024];
    const int t = threadIdx.x;
    const long long per = (n + 1023) / 1024, i0 = min(n, t * per), i1 = min(n, i0 + per);
    double mine = 0.0;
    for (long long i = i0; i < i1; ++i) mine += (double)max(asked[i], min_records) * 1.25;
    s_sum[t] = mine;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (t < o) s_sum[t] += s_sum[t + o]; __syncthreads(); }
    // one record per sub-range is set aside before the rest is divided in proportion: the floor of a share below 1 used to be raised
    // to 1 AFTER scaling, so the capacities of a very skewed demand could add up to more than `total`
    const double scale = (total - (double)n) / s_sum[0];
    __syncthreads();
    double caps = 0.0;
    for (long long i = i0; i < i1; ++i) caps += 1.0 + floor((double)max(asked[i], min_records) * 1.25 * scale);
    s_sum[t] = caps;
    __syncthreads();
    // exclusive scan of the 1024 piece totals (Hillis-Steele in place, double buffered by the barrier pairs)
    for (int o = 1; o < 1024; o <<= 1) {
        const double v = t >= o ? s_sum[t - o] : 0.0;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    double first = s_sum[t] - caps;
    for (long long i = i0; i < i1; ++i) {
        const double c = 1.0 + floor((double)max(asked[i], min_records) * 1.25 * scale);
        layout[i * 2] = (unsigned)(long long)first;
        layout[i * 2 + 1] = (unsigned)(long long)c;
        first += c;
    }
}
extern "C" int dbw_bin_layout(const int32_t *asked, int64_t n, double total_records, int min_records, uint32_t *layout, dbw_stream_t stream) {
    DBW_REQUIRE(asked && layout, "null pointer");
    DBW_REQUIRE(n >= 0 && min_records >= 1 && total_records >= 1.0 && total_records < 4294967296.0, "bad size (records are indexed with 32 bits)");
    DBW_REQUIRE((double)n * 1.0 <= total_records, "fewer records than sub-ranges");
    if (n == 0) return DBW_OK;
    hipLaunchKernelGGL(bin_layout_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, asked, (long long)n, total_records, min_records, layout);
    return dbw_check_launch("bin_layout_kernel");
}

// Ablation hook for profiling scripts (tools/) and for the parity tests, which run the product kernels' alternative code paths against the
// oracle too: not part of the rendering contract.  The switches belong to the CALLING HOST THREAD (thread_local, read when a launch is
// enqueued): two threads with different settings do not see each other's.
void dbw_set_raster_dbg(int flags);
void dbw_set_render_dbg(int v);
// bits 0-7: shading/blend ablations (ShadeArgs::dbg) and rasteriser ablations (16, 128); bit 8: plain IEEE divisions in the
// rasteriser, bit 9: no tile culling in the binning (the parity tests run these variants against the oracle too)
extern "C" void dbw_debug_set_flags(int flags) { g_dbg_flags = (flags & 0xff) | (flags & ~0xffff);   // bits 16+: backward experiments (ShadeArgs::dbg)
    dbw_set_raster_dbg(flags); dbw_set_render_dbg(flags >> 8); }
