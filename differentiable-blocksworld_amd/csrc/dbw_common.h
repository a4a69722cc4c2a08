// Shared device helpers for libdbw_hip.so (gfx950 only).
//
// The per-(pixel, face) arithmetic below is held BIT-EXACT to oracle/raster_ref.c (the canonical restatement of the
// PyTorch3D 0.7.1 CPU rasteriser): same operations, same order, one rounding per operation.
// That is why this library is built with -ffp-contract=off and IEEE fp32 division/sqrt (hipcc default), and why the
// helpers are written with explicit ternaries instead of fmaxf/fminf.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "raster_math.h"      // f2 / f3, pix_to_ndc, edge_fn, FaceRec, eval_pair, TopK (host + device)

#pragma clang fp contract(off)

#define DBW_WAVE 64

namespace dbw {

__device__ __forceinline__ void edge_fn_bwd(f2 p, f2 a, f2 b, float g, f2 &gp, f2 &ga, f2 &gb) {
    gp.x = g * (b.y - a.y); gp.y = g * (a.x - b.x);
    ga.x = g * (p.y - b.y); ga.y = g * (b.x - p.x);
    gb.x = g * (a.y - p.y); gb.y = g * (p.x - a.x);
}

// FAST = true replaces the IEEE divisions by v_rcp_f32 (1 ulp): used by the fused backward only, whose gradients are compared at
// 1e-4 relative tolerance -- the forward rasteriser keeps the exact arithmetic that decides the face indices.
template <bool FAST>
__device__ __forceinline__ float div_(float a, float b) { return FAST ? a * __builtin_amdgcn_rcpf(b) : a / b; }

template <bool FAST = false>
__device__ __forceinline__ f3 bary_fwd(f2 p, f2 v0, f2 v1, f2 v2) {
    const float area = DBW_AREA_EPS(edge_fn(v2, v0, v1));
    f3 w;
    if (FAST) {
        const float r = __builtin_amdgcn_rcpf(area);
        w.x = edge_fn(p, v1, v2) * r; w.y = edge_fn(p, v2, v0) * r; w.z = edge_fn(p, v0, v1) * r;
        return w;
    }
    w.x = edge_fn(p, v1, v2) / area;
    w.y = edge_fn(p, v2, v0) / area;
    w.z = edge_fn(p, v0, v1) / area;
    return w;
}

template <bool FAST = false>
__device__ __forceinline__ void bary_bwd(f2 p, f2 v0, f2 v1, f2 v2, f3 g, f2 &g0, f2 &g1, f2 &g2) {
    const float area = DBW_AREA_EPS(edge_fn(v2, v0, v1));
    const float area_inv = FAST ? __builtin_amdgcn_rcpf(area) : 1.0f / area;
    const float area2_inv = FAST ? area_inv * area_inv : 0.f;
    const float area2 = area * area;
    const float e0 = edge_fn(p, v1, v2);
    const float e1 = edge_fn(p, v2, v0);
    const float e2 = edge_fn(p, v0, v1);
    f2 gp, ga, gb, hp, ha, hb;
    g0.x = g0.y = g1.x = g1.y = g2.x = g2.y = 0.f;
    edge_fn_bwd(p, v1, v2, g.x * area_inv, gp, ga, gb);
    edge_fn_bwd(v2, v0, v1, g.x * (FAST ? -e0 * area2_inv : -e0 / area2), hp, ha, hb);
    g0.x += ha.x;        g0.y += ha.y;
    g1.x += ga.x + hb.x; g1.y += ga.y + hb.y;
    g2.x += gb.x + hp.x; g2.y += gb.y + hp.y;
    edge_fn_bwd(p, v2, v0, g.y * area_inv, gp, ga, gb);
    edge_fn_bwd(v2, v0, v1, g.y * (FAST ? -e1 * area2_inv : -e1 / area2), hp, ha, hb);
    g0.x += gb.x + ha.x; g0.y += gb.y + ha.y;
    g1.x += hb.x;        g1.y += hb.y;
    g2.x += ga.x + hp.x; g2.y += ga.y + hp.y;
    edge_fn_bwd(p, v0, v1, g.z * area_inv, gp, ga, gb);
    edge_fn_bwd(v2, v0, v1, g.z * (FAST ? -e2 * area2_inv : -e2 / area2), hp, ha, hb);
    g0.x += ga.x + ha.x; g0.y += ga.y + ha.y;
    g1.x += gb.x + hb.x; g1.y += gb.y + hb.y;
    g2.x += hp.x;        g2.y += hp.y;
}

template <bool FAST = false>
__device__ __forceinline__ f3 persp_fwd(f3 b, float z0, float z1, float z2) {
    const float t0 = b.x * z1 * z2;
    const float t1 = z0 * b.y * z2;
    const float t2 = z0 * z1 * b.z;
    float denom = t0 + t1 + t2;
    if (!(denom > DBW_EPS)) denom = DBW_EPS;
    f3 w;
    if (FAST) { const float r = __builtin_amdgcn_rcpf(denom); w.x = t0 * r; w.y = t1 * r; w.z = t2 * r; return w; }
    w.x = t0 / denom; w.y = t1 / denom; w.z = t2 / denom;
    return w;
}

template <bool FAST = false>
__device__ __forceinline__ f3 persp_bwd(f3 b, float z0, float z1, float z2, f3 g, float &gz0, float &gz1, float &gz2) {
    const float t0 = b.x * z1 * z2;
    const float t1 = z0 * b.y * z2;
    const float t2 = z0 * z1 * b.z;
    float denom = t0 + t1 + t2;
    if (!(denom > DBW_EPS)) denom = DBW_EPS;
    const float g_denom_top = -t0 * g.x - t1 * g.y - t2 * g.z;
    const float rd = FAST ? __builtin_amdgcn_rcpf(denom) : 0.f;
    const float g_denom = FAST ? g_denom_top * rd * rd : g_denom_top / (denom * denom);
    const float gt0 = g_denom + (FAST ? g.x * rd : g.x / denom);
    const float gt1 = g_denom + (FAST ? g.y * rd : g.y / denom);
    const float gt2 = g_denom + (FAST ? g.z * rd : g.z / denom);
    f3 gb; gb.x = gt0 * z1 * z2; gb.y = gt1 * z0 * z2; gb.z = gt2 * z0 * z1;
    gz0 = gt1 * b.y * z2 + gt2 * b.z * z1;
    gz1 = gt0 * b.x * z2 + gt2 * b.z * z0;
    gz2 = gt0 * b.x * z1 + gt1 * b.y * z0;
    return gb;
}

__device__ __forceinline__ f3 clip_fwd(f3 b) {
    f3 w;
    w.x = b.x > 0.f ? b.x : 0.f; w.y = b.y > 0.f ? b.y : 0.f; w.z = b.z > 0.f ? b.z : 0.f;
    float s = w.x + w.y + w.z;
    if (!(s > 1e-5f)) s = 1e-5f;
    w.x /= s; w.y /= s; w.z /= s;
    return w;
}

template <bool FAST = false>
__device__ __forceinline__ f3 clip_bwd(f3 b, f3 g) {
    f3 w;
    w.x = b.x > 0.f ? b.x : 0.f; w.y = b.y > 0.f ? b.y : 0.f; w.z = b.z > 0.f ? b.z : 0.f;
    float s = w.x + w.y + w.z;
    float gsc = 1.f;
    if (s < 1e-5f) { gsc = 0.f; s = 1e-5f; }
    const float cx = b.x < 0.f ? 0.f : 1.f, cy = b.y < 0.f ? 0.f : 1.f, cz = b.z < 0.f ? 0.f : 1.f;
    const float s2 = s * s;
    if (FAST) {
        const float rs = __builtin_amdgcn_rcpf(s), rs2 = rs * rs;
        const float common = (g.x * -w.x + g.y * -w.y + g.z * -w.z) * rs2 * gsc;
        f3 o;
        o.x = cx * (g.x * rs + common); o.y = cy * (g.y * rs + common); o.z = cz * (g.z * rs + common);
        return o;
    }
    const float gsx = -w.x / s2 * gsc, gsy = -w.y / s2 * gsc, gsz = -w.z / s2 * gsc;
    const float common = g.x * gsx + g.y * gsy + g.z * gsz;
    f3 o;
    o.x = cx * (g.x / s + common);
    o.y = cy * (g.y / s + common);
    o.z = cz * (g.z / s + common);
    return o;
}

__device__ __forceinline__ float point_line_dist(f2 p, f2 a, f2 b) {
    const float dx = b.x - a.x, dy = b.y - a.y;
    const float l2 = dx * dx + dy * dy;
    if (l2 <= DBW_EPS) return (p.x - b.x) * (p.x - b.x) + (p.y - b.y) * (p.y - b.y);
    const float t = (dx * (p.x - a.x) + dy * (p.y - a.y)) / l2;
    const float tt = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
    const float qx = a.x + tt * dx, qy = a.y + tt * dy;
    return (p.x - qx) * (p.x - qx) + (p.y - qy) * (p.y - qy);
}

template <bool FAST = false>
__device__ __forceinline__ void point_line_dist_bwd(f2 p, f2 a, f2 b, float g, f2 &ga, f2 &gb) {
    const float dx = b.x - a.x, dy = b.y - a.y;
    const float t_bot = dx * dx + dy * dy;
    const float t_top = dx * (p.x - a.x) + dy * (p.y - a.y);
    const float t = div_<FAST>(t_top, t_bot);
    const float tt = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
    const float qx = (1.f - tt) * a.x + tt * b.x, qy = (1.f - tt) * a.y + tt * b.y;
    ga.x = g * (1.f - tt) * 2.f * (qx - p.x); ga.y = g * (1.f - tt) * 2.f * (qy - p.y);
    gb.x = g * tt * 2.f * (qx - p.x);         gb.y = g * tt * 2.f * (qy - p.y);
}

template <bool FAST = false>
__device__ __forceinline__ float point_line_dist_t(f2 p, f2 a, f2 b) {
    const float dx = b.x - a.x, dy = b.y - a.y;
    const float l2 = dx * dx + dy * dy;
    if (l2 <= DBW_EPS) return (p.x - b.x) * (p.x - b.x) + (p.y - b.y) * (p.y - b.y);
    const float t = div_<FAST>(dx * (p.x - a.x) + dy * (p.y - a.y), l2);
    const float tt = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
    const float qx = a.x + tt * dx, qy = a.y + tt * dy;
    return (p.x - qx) * (p.x - qx) + (p.y - qy) * (p.y - qy);
}

__device__ __forceinline__ float point_tri_dist(f2 p, f2 v0, f2 v1, f2 v2) {
    const float e01 = point_line_dist(p, v0, v1);
    const float e02 = point_line_dist(p, v0, v2);
    const float e12 = point_line_dist(p, v1, v2);
    const float m = e01 < e02 ? e01 : e02;
    return m < e12 ? m : e12;
}

template <bool FAST = false>
__device__ __forceinline__ void point_tri_dist_bwd(f2 p, f2 v0, f2 v1, f2 v2, float g, f2 &g0, f2 &g1, f2 &g2) {
    const float e01 = point_line_dist_t<FAST>(p, v0, v1);
    const float e02 = point_line_dist_t<FAST>(p, v0, v2);
    const float e12 = point_line_dist_t<FAST>(p, v1, v2);
    g0.x = g0.y = g1.x = g1.y = g2.x = g2.y = 0.f;
    if (e01 <= e02 && e01 <= e12) point_line_dist_bwd<FAST>(p, v0, v1, g, g0, g1);
    else if (e02 <= e01 && e02 <= e12) point_line_dist_bwd<FAST>(p, v0, v2, g, g0, g2);
    else if (e12 <= e01 && e12 <= e02) point_line_dist_bwd<FAST>(p, v1, v2, g, g1, g2);
}

// ---- wave-level helpers (wave64) ---------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Same sum on the VALU only (DPP row shifts + row broadcasts, the gfx9 reduction idiom): __shfl_xor lowers to ds_bpermute, i.e.
// six LDS-pipe instructions per value, which is exactly the pipe the LDS tables of the backward are short of.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, BANK_MASK, false));
}
// 2x2-pixel quad helpers (lanes 4i..4i+3 of the 8x8 patch are not a spatial quad, but they are 4 horizontally adjacent pixels):
// broadcast of the quad's first lane, and the sum over the quad in every lane
__device__ __forceinline__ int quad_first(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x00, 0xf, 0xf, false); }
__device__ __forceinline__ int quad_and(int x) {
    x &= __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
    x &= __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
    return x;
}
__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_<0xB1, 0xf, 0xf>(v);
    v += dpp_<0x4E, 0xf, 0xf>(v);
    return v;
}
// Neighbouring lanes that update the same key of an LDS table are merged before the table sees them.  An LDS atomic replays once per
// lane of the most contended address of each 16-lane group (profiles/r01_lds_atomic_ubench.txt: ~3 clk per replay for ds_add_f64), and
// the lanes that share a face or a texel are neighbouring pixels: in the 8x8-tile layout lanes 4i..4i+3 are four horizontally
// adjacent pixels, lanes i and i + 8 of a 16-lane row are vertical neighbours.  STEPS merging steps, each halving the candidates:
// lane ^ 1, lane ^ 2 (quad permutes), lane + 4, lane + 8 (row shifts) -- a lane whose partner (the one with the step's bit clear)
// is active with the same key hands its values over and drops out.  VALU only (DPP); `key` >= 0.
template <int NV, int STEPS>
__device__ __forceinline__ void lane_merge(int key, bool &active, float (&v)[NV]) {
    if (STEPS <= 0) return;
    const int lane = threadIdx.x & 63;
    int k = active ? key : -1000 - lane;              // an inactive lane matches nobody
#define DBW_MERGE_STEP(CTRL_PARTNER, CTRL_FROM_GIVER, BIT)                                                        \
    {                                                                                                             \
        const int kp = __builtin_amdgcn_update_dpp(-1, k, CTRL_PARTNER, 0xf, 0xf, false);                         \
        const bool give = (lane & (BIT)) && kp == k;                                                              \
        _Pragma("unroll") for (int c = 0; c < NV; ++c) {                                                          \
            const float t = give ? v[c] : 0.f;                                                                    \
            v[c] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), CTRL_FROM_GIVER, 0xf, 0xf, false)); \
        }                                                                                                         \
        if (give) { active = false; k = -1000 - lane; }                                                           \
    }
    DBW_MERGE_STEP(0xB1, 0xB1, 1)                     // quad_perm [1,0,3,2]
    if (STEPS >= 2) DBW_MERGE_STEP(0x4E, 0x4E, 2)     // quad_perm [2,3,0,1]
    if (STEPS >= 3) DBW_MERGE_STEP(0x114, 0x104, 4)   // the giver looks at lane - 4 (row_shr:4), the receiver takes from lane + 4 (row_shl:4)
    if (STEPS >= 4) DBW_MERGE_STEP(0x118, 0x108, 8)
#undef DBW_MERGE_STEP
}

__device__ __forceinline__ float wave_sum_dpp(float v) {
    const float t = v;
    v += dpp_<0x111, 0xf, 0xf>(t);           // row_shr:1
    v += dpp_<0x112, 0xf, 0xf>(t);           // row_shr:2
    v += dpp_<0x113, 0xf, 0xf>(t);           // row_shr:3   -> every lane: itself + the 3 lanes below it in its row
    // (all lanes take part in every step: the row / bank masks of the textbook form only keep lanes that nobody reads from adding what
    // they do not need -- lane 15 of a row, lane 31, lane 63 get the same operands either way -- and a masked DPP operand cannot be folded
    // into the add: v_mov 0, v_mov_dpp, v_add per step instead of one v_add_f32_dpp)
    v += dpp_<0x114, 0xf, 0xf>(v);           // row_shr:4
    v += dpp_<0x118, 0xf, 0xf>(v);           // row_shr:8   -> lane 15 of every row: the row's sum
    v += dpp_<0x142, 0xf, 0xf>(v);           // row_bcast:15 -> lane 31: rows 0 + 1, lane 63: rows 2 + 3
    v += dpp_<0x143, 0xf, 0xf>(v);           // row_bcast:31 -> lane 63: the wave's sum
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Wave-aggregated atomic accumulation into base[idx*NC + c]: lanes of a wave that target the same index (neighbouring
// pixels usually hit the same face / texel) are summed with a butterfly first, so memory sees ~1 atomic per
// (wave, index) instead of one per lane.  Must be called by ALL lanes of the wave (inactive lanes pass active=false).
// Bounded: after 8 distinct indices the remaining lanes fall back to plain atomics.
template <int NC>
__device__ __forceinline__ void wave_agg_atomic(float *__restrict__ base, long long idx, bool active,
                                                const float (&g)[NC], int lane) {
    unsigned long long rem = __ballot(active);
    int iter = 0;
    while (rem) {
        if (iter >= 8) {
            if (active && ((rem >> lane) & 1ull)) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (g[c] != 0.f) unsafeAtomicAdd(base + idx * NC + c, g[c]);
            }
            break;
        }
        const int leader = __ffsll((long long)rem) - 1;
        const long long i0 = __shfl(idx, leader, 64);
        const bool match = active && (idx == i0);
        const unsigned long long mm = __ballot(match);
        if (__popcll(mm) > 1) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float s = wave_sum(match ? g[c] : 0.f);
                if (lane == leader && s != 0.f) unsafeAtomicAdd(base + i0 * NC + c, s);
            }
        } else if (match) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (g[c] != 0.f) unsafeAtomicAdd(base + idx * NC + c, g[c]);
        }
        rem &= ~mm;
        ++iter;
    }
}

// Workgroup-level pre-aggregation of scattered gradient updates in LDS (open-addressing hash keyed by the destination
// index).  Measured on MI355X (profiles/r01_atomic_scope_ubench.txt): scattered global fp32 atomics run at a flat
// ~21 G lane-ops/s whatever the scope / working set, while LDS atomics are far cheaper -- so every update first lands
// in LDS (ds_cmpst + ds_add_f64) and each DISTINCT destination of the tile is flushed to memory once.  Updates that do
// not find a slot within 8 probes go straight to memory, so the result is exact regardless of table pressure.
// The table accumulates in fp64 because of how gfx950 executes LDS float atomics (profiles/r01_lds_atomic_ubench.txt):
// ds_add_f32 retires one lane per ~3 clk whatever the addresses (193 clk for a full wave), ds_add_f64 costs ~6-8 clk per wave
// when the lanes hit distinct slots and ~3 clk per extra lane sharing a slot within a 16-lane group.
template <int NV, int LOG2_SLOTS>
struct LdsAgg {
    static constexpr int NSLOT = 1 << LOG2_SLOTS;
    static constexpr size_t BYTES = (size_t)NSLOT * (NV * 8 + 4);
    static_assert(NSLOT % 2 == 0, "keys must leave the fp64 values 8-byte aligned");
    int *keys;
    double *vals;
    __device__ __forceinline__ void bind(void *lds) { keys = (int *)lds; vals = (double *)((int *)lds + NSLOT); }
    __device__ __forceinline__ void clear(int tid, int nthreads) {
        for (int i = tid; i < NSLOT; i += nthreads) keys[i] = -1;
        for (int i = tid; i < NSLOT * NV; i += nthreads) vals[i] = 0.0;
    }
    __device__ __forceinline__ void add(float *__restrict__ gbase, int key, const float (&v)[NV]) {
        unsigned h = ((unsigned)key * 2654435761u) >> (32 - LOG2_SLOTS);
#pragma unroll 1
        for (int p = 0; p < 8; ++p) {
            const int old = atomicCAS(&keys[h], -1, key);
            if (old == -1 || old == key) {
#pragma unroll
                for (int c = 0; c < NV; ++c)
                    if (v[c] != 0.f) atomicAdd(&vals[h * NV + c], (double)v[c]);
                return;
            }
            h = (h + 1) & (NSLOT - 1);
        }
#pragma unroll
        for (int c = 0; c < NV; ++c)
            if (v[c] != 0.f) unsafeAtomicAdd(gbase + (long long)key * NV + c, v[c]);
    }
    // Wave-collective add (every lane of the wave calls it; `active` marks the lanes with an update).  When all active lanes
    // carry the same key -- a wave inside one large face, one block's opacity -- the values are summed across the wave first
    // and one lane does the table update: 64 lanes hitting one slot would serialise at ~3 clk per lane and value.
    __device__ __forceinline__ void add_wave(float *__restrict__ gbase, int key, const float (&v)[NV], bool active) {
        const unsigned long long am = __ballot(active);
        if (am == 0ull) return;
        const int leader = __ffsll((long long)am) - 1;
        const int k0 = __builtin_amdgcn_readlane(key, leader);          // (the leader is wave-uniform: no LDS round trip for the broadcast)
        if (__popcll(am) >= 8 && __ballot(active && key == k0) == am) {
            float s[NV];
#pragma unroll
            for (int c = 0; c < NV; ++c) s[c] = wave_sum_dpp(active ? v[c] : 0.f);
            if ((int)(threadIdx.x & 63) == leader) add(gbase, k0, s);
        } else if (active) {
            add(gbase, key, v);
        }
    }
    // add_wave with the merge of neighbouring lanes (lane_merge, STEPS steps) in front of the table -- but only where it is needed: a wave whose
    // active lanes all carry ONE key (magnified / decimated maps: most waves; a wave inside one large face) goes straight to the wave sum,
    // where the merge steps -- a dozen instructions each -- would only have halved what the sum adds up anyway
    template <int STEPS>
    __device__ __forceinline__ void add_wave_merged(float *__restrict__ gbase, int key, float (&v)[NV], bool active) {
        const unsigned long long am = __ballot(active);
        if (am == 0ull) return;
        const int leader = __ffsll((long long)am) - 1;
        const int k0 = __builtin_amdgcn_readlane(key, leader);
        if (__popcll(am) >= 8 && __ballot(active && key == k0) == am) {
            float s[NV];
#pragma unroll
            for (int c = 0; c < NV; ++c) s[c] = wave_sum_dpp(active ? v[c] : 0.f);
            if ((int)(threadIdx.x & 63) == leader) add(gbase, k0, s);
            return;
        }
        lane_merge<NV, STEPS>(key, active, v);
        if (active) add(gbase, key, v);
    }
    __device__ __forceinline__ void flush(float *__restrict__ gbase, int tid, int nthreads) {
        for (int i = tid; i < NSLOT; i += nthreads) {
            const int k = keys[i];
            if (k >= 0) {
#pragma unroll
                for (int c = 0; c < NV; ++c) {
                    const float x = (float)vals[i * NV + c];
                    if (x != 0.f) unsafeAtomicAdd(gbase + (long long)k * NV + c, x);
                }
            }
        }
    }
};

// XCD-aware block remap (cdna_hip_programming.md T1): hardware places block b on XCD b % 8; we want consecutive
// LOGICAL blocks (tiles of one view: same face list, same texture footprint, adjacent output rows) on one XCD/L2.
// Grid must be launched with 8*ceil(total/8) blocks; returns -1 for padding blocks.
__device__ __forceinline__ long long xcd_remap(long long b, long long total) {
    const long long per = (total + 7) / 8;
    const long long logical = (b % 8) * per + (b / 8);
    return logical < total ? logical : -1;
}

}  // namespace dbw

// ---- host-side helpers ------------------------------------------------------------------------------------------
void dbw_set_error(const char *fmt, ...);
int dbw_check_launch(const char *what);
#define DBW_REQUIRE(cond, msg)                          \
    do {                                                \
        if (!(cond)) {                                  \
            dbw_set_error("%s: %s", __func__, msg);     \
            return DBW_ERR_INVALID;                     \
        }                                               \
    } while (0)
static inline unsigned dbw_xcd_grid(long long total) { return (unsigned)(8 * ((total + 7) / 8)); }
