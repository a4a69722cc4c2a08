// Per-(pixel, face) arithmetic of the rasteriser and the per-pixel top-K list, written once for the device (hipcc) and for the
// host (g++, tests/test_host_raster_math.py builds it into a checker-side shared object and compares it with oracle/raster_ref.c
// without a GPU).  Everything here is held BIT-EXACT to the oracle: same operations, same order, one rounding per
// operation -- both sides are built with -ffp-contract=off.
//
// What is new relative to a literal restatement, and why it does not change a bit:
//  * FaceRec: every pixel-independent sub-expression (edge deltas, area, squared edge lengths) is evaluated ONCE per face by
//    face_setup_kernel with the oracle's own expression and rounding, and read back by the tiles as wave-uniform scalar loads.
//    Negated deltas are exact in IEEE arithmetic (fl(-x) = -fl(x)), so edge (v0, v2) of the distance reuses the (v2 -> v0) delta.
//  * div_fast: the three quotients of a stage that share a denominator share ONE v_rcp_f32 + Newton refinement.  The sequence is
//    the unscaled core of the compiler's own IEEE fp32 division (v_div_scale / v_rcp / 4 fma / v_div_fmas / v_div_fixup): inside
//    the operand range where v_div_scale does not scale and v_div_fixup passes through (checked per lane by the guards below) it
//    IS that sequence, instruction for instruction, so the quotient has the same bits; outside it the lane takes the `/` path.
//  * TopK: the sorted list holds 64-bit keys (depth bits | face id | payload slot) in registers and moves nothing else; the
//    payload (signed distance + barycentrics) of an entry sits in a fixed slot of an LDS home array and is written once.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define DBW_HD __host__ __device__ __forceinline__
#else
#define DBW_HD inline
#endif
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#define DBW_EPS 1e-8f
// PyTorch3D's kEpsilon is a double (`const auto kEpsilon = 1e-8;`): the one place it enters ARITHMETIC -- the barycentric denominator
// area = EdgeFunction(v2, v0, v1) + kEpsilon -- promotes the float edge function, adds in double and rounds to float once
// (oracle/raster_ref.c header); comparisons against it decide the same in float
#define DBW_AREA_EPS(e) ((float)((double)(e) + 1e-8))
#ifndef DBW_CLAMP_MED3
#define DBW_CLAMP_MED3 1        // the t clamp of the point-segment distance as one v_med3_f32 (same value for every finite t)
#endif
#ifndef DBW_TOPK_ORDERED
#define DBW_TOPK_ORDERED 1      // TopK::insert_ordered instead of TopK::insert in the kernels
#endif

namespace dbw {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct __attribute__((aligned(16))) pay4 { float x, y, z, w; };       // payload of a list entry: signed distance, clipped barycentrics

DBW_HD uint32_t f2u(float x) { union { float f; uint32_t u; } c; c.f = x; return c.u; }
DBW_HD float u2f(uint32_t x) { union { float f; uint32_t u; } c; c.u = x; return c.f; }

// wave-level "any lane" (the host build evaluates one pixel at a time), median of three unsigned / clamp of a float to [0, 1]
#if defined(__HIP_DEVICE_COMPILE__)
DBW_HD bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
DBW_HD int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
DBW_HD uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
DBW_HD float clamp01(float t) {
#if DBW_CLAMP_MED3
    return __builtin_amdgcn_fmed3f(t, 0.f, 1.f);
#else
    return t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
#endif
}
#else
inline float clamp01(float t) { return t < 0.f ? 0.f : (t > 1.f ? 1.f : t); }
inline bool wave_any(bool p) { return p; }
inline int wave_uniform(int x) { return x; }
inline uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
}
#endif

// SURVEY A.1 NonSquarePixToNdc
DBW_HD float pix_to_ndc(int i, int S1, int S2) {
    float range = 2.0f;
    if (S1 > S2) range = ((float)S1 * range) / (float)S2;
    const float offset = range / 2.0f;
    return -offset + (range * (float)i + offset) / (float)S1;
}

DBW_HD float edge_fn(f2 p, f2 a, f2 b) {
    return (p.x - a.x) * (b.y - a.y) - (p.y - a.y) * (b.x - a.x);
}

// ---- division ------------------------------------------------------------------------------------------------------------------
#if !defined(__HIP_DEVICE_COMPILE__)
// host stand-in for v_rcp_f32 (1 ulp): the correctly rounded reciprocal, optionally pushed one ulp off by the test harness to show
// that the refined quotient does not depend on the last bit of the seed
static int g_host_rcp_perturb = 0;
#endif
DBW_HD float rcp_seed(float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(d);
#else
    float r = (float)(1.0 / (double)d);
    if (g_host_rcp_perturb > 0) r = nextafterf(r, INFINITY);
    if (g_host_rcp_perturb < 0) r = nextafterf(r, -INFINITY);
    return r;
#endif
}
// Fma0, Fma1 of the compiler's fdiv32 expansion: the refined reciprocal shared by every quotient over d
DBW_HD float rcp_refined(float d) {
    const float r = rcp_seed(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
// Mul, Fma2, Fma3, Fma4, Fmas of the same expansion (r = rcp_refined(d))
DBW_HD float div_fast(float n, float d, float r) {
    float q = n * r;
    float e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r, q);
}
// Guard of a numerator: 0, or 2^-60 <= |n| (the expansion is only unscaled for |n| >= 2^-103; upper bounds follow from the
// per-face coordinate bound REC_FAST, see make_face_rec).  guard_key(0) wraps to 0xffffffff, so min over the keys >= GUARD_LO.
DBW_HD uint32_t guard_key(float n) { return (f2u(n) << 1) - 1u; }
constexpr uint32_t GUARD_LO = ((127u - 60u) << 24) - 1u;
DBW_HD uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { const uint32_t m = a < b ? a : b; return m < c ? m : c; }
DBW_HD bool guard3(float n0, float n1, float n2) { return umin3(guard_key(n0), guard_key(n1), guard_key(n2)) >= GUARD_LO; }
// positive denominator in [2^-27, 2^40)  (the eps clamp of the perspective denominator, 1e-8, lies inside)
DBW_HD bool guard_den(float d) { return (f2u(d) - ((127u - 27u) << 23)) < (67u << 23); }

// pix_to_ndc with the division by S1 done by div_fast on a reciprocal shared by all pixels of an axis: the numerator
// |range * i + offset| >= range / 2 >= 1 and the denominator S1 >= 1 are always inside the guarded range, so the bits are pix_to_ndc's
struct NdcAxis { float range, offset, s1, r; };
DBW_HD NdcAxis ndc_axis(int S1, int S2) {
    NdcAxis a;
    a.range = 2.0f;
    if (S1 > S2) a.range = ((float)S1 * a.range) / (float)S2;
    a.offset = a.range / 2.0f;
    a.s1 = (float)S1;
    a.r = rcp_refined(a.s1);
    return a;
}
// the same axis from its host-computed range and offset (IEEE divisions: the same bits on either side; CoarseBins::ndc): a tile then pays
// no division for them -- only the refined reciprocal of S1, which stays a device value (v_rcp_f32 seed)
DBW_HD NdcAxis ndc_axis_given(int S1, float range, float offset) {
    NdcAxis a;
    a.range = range; a.offset = offset;
    a.s1 = (float)S1;
    a.r = rcp_refined(a.s1);
    return a;
}
DBW_HD float pix_to_ndc_fast(int i, const NdcAxis &a) { return -a.offset + div_fast(a.range * (float)i + a.offset, a.s1, a.r); }

// ---- per-face record -----------------------------------------------------------------------------------------------------------
enum { REC_FAST = 1, REC_AREA_POS = 2, REC_DEG_AB = 4, REC_DEG_AC = 8, REC_DEG_BC = 16, REC_CULL = 32 };

struct __attribute__((aligned(128))) FaceRec {
    float ax, ay, bx, by, cx, cy;                        // NDC vertices v0 = a, v1 = b, v2 = c
    float z0, z1, z2;                                    // view depths
    float area, r_area;                                  // edge_fn(c, a, b) + eps  (bary_fwd's denominator), its refined reciprocal
    float dab_x, dab_y, dbc_x, dbc_y, dca_x, dca_y;      // b - a, c - b, a - c
    float l2_ab, l2_ac, l2_bc;                           // squared edge lengths as point_line_dist computes them
    float r_ab, r_ac, r_bc;                              // their refined reciprocals
    float xlo, xhi, ylo, yhi;                            // blur-expanded box (empty for faces that can never be hit)
    int nb;                                              // clip sibling (packed face index) or -1
    int flags;                                           // REC_*
    float thr_ab, thr_bc, thr_ca;                        // conservative tile-culling thresholds: (1.01 sqrt(blur) + 1e-4) * edge length
};
static_assert(sizeof(FaceRec) == 128, "FaceRec must be 128 B");

// p = the face's 9 floats.  box4 = {xlo, xhi, ylo, yhi} (also stored in the record).  Same dead-face rules as the oracle's per-pixel
// tests (zmin < eps, |edge_fn(a, b, c)| <= eps, back-face culling): a dead face gets an empty box and is never staged.
DBW_HD void make_face_rec(const float *p, float margin, int cull, int nb, FaceRec &r, float box4[4]) {
    const f2 a{p[0], p[1]}, b{p[3], p[4]}, c{p[6], p[7]};
    const float z0 = p[2], z1 = p[5], z2 = p[8];
    float xmin = a.x < b.x ? a.x : b.x; xmin = xmin < c.x ? xmin : c.x;
    float xmax = a.x > b.x ? a.x : b.x; xmax = xmax > c.x ? xmax : c.x;
    float ymin = a.y < b.y ? a.y : b.y; ymin = ymin < c.y ? ymin : c.y;
    float ymax = a.y > b.y ? a.y : b.y; ymax = ymax > c.y ? ymax : c.y;
    float zmin = z0 < z1 ? z0 : z1; zmin = zmin < z2 ? zmin : z2;
    float zmax = z0 > z1 ? z0 : z1; zmax = zmax > z2 ? zmax : z2;
    const float face_area = edge_fn(a, b, c);
    const bool dead = (zmin < DBW_EPS) || (face_area <= DBW_EPS && face_area >= -DBW_EPS) || (cull && face_area < 0.f);
    r.ax = a.x; r.ay = a.y; r.bx = b.x; r.by = b.y; r.cx = c.x; r.cy = c.y;
    r.z0 = z0; r.z1 = z1; r.z2 = z2;
    r.area = DBW_AREA_EPS(edge_fn(c, a, b));
    r.dab_x = b.x - a.x; r.dab_y = b.y - a.y;
    r.dbc_x = c.x - b.x; r.dbc_y = c.y - b.y;
    r.dca_x = a.x - c.x; r.dca_y = a.y - c.y;
    r.l2_ab = r.dab_x * r.dab_x + r.dab_y * r.dab_y;
    r.l2_ac = r.dca_x * r.dca_x + r.dca_y * r.dca_y;      // (c - a)^2 == (a - c)^2 bit for bit
    r.l2_bc = r.dbc_x * r.dbc_x + r.dbc_y * r.dbc_y;
    int flags = (r.area > 0.f) ? REC_AREA_POS : 0;
    if (r.l2_ab <= DBW_EPS) flags |= REC_DEG_AB;
    if (r.l2_ac <= DBW_EPS) flags |= REC_DEG_AC;
    if (r.l2_bc <= DBW_EPS) flags |= REC_DEG_BC;
    float cmax = fabsf(a.x);
    cmax = fmaxf(cmax, fabsf(a.y)); cmax = fmaxf(cmax, fabsf(b.x)); cmax = fmaxf(cmax, fabsf(b.y));
    cmax = fmaxf(cmax, fabsf(c.x)); cmax = fmaxf(cmax, fabsf(c.y));
    const float aabs = fabsf(r.area);
    // REC_FAST: |x|, |y|, z <= 2^10 and |area| >= 2^-20 bound every numerator of the four division stages inside the unscaled
    // range of the fdiv32 expansion: edge functions <= 2^22, barycentrics <= 2^42, perspective terms <= 2^62 over denominators
    // >= 2^-27 (exponent difference < 96, no overflow); the lower bounds are checked per lane (guard3 / guard_den)
    if (!dead && cmax <= 1024.f && zmax <= 1024.f && aabs >= 9.5367432e-7f && aabs <= 8388608.f) flags |= REC_FAST;
    if (!dead && cmax <= 64.f && !(flags & (REC_DEG_AB | REC_DEG_AC | REC_DEG_BC))) flags |= REC_CULL;
    r.r_area = (flags & REC_FAST) ? rcp_refined(r.area) : 0.f;
    r.r_ab = (flags & REC_DEG_AB) ? 0.f : rcp_refined(r.l2_ab);
    r.r_ac = (flags & REC_DEG_AC) ? 0.f : rcp_refined(r.l2_ac);
    r.r_bc = (flags & REC_DEG_BC) ? 0.f : rcp_refined(r.l2_bc);
    if (dead) { r.xlo = INFINITY; r.xhi = -INFINITY; r.ylo = INFINITY; r.yhi = -INFINITY; flags = 0; }
    else { r.xlo = xmin - margin; r.xhi = xmax + margin; r.ylo = ymin - margin; r.yhi = ymax + margin; }
    r.nb = nb;
    r.flags = flags;
    const float k = 1.01f * margin + 1e-4f;
    r.thr_ab = k * sqrtf(r.l2_ab); r.thr_bc = k * sqrtf(r.l2_bc); r.thr_ca = k * sqrtf(r.l2_ac);
    box4[0] = r.xlo; box4[1] = r.xhi; box4[2] = r.ylo; box4[3] = r.yhi;
}

// Conservative tile test: true when NO pixel centre of the box [xmin, xmax] x [ymin, ymax] can be accepted for the face, because the
// whole box lies further than sqrt(blur) (+ 1 % + 1e-4: far above the fp32 error of this evaluation and of the exact per-pixel
// arithmetic for |coordinates| <= 64, REC_CULL) outside one of the three edge lines.  A pixel is accepted only if it lies inside the
// triangle or within sqrt(blur) of it, i.e. on the inner side of every edge line moved out by sqrt(blur).
DBW_HD bool tile_outside_edge(float s, float ox, float oy, float dx, float dy, float thr, float xmin, float xmax, float ymin,
                              float ymax) {
    const float A = s * dy, B = -(s * dx);            // s * edge_fn(p; o, o + d) = A (p.x - ox) + B (p.y - oy)
    const float m = A * ((A >= 0.f ? xmax : xmin) - ox) + B * ((B >= 0.f ? ymax : ymin) - oy);
    return m < -thr;
}
DBW_HD bool tile_culled(const FaceRec &r, float xmin, float xmax, float ymin, float ymax) {
    if (!(r.flags & REC_CULL)) return false;
    const float s = (r.flags & REC_AREA_POS) ? 1.f : -1.f;
    return tile_outside_edge(s, r.bx, r.by, r.dbc_x, r.dbc_y, r.thr_bc, xmin, xmax, ymin, ymax) ||     // e0 = edge_fn(p, b, c)
           tile_outside_edge(s, r.cx, r.cy, r.dca_x, r.dca_y, r.thr_ca, xmin, xmax, ymin, ymax) ||     // e1 = edge_fn(p, c, a)
           tile_outside_edge(s, r.ax, r.ay, r.dab_x, r.dab_y, r.thr_ab, xmin, xmax, ymin, ymax);       // e2 = edge_fn(p, a, b)
}

// ---- one (pixel, face) evaluation ------------------------------------------------------------------------------------------------
// squared distance of p to the segment o -> o + d (point_line_dist of the oracle with a = o, b = o + d given by its delta);
// `far` = the segment's end point (used when the edge is degenerate)
template <bool FAST>
DBW_HD float seg_dist(f2 p, float ox, float oy, float dx, float dy, float l2, float rr, bool degenerate, float farx, float fary,
                      float pox, float poy, bool &unsafe) {
    if (degenerate) return (p.x - farx) * (p.x - farx) + (p.y - fary) * (p.y - fary);
    const float num = dx * pox + dy * poy;            // pox = p.x - ox, poy = p.y - oy
    float t;
    if (FAST) { t = div_fast(num, l2, rr); unsafe |= guard_key(num) < GUARD_LO; }
    else t = num / l2;
    // (FAST: t is finite -- guarded numerator over a squared length > 1e-8 -- so the clamp is one median instruction)
    const float tt = FAST ? clamp01(t) : (t < 0.f ? 0.f : (t > 1.f ? 1.f : t));
    const float qx = ox + tt * dx, qy = oy + tt * dy;
    return (p.x - qx) * (p.x - qx) + (p.y - qy) * (p.y - qy);
}

// Steps 3-8 of SURVEY A.5 for a face that passed the box test.  Returns whether the pixel keeps the face; pz, sd (signed squared
// distance) and bc (the stored barycentrics) are only meaningful then.  FAST: shared-reciprocal divisions; `unsafe` comes back true
// when an operand left the guarded range, in which case the caller re-evaluates with FAST = false (plain IEEE divisions).
template <bool FAST>
DBW_HD bool eval_pair(const FaceRec &r, f2 p, float blur, int persp, int clipb, float &pz, float &sd, f3 &bc, bool &unsafe,
                      bool sign_only = false) {
    unsafe = false;
    const float pax = p.x - r.ax, pay = p.y - r.ay, pbx = p.x - r.bx, pby = p.y - r.by, pcx = p.x - r.cx, pcy = p.y - r.cy;
    const float e0 = pbx * r.dbc_y - pby * r.dbc_x;       // edge_fn(p, b, c)
    const float e1 = pcx * r.dca_y - pcy * r.dca_x;       // edge_fn(p, c, a)
    const float e2 = pax * r.dab_y - pay * r.dab_x;       // edge_fn(p, a, b)
    if (blur == 0.f) {
        // hard pass: a pixel outside the triangle can never be kept, and "outside" is decided exactly by the signs of the edge
        // functions (b_i = e_i / area <= 0 for some i) before any division
        const bool pos = r.area > 0.f;
        if (e0 == 0.f || e1 == 0.f || e2 == 0.f || (e0 > 0.f) != pos || (e1 > 0.f) != pos || (e2 > 0.f) != pos) return false;
    }
    f3 b0;
    if (FAST) {
        b0.x = div_fast(e0, r.area, r.r_area); b0.y = div_fast(e1, r.area, r.r_area); b0.z = div_fast(e2, r.area, r.r_area);
        unsafe |= !guard3(e0, e1, e2);
    } else { b0.x = e0 / r.area; b0.y = e1 / r.area; b0.z = e2 / r.area; }
    f3 bp = b0;
    if (persp) {
        const float t0 = b0.x * r.z1 * r.z2;
        const float t1 = r.z0 * b0.y * r.z2;
        const float t2 = r.z0 * r.z1 * b0.z;
        float denom = t0 + t1 + t2;
        if (!(denom > DBW_EPS)) denom = DBW_EPS;
        if (FAST) {
            const float rd = rcp_refined(denom);
            bp.x = div_fast(t0, denom, rd); bp.y = div_fast(t1, denom, rd); bp.z = div_fast(t2, denom, rd);
            unsafe |= (int)!guard3(t0, t1, t2) | (int)!guard_den(denom);
        } else { bp.x = t0 / denom; bp.y = t1 / denom; bp.z = t2 / denom; }
    }
    bc = bp;
    if (clipb) {
        const float c0 = bp.x > 0.f ? bp.x : 0.f, c1 = bp.y > 0.f ? bp.y : 0.f, c2 = bp.z > 0.f ? bp.z : 0.f;
        float s = c0 + c1 + c2;
        if (!(s > 1e-5f)) s = 1e-5f;
        if (FAST) {
            const float rs = rcp_refined(s);
            bc.x = div_fast(c0, s, rs); bc.y = div_fast(c1, s, rs); bc.z = div_fast(c2, s, rs);
            unsafe |= (int)!guard3(c0, c1, c2) | (int)!(s < 1.0995116e12f);
        } else { bc.x = c0 / s; bc.y = c1 / s; bc.z = c2 / s; }
    }
    pz = bc.x * r.z0 + bc.y * r.z1 + bc.z * r.z2;
    if (pz < 0.f) return false;
    // hard passes whose consumer only looks at the SIGN of the distance (fused forward, blur == 0: a kept pixel is inside, its opacity
    // is 1 and the distance carries no gradient): any negative number will do
    if (sign_only && blur == 0.f) { sd = -1.f; return true; }
    // point_tri_dist: edges (v0, v1), (v0, v2), (v1, v2); the middle one runs along -(a - c)
    const float e01 = seg_dist<FAST>(p, r.ax, r.ay, r.dab_x, r.dab_y, r.l2_ab, r.r_ab, (r.flags & REC_DEG_AB) != 0, r.bx, r.by, pax, pay, unsafe);
    const float e02 = seg_dist<FAST>(p, r.ax, r.ay, -r.dca_x, -r.dca_y, r.l2_ac, r.r_ac, (r.flags & REC_DEG_AC) != 0, r.cx, r.cy, pax, pay, unsafe);
    const float e12 = seg_dist<FAST>(p, r.bx, r.by, r.dbc_x, r.dbc_y, r.l2_bc, r.r_bc, (r.flags & REC_DEG_BC) != 0, r.cx, r.cy, pbx, pby, unsafe);
    const float m = e01 < e02 ? e01 : e02;
    const float dist = m < e12 ? m : e12;
    const bool inside = bp.x > 0.f && bp.y > 0.f && bp.z > 0.f;
    if (!inside && dist >= blur) return false;
    sd = inside ? -dist : dist;
    return true;
}

// ---- launch order of a pass's tiles ------------------------------------------------------------------------------------------------
// Position, inside its XCD segment of `len` tiles, of the tile with rank r among the segment's O OCCUPIED tiles (ordered by face-count
// class, heaviest first) or, occupied == false, among its len - O EMPTY ones (raster_bin.h: the ranks; raster.hip: work_scatter_kernel).
//   1. The occupied tiles are spread evenly -- slot r at ceil((r + 1) len / O) - 1 -- and the empty ones fill what is left, the e-th at
//      floor(e len / (len - O)): the two sets partition [0, len) (the empty tiles' loss epilogue is pure memory traffic, which hides
//      behind the arithmetic of the occupied ones instead of piling up at the end).
//   2. The positions inside every window of 64 are then bit-reversed and XORed with a hash of the window's number.  Evenly spaced slots are
//      PERIODIC, and the hardware deals the workgroups of an XCD to its four shader engines in launch order, round robin: with one tile in four
//      occupied -- config 2 -- the slots sit at positions 4 r + 3, every occupied tile of the segment lands on the same shader engine,
//      that engine's 8 CUs fill up (five waves per SIMD), the in-order dispatcher waits for them and the other three engines idle.  Measured
//      (profiles/r06_experiments.md): 144 workgroups in flight on that XCD instead of 550, the pass 0.78 instead of 0.27 ms in the
//      scene states where len / O came within 1e-3 of 4, 15-20 % on the slowest segment within a few percent of it.  The bit reversal
//      maps a stride-2^k comb onto a run of consecutive positions (which round robin spreads over the engines), the hash moves the run
//      from window to window, and the mix of light and heavy tiles stays what it was at the scale of the ~550 workgroups an XCD holds.
//      A bijection of every full window; the last, partial window keeps step 1's positions.
constexpr unsigned WORK_WINDOW_LOG2 = 6;
DBW_HD unsigned bit_reverse32(unsigned x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x);
#else
    x = (x >> 16) | (x << 16);
    x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
    x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
    return ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
#endif
}
// step 2 on its own: position p of a list of `len` -> its place after the scramble (a bijection of [0, len)).  Also the order in which
// texbin_reduce_kernel takes the texture bins: neighbouring bins differ by orders of magnitude in records, and hot bins at a regular
// stride are the same comb (shade_blend.hip)
DBW_HD unsigned window_scramble(unsigned p, unsigned len) {
    constexpr unsigned WL = WORK_WINDOW_LOG2, WM = (1u << WL) - 1u;
#ifndef DBW_WORK_NO_SCRAMBLE      // (tools/diag/r06_spike4.py: the comb itself, to see where the hardware puts it)
    if (p < (len & ~WM)) p = (p & ~WM) | ((bit_reverse32(p & WM) >> (32 - WL)) ^ (((p >> WL) * 2654435761u) >> (32 - WL)));
#endif
    return p;
}
DBW_HD unsigned work_position(bool occupied, int r, int O, int len) {
    unsigned p;
    const bool small = (unsigned long long)len * (unsigned long long)(len + 1) < (1ull << 32);
    if (occupied) {
        if (small) p = ((unsigned)(r + 1) * (unsigned)len + (unsigned)O - 1u) / (unsigned)O - 1u;
        else p = (unsigned)((((unsigned long long)(r + 1)) * (unsigned long long)len + (unsigned long long)O - 1ull) / (unsigned long long)O - 1ull);
    } else {
        const int E = len - O;
        if (small) p = ((unsigned)r * (unsigned)len) / (unsigned)E;
        else p = (unsigned)(((unsigned long long)r * (unsigned long long)len) / (unsigned long long)E);
    }
    return window_scramble(p, (unsigned)len);
}

// ---- per-pixel top-K list --------------------------------------------------------------------------------------------------------
// key = depth bits (32) | packed face id (27) | payload slot (5): unsigned 64-bit order == the oracle's (pz, face id) tuple order
// for pz >= +0 (pz < 0 is rejected; -0 is canonicalised to +0).  Empty entries are sentinels (all ones above the slot field) that
// each own a distinct payload slot, so the entry that falls off the end of the list always hands a free slot to the one that enters.
constexpr int TOPK_ID_BITS = 27;
constexpr uint32_t TOPK_ID_MASK = (1u << TOPK_ID_BITS) - 1u;
constexpr uint64_t TOPK_EMPTY = 0xffffffffffffffe0ull;

// PAY3: the payload home keeps (signed distance, b0, b1) only, 12 B per entry in planes of `stride` floats, and b2 comes back as
// (1 - b0) - b1.  Only for consumers that resolve the barycentrics into texture coordinates and never output them (the fused
// forward's uv-fragment path: clipped barycentrics sum to 1 within 2 ulp, far inside its 1e-4 bar); it cuts the LDS per pixel by a
// quarter, which is what bounds the number of resident waves.
template <int KMAX, bool PAY3 = false>
struct TopK {
    // the two halves of the keys are kept as separate 32-bit arrays: every select below is then a plain v_cndmask on the mask of ONE
    // 64-bit compare (selects of whole 64-bit values are canonicalised into umin / umax, each lowered with a compare of its own)
    uint32_t khi[KMAX], klo[KMAX];
    pay4 pay1;                   // KMAX == 1: the single payload stays in registers
    int cnt;                     // valid entries (maintained by insert_ordered only)

    DBW_HD static uint64_t cat(uint32_t hi, uint32_t lo) { return ((uint64_t)hi << 32) | (uint64_t)lo; }
    DBW_HD void init() {
#pragma unroll
        for (int i = 0; i < KMAX; ++i) { khi[i] = 0xffffffffu; klo[i] = (uint32_t)TOPK_EMPTY | (uint32_t)i; }
        pay1.x = pay1.y = pay1.z = pay1.w = -1.f;
        cnt = 0;
    }
    DBW_HD static uint32_t key_hi(float pz) { return f2u(pz + 0.0f); }
    DBW_HD static uint32_t key_lo(int id, uint32_t slot) { return ((uint32_t)id << 5) | slot; }
    DBW_HD void last(int K, uint32_t &lhi, uint32_t &llo) const {
        lhi = khi[KMAX - 1]; llo = klo[KMAX - 1];
        if (K != KMAX) {
#pragma unroll
            for (int i = 0; i < KMAX - 1; ++i) if (i == K - 1) { lhi = khi[i]; llo = klo[i]; }
        }
    }
    // entry i of `home` for this pixel: home[slot * stride + lane]
    DBW_HD void store(pay4 *home, int stride, int lane, uint32_t slot, const pay4 &v, bool on) {
        if (KMAX == 1) { if (on) pay1 = v; }
        else if (PAY3) {
            float *h = (float *)home + slot * 3 * stride + lane;
            if (on) { h[0] = v.x; h[stride] = v.y; h[2 * stride] = v.z; }
        } else if (on) home[slot * stride + lane] = v;
    }
    DBW_HD pay4 load(const pay4 *home, int stride, int lane, uint32_t slot) const {
        if (KMAX == 1) return pay1;
        if (PAY3) {
            const float *h = (const float *)home + slot * 3 * stride + lane;
            pay4 v;
            v.x = h[0]; v.y = h[stride]; v.z = h[2 * stride];
            v.w = (1.f - v.y) - v.z;
            return v;
        }
        return home[slot * stride + lane];
    }
    // sorted insert; the displaced largest entry falls off the end (== emplace_back, sort, pop_back if size > K).  Rank-and-shift:
    // m_i = [cand < key_i] is monotone in i because the list is sorted, so key_i' = m_i ? (m_{i-1} ? key_{i-1} : cand) : key_i --
    // one 64-bit compare and two selects per half and slot, all slots independent (no compare-exchange dependency chain).  Entries
    // at and beyond K only ever hold sentinels or stale copies that nothing reads.
    DBW_HD void insert(int K, bool on, float pz, int id, const pay4 &v, pay4 *home, int stride, int lane) {
        uint32_t lhi, llo;
        last(K, lhi, llo);
        const uint32_t slot = llo & 31u;
        uint32_t chi = key_hi(pz), clo = key_lo(id, slot);
        const bool ins = on && cat(chi, clo) < cat(lhi, llo);
        if (!ins) { chi = 0xffffffffu; clo = 0xffffffffu; }
        bool m[KMAX];
#pragma unroll
        for (int i = 0; i < KMAX; ++i) m[i] = cat(chi, clo) < cat(khi[i], klo[i]);
#pragma unroll
        for (int i = KMAX - 1; i >= 0; --i) {
            const int j = i > 0 ? i - 1 : 0;
            const bool mb = i > 0 && m[j];
            const uint32_t bhi = mb ? khi[j] : chi, blo = mb ? klo[j] : clo;
            khi[i] = m[i] ? bhi : khi[i];
            klo[i] = m[i] ? blo : klo[i];
        }
        store(home, stride, lane, slot, v, ins);
    }
    // ---- ordered insert (the one the kernels use) ---------------------------------------------------------------------------------
    // Same result as insert() under the one precondition every caller meets: candidates arrive in ASCENDING face id (tiles walk
    // their face lists in face order), so a candidate whose depth equals an entry's goes behind it -- the (pz, id) order of the
    // oracle -- and the 64-bit key compare reduces to a 32-bit compare of the depth words; the depth words of the shifted list are
    // then one median each (sorted a[i-1] <= a[i]: the new a[i] is the candidate clamped into [a[i-1], a[i]]): 4 instructions per
    // slot instead of 6.  Payload slots are handed out in arrival order (`cnt`) until the list is full, after that the entry that
    // falls off the end hands over its slot.  Do not mix with insert() on one list (that one takes free slots from the sentinels).
    // (Measured and dropped, profiles/r03_experiments.md: shifting only the slots below the wave's deepest list -- 6.9 of 10 on
    // average at config 2 -- through a switch over the depth; the register copies at the join cost more than the skipped slots.)
    DBW_HD void insert_ordered(int K, bool on, float pz, int id, const pay4 &v, pay4 *home, int stride, int lane) {
        uint32_t lhi, llo;
        last(K, lhi, llo);
        uint32_t chi = key_hi(pz);
        const bool ins = on && chi < lhi;                 // (a list that is not full ends in a sentinel: always admitted)
        const uint32_t slot = cnt < K ? (uint32_t)cnt : (llo & 31u);
        const uint32_t clo = key_lo(id, slot);
        if (!ins) chi = 0xffffffffu;
        bool m[KMAX];
#pragma unroll
        for (int i = 0; i < KMAX; ++i) m[i] = chi < khi[i];
#pragma unroll
        for (int i = KMAX - 1; i > 0; --i) {
            klo[i] = m[i] ? (m[i - 1] ? klo[i - 1] : clo) : klo[i];
            khi[i] = umed3(khi[i - 1], chi, khi[i]);
        }
        klo[0] = m[0] ? clo : klo[0];
        khi[0] = m[0] ? chi : khi[0];
        store(home, stride, lane, slot, v, ins);
        cnt += (ins && cnt < K) ? 1 : 0;
    }
    DBW_HD void cswap(int i) {
        const bool c = cat(khi[i + 1], klo[i + 1]) < cat(khi[i], klo[i]);
        const uint32_t ah = khi[i], al = klo[i], bh = khi[i + 1], bl = klo[i + 1];
        khi[i] = c ? bh : ah; klo[i] = c ? bl : al;
        khi[i + 1] = c ? ah : bh; klo[i + 1] = c ? al : bl;
    }
    // sibling rule (clipped split quads): if face `nb` is already in the list, keep whichever of the two lies closer to the pixel
    // (in place, then re-sort) and return true -- the caller must not insert then
    DBW_HD bool sibling(int K, bool on, int nb, float dist, float pz, int id, const pay4 &v, pay4 *home, int stride, int lane) {
        bool found = false;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
            if (i < K) {
                const bool m = on && !found && khi[i] != 0xffffffffu && ((klo[i] >> 5) & TOPK_ID_MASK) == (uint32_t)nb;
                if (m) {
                    found = true;
                    const uint32_t slot = klo[i] & 31u;
                    const float od = load(home, stride, lane, slot).x;
                    const float nd = od < 0.f ? -od : od;
                    if (dist < nd) {
                        khi[i] = key_hi(pz); klo[i] = key_lo(id, slot);
                        store(home, stride, lane, slot, v, true);
                    }
                }
            }
        }
        // one entry may be out of place: one forward + one backward adjacent pass restores the order (a no-op where nothing changed)
#pragma unroll
        for (int i = 0; i < KMAX - 1; ++i) if (i + 1 < K) cswap(i);
#pragma unroll
        for (int i = KMAX - 2; i >= 0; --i) if (i + 1 < K) cswap(i);
        return found;
    }
    // entry k (compile-time constant in the unrolled consumers): false for an empty slot
    DBW_HD bool get(int k, const pay4 *home, int stride, int lane, float &pz, int &fi, pay4 &v) const {
        if (khi[k] == 0xffffffffu) return false;
        pz = u2f(khi[k]);
        fi = (int)((klo[k] >> 5) & TOPK_ID_MASK);
        v = load(home, stride, lane, klo[k] & 31u);
        return true;
    }
    DBW_HD bool valid(int k) const { return khi[k] != 0xffffffffu; }
};

}  // namespace dbw
