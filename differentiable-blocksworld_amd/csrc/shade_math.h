// Shading arithmetic shared by the device kernels (hipcc) and the host (g++: tests/test_host_shade_math.py builds it into a
// checker-side shared object and compares it, without a GPU, with torch's grid_sample and with golden vectors of the reference's own
// layered_rgb_blend): the bilinear texture footprint (grid_sample semantics, v flip, circular u wrap, decimation shift), the
// barycentric back-conversion of clipped faces, and the layered blend with its back-to-front backward recurrences.
#pragma once
#include "raster_math.h"      // DBW_HD

namespace dbw {

struct Sample {   // bilinear footprint of one fragment
    int a00, a01, a10, a11;         // float offsets of the 4 texels (RGB triplets) in `maps` (map_desc offsets are int32: < 2^31 floats)
    float w00, w01, w10, w11;
    float dudx, dvdy;               // d(ix)/du, d(iy)/dv (0 when clamped at the border)
    float wx0, wx1, wy0, wy1;
    int r0, c0, r1, c1, ws;         // stored-resolution texel coordinates of the footprint and stored row width
};

DBW_HD float sel3(int i, float a, float b, float c) { return i == 0 ? a : (i == 1 ? b : c); }

// Experiment switch (profiles/r03_experiments.md): with 1 the bilinear fetch and the front-to-back blend contract a*b+c into one fma
// (~15 VALU instructions per layer, forward fg pass 0.306 -> 0.302 ms: inside the noise).  Off: the operator-level and the fused kernels
// then stay bit-equal on the hard pass (the contraction the compiler picks depends on the code around the inlined function).  Never
// for the texture coordinates (convert_bary, interp_uv, footprint_desc): floor() of the sample position picks the texel cell, the
// gradient to (u, v) is piecewise constant per cell, and one ulp of difference to the reference's unfused arithmetic moves pixels
// across cell borders (measured: 0.7 % of the env pass's vertex gradient at config 2).
#ifndef DBW_SHADE_FMA
#define DBW_SHADE_FMA 0
#endif
#if DBW_SHADE_FMA && defined(__clang__)
#define DBW_FMA_SCOPE _Pragma("clang fp contract(fast)")
#else
#define DBW_FMA_SCOPE
#endif

DBW_HD void convert_bary(int cd, float w2, float w3, const float b[3], float bo[3]) {
    if (cd < 0) { bo[0] = b[0]; bo[1] = b[1]; bo[2] = b[2]; return; }
    const int i1 = cd & 3, kind = cd >> 2;
    float o1, o2, o3;
    if (kind == 0) { o1 = b[0] * (1.f - w2) + b[1] * (1.f - w3) + b[2]; o2 = b[0] * w2; o3 = b[1] * w3; }
    else if (kind == 1) { o1 = b[0] * (1.f - w2) + b[2] * (1.f - w3); o2 = b[0] * w2 + b[1]; o3 = b[2] * w3; }
    else { o1 = b[0] * (1.f - w3); o2 = b[1]; o3 = b[0] * w3 + b[2]; }
    // slot i1 <- o1, slot i1+1 <- o2, slot i1+2 <- o3 (mod 3)
    bo[0] = sel3(i1, o1, o3, o2);
    bo[1] = sel3(i1, o2, o1, o3);
    bo[2] = sel3(i1, o3, o2, o1);
}

// texture coordinates of a fragment: the (clip-converted) barycentrics times the face's three (u, v) vertices
DBW_HD void interp_uv(const float bo[3], const float uv[6], float &u, float &v) {
    u = bo[0] * uv[0] + bo[1] * uv[2] + bo[2] * uv[4];
    v = bo[0] * uv[1] + bo[1] * uv[3] + bo[2] * uv[5];
}

DBW_HD void convert_bary_bwd(int cd, float w2, float w3, const float go[3], float gb[3]) {
    if (cd < 0) { gb[0] = go[0]; gb[1] = go[1]; gb[2] = go[2]; return; }
    const int i1 = cd & 3, kind = cd >> 2;
    const float g1 = sel3(i1, go[0], go[1], go[2]), g2 = sel3(i1, go[1], go[2], go[0]), g3 = sel3(i1, go[2], go[0], go[1]);
    if (kind == 0) { gb[0] = g1 * (1.f - w2) + g2 * w2; gb[1] = g1 * (1.f - w3) + g3 * w3; gb[2] = g1; }
    else if (kind == 1) { gb[0] = g1 * (1.f - w2) + g2 * w2; gb[1] = g2; gb[2] = g1 * (1.f - w3) + g3 * w3; }
    else { gb[0] = g1 * (1.f - w3) + g3 * w3; gb[1] = g2; gb[2] = g3; }
}

// c mod w for c in [-pad_left, w + pad_right): one conditional add/subtract when the pads do not exceed the width (the integer
// modulo is ~25 instructions), the general form otherwise
DBW_HD int wrap_col(int c, int w) {
    if (c < 0) c += w;
    if (c >= w) c -= w;
    if ((unsigned)c >= (unsigned)w) { c %= w; if (c < 0) c += w; }
    return c;
}

// grid_sample(bilinear, align_corners=True, padding_mode='border') on the v-flipped, circularly u-padded map whose descriptor is
// (off, h, w, pl, pr, sh)
DBW_HD void footprint_desc(float u, float v, int off, int h, int w, int pl, int pr, int sh, Sample &s) {
    const int wp = w + pl + pr;
    float ix = ((u * 2.f - 1.f) + 1.f) / 2.f * (float)(wp - 1);
    float iy = ((v * 2.f - 1.f) + 1.f) / 2.f * (float)(h - 1);
    s.dudx = (float)(wp - 1); s.dvdy = (float)(h - 1);
    // clip_coordinates_set_grad of torch's grid_sampler: no gradient at or beyond the border
    if (!(ix > 0.f)) { ix = 0.f; s.dudx = 0.f; } else if (ix >= (float)(wp - 1)) { ix = (float)(wp - 1); s.dudx = 0.f; }
    if (!(iy > 0.f)) { iy = 0.f; s.dvdy = 0.f; } else if (iy >= (float)(h - 1)) { iy = (float)(h - 1); s.dvdy = 0.f; }
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const int x1 = x0 + 1 < wp - 1 ? x0 + 1 : wp - 1, y1 = y0 + 1 < h - 1 ? y0 + 1 : h - 1;
    s.wx1 = ix - fx; s.wx0 = 1.f - s.wx1;
    s.wy1 = iy - fy; s.wy0 = 1.f - s.wy1;
    // padded column -> source column (circular pad), flipped row -> source row
    int c0 = wrap_col(x0 - pl, w), c1 = wrap_col(x1 - pl, w);
    // stored resolution = (h >> sh, w >> sh): a decimated map (avg_pool d + nearest upsample, dbw.py:276-278,331-334) is
    // kept at cell resolution and the nearest upsampling is this shift
    const int r0 = (h - 1 - y0) >> sh, r1 = (h - 1 - y1) >> sh, ws = w >> sh;
    c0 >>= sh; c1 >>= sh;
    s.a00 = off + (r0 * ws + c0) * 3; s.a01 = off + (r0 * ws + c1) * 3;
    s.a10 = off + (r1 * ws + c0) * 3; s.a11 = off + (r1 * ws + c1) * 3;
    s.w00 = s.wx0 * s.wy0; s.w01 = s.wx1 * s.wy0; s.w10 = s.wx0 * s.wy1; s.w11 = s.wx1 * s.wy1;
    s.r0 = r0; s.c0 = c0; s.r1 = r1; s.c1 = c1; s.ws = ws;
}

DBW_HD void fetch(const float *maps, const Sample &s, float c[3]) {
    DBW_FMA_SCOPE
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
        c[ch] = maps[s.a00 + ch] * s.w00 + maps[s.a01 + ch] * s.w01 + maps[s.a10 + ch] * s.w10 + maps[s.a11 + ch] * s.w11;
}

// The same footprint for consumers that only SAMPLE (the fused forward: no gradient to (u, v) here, the backward rebuilds its own): no
// border-gradient bookkeeping, the border clamp as max / min (same value for every input, NaN included: !(ix > 0) -> 0), and the four
// texels as 32-bit BYTE offsets from `maps` -- a wave-uniform base plus a 32-bit per-lane offset is one load instruction with no 64-bit
// address arithmetic in front of it (contract of the uv-fragment passes, include/dbw_hip.h: the maps buffer is smaller than 2^30 floats).
struct SampleFwd { unsigned b00, b01, b10, b11; float w00, w01, w10, w11; };
DBW_HD void footprint_fwd(float u, float v, int off, int h, int w, int pl, int pr, int sh, SampleFwd &s) {
    const int wp = w + pl + pr;
    float ix = ((u * 2.f - 1.f) + 1.f) / 2.f * (float)(wp - 1);
    float iy = ((v * 2.f - 1.f) + 1.f) / 2.f * (float)(h - 1);
    ix = __builtin_fminf(__builtin_fmaxf(ix, 0.f), (float)(wp - 1));
    iy = __builtin_fminf(__builtin_fmaxf(iy, 0.f), (float)(h - 1));
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const int x1 = x0 + 1 < wp - 1 ? x0 + 1 : wp - 1, y1 = y0 + 1 < h - 1 ? y0 + 1 : h - 1;
    const float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
    int c0 = wrap_col(x0 - pl, w), c1 = wrap_col(x1 - pl, w);
    const int r0 = (h - 1 - y0) >> sh, r1 = (h - 1 - y1) >> sh, ws = w >> sh;
    c0 >>= sh; c1 >>= sh;
    s.b00 = ((unsigned)off + (unsigned)(r0 * ws + c0) * 3u) * 4u; s.b01 = ((unsigned)off + (unsigned)(r0 * ws + c1) * 3u) * 4u;
    s.b10 = ((unsigned)off + (unsigned)(r1 * ws + c0) * 3u) * 4u; s.b11 = ((unsigned)off + (unsigned)(r1 * ws + c1) * 3u) * 4u;
    s.w00 = wx0 * wy0; s.w01 = wx1 * wy0; s.w10 = wx0 * wy1; s.w11 = wx1 * wy1;
}
DBW_HD void fetch_fwd(const float *maps, const SampleFwd &s, float c[3]) {
    DBW_FMA_SCOPE
    const char *m = (const char *)maps;
    const float *t00 = (const float *)(m + s.b00), *t01 = (const float *)(m + s.b01), *t10 = (const float *)(m + s.b10), *t11 = (const float *)(m + s.b11);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) c[ch] = t00[ch] * s.w00 + t01[ch] * s.w01 + t10[ch] * s.w10 + t11[ch] * s.w11;
}


// d (sum_ch gc[ch] * colour[ch]) / d (u, v) of the bilinear sample (grid_sampler_2d_backward's gradient to the grid, times the
// align_corners scale; zero where the coordinate was clamped at the border)
DBW_HD void sample_grad_uv(const float *maps, const Sample &s, const float gc[3], float &gu, float &gv) {
    float gix = 0.f, giy = 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float t00 = maps[s.a00 + ch], t01 = maps[s.a01 + ch], t10 = maps[s.a10 + ch], t11 = maps[s.a11 + ch];
        gix += gc[ch] * ((t01 - t00) * s.wy0 + (t11 - t10) * s.wy1);
        giy += gc[ch] * ((t10 - t00) * s.wx0 + (t11 - t01) * s.wx1);
    }
    gu = gix * s.dudx;
    gv = giy * s.dvdy;
}

// ---- layered blend (renderer.py:241-273): rgb = sum_k T_k a_k c_k + T_K bg, A = 1 - T_K, T_k = prod_{j<k} (1 - a_j) ---------------
// geometric alpha of a fragment from its signed squared distance (renderer.py:252-258): exp(-max(d, 0) / sigma), hard indicator at 0
DBW_HD float geometric_alpha(float d, float sigma) {
    if (sigma == 0.f) return d <= 0.f ? 1.f : 0.f;
    if (sigma < 0.f) return 1.f / (1.f + expf(d / -sigma));          // clip_inside = False (renderer.py:257-258): sigmoid(-d / |sigma|)
    return expf(-(d > 0.f ? d : 0.f) / sigma);
}

struct BlendFront { float T, r, g, b; };     // forward, front to back: transmittance in front of the next layer, colour so far
DBW_HD void blend_front_init(BlendFront &s) { s.T = 1.f; s.r = s.g = s.b = 0.f; }
DBW_HD void blend_front_step(BlendFront &s, float a, const float c[3]) {
    DBW_FMA_SCOPE
    const float wgt = s.T * a;
    s.r += wgt * c[0]; s.g += wgt * c[1]; s.b += wgt * c[2];
    s.T *= (1.f - a);
}
DBW_HD void blend_front_finish(const BlendFront &s, const float bg[3], float out[4]) {
    DBW_FMA_SCOPE
    out[0] = s.r + s.T * bg[0]; out[1] = s.g + s.T * bg[1]; out[2] = s.b + s.T * bg[2]; out[3] = 1.f - s.T;
}

// backward, back to front, without divisions: U = colour behind the current layer (initially the background), V = transmittance
// behind it.  d rgb / d a_k = T_k (c_k - U_k), d A / d a_k = T_k V_k; d rgb / d c_k = T_k a_k.
struct BlendBack { float U0, U1, U2, V; };
DBW_HD void blend_back_init(BlendBack &s, const float bg[3]) { s.U0 = bg[0]; s.U1 = bg[1]; s.U2 = bg[2]; s.V = 1.f; }
// -> d loss / d a_k for layer k with transmittance Tk in front of it, opacity ak, colour c; (gr, gg, gb, gA) = d loss / d (rgb, A) of
// the pixel.  Moves the state behind layer k - 1.
DBW_HD float blend_back_step(BlendBack &s, float Tk, float ak, float c0, float c1, float c2, float gr, float gg, float gb, float gA) {
    const float ga = Tk * (gr * (c0 - s.U0) + gg * (c1 - s.U1) + gb * (c2 - s.U2) + gA * s.V);
    s.U0 = ak * c0 + (1.f - ak) * s.U0;
    s.U1 = ak * c1 + (1.f - ak) * s.U1;
    s.U2 = ak * c2 + (1.f - ak) * s.U2;
    s.V = (1.f - ak) * s.V;
    return ga;
}

}  // namespace dbw
