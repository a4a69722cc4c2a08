// Backward of ONE pixel of the hard single-layer pass (sky + ground: K = 1, sigma = 0, no learned opacity): the body of
// render_bwd_hard_kernel (shade_blend.hip), which loads the hard uv-fragment and the pixel's image gradient and calls it.  A function
// of its own since round 6, when it also ran in the epilogue of the training step's fused forward (the folded env layer has the
// fragment in registers, the composite has just produced d loss / d env colour): bit-equal gradients, and SLOWER -- 0.891 -> 1.000 ms
// per step at 49 views, of which 0.144 ms are the flushes of per-WAVE tables (four times the global atomics of this kernel's 256-pixel
// workgroups, on the same few ground faces and texel cells) and the rest simply the same instructions in a kernel that is bound by
// instruction issue already (profiles/r06_experiments.md).  A kept pixel lies inside its face, its opacity is 1, nothing flows through
// the distance:
//   colour -> texels:  the footprint's texels, merged where they fall into one stored cell, neighbouring pixels of one texel merged in
//                      registers (lane_merge), then the workgroup's LDS texel table (what does not fit goes straight to memory);
//   colour -> uv -> barycentrics -> vertices, only for faces whose vertices are variables (j >= geom_begin: the sky dome is a buffer),
//                      barycentrics rebuilt from the pixel position as the rasteriser backward does; gradient-only arithmetic on v_rcp_f32.
// Same mathematics as shade_blend_bwd_kernel<true, false, true>.  Wave-collective: every lane of the wave calls it.
#pragma once
#include "shade_common.h"

namespace dbw {

// (a 64-slot texel table and a 32-slot face table: a 16x16-pixel tile of the magnified env maps touches a few cells and a handful of
// large faces; with the soft pass's 512 / 128 slots the clears, the flush scans and the lost residency cost a quarter of the kernel:
// 0.22 -> 0.16 ms with decimated maps, 0.31 -> 0.26 ms at full resolution; 16 slots and fewer overflow at full resolution (0.9 ms).
// What does not fit goes straight to memory, as always)
#ifndef DBW_HARD_TEX_LOG2
#define DBW_HARD_TEX_LOG2 6
#endif
#ifndef DBW_HARD_FACE_LOG2
#define DBW_HARD_FACE_LOG2 5
#endif
typedef LdsAgg<3, DBW_HARD_TEX_LOG2> HardTexAgg;
typedef LdsAgg<9, DBW_HARD_FACE_LOG2> HardFaceAgg;      // a tile of the hard pass sees a handful of (large) faces

struct EnvBwdArgs {
    const int *map_desc; const float *maps; const float *face_uvs;      // the pass's maps and per-face texture coordinates
    const int *code; const float *cw;                                   // clip conversion of the clipped faces (NULL: unclipped scene)
    const float *fv;                                                    // clipped face vertices (F_total, 3, 3)
    float *gmaps, *gfv;                                                 // out: d loss / d maps, d loss / d clipped face vertices
    int H, W, geom_begin, want_bary, persp;
    float ndc[4];                                                       // pixel -> NDC constants from the host (CoarseBins::ndc)
};

// valid: the pixel holds a fragment -- clipped face fc, texture coordinates (u, v), jm = original face | map << 20 -- with colour gradient gc
__device__ __forceinline__ void env_bwd_pixel(const EnvBwdArgs &A, HardTexAgg &tex_agg, HardFaceAgg &face_agg, bool valid, int fc, float u, float v,
                                              int jm, const float (&gc)[3], int xi, int yi) {
    const int j = jm & 0xfffff, map = jm >> 20;
    const bool tex = valid && (gc[0] != 0.f || gc[1] != 0.f || gc[2] != 0.f);
    Sample s;
    s.a00 = s.a01 = s.a10 = s.a11 = 0;
    s.w00 = s.w01 = s.w10 = s.w11 = 0.f;
    if (__ballot(tex) != 0ull) {
        const int *md = A.map_desc + (valid ? map : 0) * 8;
        footprint_desc(u, v, md[0], md[1], md[2], md[3], md[4], md[5], s);
        // colour -> texels: merge the footprint's texels that fall into the same stored cell
        float w00 = s.w00, w01 = s.w01, w10 = s.w10, w11 = s.w11;
        if (s.a01 == s.a00) { w00 += w01; w01 = 0.f; }
        if (s.a10 == s.a00) { w00 += w10; w10 = 0.f; }
        if (s.a11 == s.a00) { w00 += w11; w11 = 0.f; }
        else if (s.a11 == s.a01) { w01 += w11; w11 = 0.f; }
        else if (s.a11 == s.a10) { w10 += w11; w11 = 0.f; }
        const int ad[4] = {s.a00, s.a01, s.a10, s.a11};
        const float wt[4] = {w00, w01, w10, w11};
        // neighbouring pixels that hit the same texel (magnified maps: most of them) are merged in registers first (lane_merge, up to 16
        // lanes into one; full-resolution env maps 0.27 -> 0.24 ms, decimated ones unchanged)
        // (tap 0 and the lane's first other tap with weight as wave-wide passes; the rest -- footprints that cross a cell border in x AND y:
        // few lanes on magnified / decimated maps, every lane on full-resolution ones -- lane by lane or wave-wide accordingly.  See the uv
        // backward)
        const int f = wt[1] != 0.f ? 1 : (wt[2] != 0.f ? 2 : 3);
        const int a2[2] = {ad[0], f == 1 ? ad[1] : (f == 2 ? ad[2] : ad[3])};
        const float w2[2] = {wt[0], f == 1 ? wt[1] : (f == 2 ? wt[2] : wt[3])};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float val[3] = {gc[0] * w2[q], gc[1] * w2[q], gc[2] * w2[q]};
            const bool on = tex && w2[q] != 0.f;
            tex_agg.template add_wave_merged<4>(A.gmaps, (int)((unsigned)a2[q] / 3u), val, on);
        }
        const bool rest = tex && ((f == 1 && (wt[2] != 0.f || wt[3] != 0.f)) || (f == 2 && wt[3] != 0.f));
        const unsigned long long rm = __ballot(rest);
        if (rm != 0ull) {
            const bool wide = __popcll(rm) > 16;
#pragma unroll
            for (int q = 2; q < 4; ++q) {
                float val[3] = {gc[0] * wt[q], gc[1] * wt[q], gc[2] * wt[q]};
                const bool on = rest && q > f && wt[q] != 0.f;
                const int key = (int)((unsigned)ad[q] / 3u);
                if (wide) tex_agg.template add_wave_merged<4>(A.gmaps, key, val, on);
                else if (on) tex_agg.add(A.gmaps, key, val);
            }
        }
    }
    // colour -> uv -> barycentrics -> vertices, for the faces whose vertices are variables
    const bool geom = tex && A.want_bary != 0 && j >= A.geom_begin;
    float g9[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool has_g9 = false;
    if (__ballot(geom) != 0ull) {
        if (geom) {
            float gu, gv;
            sample_grad_uv(A.maps, s, gc, gu, gv);
            const float *uv = A.face_uvs + (long long)j * 6;
            const float go[3] = {gu * uv[0] + gv * uv[1], gu * uv[2] + gv * uv[3], gu * uv[4] + gv * uv[5]};
            int cd = -1;
            float w2 = 0.f, w3 = 0.f;
            if (A.code) {
                cd = A.code[fc]; w2 = A.cw[(long long)fc * 2]; w3 = A.cw[(long long)fc * 2 + 1];
            }
            float gb[3] = {0.f, 0.f, 0.f};
            convert_bary_bwd(cd, w2, w3, go, gb);
            if (gb[0] != 0.f || gb[1] != 0.f || gb[2] != 0.f) {
                has_g9 = true;
                f2 pndc;          // (same bits as pix_to_ndc: the shared-reciprocal division is exact for these operands, raster_math.h)
                pndc.x = pix_to_ndc_fast(A.W - 1 - xi, ndc_axis_given(A.W, A.ndc[0], A.ndc[1]));
                pndc.y = pix_to_ndc_fast(A.H - 1 - yi, ndc_axis_given(A.H, A.ndc[2], A.ndc[3]));
                const float *q = A.fv + (long long)fc * 9;
                const f2 a{q[0], q[1]}, b{q[3], q[4]}, c{q[6], q[7]};
                const float z0 = q[2], z1 = q[5], z2 = q[8];
                // (gradient-only arithmetic: v_rcp_f32 instead of ~15 IEEE divisions per pixel, as in the soft backward -- held at 1e-4)
                const f3 bary0 = bary_fwd<true>(pndc, a, b, c);
                const f3 bp = A.persp ? persp_fwd<true>(bary0, z0, z1, z2) : bary0;
                f3 gg3{gb[0], gb[1], gb[2]};
                gg3 = clip_bwd<true>(bp, gg3);
                float pz0 = 0.f, pz1 = 0.f, pz2 = 0.f;
                if (A.persp) gg3 = persp_bwd<true>(bary0, z0, z1, z2, gg3, pz0, pz1, pz2);
                f2 e0, e1, e2;
                bary_bwd<true>(pndc, a, b, c, gg3, e0, e1, e2);
                g9[0] = e0.x; g9[1] = e0.y; g9[2] = pz0;
                g9[3] = e1.x; g9[4] = e1.y; g9[5] = pz1;
                g9[6] = e2.x; g9[7] = e2.y; g9[8] = pz2;
            }
        }
        face_agg.add_wave(A.gfv, valid ? fc : 0, g9, has_g9);
    }
}

}  // namespace dbw
