// Fused forward of one render pass: tile rasterisation -> per-pixel top-K (registers) -> shading + layered blend, in ONE
// kernel.  The image is produced straight from the register-resident fragment lists; the fragments are also stored (once)
// because the backward pass consumes them, but they are never read back in the forward direction -- the reference path
// (PyTorch3D rasterize -> interpolate -> grid_sample -> ~12 blend kernels, renderer.py:92-94,219-273) re-reads them ~15x.
#include "raster_common.h"
#include "shade_common.h"
#include "../../include/dbw_hip.h"

#include <math.h>

using namespace dbw;

// implemented in raster.hip / shade_blend.hip
int dbw_prepare_raster(const float *face_verts, const int *first_idx, const int *num_faces, const int *neighbor, int N, long long F_total,
                       int H, int W, float margin, int cull, void *workspace, size_t workspace_bytes, dbw::CoarseBins &cb, hipStream_t s);
const dbw::FaceRec *dbw_workspace_recs(const void *workspace, long long F_total);
int dbw_fill_shade_args(ShadeArgs &A, const int32_t *pix_to_face, const float *bary, const float *dists, const int32_t *c2o,
                        const int32_t *clip_code, const float *clip_w, int Fc_stride, const float *face_uvs,
                        const int32_t *face_map, const int32_t *map_desc, const float *maps, const float *faces_alpha,
                        int alpha_len, int N, int H, int W, int K, int F, float sigma, const float *background3);

int g_render_variant = 0;
int g_render_dbg = 0;       // bit 0: plain IEEE divisions in the rasteriser, bit 1: no tile culling (dbw_debug_set_flags >> 8)
extern "C" void dbw_debug_set_render_variant(int v) { g_render_variant = v; }
void dbw_set_render_dbg(int v) { g_render_dbg = v; }

namespace {

template <int KMAX, int TW, int TH, int GROUP>
__global__ __launch_bounds__(TW * TH, DBW_RASTER_WAVES(KMAX)) void render_fwd_kernel(const FaceRec *__restrict__ recs, const float4 *__restrict__ bbox,
                                                             const int *__restrict__ first_idx, const int *__restrict__ num_faces,
                                                             float blur, int persp, int dbg,
                                                             long long total_blocks, ShadeArgs A, CoarseBins cb, int *__restrict__ p2f,
                                                             float *__restrict__ bary, float *__restrict__ dists,
                                                             float *__restrict__ image) {
    int n, xi, yi;
    TopK<KMAX> q;
    pay4 *home;
    if (!raster_tile<KMAX, TW, TH, GROUP>(recs, bbox, first_idx, num_faces, A.H, A.W, A.K, blur, persp, 1, total_blocks, cb, dbg, n, xi, yi, q, home)) return;
    if (xi >= A.W || yi >= A.H) return;
    constexpr int NT = TW * TH;
    float T = 1.f, r = 0.f, g = 0.f, b = 0.f;
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
                p2f[o.s] = valid ? ((A.tiled == 2 && k == 0) ? (fik | (cnt << FRAG_COUNT_SHIFT)) : fik) : -1;
            // internal layouts: empty slots carry only the -1 face id (the backward never reads the rest; a wave whose 64
            // pixels are all empty at this depth issues no store at all); the PyTorch3D-shaped layout 0 is filled with -1
            if (valid || A.tiled == 0) {
                dists[o.s] = v.x;
                if (A.tiled != 2) {
                    bary[o.b] = v.y;
                    bary[o.b + o.bstride] = v.z;
                    bary[o.b + 2 * o.bstride] = v.w;
                }
            }
            if (valid) {
                Frag fr;
                const float bc[3] = {v.y, v.z, v.w};
                decode_frag(A, n, fik, bc, v.x, fr);
                if (A.tiled == 2) {       // hand the resolved shading inputs to the backward pass
                    bary[o.b] = fr.u;
                    bary[o.b + o.bstride] = fr.v;
                    bary[o.b + 2 * o.bstride] = __int_as_float(fr.j | (fr.map << 20));
                }
                const float a = fr.e * fr.fa;
                float c[3] = {0.f, 0.f, 0.f};
                if (a != 0.f) {
                    Sample s;
                    footprint(A, fr, s);
                    fetch(A.maps, s, c);
                    const float wgt = T * a;
                    r += wgt * c[0]; g += wgt * c[1]; b += wgt * c[2];
                }
                if (A.tiled == 2) {       // ... together with the blend opacity, the sampled colour (0 where the opacity is 0) and
                                          // the transmittance in front of the fragment
                    bary[o.b + 7 * o.bstride] = T;
                    bary[o.b + 3 * o.bstride] = a;
                    bary[o.b + 4 * o.bstride] = c[0]; bary[o.b + 5 * o.bstride] = c[1]; bary[o.b + 6 * o.bstride] = c[2];
                }
                T *= (1.f - a);
            }
        }
    }
    const long long plane = (long long)A.H * A.W;
    float *out = image + (long long)n * 4 * plane + (long long)yi * A.W + xi;
    out[0] = r + T * A.bg[0];
    out[plane] = g + T * A.bg[1];
    out[2 * plane] = b + T * A.bg[2];
    out[3 * plane] = 1.f - T;
}

template <int KMAX, int TW, int TH, int GROUP>
int launch_v(const FaceRec *recs, const float4 *bbox, const int *first_idx, const int *num_faces, float blur,
             int persp, ShadeArgs &A, const CoarseBins &cb, int *p2f, float *bary, float *dists, float *image, hipStream_t s) {
    const long long total = (long long)A.N * ((A.W + TW - 1) / TW) * ((A.H + TH - 1) / TH);
    hipLaunchKernelGGL((render_fwd_kernel<KMAX, TW, TH, GROUP>), dim3(dbw_xcd_grid(total)), dim3(TW * TH), 0, s, recs, bbox, first_idx,
                       num_faces, blur, persp, g_render_dbg, total, A, cb, p2f, bary, dists, image);
    return dbw_check_launch("render_fwd_kernel");
}

template <int KMAX>
int launch(const FaceRec *recs, const float4 *bbox, const int *first_idx, const int *num_faces, float blur,
           int persp, ShadeArgs &A, const CoarseBins &cb, int *p2f, float *bary, float *dists, float *image, hipStream_t s) {
#define DBW_V(TW, TH, G) launch_v<KMAX, TW, TH, G>(recs, bbox, first_idx, num_faces, blur, persp, A, cb, p2f, bary, dists, image, s)
    if constexpr (KMAX == 1) return DBW_V(16, 16, 2);   // hard K=1 pass: large faces (sky dome, ground), fewer tiles re-scan the face list; the
                                              // single payload stays in registers
    else {
#ifdef DBW_TUNE_VARIANTS
    switch (g_render_variant) {               // tile-shape / load-batching sweep (tools/sweep_render_fwd.py)
        case 1: return DBW_V(8, 8, 1);
        case 2: return DBW_V(8, 8, 2);
        case 3: return DBW_V(16, 8, 2);
        case 4: return DBW_V(16, 16, 2);
        default: break;
    }
#endif
    return DBW_V(8, 8, 4);                    // soft K-layer passes: one wave64 per 8x8 tile
    }
#undef DBW_V
}

}  // namespace

extern "C" int dbw_render_fwd_fused(const float *face_verts_c, const int32_t *first_idx, const int32_t *num_faces,
                                    const int32_t *neighbor, const int32_t *c2o, const int32_t *clip_code,
                                    const float *clip_w, int Fc_stride, const float *face_uvs, const int32_t *face_map,
                                    const int32_t *map_desc, const float *maps, const float *faces_alpha, int alpha_len,
                                    int N, int64_t F_total, int H, int W, int K, int F, float sigma, float blur_radius,
                                    int perspective_correct, const float *background3, int32_t *pix_to_face, float *bary,
                                    float *dists, float *image, void *workspace, size_t workspace_bytes,
                                    int frag_layout, dbw_stream_t stream) {
    DBW_REQUIRE(face_verts_c && first_idx && num_faces && pix_to_face && bary && dists && image && workspace, "null pointer");
    DBW_REQUIRE(workspace_bytes >= dbw_rasterize_workspace_bytes(F_total), "workspace too small");
    DBW_REQUIRE(blur_radius >= 0.f && F_total >= 0, "bad blur_radius / F_total");
    ShadeArgs A;
    int rc = dbw_fill_shade_args(A, pix_to_face, bary, dists, c2o, clip_code, clip_w, Fc_stride, face_uvs, face_map, map_desc, maps,
                                 faces_alpha, alpha_len, N, H, W, K, F, sigma, background3);
    if (rc) return rc;
    DBW_REQUIRE(frag_layout >= 0 && frag_layout <= 2, "frag_layout must be 0 (N,H,W,K), 1 (8x8-tile planar) or 2 (planar, uv)");
    DBW_REQUIRE(frag_layout != 2 || F < (1 << 20), "frag_layout 2 packs the face id in 20 bits");
    DBW_REQUIRE(frag_layout != 2 || F_total < (1LL << FRAG_COUNT_SHIFT), "frag_layout 2 packs the clipped face id in 26 bits");
    A.tiled = frag_layout;
    if (K > DBW_MAX_FACES_PER_PIXEL) {
        dbw_set_error("dbw_render_fwd_fused: faces_per_pixel=%d > %d", K, DBW_MAX_FACES_PER_PIXEL);
        return DBW_ERR_UNSUPPORTED;
    }
    if (N == 0) return DBW_OK;
    hipStream_t s = (hipStream_t)stream;
    const float margin = (float)sqrt((double)blur_radius);
    CoarseBins cb;
    rc = dbw_prepare_raster(face_verts_c, first_idx, num_faces, neighbor, N, F_total, H, W, margin, 0, workspace, workspace_bytes, cb, s);
    if (rc) return rc;
    const float4 *bbox = (const float4 *)workspace;
    const FaceRec *recs = dbw_workspace_recs(workspace, F_total);
#define DBW_RF(KM) launch<KM>(recs, bbox, first_idx, num_faces, blur_radius, perspective_correct, A, cb, pix_to_face, bary, dists, image, s)
    if (K == 1) return DBW_RF(1);
    if (K <= 4) return DBW_RF(4);
    if (K <= 10) return DBW_RF(10);
    if (K <= 16) return DBW_RF(16);
    return DBW_RF(25);
#undef DBW_RF
}
