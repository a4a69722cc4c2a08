// Checker-side build of the product's shading arithmetic (differentiable-blocksworld_amd/csrc/shade_math.h, the header the HIP kernels
// compile) for the host: tests/test_host_shade_math.py compares it, without a GPU, with torch's grid_sample and with golden vectors of
// the reference's own layered_rgb_blend (tests/golden/blend.npz).  Test infrastructure, not product code.
#include <vector>
#include "../differentiable-blocksworld_amd/csrc/shade_math.h"

using namespace dbw;

extern "C" {

// bilinear sampling of n points: rgb (n,3); guv (n,2) = d(sum_ch gc * rgb)/d(u, v); gmaps += the texel gradient (same size as maps)
int host_sample(const float *maps, const int *desc, int n, const float *uv, const float *gc, float *rgb, float *guv, float *gmaps) {
    for (int i = 0; i < n; ++i) {
        Sample s;
        footprint_desc(uv[2 * i], uv[2 * i + 1], desc[0], desc[1], desc[2], desc[3], desc[4], desc[5], s);
        fetch(maps, s, rgb + 3 * i);
        sample_grad_uv(maps, s, gc + 3 * i, guv[2 * i], guv[2 * i + 1]);
        for (int ch = 0; ch < 3; ++ch) {
            gmaps[s.a00 + ch] += gc[3 * i + ch] * s.w00;
            gmaps[s.a01 + ch] += gc[3 * i + ch] * s.w01;
            gmaps[s.a10 + ch] += gc[3 * i + ch] * s.w10;
            gmaps[s.a11 + ch] += gc[3 * i + ch] * s.w11;
        }
    }
    return 0;
}

// layered blend of (N,H,W,K) fragments, forward + backward for loss = sum(out * w): the way the kernels do it -- forward front to
// back keeping T_k, backward back to front with the division-free recurrences
int host_blend(const long long *p2f, const float *dists, const float *colors, const float *faces_alpha, int N, int H, int W, int K,
               float sigma, const float *bg, const float *w, float *out, float *g_colors, float *g_dists, float *g_fa) {
    const long long plane = (long long)H * W;
    std::vector<float> T(K), a(K), e(K);
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const long long pix = ((long long)n * H + y) * W + x;
                BlendFront f;
                blend_front_init(f);
                for (int k = 0; k < K; ++k) {
                    const long long o = pix * K + k;
                    const bool valid = p2f[o] >= 0;
                    e[k] = valid ? geometric_alpha(dists[o], sigma) : 0.f;
                    a[k] = e[k] * ((valid && faces_alpha) ? faces_alpha[p2f[o]] : 1.f);
                    T[k] = f.T;
                    blend_front_step(f, a[k], colors + o * 3);
                }
                float px[4];
                blend_front_finish(f, bg, px);
                float g[4];
                for (int c = 0; c < 4; ++c) {
                    out[((long long)n * 4 + c) * plane + (long long)y * W + x] = px[c];
                    g[c] = w[((long long)n * 4 + c) * plane + (long long)y * W + x];
                }
                BlendBack b;
                blend_back_init(b, bg);
                for (int k = K - 1; k >= 0; --k) {
                    const long long o = pix * K + k;
                    const float *c = colors + o * 3;
                    const float ga = blend_back_step(b, T[k], a[k], c[0], c[1], c[2], g[0], g[1], g[2], g[3]);
                    const bool valid = p2f[o] >= 0;
                    const float wgt = T[k] * a[k];
                    for (int ch = 0; ch < 3; ++ch) g_colors[o * 3 + ch] = wgt * g[ch];
                    // (sigma < 0: the sigmoid opacity of clip_inside = False, d e / d d = -e (1 - e) / |sigma| at every d -- as the kernels have it)
                    g_dists[o] = !valid ? 0.f : (sigma > 0.f ? (dists[o] >= 0.f ? ga * a[k] * (-1.f / sigma) : 0.f) : (sigma < 0.f ? ga * a[k] * (1.f - e[k]) * (1.f / sigma) : 0.f));
                    if (valid && faces_alpha) g_fa[p2f[o]] += ga * e[k];
                }
            }
    return 0;
}

// barycentric back-conversion of a clipped face and its backward
int host_convert_bary(int cd, float w2, float w3, const float *b, float *bo, const float *go, float *gb) {
    convert_bary(cd, w2, w3, b, bo);
    convert_bary_bwd(cd, w2, w3, go, gb);
    return 0;
}

}  // extern "C"
