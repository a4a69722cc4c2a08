// Checker-side build of the product's loss / optimiser arithmetic (differentiable-blocksworld_amd/csrc/loss_math.h, the header texture.hip
// compiles) for the host; tests/test_host_loss_math.py compares it with the reference's golden TV vectors, with autograd and with
// torch.optim.Adam, without a GPU.  Test infrastructure only.
#include "../differentiable-blocksworld_amd/csrc/loss_math.h"

using namespace dbw;

extern "C" {

// TV of n maps (n,h,w,3): value (double accumulation of the per-texel shares) and gradient
double host_tv(const float *maps, int n, int h, int w, int wrap, float *grad) {
    const float sx = 1.f / ((float)h * (float)(w - 1 + (wrap ? 1 : 0))), sy = 1.f / ((float)(h - 1) * (float)w);
    double tot = 0.0;
    for (int row = 0; row < n * h; ++row) {
        const float *r = maps + (long long)row * w * 3;
        for (int x = 0; x < w; ++x) tot += tv_l2sq_texel(r, x, row % h, w, h, wrap, sx, sy, grad + ((long long)row * w + x) * 3);
    }
    return tot;
}

// decoupled composite + MSE over P pixels of planar images: fg (4,P), env (4,P: rgb used), target (3,P) -> rec (3,P), g_fg (4,P), g_env (3,P)
double host_composite_mse(const float *fg, const float *env, const float *target, int P, float scale, float *rec, float *g_fg, float *g_env) {
    double tot = 0.0;
    for (int p = 0; p < P; ++p) {
        const float fc[3] = {fg[p], fg[P + p], fg[2 * P + p]}, ec[3] = {env[p], env[P + p], env[2 * P + p]};
        const float t[3] = {target[p], target[P + p], target[2 * P + p]};
        float r[3], gf[3], ge[3], gm;
        tot += composite_mse_pixel(fc, fg[3 * P + p], ec, t, true, 2.f * scale, r, gf, ge, gm);
        for (int c = 0; c < 3; ++c) { rec[c * P + p] = r[c]; g_fg[c * P + p] = gf[c]; g_env[c * P + p] = ge[c]; }
        g_fg[3 * P + p] = gm;
    }
    return tot * scale;
}

int host_adam(float *p, const float *g, float *m, float *v, int n, float lr, float beta1, float beta2, float eps, int step) {
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    for (int i = 0; i < n; ++i) adam_update(p[i], g[i], m[i], v[i], (float)(lr / bc1), beta1, beta2, eps, (float)sqrt(bc2));
    return 0;
}

}  // extern "C"
