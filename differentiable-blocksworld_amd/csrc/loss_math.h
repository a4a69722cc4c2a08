// Loss / optimiser arithmetic shared by the device kernels (hipcc, texture.hip) and the host (g++: tests/test_host_loss_math.py builds it
// into a checker-side shared object and compares it, without a GPU, with the golden vectors of the reference's TV term, with autograd of
// the decoupled composite + MSE and with torch.optim.Adam).
#pragma once
#include "raster_math.h"      // DBW_HD

namespace dbw {

// TV regulariser (dbw.py:378-387, loss.py:46: l2sq of the forward differences, mean over the differences of each direction) for ONE
// texel x of row y (r = the row's first float; rows are w * 3 floats apart): returns the texel's share of the value -- the forward
// differences it owns, x -> x + 1 (wrapping to column 0 when `wrap`) and y -> y + 1 -- and g[3] = d value / d texel (both differences
// the texel takes part in, in each direction).  sx, sy = 1 / number of differences per channel in x and in y.
DBW_HD float tv_l2sq_texel(const float *r, int x, int y, int w, int h, int wrap, float sx, float sy, float g[3]) {
    const int xr = x + 1 < w ? x + 1 : (wrap ? 0 : -1), xl = x > 0 ? x - 1 : (wrap ? w - 1 : -1);
    float part = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = r[x * 3 + c];
        float gc = 0.f;
        if (xr >= 0) { const float dxf = r[xr * 3 + c] - v; part += dxf * dxf * sx; gc -= 2.f * dxf * sx; }
        if (xl >= 0) { const float dxb = v - r[xl * 3 + c]; gc += 2.f * dxb * sx; }
        if (y + 1 < h) { const float dyf = r[(x + w) * 3 + c] - v; part += dyf * dyf * sy; gc -= 2.f * dyf * sy; }
        if (y > 0) { const float dyb = v - r[(x - w) * 3 + c]; gc += 2.f * dyb * sy; }
        g[c] = gc;
    }
    return part;
}

// decoupled composite + squared error of one pixel (dbw.py:223,366-367): rec = fg_rgb * mask + (1 - mask) * env_rgb (the fg colour is
// premultiplied AND multiplied by the mask again).  Returns the pixel's sum of squared differences to `target`;
// with two_scale = 2 * (weight / count): gf = d loss / d fg_rgb, ge = d loss / d env_rgb, gmask = d loss / d mask.
// extra (optional): d(another loss term) / d rec of this pixel -- the perceptual term, which a network outside this path differentiates --
// added to the MSE's before the chain rule through the composite (has_extra: a flag next to the three values, not a nullable pointer --
// a run-time-selected pointer to a local array keeps that array in scratch memory).
DBW_HD float composite_mse_pixel(const float fc[3], float mask, const float ec[3], const float target[3], bool has_target, float two_scale,
                                 float rec[3], float gf[3], float ge[3], float &gmask, const float extra[3], bool has_extra) {
    float part = 0.f;
    gmask = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        rec[c] = fc[c] * mask + (1.f - mask) * ec[c];
        const float d = has_target ? rec[c] - target[c] : 0.f;
        part += d * d;
        const float gr = has_extra ? two_scale * d + extra[c] : two_scale * d;
        gf[c] = gr * mask;
        ge[c] = gr * (1.f - mask);
        gmask += gr * (fc[c] - ec[c]);
    }
    return part;
}

DBW_HD float composite_mse_pixel(const float fc[3], float mask, const float ec[3], const float target[3], bool has_target, float two_scale,
                                 float rec[3], float gf[3], float ge[3], float &gmask) {
    const float none[3] = {0.f, 0.f, 0.f};
    return composite_mse_pixel(fc, mask, ec, target, has_target, two_scale, rec, gf, ge, gmask, none, false);
}

// torch.optim.Adam (no weight decay, no amsgrad; optimizer.py:6-18) for one element: step_size = lr / (1 - beta1^t),
// bc2_sqrt = sqrt(1 - beta2^t)
DBW_HD void adam_update(float &p, float g, float &m, float &v, float step_size, float beta1, float beta2, float eps, float bc2_sqrt) {
    const float mi = beta1 * m + (1.f - beta1) * g;
    const float vi = beta2 * v + (1.f - beta2) * g * g;
    m = mi; v = vi;
    p -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
}

}  // namespace dbw
