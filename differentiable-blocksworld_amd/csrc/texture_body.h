// Forward of the texture preparation (sigmoid + decimation to cell resolution, dbw.py:273-278,288-293,306,331-334) as a device function
// of a 256-thread grid-stride launch: shared by texture.hip's kernels and by the training step's prologue (model_ops.hip).
#pragma once
#include "dbw_common.h"

namespace dbw {

__device__ __forceinline__ float tex_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// d == 1: elementwise.  d > 1: one WAVE per (d x d) cell (lane <-> texel, wave-sum for the cell mean); `maps` is written
// at CELL resolution (n, h/d, w/d, 3).
__device__ __forceinline__ void texture_prep_fwd_body(const float *__restrict__ tex, int n, int h, int w, int d,
                                                      float *__restrict__ maps, float *__restrict__ sig) {
    if (d <= 1) {
        const long long total = (long long)n * h * w * 3;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
            const float s = tex_sigmoid(tex[i]);
            maps[i] = s;
            if (sig) sig[i] = s;
        }
        return;
    }
    const int ch_ = h / d, cw_ = w / d, lane = threadIdx.x & 63;
    const long long cells = (long long)n * ch_ * cw_;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long c = wave; c < cells; c += nwaves) {
        const int m = (int)(c / (ch_ * cw_));
        const int r = (int)(c % (ch_ * cw_));
        const int cy = r / cw_, cx = r % cw_;
        float acc[3] = {0.f, 0.f, 0.f};
        for (int t = lane; t < d * d; t += 64) {
            const long long o = (((long long)m * h + cy * d + t / d) * w + cx * d + t % d) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float s = tex_sigmoid(tex[o + k]);
                sig[o + k] = s;
                acc[k] += s;
            }
        }
        const float inv = 1.f / (float)(d * d);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float tot = wave_sum(acc[k]);
            if (lane == 0) maps[c * 3 + k] = tot * inv;
        }
    }
}


}  // namespace dbw
