// Forward of the texture preparation (sigmoid + decimation to cell resolution, dbw.py:273-278,288-293,306,331-334) as a device function
// of a 256-thread grid-stride launch: shared by texture.hip's kernels and by the training step's prologue (model_ops.hip).
#pragma once
#include "dbw_common.h"

namespace dbw {

__device__ __forceinline__ float tex_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// d == 1: elementwise.  d > 1: one WAVE per (d x d) cell (lane <-> texel, wave-sum for the cell mean); `maps` is written
// at CELL resolution (n, h/d, w/d, 3).
// (blocks bid of nblk: a kernel that runs the preparation next to something else gives it a slice of its own grid)
__device__ __forceinline__ void texture_prep_fwd_body(const float *__restrict__ tex, int n, int h, int w, int d,
                                                      float *__restrict__ maps, float *__restrict__ sig, long long bid, long long nblk) {
    if (d <= 1) {
        const long long total = (long long)n * h * w * 3;
        for (long long i = bid * blockDim.x + threadIdx.x; i < total; i += nblk * blockDim.x) {
            const float s = tex_sigmoid(tex[i]);
            maps[i] = s;
            if (sig) sig[i] = s;
        }
        return;
    }
    const int ch_ = h / d, cw_ = w / d, lane = threadIdx.x & 63;
    const long long cells = (long long)n * ch_ * cw_;
    const long long wave = (bid * blockDim.x + threadIdx.x) >> 6, nwaves = (nblk * blockDim.x) >> 6;
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

// Backward of the preparation, elementwise: gtex = (gcell[cell of the texel] / d^2 + gsig) * s (1 - s).  A grid-stride loop over blocks
// bid of nblk (a kernel that runs it next to something else gives it a slice of its own grid)
__device__ __forceinline__ void texture_prep_bwd_body(const float *__restrict__ tex, int n, int h, int w, int d,
                                                      const float *__restrict__ gmaps, const float *__restrict__ gsig,
                                                      float *__restrict__ gtex, long long bid, long long nblk) {
    const long long total = (long long)n * h * w * 3;
    const float inv = 1.f / (float)(d * d);
    const int ch_ = h / d, cw_ = w / d;
    for (long long i = bid * blockDim.x + threadIdx.x; i < total; i += nblk * blockDim.x) {
        const float s = tex_sigmoid(tex[i]);
        float g;
        if (d <= 1) g = gmaps[i];
        else {
            const int k = (int)(i % 3);
            const long long t = i / 3;
            const int x = (int)(t % w), y = (int)((t / w) % h), m = (int)(t / ((long long)w * h));
            g = gmaps[(((long long)m * ch_ + y / d) * cw_ + x / d) * 3 + k] * inv;
        }
        if (gsig) g += gsig[i];
        gtex[i] = g * s * (1.f - s);
    }
}

}  // namespace dbw
