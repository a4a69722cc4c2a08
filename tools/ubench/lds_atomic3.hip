// Microbenchmark (profiling aid, not product code): ds_add_f64 into the (33 x 33 texel) x 3 fp64 tile of texbin_reduce_kernel, 12 adds
// per record (4 texels x 3 channels) -- with the texel of a lane (a) random, as records arrive, (b) random but of bank class
// (texel mod 16) == lane mod 16: a double is two of the 32 banks, a texel 24 B, so sixteen consecutive lanes with sixteen different
// classes touch every bank exactly once.  256-thread workgroups, 3 per CU as the kernel runs.
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int ITER = 400;
__global__ __launch_bounds__(256) void k(float *out, int sorted) {
    __shared__ double tile[33 * 33 * 3];
    __shared__ int pad[4096];                   // (the kernel's staging area: 3 workgroups per CU)
    for (int i = threadIdx.x; i < 33 * 33 * 3; i += 256) tile[i] = 0.0;
    if (out == nullptr) pad[threadIdx.x] = 1;
    __syncthreads();
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < ITER; ++it) {
        h = h * 1664525u + 1013904223u;
        int r = 1 + (int)((h >> 8) % 32u), c = (int)((h >> 16) % 32u);
        int t = r * 33 + c;
        if (sorted) {                            // move to the nearest texel of the lane's class (same row where possible)
            const int want = lane & 15, d = (want - (t & 15)) & 15;
            t += d; if (t >= 33 * 33 - 34) t -= 16;
        }
        const int idx[4] = {t * 3, (t + 1) * 3, (t - 33) * 3, (t - 32) * 3};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            atomicAdd(&tile[idx[q]], 1.0); atomicAdd(&tile[idx[q] + 1], 1.0); atomicAdd(&tile[idx[q] + 2], 1.0);
        }
    }
    __syncthreads();
    if (tile[threadIdx.x] == -1.0) out[0] = 1.f;
}
int main() {
    float *out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int sorted = 0; sorted < 2; ++sorted) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256 * 3), dim3(256), 0, 0, out, sorted);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        // per CU: 3 workgroups x 4 waves x ITER x 12 wave-instructions
        printf("ds_add_f64 into the bin tile, texels %s: %6.1f clk per wave-instruction per CU\n", sorted ? "of the lane's bank class" : "random", 
               best * 1e6 * 2.4 / ((double)ITER * 12 * 12));
    }
    return 0;
}
