// Microbenchmark (profiling aid, not product code): ds_add_f64 / ds_add_f32 / ds_add_u32 cost vs. lanes-per-address and
// active-lane count (16 waves per CU, 256 blocks).
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int ITER = 1000, SLOTS = 4096;
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int per_addr, int active) {
    __shared__ __attribute__((aligned(8))) float s[SLOTS * 2];
    for (int i = threadIdx.x; i < SLOTS * 2; i += blockDim.x) s[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int a = ((threadIdx.x >> 6) * 64 + lane / per_addr) * 13 % (SLOTS - 16);
    if (lane < active) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                if (MODE == 0) atomicAdd(&s[a + q], 1.0f);
                else if (MODE == 1) atomicAdd((unsigned *)&s[a + q], 1u);
                else atomicAdd((double *)&s[(a + q) * 2], 1.0);
            }
            a = (a + 12 * 7) % (SLOTS - 16);
        }
    }
    __syncthreads();
    if (s[threadIdx.x] == -1.f) out[0] = 1.f;
}
int main() {
    float *out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char *mn[] = {"ds_add_f32", "ds_add_u32", "ds_add_f64"};
    const int waves = 16;
    for (int mode = 0; mode < 3; ++mode)
        for (int active : {64, 32, 16, 8, 4})
            for (int per : {1, 2, 4, 8, 16, 64}) {
                if (per > active) continue;
                float best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    (void)hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, out, per, active);
                    else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, out, per, active);
                    else hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), 0, 0, out, per, active);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                printf("%-11s active %2d lanes/addr %2d : %6.1f clk per wave-instruction per CU\n", mn[mode], active, per,
                       best * 1e6 / ((double)ITER * 12 * waves) * 2.4);
            }
    return 0;
}
