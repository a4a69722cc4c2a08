// Microbenchmark (profiling aid, not product code): cost of LDS atomics on gfx950 by type, active-lane count and address
// pattern.  One block of `waves` waves per CU (256 blocks); every wave issues ITER x 12 LDS operations.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int ITER = 2000, SLOTS = 4096;

// MODE 5 ds_add_f64 | 6 ds_add_u64 | 7 CAS loop f32
// MODE 0 ds_add_f32 | 1 ds_add_u32 | 2 ds_add_rtn_u32 (value used) | 3 read+add+write (non-atomic) | 4 ds_add_rtn_f32
// PAT  0 distinct address per lane | 1 all lanes one address | 2 ~10 lanes per address | ACTIVE lanes per wave
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int pat, int active) {
    __shared__ __attribute__((aligned(8))) float s[SLOTS];
    for (int i = threadIdx.x; i < SLOTS; i += blockDim.x) s[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int a = pat == 0 ? (threadIdx.x * 13) % (SLOTS - 16) : pat == 1 ? (threadIdx.x >> 6) * 16 : ((threadIdx.x >> 6) * 64 + lane / 10) * 13 % (SLOTS - 16);
    float acc = 0.f;
    unsigned *u = (unsigned *)s;
    if (lane < active) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                if (MODE == 0) atomicAdd(&s[a + q], 1.0f);
                else if (MODE == 1) atomicAdd(&u[a + q], 1u);
                else if (MODE == 2) acc += (float)atomicAdd(&u[a + q], 1u);
                else if (MODE == 3) s[a + q] += 1.0f;
                else if (MODE == 4) acc += atomicAdd(&s[a + q], 1.0f);
                else if (MODE == 5) atomicAdd((double *)&s[(a + q) * 2 % (SLOTS - 2) & ~1], 1.0);
                else if (MODE == 6) atomicAdd((unsigned long long *)&s[(a + q) * 2 % (SLOTS - 2) & ~1], 1ull);
                else {
                    unsigned *p = &u[a + q];
                    unsigned old = *p, assumed;
                    do { assumed = old; old = atomicCAS(p, assumed, __float_as_uint(__uint_as_float(assumed) + 1.0f)); } while (old != assumed);
                }
            }
            a = (a + 12 * 7) % (SLOTS - 16);
        }
    }
    __syncthreads();
    if (acc == 12345.f || s[threadIdx.x] == -1.f) out[0] = acc;
}

int main() {
    float *out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *mn[] = {"ds_add_f32", "ds_add_u32", "ds_add_rtn_u32", "read+add+write", "ds_add_rtn_f32", "ds_add_f64", "ds_add_u64", "cas-loop f32"};
    const char *pn[] = {"distinct", "same-addr", "10/addr"};
    for (int waves : {4})
        for (int mode = 0; mode < 1; ++mode)
            for (int pat = 0; pat < 3; ++pat)
                for (int active : {64, 16, 1}) {
                    if (pat && active != 64) continue;
                    float best = 1e9;
                    for (int rep = 0; rep < 3; ++rep) {
                        hipEventRecord(e0);
                        switch (mode) {
                            case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, out, pat, active); break;
                            case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, out, pat, active); break;
                            case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), 0, 0, out, pat, active); break;
                            case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(64 * waves), 0, 0, out, pat, active); break;
                            case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(64 * waves), 0, 0, out, pat, active); break;
                            case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(64 * waves), 0, 0, out, pat, active); break;
                            case 6: hipLaunchKernelGGL(k<6>, dim3(256), dim3(64 * waves), 0, 0, out, pat, active); break;
                            case 7: hipLaunchKernelGGL(k<7>, dim3(256), dim3(64 * waves), 0, 0, out, pat, active); break;
                        }
                        hipEventRecord(e1); hipEventSynchronize(e1);
                        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                    }
                    const double instr = (double)ITER * 12 * waves;       // wave-instructions per CU
                    printf("waves/CU %2d %-15s %-9s active %2d : %7.3f ms  %7.1f ns per wave-instruction per CU (%.1f clk @2.4GHz)\n", waves,
                           mn[mode], pn[pat], active, best, best * 1e6 / instr, best * 1e6 / instr * 2.4);
                }
    return 0;
}
