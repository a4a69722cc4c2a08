// Microbenchmark (profiling aid, not product code): throughput of scattered fp32 atomic adds by memory scope, and a
// correctness check of the "one private accumulation buffer per XCD, selected by HW_REG_XCC_ID" scheme.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int MODE>
__global__ void k(float* buf, long long n_elems, const unsigned* idx, long long n_ops, long long stride_copy) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_ops) return;
    const unsigned a = idx[i];
    float* p = buf + a;
    if (MODE == 0) unsafeAtomicAdd(p, 1.0f);
    else if (MODE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 2) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 3) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    else if (MODE == 4) { p = buf + (long long)xcc_id() * stride_copy + a; __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    else if (MODE == 5) { p = buf + (long long)xcc_id() * stride_copy + a; __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

__global__ void census(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

int main(int argc, char** argv) {
    const long long n_elems = argc > 1 ? atoll(argv[1]) : 2400000, n_ops = 64LL << 20;
    const int local = argc > 2 ? atoi(argv[2]) : 0;   // >0: lanes of a wave hit a window of `local` floats   // 9.6 MB buffer (like the texture grads), 64M atomics
    float* buf; unsigned* idx;
    hipMalloc(&buf, n_elems * 8 * sizeof(float));
    hipMalloc(&idx, n_ops * sizeof(unsigned));
    std::vector<unsigned> h(n_ops);
    unsigned s = 12345;
    unsigned base = 0;
    for (long long i = 0; i < n_ops; ++i) { s = s * 1664525u + 1013904223u; if (local && (i % 64) == 0) base = (s >> 8) % (n_elems - local); s = s * 1664525u + 1013904223u; h[i] = local ? base + (s >> 8) % local : (s >> 8) % n_elems; }
    hipMemcpy(idx, h.data(), n_ops * sizeof(unsigned), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"unsafeAtomicAdd(agent)", "fetch_add agent", "fetch_add workgroup", "fetch_add wavefront", "per-XCC copy + workgroup", "per-XCC copy + agent"};
    std::vector<float> hb(n_elems * 8);
    printf("n_elems=%lld (%.1f MB) local=%d\n", n_elems, n_elems * 4 / 1e6, local);
    for (int mode = 0; mode < 6; mode += 4) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(buf, 0, n_elems * 8 * sizeof(float));
            hipEventRecord(e0);
            dim3 g((unsigned)((n_ops + 255) / 256));
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, g, dim3(256), 0, 0, buf, n_elems, idx, n_ops, n_elems); break;
                case 1: hipLaunchKernelGGL(k<1>, g, dim3(256), 0, 0, buf, n_elems, idx, n_ops, n_elems); break;
                case 2: hipLaunchKernelGGL(k<2>, g, dim3(256), 0, 0, buf, n_elems, idx, n_ops, n_elems); break;
                case 3: hipLaunchKernelGGL(k<3>, g, dim3(256), 0, 0, buf, n_elems, idx, n_ops, n_elems); break;
                case 4: hipLaunchKernelGGL(k<4>, g, dim3(256), 0, 0, buf, n_elems, idx, n_ops, n_elems); break;
                case 5: hipLaunchKernelGGL(k<5>, g, dim3(256), 0, 0, buf, n_elems, idx, n_ops, n_elems); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        hipMemcpy(hb.data(), buf, n_elems * 8 * sizeof(float), hipMemcpyDeviceToHost);
        double tot = 0; for (float v : hb) tot += v;
        printf("%-28s %8.3f ms  %7.2f G atomics/s   sum=%.0f (expect %lld)%s\n", names[mode], best, n_ops / best / 1e6, tot, n_ops,
               tot == (double)n_ops ? "" : "   <-- LOST UPDATES");
    }
    unsigned* c; hipMalloc(&c, 64 * 4); hipLaunchKernelGGL(census, dim3(64), dim3(64), 0, 0, c);
    unsigned hc[64]; hipMemcpy(hc, c, 256, hipMemcpyDeviceToHost);
    printf("xcc of blocks 0..15:"); for (int i = 0; i < 16; ++i) printf(" %u", hc[i]); printf("\n");
    return 0;
}
