// Shader-clock probe (measurement aid, not part of the library): one wave spins a dependent chain and reports shader cycles per 100 MHz
// wall tick -> the clock it ran at.  Built into a tiny shared object that tools/diag scripts call through ctypes on a given stream:
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench/clock_probe.hip -o tools/ubench/libclock_probe.so
#include <hip/hip_runtime.h>
__global__ void clock_probe_kernel(float *out, int iters) {
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    float x = (float)threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0000001f + 1e-7f;
    const unsigned long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[0] = (float)(c1 - c0) / (float)(w1 - w0) * 100.f; out[1] = (float)(w1 - w0) * 0.01f; out[2] = x; }
}
extern "C" int clock_probe(float *out3, int iters, void *stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out3, iters);
    return (int)hipGetLastError();
}
