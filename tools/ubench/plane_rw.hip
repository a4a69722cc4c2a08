// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access pattern of the render kernels: planes read and written 4 bytes per
// lane (one coalesced 256 B line per wave and instruction), far past the 256 MiB Infinity Cache.  The microarchitecture guide calibrates
// FETCH_SIZE (x2 on gfx950) for 16 B/lane streaming reads only and says other widths are uncalibrated: run this under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/ubench/plane_rw     and     ... --pmc WRITE_SIZE -- ...
// and divide the known byte counts (printed) by what the counters report for plane_read_kernel / plane_write_kernel.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/plane_rw tools/ubench/plane_rw.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void plane_read_kernel(const float *__restrict__ in, long long n, float *__restrict__ out) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) acc += __builtin_nontemporal_load(in + i);
    if (acc == 123.456f) out[0] = acc;          // (keeps the loads alive, never true)
}

__global__ __launch_bounds__(256) void plane_write_kernel(float *__restrict__ out, long long n, float v) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) __builtin_nontemporal_store(v, out + i);
}

int main() {
    const long long n = 512LL << 20;            // 2 GiB of floats
    float *buf, *out;
    if (hipMalloc(&buf, n * 4) != hipSuccess || hipMalloc(&out, 256) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(buf, 0, n * 4);
    hipDeviceSynchronize();
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(plane_read_kernel, dim3(8192), dim3(256), 0, 0, buf, n, out);
        hipLaunchKernelGGL(plane_write_kernel, dim3(8192), dim3(256), 0, 0, buf, n / 2, 1.f);
    }
    hipDeviceSynchronize();
    printf("plane_read_kernel reads %lld bytes per launch, plane_write_kernel writes %lld bytes per launch\n", n * 4, n * 2);
    return 0;
}
