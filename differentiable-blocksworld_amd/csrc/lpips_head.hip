// The head of the perceptual criterion (lpips 0.1.4 as src/model/loss.py:32-40 calls it): for one feature tap of the
// frozen VGG16, per image n
//     value[n] = mean_pixels sum_c w_c (a_c - f_c / (|f| + 1e-10))^2,   |f| = sqrt(sum_c f_c^2)
// with f (N, C, HW) the tap of the reconstruction (it carries the gradient), a the UNIT-NORMALISED tap of the target image (a constant:
// computed once per training view, dbw_amd/lpips_vgg.py) and w >= 0 the 1x1 head.  In torch this is a dozen element-wise / reduction
// kernels over the largest tensors of the network in each direction (2.0 of the 9.5 ms of a 4-view step at 400x300, next to 6.8 ms of
// MIOpen convolutions); here it is one streaming pass each way.  HBM-bound: a thread owns a pixel and walks its channels (NCHW: the
// lanes of a wave read consecutive pixels of one channel plane), twice -- first the norm, then the terms -- the second walk out of L2.
//   forward : reads 2 f + a,            writes one partial sum per workgroup      (algorithmic bytes: 8 N C HW, f and a once)
//   backward: reads 2 f + 2 a,          writes g_f                                (12 N C HW)
#include "dbw_common.h"
#include "../../include/dbw_hip.h"

using namespace dbw;

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float block_sum(float v, float *s_red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) s_red[wv] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NT / DBW_WAVE; ++w) t += s_red[w];
    return t;
}

// d value[n] / d f_k = (1 / HW) [ r q_k - f_k (q . f) / (|f| (|f| + eps)^2) ],  q_c = -2 w_c (a_c - f_c r),  r = 1 / (|f| + eps);
// q . f = -2 (sum w a f - r sum w f^2) comes out of the first walk together with the norm.  A pixel whose tap is all zero has no
// second term (torch's sqrt backward gives 0 * inf there; the published network never produces it behind its biases).
// VEC = 4: a thread owns four consecutive pixels (16 B loads; taps whose plane is a multiple of four pixels -- the three large ones of a
// 400x300 image), VEC = 1 otherwise.  The walks are unrolled by eight channels: the loads in flight, not the arithmetic, set the pace.
typedef float v4f __attribute__((ext_vector_type(4)));
template <int VEC> struct Px { float v[VEC]; };
template <int VEC>
__device__ __forceinline__ Px<VEC> ldpx(const float *p) {
    Px<VEC> r;
    if (VEC == 4) { const float4 t = *reinterpret_cast<const float4 *>(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[VEC - 1] = t.w; }
    else r.v[0] = *p;
    return r;
}

// CG > 1 (small taps: few pixels, 512 channels): the workgroup's 256 threads are 256 / CG pixels x CG channel groups, a thread walks every
// CG-th channel and the per-pixel sums meet in LDS -- CG times the threads, walks CG times shorter.
template <bool BWD, int VEC, int CG>
__global__ __launch_bounds__(NT) void lpips_head_kernel(const float *__restrict__ f, const float *__restrict__ a, const long long *__restrict__ ids,
                                                        const float *__restrict__ w, int V, int C, int HW, float inv_hw, const float *__restrict__ gout,
                                                        float *__restrict__ partial, float *__restrict__ gf) {
    static_assert(VEC == 1 || CG == 1, "the vector form owns whole pixels");
    constexpr int PX = NT / CG;
    __shared__ float s_red[NT / DBW_WAVE];
    __shared__ float s_part[CG > 1 ? 3 * NT : 1];
    const int n = blockIdx.y, px = threadIdx.x % PX, cg = threadIdx.x / PX;
    const int p = (blockIdx.x * PX + px) * VEC;
    const bool on = p < HW;                              // (VEC == 4: HW is a multiple of four, a thread's pixels are all in or all out)
    const long long plane = HW;
    long long row = ids ? ids[n] : (long long)n;
    const bool bad = row < 0 || row >= V;              // (a view id outside the cache: the value and the gradient come out as NaN, nothing is read out of bounds)
    if (bad) row = 0;
    const float *fp = f + (long long)n * C * plane + (on ? p : 0);
    const float *ap = a + row * C * plane + (on ? p : 0);
    float ss[VEC], waf[VEC], wff[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) ss[i] = waf[i] = wff[i] = 0.f;
    if (on) {
#pragma unroll 8
        for (int c = cg; c < C; c += CG) {
            const Px<VEC> fc = ldpx<VEC>(fp + c * plane);
            if (BWD) {
                const Px<VEC> ac = ldpx<VEC>(ap + c * plane);
                const float wc = w[c];
#pragma unroll
                for (int i = 0; i < VEC; ++i) { waf[i] += wc * ac.v[i] * fc.v[i]; wff[i] += wc * fc.v[i] * fc.v[i]; }
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) ss[i] += fc.v[i] * fc.v[i];
        }
    }
    if (CG > 1) {
        s_part[threadIdx.x] = ss[0]; s_part[NT + threadIdx.x] = waf[0]; s_part[2 * NT + threadIdx.x] = wff[0];
        __syncthreads();
        ss[0] = waf[0] = wff[0] = 0.f;
#pragma unroll
        for (int g = 0; g < CG; ++g) { ss[0] += s_part[g * PX + px]; waf[0] += s_part[NT + g * PX + px]; wff[0] += s_part[2 * NT + g * PX + px]; }
    }
    float s[VEC], r[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { s[i] = sqrtf(ss[i]); r[i] = 1.f / (s[i] + 1e-10f); }
    if (!BWD) {
        float acc = 0.f;
        if (on) {
#pragma unroll 8
            for (int c = cg; c < C; c += CG) {
                const Px<VEC> fc = ldpx<VEC>(fp + c * plane), ac = ldpx<VEC>(ap + c * plane);
                const float wc = w[c];
#pragma unroll
                for (int i = 0; i < VEC; ++i) { const float d = ac.v[i] - fc.v[i] * r[i]; acc += wc * d * d; }
            }
        }
        const float tot = block_sum(acc, s_red);
        if (threadIdx.x == 0) partial[(long long)n * gridDim.x + blockIdx.x] = bad ? __builtin_nanf("") : tot * inv_hw;
    } else if (on) {
        const float scale = bad ? __builtin_nanf("") : gout[n] * inv_hw;
        float k2[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float qf = -2.f * (waf[i] - r[i] * wff[i]);
            k2[i] = s[i] > 0.f ? qf / (s[i] * (s[i] + 1e-10f) * (s[i] + 1e-10f)) : 0.f;
        }
        float *gp = gf + (long long)n * C * plane + p;
#pragma unroll 8
        for (int c = cg; c < C; c += CG) {
            const Px<VEC> fc = ldpx<VEC>(fp + c * plane), ac = ldpx<VEC>(ap + c * plane);
            const float wc = w[c];
            float o[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) o[i] = scale * (r[i] * (-2.f * wc * (ac.v[i] - fc.v[i] * r[i])) - fc.v[i] * k2[i]);
            if (VEC == 4) { const v4f t = {o[0], o[1], o[2], o[VEC - 1]}; __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(gp + c * plane)); }
            else __builtin_nontemporal_store(o[0], gp + c * plane);
        }
    }
}

// workgroups of a launch: one pixel per thread, or 256 / 16 pixels per workgroup for the small taps (the caller sizes `partial` with
// dbw_lpips_head_blocks: the same rule)
constexpr int SMALL_CG = 16;
inline bool small_tap(int N, int HW) { return (long long)N * HW < 65536; }
inline int blocks_of(int N, int HW) { return small_tap(N, HW) ? (HW + NT / SMALL_CG - 1) / (NT / SMALL_CG) : (HW + NT - 1) / NT; }

template <bool BWD>
int launch(const float *feat, const float *target_unit, const int64_t *view_ids, const float *lin_w, int N, int V, int C, int HW,
           const float *grad_value, float *partial, float *grad_feat, dbw_stream_t stream) {
    const bool vec = HW % 4 == 0 && ((uintptr_t)feat % 16 == 0) && ((uintptr_t)target_unit % 16 == 0) && (!BWD || (uintptr_t)grad_feat % 16 == 0);
    const dim3 grid((unsigned)blocks_of(N, HW), (unsigned)N);
#define DBW_LH(VEC, CG) hipLaunchKernelGGL((lpips_head_kernel<BWD, VEC, CG>), grid, dim3(NT), 0, (hipStream_t)stream, feat, target_unit, (const long long *)view_ids, \
                                           lin_w, V, C, HW, 1.f / (float)HW, grad_value, partial, grad_feat)
    if (small_tap(N, HW)) DBW_LH(1, SMALL_CG);
    else if (vec) DBW_LH(4, 1);          // (a quarter of the workgroups have pixels; the others write a zero partial)
    else DBW_LH(1, 1);
#undef DBW_LH
    return dbw_check_launch(BWD ? "lpips_head_kernel<bwd>" : "lpips_head_kernel<fwd>");
}

// ---- the two element-wise layers between the frozen network's convolutions (torch runs each as several kernels) ------------------------
// bias + ReLU behind a convolution that was run WITHOUT its bias: y = max(x + b[c], 0) in one pass (torch: the convolution's bias add, then
// clamp -- two passes over the largest tensors of the network).  x == y allowed.
__global__ __launch_bounds__(NT) void bias_relu_kernel(const float *__restrict__ x, const float *__restrict__ b, int C, int HW, float *__restrict__ y, int vec) {
    const long long plane = (long long)blockIdx.y;                      // n * C + c
    const float bc = b[(int)(plane % C)];
    const long long o = plane * HW;
    if (vec) {
        const int i = (blockIdx.x * NT + threadIdx.x) * 4;
        if (i >= HW) return;
        const v4f t = *reinterpret_cast<const v4f *>(x + o + i);
        const v4f r = {fmaxf(t.x + bc, 0.f), fmaxf(t.y + bc, 0.f), fmaxf(t.z + bc, 0.f), fmaxf(t.w + bc, 0.f)};
        *reinterpret_cast<v4f *>(y + o + i) = r;
    } else {
        const int i = blockIdx.x * NT + threadIdx.x;
        if (i < HW) y[o + i] = fmaxf(x[o + i] + bc, 0.f);
    }
}

// 2x2 / stride 2 max pooling (floor mode: an odd last row / column is dropped), and its backward WITHOUT stored indices: the window's
// maximum is found again from the input -- the FIRST one in row-major window order, `>` comparisons, as torch's forward picks it (behind a
// ReLU most windows are ties of zeros) -- and a thread per INPUT pixel writes its own gradient (zeros included: gx is fully written).
__global__ __launch_bounds__(NT) void maxpool2_fwd_kernel(const float *__restrict__ x, int H, int W, int Ho, int Wo, float *__restrict__ y) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= Ho * Wo) return;
    const int ho = i / Wo, wo = i - ho * Wo;
    const float *p = x + (long long)blockIdx.y * H * W + (long long)(2 * ho) * W + 2 * wo;
    float m = p[0];
    if (p[1] > m || p[1] != p[1]) m = p[1];
    if (p[W] > m || p[W] != p[W]) m = p[W];
    if (p[W + 1] > m || p[W + 1] != p[W + 1]) m = p[W + 1];
    y[(long long)blockIdx.y * Ho * Wo + i] = m;
}
__global__ __launch_bounds__(NT) void maxpool2_bwd_kernel(const float *__restrict__ x, const float *__restrict__ gy, int H, int W, int Ho, int Wo, float *__restrict__ gx) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= H * W) return;
    const int h = i / W, w = i - h * W, ho = h >> 1, wo = w >> 1;
    float g = 0.f;
    if (ho < Ho && wo < Wo) {
        const float *p = x + (long long)blockIdx.y * H * W + (long long)(2 * ho) * W + 2 * wo;
        float m = p[0];
        int arg = 0;
        if (p[1] > m || p[1] != p[1]) { m = p[1]; arg = 1; }
        if (p[W] > m || p[W] != p[W]) { m = p[W]; arg = 2; }
        if (p[W + 1] > m || p[W + 1] != p[W + 1]) { m = p[W + 1]; arg = 3; }
        if (arg == ((h & 1) << 1 | (w & 1))) g = gy[(long long)blockIdx.y * Ho * Wo + (long long)ho * Wo + wo];
    }
    gx[(long long)blockIdx.y * H * W + i] = g;
}

}  // namespace

extern "C" int dbw_bias_relu(const float *x, const float *bias, int N, int C, int HW, float *y, dbw_stream_t stream) {
    DBW_REQUIRE(x && bias && y, "null pointer");
    DBW_REQUIRE(N >= 0 && C > 0 && HW > 0 && (long long)N * C < 65536, "bad sizes");
    if (N == 0) return 0;
    const int vec = HW % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0;
    const int per = vec ? NT * 4 : NT;
    hipLaunchKernelGGL(bias_relu_kernel, dim3((unsigned)((HW + per - 1) / per), (unsigned)(N * C)), dim3(NT), 0, (hipStream_t)stream, x, bias, C, HW, y, vec);
    return dbw_check_launch("bias_relu_kernel");
}

extern "C" int dbw_maxpool2_fwd(const float *x, int planes, int H, int W, float *y, dbw_stream_t stream) {
    DBW_REQUIRE(x && y, "null pointer");
    DBW_REQUIRE(planes >= 0 && planes < 65536 && H >= 2 && W >= 2, "bad sizes");
    if (planes == 0) return 0;
    const int Ho = H / 2, Wo = W / 2;
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3((unsigned)((Ho * Wo + NT - 1) / NT), (unsigned)planes), dim3(NT), 0, (hipStream_t)stream, x, H, W, Ho, Wo, y);
    return dbw_check_launch("maxpool2_fwd_kernel");
}

extern "C" int dbw_maxpool2_bwd(const float *x, const float *grad_y, int planes, int H, int W, float *grad_x, dbw_stream_t stream) {
    DBW_REQUIRE(x && grad_y && grad_x, "null pointer");
    DBW_REQUIRE(planes >= 0 && planes < 65536 && H >= 2 && W >= 2, "bad sizes");
    if (planes == 0) return 0;
    const int Ho = H / 2, Wo = W / 2;
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3((unsigned)((H * W + NT - 1) / NT), (unsigned)planes), dim3(NT), 0, (hipStream_t)stream, x, grad_y, H, W, Ho, Wo, grad_x);
    return dbw_check_launch("maxpool2_bwd_kernel");
}

extern "C" int dbw_lpips_head_blocks(int N, int HW) { return blocks_of(N, HW); }

extern "C" int dbw_lpips_head_fwd(const float *feat, const float *target_unit, const int64_t *view_ids, const float *lin_w, int N, int V, int C, int HW,
                                  float *partial, dbw_stream_t stream) {
    DBW_REQUIRE(feat && target_unit && lin_w && partial, "null pointer");
    DBW_REQUIRE(N >= 0 && V >= (view_ids ? 1 : N) && C > 0 && HW > 0 && N < 65536, "bad sizes");
    if (N == 0) return 0;
    return launch<false>(feat, target_unit, view_ids, lin_w, N, V, C, HW, nullptr, partial, nullptr, stream);
}

extern "C" int dbw_lpips_head_bwd(const float *feat, const float *target_unit, const int64_t *view_ids, const float *lin_w, int N, int V, int C, int HW,
                                  const float *grad_value, float *grad_feat, dbw_stream_t stream) {
    DBW_REQUIRE(feat && target_unit && lin_w && grad_value && grad_feat, "null pointer");
    DBW_REQUIRE(N >= 0 && V >= (view_ids ? 1 : N) && C > 0 && HW > 0 && N < 65536, "bad sizes");
    if (N == 0) return 0;
    return launch<true>(feat, target_unit, view_ids, lin_w, N, V, C, HW, grad_value, nullptr, grad_feat, stream);
}
