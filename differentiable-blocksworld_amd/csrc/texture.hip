// Texture-side kernels: sigmoid + decimation (dbw.py:273-278,288-293,306,331-334), TV regulariser with gradient
// (dbw.py:378-387, loss.py:46) and the fused decoupled composite + MSE forward/backward (dbw.py:223,366-367).
// All are streaming, HBM-bound passes over O(10 MB) per optimisation step (not per view).
#include "dbw_common.h"
#include "loss_math.h"
#include "texture_body.h"
#include "../../include/dbw_hip.h"

#include <string.h>

using namespace dbw;

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ float block_sum(float v, float *s_red) {  // NT threads, returns the total in every thread
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) s_red[wv] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NT / DBW_WAVE; ++w) t += s_red[w];
    return t;
}

__global__ __launch_bounds__(256) void texture_prep_fwd_kernel(const float *__restrict__ tex, int n, int h, int w, int d,
                                                               float *__restrict__ maps, float *__restrict__ sig) {
    texture_prep_fwd_body(tex, n, h, w, d, maps, sig, blockIdx.x, gridDim.x);
}

// several texture sets in ONE launch (blockIdx.y = set): the three maps of a scene (blocks, sky, ground) differ in shape, and one
// 5 us launch each is all they cost
constexpr int MAX_SETS = 4;
struct TextureSets { dbw_texture_set s[MAX_SETS]; };

__global__ __launch_bounds__(256) void texture_prep_fwd_sets_kernel(const TextureSets S) {
    const dbw_texture_set &t = S.s[blockIdx.y];
    texture_prep_fwd_body(t.texture, t.n, t.h, t.w, t.decim, t.maps, t.sig, blockIdx.x, gridDim.x);
}

__global__ void texture_prep_bwd_kernel(const float *__restrict__ tex, int n, int h, int w, int d,
                                        const float *__restrict__ gmaps, const float *__restrict__ gsig,
                                        float *__restrict__ gtex) {
    texture_prep_bwd_body(tex, n, h, w, d, gmaps, gsig, gtex, blockIdx.x, gridDim.x);
}

__global__ void texture_prep_bwd_sets_kernel(const TextureSets S) {
    const dbw_texture_set &t = S.s[blockIdx.y];
    texture_prep_bwd_body(t.texture, t.n, t.h, t.w, t.decim, t.grad_maps, t.grad_sig, t.grad_texture, blockIdx.x, gridDim.x);
}

// one thread per texel (its three channels); blockIdx.y = row (map * h + y): no integer division anywhere
__device__ __forceinline__ void tv_l2sq_body(const float *__restrict__ m, int n, int h, int w, int wrap,
                                             float scale, float *__restrict__ loss, float *__restrict__ gm, float *s_red) {
    const float sx = 1.f / ((float)h * (float)(w - 1 + (wrap ? 1 : 0)));
    const float sy = 1.f / ((float)(h - 1) * (float)w);
    float part = 0.f;
    for (int row = blockIdx.y; row < n * h; row += gridDim.y) {
        const int y = row % h;
        const float *r = m + (long long)row * w * 3;
        for (int x = blockIdx.x * NT + threadIdx.x; x < w; x += gridDim.x * NT) {
            float g3[3];
            part += tv_l2sq_texel(r, x, y, w, h, wrap, sx, sy, g3);
            if (gm) {
#pragma unroll
                for (int c = 0; c < 3; ++c) gm[((long long)row * w + x) * 3 + c] = scale * g3[c];
            }
        }
    }
    const float tot = block_sum(part, s_red);
    if (threadIdx.x == 0 && tot != 0.f) unsafeAtomicAdd(loss, scale * tot);
}

__global__ __launch_bounds__(NT) void tv_l2sq_kernel(const float *__restrict__ m, int n, int h, int w, int wrap,
                                                     float scale, float *__restrict__ loss, float *__restrict__ gm) {
    __shared__ float s_red[NT / DBW_WAVE];
    tv_l2sq_body(m, n, h, w, wrap, scale, loss, gm, s_red);
}

__global__ __launch_bounds__(NT) void tv_l2sq_sets_kernel(const TextureSets S, float *__restrict__ loss) {
    __shared__ float s_red[NT / DBW_WAVE];
    const dbw_texture_set &t = S.s[blockIdx.z];
    tv_l2sq_body(t.sig, t.n, t.h, t.w, t.wrap_x, t.tv_scale, loss, t.grad_sig_out, s_red);
}

// 1024-thread blocks: the loss is one atomic per block on a single address (~9 ns each when they queue up), so the same number of
// threads in 4x fewer blocks shortens the tail of the forward (loss-only) launch
constexpr int CNT = 1024;
__global__ __launch_bounds__(CNT) void composite_mse_kernel(const float *__restrict__ fg, const float *__restrict__ env,
                                                           const float *__restrict__ img, int N, long long plane,
                                                           float scale, const float *__restrict__ scale_dev, float *__restrict__ rec,
                                                           float *__restrict__ loss, float *__restrict__ gfg,
                                                           float *__restrict__ genv) {
    __shared__ float s_red[CNT / DBW_WAVE];
    const long long total = (long long)N * plane;
    float part = 0.f;
    const float loss_scale = scale;
    if (scale_dev) scale *= scale_dev[0];
    for (long long i0 = (long long)blockIdx.x * CNT; i0 < total; i0 += (long long)gridDim.x * CNT) {
        const long long i = i0 + threadIdx.x;
        if (i < total) {
            const long long n = i / plane, p = i % plane;
            const float *f = fg + n * 4 * plane + p, *e = env + n * 4 * plane + p;
            const float *t = img ? img + n * 3 * plane + p : nullptr;
            const float mask = f[3 * plane];
            const float fc3[3] = {f[0], f[plane], f[2 * plane]}, ec3[3] = {e[0], e[plane], e[2 * plane]};
            const float t3[3] = {t ? t[0] : 0.f, t ? t[plane] : 0.f, t ? t[2 * plane] : 0.f};
            float rec3[3], gf3[3], ge3[3], gmask = 0.f;
            part += composite_mse_pixel(fc3, mask, ec3, t3, t != nullptr, 2.f * scale, rec3, gf3, ge3, gmask);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (rec) rec[n * 3 * plane + c * plane + p] = rec3[c];
                if (gfg) { gfg[n * 4 * plane + c * plane + p] = gf3[c]; genv[n * 4 * plane + c * plane + p] = ge3[c]; }
            }
            if (gfg) { gfg[n * 4 * plane + 3 * plane + p] = gmask; genv[n * 4 * plane + 3 * plane + p] = 0.f; }
        }
    }
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0 && loss) {
        float tot = 0.f;
        for (int w = 0; w < CNT / DBW_WAVE; ++w) tot += s_red[w];
        unsafeAtomicAdd(loss, tot * loss_scale);
    }
}

__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, long long n, float step_size, float beta1, float beta2, float eps,
                            float bc2_sqrt) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        adam_update(p[i], g[i], m[i], v[i], step_size, beta1, beta2, eps, bc2_sqrt);
    }
}

struct AdamGroups { long long end[MAX_SETS]; float step_size[MAX_SETS]; };

// parameter groups that differ in learning rate only (optimizer.py:6-18: textures vs everything else), contiguous in one flat buffer
// zero_buf: scratch the NEXT step expects zero-initialised (the per-step zero arena): cleared here, at the end of a step, so that the
// next step does not open with a fill launch in front of its first kernel
__global__ void adam_groups_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                                   long long n, const AdamGroups G, float beta1, float beta2, float eps, float bc2_sqrt,
                                   uint4 *__restrict__ zero_buf, long long zero_vec, const float *__restrict__ skip_flag) {
    // (skip_flag: a training step whose cross-stream wait gave up -- its gradients may be incomplete -- moves nothing; the flag lies
    // outside zero_buf and nobody writes it while this launch runs)
    if (skip_flag && *skip_flag != 0.f) n = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float step_size = G.step_size[0];
#pragma unroll
        for (int k = 1; k < MAX_SETS; ++k) step_size = i >= G.end[k - 1] ? G.step_size[k] : step_size;
        adam_update(p[i], g[i], m[i], v[i], step_size, beta1, beta2, eps, bc2_sqrt);
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < zero_vec; i += (long long)gridDim.x * blockDim.x)
        zero_buf[i] = make_uint4(0u, 0u, 0u, 0u);
}

inline unsigned grid_for(long long work) {
    long long b = (work + NT - 1) / NT;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;   // 256 CUs x 8 resident blocks, grid-stride beyond that
    return (unsigned)b;
}

}  // namespace

extern "C" int dbw_texture_prep_fwd(const float *texture, int n, int h, int w, int decim, float *maps_out,
                                    float *sig_out, dbw_stream_t stream) {
    DBW_REQUIRE(texture && maps_out, "null pointer");
    DBW_REQUIRE(n > 0 && h > 0 && w > 0 && decim >= 1, "bad size");
    DBW_REQUIRE(decim == 1 || sig_out, "sig_out is required when decimating");
    DBW_REQUIRE(decim == 1 || (h % decim == 0 && w % decim == 0), "map size must be a multiple of the decimation factor");
    const long long work = decim > 1 ? (long long)n * (h / decim) * (w / decim) * 64 : (long long)n * h * w * 3;
    hipLaunchKernelGGL(texture_prep_fwd_kernel, dim3(grid_for(work)), dim3(NT), 0, (hipStream_t)stream, texture, n, h,
                       w, decim, maps_out, sig_out);
    return dbw_check_launch("texture_prep_fwd_kernel");
}

extern "C" int dbw_texture_prep_bwd(const float *texture, int n, int h, int w, int decim, const float *grad_maps,
                                    const float *grad_sig, float *grad_texture, dbw_stream_t stream) {
    DBW_REQUIRE(texture && grad_maps && grad_texture, "null pointer");
    DBW_REQUIRE(n > 0 && h > 0 && w > 0 && decim >= 1, "bad size");
    DBW_REQUIRE(decim == 1 || (h % decim == 0 && w % decim == 0), "map size must be a multiple of the decimation factor");
    const long long work = (long long)n * h * w * 3;
    hipLaunchKernelGGL(texture_prep_bwd_kernel, dim3(grid_for(work)), dim3(NT), 0, (hipStream_t)stream, texture, n, h,
                       w, decim, grad_maps, grad_sig, grad_texture);
    return dbw_check_launch("texture_prep_bwd_kernel");
}

extern "C" int dbw_tv_l2sq(const float *maps, int n, int h, int w, int wrap_x, float scale, float *loss,
                           float *grad_maps, dbw_stream_t stream) {
    DBW_REQUIRE(maps && loss, "null pointer");
    DBW_REQUIRE(n > 0 && h > 1 && w > 1, "bad size");
    const long long rows = (long long)n * h;
    hipLaunchKernelGGL(tv_l2sq_kernel, dim3((unsigned)((w + NT - 1) / NT), (unsigned)(rows < 512 ? rows : 512)), dim3(NT), 0,
                       (hipStream_t)stream, maps, n, h, w, wrap_x, scale, loss, grad_maps);
    return dbw_check_launch("tv_l2sq_kernel");
}

namespace {
int check_sets(const dbw_texture_set *sets, int nsets) {
    DBW_REQUIRE(sets && nsets >= 1 && nsets <= MAX_SETS, "1..4 texture sets");
    for (int i = 0; i < nsets; ++i) {
        const dbw_texture_set &t = sets[i];
        DBW_REQUIRE(t.n > 0 && t.h > 1 && t.w > 1 && t.decim >= 1, "bad size");
        DBW_REQUIRE(t.decim == 1 || (t.h % t.decim == 0 && t.w % t.decim == 0), "map size must be a multiple of the decimation factor");
    }
    return DBW_OK;
}
TextureSets pack_sets(const dbw_texture_set *sets, int nsets) {
    TextureSets S;
    for (int i = 0; i < MAX_SETS; ++i) S.s[i] = sets[i < nsets ? i : 0];
    return S;
}
}  // namespace

extern "C" int dbw_texture_prep_fwd_sets(const dbw_texture_set *sets, int nsets, dbw_stream_t stream) {
    if (int rc = check_sets(sets, nsets)) return rc;
    long long work = 0;
    for (int i = 0; i < nsets; ++i) {
        const dbw_texture_set &t = sets[i];
        DBW_REQUIRE(t.texture && t.maps, "null pointer");
        DBW_REQUIRE(t.decim == 1 || t.sig, "sig is required when decimating");
        const long long wk = t.decim > 1 ? (long long)t.n * (t.h / t.decim) * (t.w / t.decim) * 64 : (long long)t.n * t.h * t.w * 3;
        work = wk > work ? wk : work;
    }
    hipLaunchKernelGGL(texture_prep_fwd_sets_kernel, dim3(grid_for(work), (unsigned)nsets), dim3(NT), 0, (hipStream_t)stream,
                       pack_sets(sets, nsets));
    return dbw_check_launch("texture_prep_fwd_sets_kernel");
}

extern "C" int dbw_texture_prep_bwd_sets(const dbw_texture_set *sets, int nsets, dbw_stream_t stream) {
    if (int rc = check_sets(sets, nsets)) return rc;
    long long work = 0;
    for (int i = 0; i < nsets; ++i) {
        const dbw_texture_set &t = sets[i];
        DBW_REQUIRE(t.texture && t.grad_maps && t.grad_texture, "null pointer");
        const long long wk = (long long)t.n * t.h * t.w * 3;
        work = wk > work ? wk : work;
    }
    hipLaunchKernelGGL(texture_prep_bwd_sets_kernel, dim3(grid_for(work), (unsigned)nsets), dim3(NT), 0, (hipStream_t)stream,
                       pack_sets(sets, nsets));
    return dbw_check_launch("texture_prep_bwd_sets_kernel");
}

extern "C" int dbw_tv_l2sq_sets(const dbw_texture_set *sets, int nsets, float *loss, dbw_stream_t stream) {
    if (int rc = check_sets(sets, nsets)) return rc;
    DBW_REQUIRE(loss, "null pointer");
    long long rows = 0;
    int w = 0;
    for (int i = 0; i < nsets; ++i) {
        DBW_REQUIRE(sets[i].sig, "null pointer");
        const long long r = (long long)sets[i].n * sets[i].h;
        rows = r > rows ? r : rows;
        w = sets[i].w > w ? sets[i].w : w;
    }
    hipLaunchKernelGGL(tv_l2sq_sets_kernel, dim3((unsigned)((w + NT - 1) / NT), (unsigned)(rows < 512 ? rows : 512), (unsigned)nsets),
                       dim3(NT), 0, (hipStream_t)stream, pack_sets(sets, nsets), loss);
    return dbw_check_launch("tv_l2sq_sets_kernel");
}

extern "C" int dbw_adam_step_groups(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, const int64_t *group_end,
                                    const float *lr, int ngroups, float beta1, float beta2, float eps, int step, void *zero_buf,
                                    int64_t zero_bytes, const float *skip_flag, dbw_stream_t stream) {
    DBW_REQUIRE(param && grad && exp_avg && exp_avg_sq && group_end && lr, "null pointer");
    DBW_REQUIRE(zero_bytes >= 0 && (zero_bytes == 0 || (zero_buf && zero_bytes % 16 == 0 && ((uintptr_t)zero_buf & 15) == 0)),
                "zero_buf: 16-byte aligned, a multiple of 16 bytes");
    DBW_REQUIRE(ngroups >= 1 && ngroups <= MAX_SETS && step >= 1, "1..4 groups, step >= 1");
    for (int k = 0; k < ngroups; ++k) DBW_REQUIRE(group_end[k] >= (k ? group_end[k - 1] : 0), "group ends must not decrease");
    const long long n = group_end[ngroups - 1];
    if (n == 0) return DBW_OK;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    AdamGroups G;
    for (int k = 0; k < MAX_SETS; ++k) {
        G.end[k] = k < ngroups ? group_end[k] : n;
        G.step_size[k] = (float)(lr[k < ngroups ? k : ngroups - 1] / bc1);
    }
    hipLaunchKernelGGL(adam_groups_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, G,
                       beta1, beta2, eps, (float)sqrt(bc2), (uint4 *)zero_buf, (long long)(zero_bytes / 16), skip_flag);
    return dbw_check_launch("adam_groups_kernel");
}

extern "C" int dbw_composite_mse(const float *fg, const float *env, const float *imgs, int N, int H, int W,
                                 float scale, const float *scale_dev, float *rec, float *loss_sum, float *grad_fg,
                                 float *grad_env, dbw_stream_t stream) {
    DBW_REQUIRE(fg && env, "null pointer");
    DBW_REQUIRE(imgs || (!loss_sum && !grad_fg && rec), "imgs may only be NULL when just `rec` is wanted");
    DBW_REQUIRE((grad_fg && grad_env) || (!grad_fg && !grad_env), "grad_fg/grad_env: both or none");
    DBW_REQUIRE(N >= 0 && H > 0 && W > 0, "bad size");
    if (N == 0) return DBW_OK;
    const long long plane = (long long)H * W;
    long long gb = ((long long)N * plane + CNT - 1) / CNT;
    if (gb > 512) gb = 512;                 // 256 CUs x 2 resident 1024-thread blocks, grid-stride beyond that
    hipLaunchKernelGGL(composite_mse_kernel, dim3((unsigned)(gb < 1 ? 1 : gb)), dim3(CNT), 0, (hipStream_t)stream, fg,
                       env, imgs, N, plane, scale, scale_dev, rec, loss_sum, grad_fg, grad_env);
    return dbw_check_launch("composite_mse_kernel");
}

extern "C" int dbw_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, int step, dbw_stream_t stream) {
    DBW_REQUIRE(param && grad && exp_avg && exp_avg_sq, "null pointer");
    DBW_REQUIRE(n >= 0 && step >= 1, "bad size/step");
    if (n == 0) return DBW_OK;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, (long long)n, (float)(lr / bc1), beta1, beta2, eps, (float)sqrt(bc2));
    return dbw_check_launch("adam_kernel");
}
