// Param -> mesh kernels (superquadric deformation + 6D-rotation posing + world transform) with hand-derived backward,
// and the overlap regulariser.  Reference: src/model/dbw.py:282-287,297-311,343-352,389-405,
// src/utils/superquadric.py:10-38, src/utils/pytorch.py:31-36, pytorch3d rotation_6d_to_matrix.
// These are latency-bound launches over K x 42 vertices / K x 1000 sample points: one wave per block primitive, the
// per-primitive gradient (17 numbers) is reduced inside the wave and written by one lane -- no atomics, no MFMA
// (~0.1 MFLOP per step, matrix cores do not pay).
#include "dbw_common.h"
#include "model_math.h"
#include "rng_math.h"
#include "texture_body.h"
#include "step_kernels.h"
#include "../../include/dbw_hip.h"

using namespace dbw;

namespace {

__device__ __forceinline__ int dense_index(const int *keep, int k) {
    if (!keep) return k;
    int c = 0;
    for (int i = 0; i < k; ++i) c += keep[i] != 0;
    return c;
}

__global__ __launch_bounds__(64) void sq_blocks_fwd_kernel(const float *sq_eps, const float *S, const float *R6, const float *T,
                                                           const float *trig, const int *keep, int dense, int Kb, int nv, float ratio,
                                                           float scale_min, float S_world, const float *Rw, const float *Tw,
                                                           float *verts) {
    const int k = blockIdx.x;
    if (keep && !keep[k]) {
        if (!dense)      // collapse the dead block to one point: zero-area faces, dropped by the rasteriser
            for (int v = threadIdx.x; v < nv * 3; v += 64) verts[(long long)k * nv * 3 + v] = 0.f;
        return;
    }
    const int ko = dense ? dense_index(keep, k) : k;
    Pose p;
    load_pose(sq_eps, S, R6, T, k, scale_min, p);
    const long long plane = (long long)Kb * nv;
    for (int v = threadIdx.x; v < nv; v += 64) {
        const long long o = (long long)k * nv + v;
        float loc[3], de1[3], de2[3];
        parametric_sq_point(trig[o], trig[plane + o], trig[2 * plane + o], trig[3 * plane + o], p.e1, p.e2, ratio, loc, de1, de2);
        pose_fwd(p, loc, S_world, Rw, Tw, verts + ((long long)ko * nv + v) * 3);
    }
}

__global__ __launch_bounds__(64) void sq_blocks_bwd_kernel(const float *sq_eps, const float *S, const float *R6, const float *T,
                                                           const float *trig, const int *keep, int dense, int Kb, int nv, float ratio,
                                                           float scale_min, float S_world, const float *Rw,
                                                           const float *gverts, float *g_sq_eps, float *g_S, float *g_R6,
                                                           float *g_T) {
    const int k = blockIdx.x;
    if (keep && !keep[k]) return;
    const int ko = dense ? dense_index(keep, k) : k;
    Pose p;
    load_pose(sq_eps, S, R6, T, k, scale_min, p);
    const long long plane = (long long)Kb * nv;
    float acc[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) acc[i] = 0.f;
    for (int v = threadIdx.x; v < nv; v += 64) {
        const long long o = (long long)k * nv + v;
        float loc[3], de1[3], de2[3];
        parametric_sq_point(trig[o], trig[plane + o], trig[2 * plane + o], trig[3 * plane + o], p.e1, p.e2, ratio, loc, de1, de2);
        float gv[3];
        pose_bwd(p, loc, S_world, Rw, gverts + ((long long)ko * nv + v) * 3, acc, gv);
        acc[0] += gv[0] * de1[0] + gv[1] * de1[1] + gv[2] * de1[2];
        acc[1] += gv[0] * de2[0] + gv[1] * de2[1] + gv[2] * de2[2];
    }
#pragma unroll
    for (int i = 0; i < 17; ++i) acc[i] = wave_sum(acc[i]);
    if (threadIdx.x == 0) finish_pose_grads(p, S, k, acc, g_sq_eps, g_S, g_R6, g_T);
}

__global__ __launch_bounds__(64) void posed_mesh_fwd_kernel(const float *base, int nv, const float *R6, const float *T,
                                                            float S_world, const float *Rw, const float *Tw, float *verts) {
    Pose p;
    load_pose(nullptr, nullptr, R6, T, 0, 0.f, p);
    for (int v = blockIdx.x * 64 + threadIdx.x; v < nv; v += gridDim.x * 64)
        pose_fwd(p, base + (long long)v * 3, S_world, Rw, Tw, verts + (long long)v * 3);
}

__global__ __launch_bounds__(64) void posed_mesh_bwd_kernel(const float *base, int nv, const float *R6, const float *T,
                                                            float S_world, const float *Rw, const float *gverts, float *g_R6,
                                                            float *g_T) {
    Pose p;
    load_pose(nullptr, nullptr, R6, T, 0, 0.f, p);
    float acc[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) acc[i] = 0.f;
    for (int v = threadIdx.x; v < nv; v += 64) {
        float gv[3];
        pose_bwd(p, base + (long long)v * 3, S_world, Rw, gverts + (long long)v * 3, acc, gv);
    }
#pragma unroll
    for (int i = 0; i < 17; ++i) acc[i] = wave_sum(acc[i]);
    if (threadIdx.x == 0) finish_pose_grads(p, nullptr, 0, acc, nullptr, nullptr, g_R6, g_T);
}

// ---- overlap ------------------------------------------------------------------------------------------------------
constexpr int MAX_KB = 64;
constexpr int NG = 18;   // raw grads per block: e1,e2 | S(3) | R(9) | T(3) | alpha

struct BlockP {
    float e1, e2, sr[3], R[9], T[3], alpha;   // sr = S*ratio
};

// one workgroup of the overlap term: 256 sample points against every block.  u == nullptr: the samples are drawn in registers
// (rng_math.h: stream 1 of (seed, rng_step), index = the point) instead of being read from a caller's torch.rand buffer
struct OverlapShared { BlockP b[MAX_KB]; float g[MAX_KB * NG]; float red[4]; };
__device__ __forceinline__ void overlap_block(const float *u, unsigned long long seed, unsigned long long rng_step, int npts, const float *sq_eps,
                                              const float *S, const float *R6, const float *T, const float *alpha, int Kb, float ratio,
                                              float scale_min, float inv_temp, float thresh, float scale_over_P, float *loss, float *ws,
                                              OverlapShared &sh) {
    BlockP *s_b = sh.b;
    float *s_g = sh.g, *s_red = sh.red;
    for (int j = threadIdx.x; j < Kb; j += blockDim.x) {
        Pose p;
        load_pose(sq_eps, S, R6, T, j, scale_min, p);
        BlockP &b = s_b[j];
        b.e1 = p.e1; b.e2 = p.e2; b.alpha = alpha[j];
        for (int i = 0; i < 3; ++i) { b.sr[i] = p.S[i] * ratio; b.T[i] = p.T[i]; b.R[i] = p.rot.b1[i]; b.R[3 + i] = p.rot.b2[i]; b.R[6 + i] = p.rot.b3[i]; }
    }
    for (int i = threadIdx.x; i < Kb * NG; i += blockDim.x) s_g[i] = 0.f;
    __syncthreads();
    const long long P = (long long)Kb * npts;
    const long long pi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float contrib = 0.f;
    if (pi < P) {
        const int k = (int)(pi / npts);
        const BlockP &bk = s_b[k];
        float l[3], pt[3];
        float u3[3];
        if (u) { u3[0] = u[pi * 3]; u3[1] = u[pi * 3 + 1]; u3[2] = u[pi * 3 + 2]; }
        else {
            const Philox4 r = step_random(seed, rng_step, 1u, (uint32_t)pi);
            u3[0] = uniform01(r.x); u3[1] = uniform01(r.y); u3[2] = uniform01(r.z);
        }
        for (int i = 0; i < 3; ++i) l[i] = (u3[i] * 2.f - 1.f) * bk.sr[i];
        for (int c = 0; c < 3; ++c) pt[c] = l[0] * bk.R[c] + l[1] * bk.R[3 + c] + l[2] * bk.R[6 + c] + bk.T[c];
        float sum = 0.f;
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && !(sum - thresh > 0.f)) break;
            for (int j = 0; j < Kb; ++j) {
                const BlockP &b = s_b[j];
                float q[3], inv[3];
                for (int c = 0; c < 3; ++c) q[c] = pt[c] - b.T[c];
                for (int i = 0; i < 3; ++i) inv[i] = (q[0] * b.R[i * 3] + q[1] * b.R[i * 3 + 1] + q[2] * b.R[i * 3 + 2]) / b.sr[i];
                float pc[3];
                bool inr[3];
                for (int i = 0; i < 3; ++i) { inr[i] = inv[i] >= -5.f && inv[i] <= 5.f; pc[i] = inv[i] < -5.f ? -5.f : (inv[i] > 5.f ? 5.f : inv[i]); }
                ImplicitSq im;
                const float sdf = implicit_sq_sdf2(pc, b.e1, b.e2, im);
                const float occ = sigmoidf(-sdf * inv_temp);
                if (pass == 0) { sum += occ * b.alpha; continue; }
                // backward of scale_over_P * (sum - thresh)
                const float go = scale_over_P;
                float *g = s_g + j * NG;
                atomicAdd(g + 17, go * occ);
                const float gsdf = go * b.alpha * occ * (1.f - occ) * (-inv_temp);
                if (gsdf == 0.f) continue;
                float ge1, ge2, gpc[3];
                implicit_sq_sdf2_bwd(pc, b.e1, b.e2, im, gsdf, ge1, ge2, gpc);
                const float gp[3] = {inr[0] ? gpc[0] : 0.f, inr[1] ? gpc[1] : 0.f, inr[2] ? gpc[2] : 0.f};
                atomicAdd(g + 0, ge1); atomicAdd(g + 1, ge2);
                for (int i = 0; i < 3; ++i) {
                    if (gp[i] == 0.f) continue;
                    const float gi = gp[i] / b.sr[i];
                    atomicAdd(g + 2 + i, -gp[i] * inv[i] / b.sr[i] * ratio);     // d/dS_i (S*ratio in the denominator)
                    for (int c = 0; c < 3; ++c) {
                        atomicAdd(g + 5 + i * 3 + c, gi * q[c]);
                        atomicAdd(g + 14 + c, -gi * b.R[i * 3 + c]);
                    }
                }
            }
        }
        const float ov = sum - thresh;
        contrib = ov > 0.f ? ov : 0.f;
    }
    contrib = wave_sum(contrib);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = contrib;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = s_red[0] + s_red[1] + s_red[2] + s_red[3];
        if (t != 0.f) unsafeAtomicAdd(loss, scale_over_P * t);
    }
    for (int i = threadIdx.x; i < Kb * NG; i += blockDim.x)
        if (s_g[i] != 0.f) unsafeAtomicAdd(ws + i, s_g[i]);
}

__global__ __launch_bounds__(256) void overlap_kernel(const float *u, int npts, const float *sq_eps, const float *S,
                                                      const float *R6, const float *T, const float *alpha, int Kb,
                                                      float ratio, float scale_min, float inv_temp, float thresh,
                                                      float scale_over_P, float *loss, float *ws) {
    __shared__ OverlapShared sh;
    overlap_block(u, 0ull, 0ull, npts, sq_eps, S, R6, T, alpha, Kb, ratio, scale_min, inv_temp, thresh, scale_over_P, loss, ws, sh);
}

// The regularisers of a training step in ONE launch: the overlap term, then -- in the workgroup that finishes last -- its per-block
// finish (raw accumulators -> parameter gradients) and the parsimony term; both add into g_alpha_full, one after the other in one thread.
__global__ __launch_bounds__(256) void regularisers_kernel(const RegulariserArgs A) {
    __shared__ OverlapShared sh;
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const bool overlap = A.overlap_scale != 0.f;
    if (overlap) {
        const long long P = (long long)A.nb * A.npts;
        overlap_block(A.u, A.seed, A.rng_step, A.npts, A.sq_eps, A.S, A.R6, A.T, A.alpha_full, A.nb, A.ratio, A.scale_min, A.inv_temp, A.thresh,
                      A.overlap_scale / (float)P, A.loss_overlap, A.ws, sh);
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(A.ticket, 1u) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (overlap && tid < A.nb) {
        Pose p;
        load_pose(A.sq_eps, A.S, A.R6, A.T, tid, A.scale_min, p);
        finish_pose_grads(p, A.S, tid, A.ws + tid * NG, A.g_sq_eps, A.g_S, A.g_R6, A.g_T);
        A.g_alpha_full[tid] += A.ws[tid * NG + 17];
    }
    __syncthreads();
    if (A.pars_scale != 0.f && tid < 64) {          // (sqrt_mean_kernel: loss += scale * mean(max(x, eps)^0.5))
        float acc = 0.f;
        for (int i = tid; i < A.nb; i += 64) {
            const float v = A.alpha_full[i], c = v > A.pars_eps ? v : A.pars_eps, r = sqrtf(c);
            acc += r;
            if (v > A.pars_eps) A.g_alpha_full[i] += A.pars_scale * 0.5f / r / (float)A.nb;
        }
        acc = dbw::wave_sum(acc);
        if (tid == 0) unsafeAtomicAdd(A.loss_parsimony, A.pars_scale * acc / (float)A.nb);
    }
    if (tid == 0) *A.ticket = 0u;
}

__global__ void overlap_finish_kernel(const float *sq_eps, const float *S, const float *R6, const float *T, int Kb,
                                      float scale_min, const float *ws, float *g_sq_eps, float *g_S, float *g_R6,
                                      float *g_T, float *g_alpha) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Kb) return;
    Pose p;
    load_pose(sq_eps, S, R6, T, j, scale_min, p);
    finish_pose_grads(p, S, j, ws + j * NG, g_sq_eps, g_S, g_R6, g_T);
    if (g_alpha) g_alpha[j] += ws[j * NG + 17];
}

// ---- block opacities (dbw.py:297-311) and the parsimony regulariser (dbw.py:373-377) ------------------------------
// One launch replaces the chain randn*s + logit -> sigmoid -> clone -> sigmoid(logit) > thresh -> mask multiply.
__global__ void block_alpha_fwd_kernel(const float *logit, const float *noise, float noise_scale, float thresh, int Kb,
                                       float *alpha, float *alpha_full, int *keep) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Kb) return;
    const float x = logit[j] + (noise ? noise_scale * noise[j] : 0.f);
    const float a = 1.f / (1.f + expf(-x));
    alpha[j] = a;
    int m = 1;
    if (thresh >= 0.f) m = (1.f / (1.f + expf(-logit[j]))) > thresh ? 1 : 0;     // the mask looks at the noise-free opacity
    alpha_full[j] = m ? a : 0.f;
    if (keep) keep[j] = m;
}

__global__ void block_alpha_bwd_kernel(const float *alpha, const int *keep, const float *g_alpha, int parts, const float *g_alpha_full,
                                       int Kb, float *g_logit) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Kb) return;
    const float a = alpha[j];
    float g = 0.f;
    if (g_alpha)
        for (int i = 0; i < parts; ++i) g += g_alpha[j * parts + i];
    if (g_alpha_full && (!keep || keep[j])) g += g_alpha_full[j];
    g_logit[j] = g * a * (1.f - a);
}

// loss += scale * mean(max(x, eps)^0.5); grad += scale * 0.5 / sqrt(x) / n where x > eps (clamp has no gradient below)
__global__ void sqrt_mean_kernel(const float *x, int n, float eps, float scale, float *loss, float *grad) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = x[i], c = v > eps ? v : eps, r = sqrtf(c);
        acc += r;
        if (grad && v > eps) grad[i] += scale * 0.5f / r / (float)n;
    }
    acc = dbw::wave_sum(acc);
    if (threadIdx.x == 0 && loss) unsafeAtomicAdd(loss, scale * acc / (float)n);
}

// The head of a training step in ONE launch (step_kernels.h): blockIdx.y < nsets -> sigmoid / decimation of texture set y (grid-stride
// over x); blockIdx.y == nsets -> x < nb: opacity + vertices of block x, x == nb: the ground's vertices and the clearing of the small
// gradient accumulators.  The same device functions as the stand-alone kernels, so the same bits.
__global__ __launch_bounds__(256) void step_prologue_kernel(const PrologueArgs P) {
    if ((int)blockIdx.y < P.nsets) {
        const dbw_texture_set &t = P.tex.s[blockIdx.y];
        texture_prep_fwd_body(t.texture, t.n, t.h, t.w, t.decim, t.maps, t.sig, blockIdx.x, gridDim.x);
        return;
    }
    const int k = blockIdx.x, tid = threadIdx.x;
    if (k < P.nb) {
        // dbw.py:297-311 (block_alpha_fwd_kernel); every thread evaluates the block's mask, thread 0 writes
        const float lg = P.alpha_logit[k];
        float nz = 0.f;
        if (P.noise_scale != 0.f) {
            if (P.noise) nz = P.noise[k];
            else { const Philox4 r = step_random(P.seed, P.rng_step, 0u, (uint32_t)k); nz = normal01(r.x, r.y); }
        }
        const float x = lg + (P.noise_scale != 0.f ? P.noise_scale * nz : 0.f);
        const float a = 1.f / (1.f + expf(-x));
        int m = 1;
        if (P.thresh >= 0.f) m = (1.f / (1.f + expf(-lg))) > P.thresh ? 1 : 0;
        if (tid == 0) { P.alpha[k] = a; P.alpha_full[k] = m ? a : 0.f; P.keep[k] = m; }
        if (!m) {            // a dead block collapses to one point: zero-area faces, dropped by the rasteriser (sq_blocks_fwd_kernel, dense = 0)
            for (int v = tid; v < P.nv * 3; v += blockDim.x) P.blk_verts[(long long)k * P.nv * 3 + v] = 0.f;
            return;
        }
        Pose p;
        load_pose(P.sq_eps, P.S, P.R6, P.T, k, P.scale_min, p);
        const long long plane = (long long)P.nb * P.nv;
        for (int v = tid; v < P.nv; v += blockDim.x) {
            const long long o = (long long)k * P.nv + v;
            float loc[3], de1[3], de2[3];
            parametric_sq_point(P.trig[o], P.trig[plane + o], P.trig[2 * plane + o], P.trig[3 * plane + o], p.e1, p.e2, P.ratio, loc, de1, de2);
            pose_fwd(p, loc, P.S_world, P.Rw, P.Tw, P.blk_verts + o * 3);
            if (P.sq_local) {          // the backward of this block (blocks_tail_kernel) re-reads what it would otherwise recompute with 8 powf / logf per vertex
                float *q = P.sq_local + o * 9;
#pragma unroll
                for (int i = 0; i < 3; ++i) { q[i] = loc[i]; q[3 + i] = de1[i]; q[6 + i] = de2[i]; }
            }
        }
    } else if (k == P.nb) {
        Pose p;
        load_pose(nullptr, nullptr, P.R6g, P.Tg, 0, 0.f, p);
        for (int v = tid; v < P.ngv; v += blockDim.x) pose_fwd(p, P.ground_base + (long long)v * 3, P.S_world, P.Rw, P.Tw, P.ground_verts + (long long)v * 3);
        for (int i = tid; i < P.nzero0; i += blockDim.x) P.zero0[i] = 0.f;
    }
}

// Backward of the blocks' pose / shape + of their opacities (step_kernels.h: BlocksTailArgs): workgroup = block, one wave.
__global__ __launch_bounds__(64) void blocks_tail_kernel(const BlocksTailArgs A) {
    const int k = blockIdx.x, lane = threadIdx.x;
    const bool alive = !A.keep || A.keep[k];
    if (k == 0 && lane == 0 && A.void_flag) *A.void_flag = *A.void_raised;       // (train_step.hip: the run's latch of the plan's sticky word)
    if (lane == 0) {          // block_alpha_bwd_kernel
        const float a = A.alpha[k];
        float g = 0.f;
        if (A.g_alpha_parts)
            for (int i = 0; i < A.alpha_parts; ++i) g += A.g_alpha_parts[k * A.alpha_parts + i];
        if (A.g_alpha_full && alive) g += A.g_alpha_full[k];
        A.g_logit[k] = g * a * (1.f - a);
    }
    if (!alive) return;
    Pose p;
    load_pose(A.sq_eps, A.S, A.R6, A.T, k, A.scale_min, p);
    float acc[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) acc[i] = 0.f;
    for (int v = lane; v < A.nv; v += 64) {
        const long long o = (long long)k * A.nv + v;
        const float *q = A.sq_local + o * 9;          // block-frame point, d / d eps1, d / d eps2 as this step's prologue computed them
        const float loc[3] = {q[0], q[1], q[2]}, de1[3] = {q[3], q[4], q[5]}, de2[3] = {q[6], q[7], q[8]};
        float gv[3];
        pose_bwd(p, loc, A.S_world, A.Rw, A.g_verts + o * 3, acc, gv);
        acc[0] += gv[0] * de1[0] + gv[1] * de1[1] + gv[2] * de1[2];
        acc[1] += gv[0] * de2[0] + gv[1] * de2[1] + gv[2] * de2[2];
    }
#pragma unroll
    for (int i = 0; i < 17; ++i) acc[i] = wave_sum(acc[i]);
    if (lane == 0) finish_pose_grads(p, A.S, k, acc, A.g_sq_eps, A.g_S, A.g_R6, A.g_T);
}

}  // namespace

int dbw::launch_blocks_tail(const BlocksTailArgs &A, hipStream_t s) {
    DBW_REQUIRE(A.sq_eps && A.S && A.R6 && A.T && A.sq_local && A.Rw && A.g_verts && A.g_sq_eps && A.g_S && A.g_R6 && A.g_T && A.alpha && A.g_logit, "null pointer");
    DBW_REQUIRE(A.nb > 0 && A.nv > 0 && A.alpha_parts >= 1, "bad size");
    hipLaunchKernelGGL(blocks_tail_kernel, dim3((unsigned)A.nb), dim3(64), 0, s, A);
    return dbw_check_launch("blocks_tail_kernel");
}

int dbw::launch_step_prologue(const PrologueArgs &P, hipStream_t s) {
    DBW_REQUIRE(P.nsets >= 0 && P.nsets <= STEP_MAX_SETS && P.nb > 0 && P.nb <= MAX_KB && P.nv > 0 && P.ngv > 0, "bad size");
    long long work = (long long)(P.nb + 1) * 256;
    for (int i = 0; i < P.nsets; ++i) {
        const dbw_texture_set &t = P.tex.s[i];
        DBW_REQUIRE(t.texture && t.maps && t.n > 0 && t.h > 1 && t.w > 1 && t.decim >= 1 && (t.decim == 1 || (t.sig && t.h % t.decim == 0 && t.w % t.decim == 0)),
                    "bad texture set");
        const long long wk = t.decim > 1 ? (long long)t.n * (t.h / t.decim) * (t.w / t.decim) * 64 : (long long)t.n * t.h * t.w * 3;
        work = wk > work ? wk : work;
    }
    long long gx = (work + 255) / 256;
    if (gx > 2048) gx = 2048;
    if (gx < P.nb + 1) gx = P.nb + 1;
    hipLaunchKernelGGL(step_prologue_kernel, dim3((unsigned)gx, (unsigned)(P.nsets + 1)), dim3(256), 0, s, P);
    return dbw_check_launch("step_prologue_kernel");
}

int dbw::launch_regularisers(const RegulariserArgs &A, hipStream_t s) {
    DBW_REQUIRE(A.nb > 0 && A.nb <= MAX_KB && A.ticket && A.g_alpha_full && A.alpha_full, "bad argument");
    const bool overlap = A.overlap_scale != 0.f;
    if (!overlap && A.pars_scale == 0.f) return DBW_OK;
    DBW_REQUIRE(!overlap || (A.npts > 0 && A.ws && A.loss_overlap && A.g_sq_eps && A.g_S && A.g_R6 && A.g_T && A.inv_temp > 0.f), "bad overlap argument");
    DBW_REQUIRE(A.pars_scale == 0.f || A.loss_parsimony, "bad parsimony argument");
    const long long P = (long long)A.nb * A.npts;
    hipLaunchKernelGGL(regularisers_kernel, dim3(overlap ? (unsigned)((P + 255) / 256) : 1u), dim3(256), 0, s, A);
    return dbw_check_launch("regularisers_kernel");
}

extern "C" int dbw_block_alpha_fwd(const float *alpha_logit, const float *noise, float noise_scale, float mask_threshold, int Kb,
                                   float *alpha, float *alpha_full, int32_t *keep, dbw_stream_t stream) {
    DBW_REQUIRE(alpha_logit && alpha && alpha_full, "null pointer");
    DBW_REQUIRE(Kb > 0, "bad size");
    hipLaunchKernelGGL(block_alpha_fwd_kernel, dim3((Kb + 63) / 64), dim3(64), 0, (hipStream_t)stream, alpha_logit, noise, noise_scale,
                       mask_threshold, Kb, alpha, alpha_full, keep);
    return dbw_check_launch("block_alpha_fwd_kernel");
}

extern "C" int dbw_block_alpha_bwd(const float *alpha, const int32_t *keep, const float *g_alpha, int g_alpha_parts,
                                   const float *g_alpha_full, int Kb, float *g_logit, dbw_stream_t stream) {
    DBW_REQUIRE(alpha && g_logit, "null pointer");
    DBW_REQUIRE(Kb > 0 && g_alpha_parts >= 1, "bad size");
    hipLaunchKernelGGL(block_alpha_bwd_kernel, dim3((Kb + 63) / 64), dim3(64), 0, (hipStream_t)stream, alpha, keep, g_alpha, g_alpha_parts,
                       g_alpha_full, Kb, g_logit);
    return dbw_check_launch("block_alpha_bwd_kernel");
}

extern "C" int dbw_sqrt_mean(const float *x, int n, float eps, float scale, float *loss, float *grad, dbw_stream_t stream) {
    DBW_REQUIRE(x && (loss || grad), "null pointer");
    DBW_REQUIRE(n > 0, "bad size");
    hipLaunchKernelGGL(sqrt_mean_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, x, n, eps, scale, loss, grad);
    return dbw_check_launch("sqrt_mean_kernel");
}

extern "C" int dbw_sq_blocks_fwd(const float *sq_eps, const float *S, const float *R6, const float *T, const float *trig,
                                 const int32_t *keep, int dense, int Kb, int nv, float ratio, float scale_min, float S_world,
                                 const float *R_world, const float *T_world, float *verts, dbw_stream_t stream) {
    DBW_REQUIRE(sq_eps && S && R6 && T && trig && R_world && verts, "null pointer");
    DBW_REQUIRE(Kb > 0 && nv > 0, "bad size");
    hipLaunchKernelGGL(sq_blocks_fwd_kernel, dim3(Kb), dim3(64), 0, (hipStream_t)stream, sq_eps, S, R6, T, trig, keep, dense, Kb,
                       nv, ratio, scale_min, S_world, R_world, T_world, verts);
    return dbw_check_launch("sq_blocks_fwd_kernel");
}

extern "C" int dbw_sq_blocks_bwd(const float *sq_eps, const float *S, const float *R6, const float *T, const float *trig,
                                 const int32_t *keep, int dense, int Kb, int nv, float ratio, float scale_min, float S_world,
                                 const float *R_world, const float *grad_verts, float *g_sq_eps, float *g_S, float *g_R6,
                                 float *g_T, dbw_stream_t stream) {
    DBW_REQUIRE(sq_eps && S && R6 && T && trig && R_world && grad_verts && g_sq_eps && g_S && g_R6 && g_T, "null pointer");
    DBW_REQUIRE(Kb > 0 && nv > 0, "bad size");
    hipLaunchKernelGGL(sq_blocks_bwd_kernel, dim3(Kb), dim3(64), 0, (hipStream_t)stream, sq_eps, S, R6, T, trig, keep, dense, Kb,
                       nv, ratio, scale_min, S_world, R_world, grad_verts, g_sq_eps, g_S, g_R6, g_T);
    return dbw_check_launch("sq_blocks_bwd_kernel");
}

extern "C" int dbw_posed_mesh_fwd(const float *base, int nv, const float *R6, const float *T, float S_world,
                                  const float *R_world, const float *T_world, float *verts, dbw_stream_t stream) {
    DBW_REQUIRE(base && R6 && T && R_world && verts, "null pointer");
    DBW_REQUIRE(nv > 0, "bad size");
    hipLaunchKernelGGL(posed_mesh_fwd_kernel, dim3((nv + 63) / 64), dim3(64), 0, (hipStream_t)stream, base, nv, R6, T,
                       S_world, R_world, T_world, verts);
    return dbw_check_launch("posed_mesh_fwd_kernel");
}

extern "C" int dbw_posed_mesh_bwd(const float *base, int nv, const float *R6, const float *T, float S_world,
                                  const float *R_world, const float *grad_verts, float *g_R6, float *g_T,
                                  dbw_stream_t stream) {
    DBW_REQUIRE(base && R6 && T && R_world && grad_verts && g_R6 && g_T, "null pointer");
    DBW_REQUIRE(nv > 0, "bad size");
    hipLaunchKernelGGL(posed_mesh_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, base, nv, R6, T, S_world,
                       R_world, grad_verts, g_R6, g_T);
    return dbw_check_launch("posed_mesh_bwd_kernel");
}

extern "C" int dbw_overlap_loss(const float *u, int npts, const float *sq_eps, const float *S, const float *R6,
                                const float *T, const float *alpha, int Kb, float ratio, float scale_min,
                                float temperature, float n_blocks_thresh, float scale, float *loss, float *g_sq_eps,
                                float *g_S, float *g_R6, float *g_T, float *g_alpha, float *workspace,
                                dbw_stream_t stream) {
    DBW_REQUIRE(u && sq_eps && S && R6 && T && alpha && loss && g_sq_eps && g_S && g_R6 && g_T && workspace, "null pointer");
    DBW_REQUIRE(Kb > 0 && Kb <= MAX_KB && npts > 0 && temperature > 0.f, "bad size (n_blocks <= 64)");
    const long long P = (long long)Kb * npts;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(overlap_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, u, npts, sq_eps, S, R6, T,
                       alpha, Kb, ratio, scale_min, 1.f / temperature, n_blocks_thresh, scale / (float)P, loss, workspace);
    int rc = dbw_check_launch("overlap_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(overlap_finish_kernel, dim3(1), dim3(MAX_KB), 0, s, sq_eps, S, R6, T, Kb, scale_min, workspace,
                       g_sq_eps, g_S, g_R6, g_T, g_alpha);
    return dbw_check_launch("overlap_finish_kernel");
}
