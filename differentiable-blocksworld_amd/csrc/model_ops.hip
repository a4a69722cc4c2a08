// Param -> mesh kernels (superquadric deformation + 6D-rotation posing + world transform) with hand-derived backward,
// and the overlap regulariser.  Reference: src/model/dbw.py:282-287,297-311,343-352,389-405,
// src/utils/superquadric.py:10-38, src/utils/pytorch.py:31-36, pytorch3d rotation_6d_to_matrix.
// These are latency-bound launches over K x 42 vertices / K x 1000 sample points: one wave per block primitive, the
// per-primitive gradient (17 numbers) is reduced inside the wave and written by one lane -- no atomics, no MFMA
// (~0.1 MFLOP per step, matrix cores do not pay).
#include "dbw_common.h"
#include "../../include/dbw_hip.h"

using namespace dbw;

namespace {

constexpr float NORM_EPS = 1e-12f;   // F.normalize eps

struct Rot6 {
    float b1[3], b2[3], b3[3], a2[3];
    float n1, n2, d;
};

__device__ __forceinline__ void rot6d_fwd(const float *a, Rot6 &r) {
    r.n1 = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    const float i1 = 1.f / (r.n1 > NORM_EPS ? r.n1 : NORM_EPS);
#pragma unroll
    for (int i = 0; i < 3; ++i) { r.b1[i] = a[i] * i1; r.a2[i] = a[3 + i]; }
    r.d = r.b1[0] * a[3] + r.b1[1] * a[4] + r.b1[2] * a[5];
    float u[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) u[i] = a[3 + i] - r.d * r.b1[i];
    r.n2 = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const float i2 = 1.f / (r.n2 > NORM_EPS ? r.n2 : NORM_EPS);
#pragma unroll
    for (int i = 0; i < 3; ++i) r.b2[i] = u[i] * i2;
    r.b3[0] = r.b1[1] * r.b2[2] - r.b1[2] * r.b2[1];
    r.b3[1] = r.b1[2] * r.b2[0] - r.b1[0] * r.b2[2];
    r.b3[2] = r.b1[0] * r.b2[1] - r.b1[1] * r.b2[0];
}

__device__ __forceinline__ void cross3(const float *a, const float *b, float *c) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}

// G: gradient w.r.t. the rotation matrix rows (b1,b2,b3), row-major 3x3 -> ga[6]
__device__ __forceinline__ void rot6d_bwd(const Rot6 &r, const float *G, float *ga) {
    float gb1[3], gb2[3], t[3];
    cross3(r.b2, G + 6, t);            // b3 = b1 x b2 : g_b1 += b2 x G3
#pragma unroll
    for (int i = 0; i < 3; ++i) gb1[i] = G[i] + t[i];
    cross3(G + 6, r.b1, t);            //                g_b2 += G3 x b1
#pragma unroll
    for (int i = 0; i < 3; ++i) gb2[i] = G[3 + i] + t[i];
    float gu[3];
    if (r.n2 > NORM_EPS) {
        const float dt = r.b2[0] * gb2[0] + r.b2[1] * gb2[1] + r.b2[2] * gb2[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) gu[i] = (gb2[i] - r.b2[i] * dt) / r.n2;
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) gu[i] = gb2[i] / NORM_EPS;
    }
    const float gd = -(gu[0] * r.b1[0] + gu[1] * r.b1[1] + gu[2] * r.b1[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        gb1[i] += -r.d * gu[i] + gd * r.a2[i];
        ga[3 + i] = gu[i] + gd * r.b1[i];
    }
    if (r.n1 > NORM_EPS) {
        const float dt = r.b1[0] * gb1[0] + r.b1[1] * gb1[1] + r.b1[2] * gb1[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) ga[i] = (gb1[i] - r.b1[i] * dt) / r.n1;
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) ga[i] = gb1[i] / NORM_EPS;
    }
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// signed_pow (pytorch.py:31-32) and its derivative w.r.t. the exponent (torch: 0 where the base is 0)
__device__ __forceinline__ float spow(float t, float e, float &dde) {
    const float ab = fabsf(t);
    const float sg = t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f);
    const float pw = powf(ab, e);
    const float r = sg * pw;
    dde = ab == 0.f ? 0.f : r * logf(ab);
    return r;
}

struct Pose {
    float e1, e2, se1, se2;   // exponents and sigmoid(sq_eps)
    float S[3];
    Rot6 rot;
    float T[3];
};

__device__ __forceinline__ void load_pose(const float *sq_eps, const float *S, const float *R6, const float *T, int k,
                                          float scale_min, Pose &p) {
    if (sq_eps) {
        p.se1 = sigmoidf(sq_eps[k * 2]); p.se2 = sigmoidf(sq_eps[k * 2 + 1]);
        p.e1 = p.se1 * 1.8f + 0.1f; p.e2 = p.se2 * 1.8f + 0.1f;
    } else { p.se1 = p.se2 = 0.f; p.e1 = p.e2 = 1.f; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { p.S[i] = S ? expf(S[k * 3 + i]) + scale_min : 1.f; p.T[i] = T[k * 3 + i]; }
    rot6d_fwd(R6 + k * 6, p.rot);
}

// local (block frame) vertex -> world:  ((v*S)@R + T) * S_world @ R_world + T_world   (row-vector convention)
__device__ __forceinline__ void pose_fwd(const Pose &p, const float *v, float S_world, const float *Rw, const float *Tw, float *out) {
    const float s[3] = {v[0] * p.S[0], v[1] * p.S[1], v[2] * p.S[2]};
    float l[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) l[j] = (s[0] * p.rot.b1[j] + s[1] * p.rot.b2[j] + s[2] * p.rot.b3[j] + p.T[j]) * S_world;
#pragma unroll
    for (int j = 0; j < 3; ++j) out[j] = l[0] * Rw[j] + l[1] * Rw[3 + j] + l[2] * Rw[6 + j] + (Tw ? Tw[j] : 0.f);
}

// acc[0..1] d/d(e1,e2) [filled by caller], acc[2..4] d/dS (post exp+min), acc[5..13] d/dR rows, acc[14..16] d/dT; returns d/dv
__device__ __forceinline__ void pose_bwd(const Pose &p, const float *v, float S_world, const float *Rw, const float *g, float *acc, float *gv) {
    float gl[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) gl[i] = (g[0] * Rw[i * 3] + g[1] * Rw[i * 3 + 1] + g[2] * Rw[i * 3 + 2]) * S_world;
    const float s[3] = {v[0] * p.S[0], v[1] * p.S[1], v[2] * p.S[2]};
    const float *rows[3] = {p.rot.b1, p.rot.b2, p.rot.b3};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        acc[14 + i] += gl[i];
        const float gs = gl[0] * rows[i][0] + gl[1] * rows[i][1] + gl[2] * rows[i][2];
        acc[2 + i] += gs * v[i];
        gv[i] = gs * p.S[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[5 + i * 3 + j] += s[i] * gl[j];
    }
}

// turn the 17 raw accumulators of one primitive into parameter gradients (+=)
__device__ __forceinline__ void finish_pose_grads(const Pose &p, const float *S, int k, const float *acc, float *g_sq_eps,
                                                  float *g_S, float *g_R6, float *g_T) {
    if (g_sq_eps) {
        g_sq_eps[k * 2] += acc[0] * 1.8f * p.se1 * (1.f - p.se1);
        g_sq_eps[k * 2 + 1] += acc[1] * 1.8f * p.se2 * (1.f - p.se2);
    }
    if (g_S) {
#pragma unroll
        for (int i = 0; i < 3; ++i) g_S[k * 3 + i] += acc[2 + i] * expf(S[k * 3 + i]);
    }
    float ga[6];
    rot6d_bwd(p.rot, acc + 5, ga);
#pragma unroll
    for (int i = 0; i < 6; ++i) g_R6[k * 6 + i] += ga[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) g_T[k * 3 + i] += acc[14 + i];
}

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
        float d;
        const float A = spow(trig[o], p.e1, d), C = spow(trig[plane + o], p.e1, d);
        const float Bc = spow(trig[2 * plane + o], p.e2, d), Bs = spow(trig[3 * plane + o], p.e2, d);
        const float loc[3] = {A * Bs * ratio, C * ratio, A * Bc * ratio};
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
        float dA, dC, dBc, dBs;
        const float A = spow(trig[o], p.e1, dA), C = spow(trig[plane + o], p.e1, dC);
        const float Bc = spow(trig[2 * plane + o], p.e2, dBc), Bs = spow(trig[3 * plane + o], p.e2, dBs);
        const float loc[3] = {A * Bs * ratio, C * ratio, A * Bc * ratio};
        float gv[3];
        pose_bwd(p, loc, S_world, Rw, gverts + ((long long)ko * nv + v) * 3, acc, gv);
        acc[0] += ratio * (gv[0] * dA * Bs + gv[1] * dC + gv[2] * dA * Bc);
        acc[1] += ratio * (gv[0] * A * dBs + gv[2] * A * dBc);
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

__device__ __forceinline__ float safe_pow_f(float t, float e, float &dt, float &de) {   // clamp(1e-6).pow(e)
    const float c = t < 1e-6f ? 1e-6f : t;
    const float r = powf(c, e);
    dt = t >= 1e-6f ? e * r / c : 0.f;
    de = r * logf(c);
    return r;
}

struct BlockP {
    float e1, e2, sr[3], R[9], T[3], alpha;   // sr = S*ratio
};

__global__ __launch_bounds__(256) void overlap_kernel(const float *u, int npts, const float *sq_eps, const float *S,
                                                      const float *R6, const float *T, const float *alpha, int Kb,
                                                      float ratio, float scale_min, float inv_temp, float thresh,
                                                      float scale_over_P, float *loss, float *ws) {
    __shared__ BlockP s_b[MAX_KB];
    __shared__ float s_g[MAX_KB * NG];
    __shared__ float s_red[4];
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
        for (int i = 0; i < 3; ++i) l[i] = (u[pi * 3 + i] * 2.f - 1.f) * bk.sr[i];
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
                const float x2 = pc[0] * pc[0], y2 = pc[1] * pc[1], z2 = pc[2] * pc[2];
                float dXt, dXe, dYt, dYe, dZt, dZe, dWt, dWe, dQt, dQe;
                const float X = safe_pow_f(x2, 1.f / b.e2, dXt, dXe), Y = safe_pow_f(y2, 1.f / b.e1, dYt, dYe);
                const float Z = safe_pow_f(z2, 1.f / b.e2, dZt, dZe);
                const float Wp = safe_pow_f(X + Z, b.e2 / b.e1, dWt, dWe);
                const float r = Wp + Y;
                const float Q = safe_pow_f(r, b.e1 / 2.f, dQt, dQe);
                const float sdf = Q - 1.f;
                const float occ = sigmoidf(-sdf * inv_temp);
                if (pass == 0) { sum += occ * b.alpha; continue; }
                // backward of scale_over_P * (sum - thresh)
                const float go = scale_over_P;
                float *g = s_g + j * NG;
                atomicAdd(g + 17, go * occ);
                const float gsdf = go * b.alpha * occ * (1.f - occ) * (-inv_temp);
                if (gsdf == 0.f) continue;
                float ge1 = gsdf * dQe * 0.5f, ge2 = 0.f;
                const float gr = gsdf * dQt;
                const float gW = gr, gY = gr;
                const float gex = gW * dWe;                       // exponent e2/e1
                ge2 += gex / b.e1; ge1 += gex * (-b.e2 / (b.e1 * b.e1));
                const float gXZ = gW * dWt;
                ge2 += (gXZ * dXe + gXZ * dZe) * (-1.f / (b.e2 * b.e2));
                ge1 += gY * dYe * (-1.f / (b.e1 * b.e1));
                const float gp[3] = {inr[0] ? gXZ * dXt * 2.f * pc[0] : 0.f, inr[1] ? gY * dYt * 2.f * pc[1] : 0.f,
                                     inr[2] ? gXZ * dZt * 2.f * pc[2] : 0.f};
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

}  // namespace

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
