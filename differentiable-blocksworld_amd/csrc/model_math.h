// Model-side arithmetic shared by the device kernels (hipcc, model_ops.hip) and the host (g++: tests/test_host_model_math.py builds
// it into a checker-side shared object and compares it, without a GPU, with golden vectors of the reference's own `parametric_sq`,
// `implicit_sq`, `safe_pow`, `signed_pow` and with the oracle's rotation_6d_to_matrix): 6D rotation, superquadric surface points,
// posing, and the implicit superquadric distance of the overlap term -- each with its hand-derived backward.
#pragma once
#include "raster_math.h"      // DBW_HD

namespace dbw {

constexpr float NORM_EPS = 1e-12f;   // F.normalize eps

struct Rot6 {
    float b1[3], b2[3], b3[3], a2[3];
    float n1, n2, d;
};

DBW_HD void rot6d_fwd(const float *a, Rot6 &r) {
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

DBW_HD void cross3(const float *a, const float *b, float *c) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}

// G: gradient w.r.t. the rotation matrix rows (b1,b2,b3), row-major 3x3 -> ga[6]
DBW_HD void rot6d_bwd(const Rot6 &r, const float *G, float *ga) {
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

DBW_HD float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// signed_pow (pytorch.py:31-32) and its derivative w.r.t. the exponent (torch: 0 where the base is 0)
DBW_HD float spow(float t, float e, float &dde) {
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

DBW_HD void load_pose(const float *sq_eps, const float *S, const float *R6, const float *T, int k,
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
DBW_HD void pose_fwd(const Pose &p, const float *v, float S_world, const float *Rw, const float *Tw, float *out) {
    const float s[3] = {v[0] * p.S[0], v[1] * p.S[1], v[2] * p.S[2]};
    float l[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) l[j] = (s[0] * p.rot.b1[j] + s[1] * p.rot.b2[j] + s[2] * p.rot.b3[j] + p.T[j]) * S_world;
#pragma unroll
    for (int j = 0; j < 3; ++j) out[j] = l[0] * Rw[j] + l[1] * Rw[3 + j] + l[2] * Rw[6 + j] + (Tw ? Tw[j] : 0.f);
}

// acc[0..1] d/d(e1,e2) [filled by caller], acc[2..4] d/dS (post exp+min), acc[5..13] d/dR rows, acc[14..16] d/dT; returns d/dv
DBW_HD void pose_bwd(const Pose &p, const float *v, float S_world, const float *Rw, const float *g, float *acc, float *gv) {
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
DBW_HD void finish_pose_grads(const Pose &p, const float *S, int k, const float *acc, float *g_sq_eps,
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


// superquadric.py:10-14 for one vertex, from the cosines / sines of its angles (tabulated once: the angles are constants):
// loc = (A Bs, C, A Bc) * ratio with A = spow(cos eta, e1), C = spow(sin eta, e1), Bc = spow(cos omega, e2), Bs = spow(sin omega, e2);
// de1, de2 = d loc / d e1, d loc / d e2
DBW_HD void parametric_sq_point(float cos_eta, float sin_eta, float cos_omega, float sin_omega, float e1, float e2, float ratio,
                                float loc[3], float de1[3], float de2[3]) {
    float dA, dC, dBc, dBs;
    const float A = spow(cos_eta, e1, dA), C = spow(sin_eta, e1, dC);
    const float Bc = spow(cos_omega, e2, dBc), Bs = spow(sin_omega, e2, dBs);
    loc[0] = A * Bs * ratio; loc[1] = C * ratio; loc[2] = A * Bc * ratio;
    de1[0] = ratio * (dA * Bs); de1[1] = ratio * dC; de1[2] = ratio * (dA * Bc);
    de2[0] = ratio * (A * dBs); de2[1] = 0.f; de2[2] = ratio * (A * dBc);
}

DBW_HD float safe_pow_f(float t, float e, float &dt, float &de) {   // clamp(1e-6).pow(e)
    const float c = t < 1e-6f ? 1e-6f : t;
    const float r = powf(c, e);
    dt = t >= 1e-6f ? e * r / c : 0.f;
    de = r * logf(c);
    return r;
}


// superquadric.py:17-38 with safe=True, as_sdf=2, on a point already clamped to [-5, 5]^3 (the caller masks the gradient of clamped
// coordinates): ((x^2)^(1/e2) + (z^2)^(1/e2))^(e2/e1) + (y^2)^(1/e1)) ^ (e1/2) - 1, every power through safe_pow
struct ImplicitSq { float dXt, dXe, dYt, dYe, dZt, dZe, dWt, dWe, dQt, dQe; };
DBW_HD float implicit_sq_sdf2(const float pc[3], float e1, float e2, ImplicitSq &m) {
    const float x2 = pc[0] * pc[0], y2 = pc[1] * pc[1], z2 = pc[2] * pc[2];
    const float X = safe_pow_f(x2, 1.f / e2, m.dXt, m.dXe), Y = safe_pow_f(y2, 1.f / e1, m.dYt, m.dYe);
    const float Z = safe_pow_f(z2, 1.f / e2, m.dZt, m.dZe);
    const float Wp = safe_pow_f(X + Z, e2 / e1, m.dWt, m.dWe);
    const float r = Wp + Y;
    const float Q = safe_pow_f(r, e1 / 2.f, m.dQt, m.dQe);
    return Q - 1.f;
}
// gsdf = d loss / d sdf -> d loss / d e1, d e2, d pc (the latter before the clamp mask)
DBW_HD void implicit_sq_sdf2_bwd(const float pc[3], float e1, float e2, const ImplicitSq &m, float gsdf, float &ge1, float &ge2, float gpc[3]) {
    ge1 = gsdf * m.dQe * 0.5f; ge2 = 0.f;
    const float gr = gsdf * m.dQt;
    const float gW = gr, gY = gr;
    const float gex = gW * m.dWe;                       // exponent e2/e1
    ge2 += gex / e1; ge1 += gex * (-e2 / (e1 * e1));
    const float gXZ = gW * m.dWt;
    ge2 += (gXZ * m.dXe + gXZ * m.dZe) * (-1.f / (e2 * e2));
    ge1 += gY * m.dYe * (-1.f / (e1 * e1));
    gpc[0] = gXZ * m.dXt * 2.f * pc[0];
    gpc[1] = gY * m.dYt * 2.f * pc[1];
    gpc[2] = gXZ * m.dZt * 2.f * pc[2];
}

}  // namespace dbw
