// Checker-side build of the product's model arithmetic (differentiable-blocksworld_amd/csrc/model_math.h, the header model_ops.hip
// compiles) for the host: tests/test_host_model_math.py compares it, without a GPU, with golden vectors of the reference's own
// parametric_sq / implicit_sq / safe_pow / signed_pow and with torch autograd of the oracle's posing.  Test infrastructure only.
#include "../differentiable-blocksworld_amd/csrc/model_math.h"
#include "../differentiable-blocksworld_amd/csrc/rng_math.h"

using namespace dbw;

extern "C" {

int host_parametric(const float *cos_eta, const float *sin_eta, const float *cos_om, const float *sin_om, int n, float e1, float e2,
                    float ratio, float *loc, float *de1, float *de2) {
    for (int i = 0; i < n; ++i)
        parametric_sq_point(cos_eta[i], sin_eta[i], cos_om[i], sin_om[i], e1, e2, ratio, loc + 3 * i, de1 + 3 * i, de2 + 3 * i);
    return 0;
}

// sdf of n points for one (e1, e2) + backward of sum(sdf * w): gpts (n,3), ge (n,2) = every point's contribution to d / d (e1, e2)
int host_implicit(const float *pts, int n, float e1, float e2, const float *w, float *sdf, float *gpts, float *ge) {
    for (int i = 0; i < n; ++i) {
        float pc[3];
        bool inr[3];
        for (int c = 0; c < 3; ++c) {
            const float v = pts[3 * i + c];
            inr[c] = v >= -5.f && v <= 5.f;
            pc[c] = v < -5.f ? -5.f : (v > 5.f ? 5.f : v);
        }
        ImplicitSq m;
        sdf[i] = implicit_sq_sdf2(pc, e1, e2, m);
        float g1, g2, gpc[3];
        implicit_sq_sdf2_bwd(pc, e1, e2, m, w[i], g1, g2, gpc);
        ge[2 * i] = g1; ge[2 * i + 1] = g2;
        for (int c = 0; c < 3; ++c) gpts[3 * i + c] = inr[c] ? gpc[c] : 0.f;
    }
    return 0;
}

int host_pows(const float *t, int n, float e, float *spow_out, float *safe_out, float *safe_dt) {
    for (int i = 0; i < n; ++i) {
        float d, de;
        spow_out[i] = spow(t[i], e, d);
        safe_out[i] = safe_pow_f(t[i], e, safe_dt[i], de);
    }
    return 0;
}

// 6D -> rotation rows (b1, b2, b3) and the backward of sum(R * G)
int host_rot6d(const float *a6, const float *G9, float *R9, float *ga6) {
    Rot6 r;
    rot6d_fwd(a6, r);
    for (int i = 0; i < 3; ++i) { R9[i] = r.b1[i]; R9[3 + i] = r.b2[i]; R9[6 + i] = r.b3[i]; }
    rot6d_bwd(r, G9, ga6);
    return 0;
}

// posing of n local vertices of ONE primitive: world = ((v * S) @ R + T) * S_world @ R_world + T_world; backward of sum(world * g):
// g_S (w.r.t. the raw S parameter), g_R6, g_T, g_v
int host_pose(const float *S_raw, const float *R6, const float *T, float scale_min, const float *v, int n, float S_world, const float *Rw,
              const float *Tw, const float *g, float *world, float *g_S, float *g_R6, float *g_T, float *g_v) {
    Pose p;
    load_pose(nullptr, S_raw, R6, T, 0, scale_min, p);
    float acc[17];
    for (int i = 0; i < 17; ++i) acc[i] = 0.f;
    for (int i = 0; i < n; ++i) {
        pose_fwd(p, v + 3 * i, S_world, Rw, Tw, world + 3 * i);
        pose_bwd(p, v + 3 * i, S_world, Rw, g + 3 * i, acc, g_v + 3 * i);
    }
    for (int i = 0; i < 3; ++i) g_S[i] = g_T[i] = 0.f;
    for (int i = 0; i < 6; ++i) g_R6[i] = 0.f;
    finish_pose_grads(p, S_raw, 0, acc, nullptr, g_S, g_R6, g_T);
    return 0;
}


// rng_math.h: the raw Philox4x32-10 block, and the step's draws (stream 0: opacity noise, stream 1: overlap samples)
int host_philox(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned *out4) {
    const Philox4 r = philox4x32_10(c0, c1, c2, c3, k0, k1);
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
    return 0;
}
int host_step_noise(unsigned long long seed, unsigned long long step, int n, float *out) {
    for (int k = 0; k < n; ++k) { const Philox4 r = step_random(seed, step, 0u, (unsigned)k); out[k] = normal01(r.x, r.y); }
    return 0;
}
int host_step_uniform(unsigned long long seed, unsigned long long step, int n, float *out3) {
    for (int i = 0; i < n; ++i) {
        const Philox4 r = step_random(seed, step, 1u, (unsigned)i);
        out3[3 * i] = uniform01(r.x); out3[3 * i + 1] = uniform01(r.y); out3[3 * i + 2] = uniform01(r.z);
    }
    return 0;
}
}  // extern "C"
