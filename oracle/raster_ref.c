/*
 * oracle/raster_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the triangle rasteriser that sits below the
 * reference's rendering call (src/model/renderer.py:53-54,92-94 ->
 * pytorch3d.renderer.MeshRasterizer -> rasterize_meshes, naive CPU path).
 *
 * The algorithm lives in a third-party dependency that is ABSENT from
 * /root/reference and from this image: pytorch3d==0.7.1 (environment.yml:25).
 * This file restates its published algorithm (naive per-pixel x per-face loop,
 * CPU tie-break rule) as recorded in SURVEY.md Appendix A.1, A.5, A.6.
 * PARITY UNPINNED: there is no PyTorch3D binary/source here and the reference
 * holds no test vectors for this boundary, so this restatement cannot be
 * checked against PyTorch3D itself; it is the canonical definition the HIP
 * kernels are held to (bit-exact face indices, fp32 arithmetic, no FMA
 * contraction -- build with -ffp-contract=off).
 *
 * Deliberate choices where SURVEY.md's from-memory notes are ambiguous:
 *  - kEpsilon: PyTorch3D declares `const auto kEpsilon = 1e-8;` (geometry_utils.h and .cuh alike), i.e. a DOUBLE.  Where it is an
 *    operand of arithmetic -- `area = EdgeFunctionForward(v2, v0, v1) + kEpsilon` in BarycentricCoordinates{Forward,Backward} --
 *    the float edge function is promoted, the sum is taken in double and rounded to float once; that is what AREA_EPS() does.
 *    Where it is only compared (`zmin < kEpsilon`, `face_area <= kEpsilon`, `l2 <= kEpsilon`) or passed through std::max<T> /
 *    fmaxf (the perspective denominator), float and double readings decide identically except for a value that equals 1e-8f
 *    to the bit, so those stay in `real`.  Rounds 1-2 added 1e-8f in float; -DKEPS_FLOAT keeps that build for
 *    tests/test_oracle_golden.py, which counts how many output slots the two readings move: none on the tiny scenes, none among the
 *    670 917 occupied slots of three config-2 views (the sums differ by 6e-17, far below an ulp of a block face's area), 106 of
 *    35 199 barycentric values on slivers whose area is within two decades of kEpsilon.
 *  - BarycentricClipForward clamps the LOWER bound only (max(b,0)) and
 *    renormalises by max(sum,1e-5): that is what pytorch3d>=0.3 ships
 *    ("Only clamp negative values to 0.0"); SURVEY A.5 step 5 wrote clamp(0,1).
 *  - the per-pixel list is kept sorted by the tuple (pz, face_idx, ...) after
 *    EVERY insertion or neighbour replacement (std::sort of tuples on CPU).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 *
 * Compiled three times: REAL=float (the oracle proper), REAL=double (used to
 * validate the hand-derived backward by finite differences) and REAL=float with
 * -DKEPS_FLOAT (suffix f32e: the float reading of kEpsilon, see above).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL float
#endif
#ifndef SUFFIX
#define SUFFIX f32
#endif
#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

typedef REAL real;

#define K_EPS ((real)1e-8)
#ifdef KEPS_FLOAT
#define AREA_EPS(e) ((e) + K_EPS)
#else
#define AREA_EPS(e) ((real)((double)(e) + 1e-8))
#endif
#define MAX_K 64

typedef struct { real x, y; } v2;
typedef struct { real x, y, z; } v3;

/* NonSquarePixToNdc (rasterization_utils: "First multiply S1 by
 * float range so that division results in a float value"). */
static real pix_to_ndc(int i, int S1, int S2) {
    real range = (real)2.0;
    if (S1 > S2) range = ((real)S1 * range) / (real)S2;
    const real offset = range / (real)2.0;
    return -offset + (range * (real)i + offset) / (real)S1;
}

/* EdgeFn(p; a, b), SURVEY A.5 step 1 */
static real edge_fn(v2 p, v2 a, v2 b) {
    return (p.x - a.x) * (b.y - a.y) - (p.y - a.y) * (b.x - a.x);
}

/* grads of edge_fn wrt (p, a, b), scaled by g */
static void edge_fn_bwd(v2 p, v2 a, v2 b, real g, v2 *gp, v2 *ga, v2 *gb) {
    gp->x = g * (b.y - a.y); gp->y = g * (a.x - b.x);
    ga->x = g * (p.y - b.y); ga->y = g * (b.x - p.x);
    gb->x = g * (a.y - p.y); gb->y = g * (p.x - a.x);
}

/* SURVEY A.5 step 3 */
static v3 bary_fwd(v2 p, v2 v0, v2 v1, v2 v2_) {
    const real area = AREA_EPS(edge_fn(v2_, v0, v1));
    v3 w;
    w.x = edge_fn(p, v1, v2_) / area;
    w.y = edge_fn(p, v2_, v0) / area;
    w.z = edge_fn(p, v0, v1) / area;
    return w;
}

static void bary_bwd(v2 p, v2 v0, v2 v1, v2 v2_, v3 g, v2 *g0, v2 *g1, v2 *g2) {
    const real area = AREA_EPS(edge_fn(v2_, v0, v1));
    const real area2 = area * area;
    const real area_inv = (real)1.0 / area;
    const real e0 = edge_fn(p, v1, v2_);
    const real e1 = edge_fn(p, v2_, v0);
    const real e2 = edge_fn(p, v0, v1);
    v2 gp, ga, gb, hp, ha, hb;
    g0->x = g0->y = g1->x = g1->y = g2->x = g2->y = 0;
    /* w0 = e0(p; v1, v2) / area(v2; v0, v1) */
    edge_fn_bwd(p, v1, v2_, g.x * area_inv, &gp, &ga, &gb);
    edge_fn_bwd(v2_, v0, v1, g.x * (-e0 / area2), &hp, &ha, &hb);
    g0->x += ha.x;        g0->y += ha.y;
    g1->x += ga.x + hb.x; g1->y += ga.y + hb.y;
    g2->x += gb.x + hp.x; g2->y += gb.y + hp.y;
    /* w1 = e1(p; v2, v0) / area */
    edge_fn_bwd(p, v2_, v0, g.y * area_inv, &gp, &ga, &gb);
    edge_fn_bwd(v2_, v0, v1, g.y * (-e1 / area2), &hp, &ha, &hb);
    g0->x += gb.x + ha.x; g0->y += gb.y + ha.y;
    g1->x += hb.x;        g1->y += hb.y;
    g2->x += ga.x + hp.x; g2->y += ga.y + hp.y;
    /* w2 = e2(p; v0, v1) / area */
    edge_fn_bwd(p, v0, v1, g.z * area_inv, &gp, &ga, &gb);
    edge_fn_bwd(v2_, v0, v1, g.z * (-e2 / area2), &hp, &ha, &hb);
    g0->x += ga.x + ha.x; g0->y += ga.y + ha.y;
    g1->x += gb.x + hb.x; g1->y += gb.y + hb.y;
    g2->x += hp.x;        g2->y += hp.y;
}

/* SURVEY A.5 step 4 */
static v3 persp_fwd(v3 b, real z0, real z1, real z2) {
    const real t0 = b.x * z1 * z2;
    const real t1 = z0 * b.y * z2;
    const real t2 = z0 * z1 * b.z;
    real denom = t0 + t1 + t2;
    if (!(denom > K_EPS)) denom = K_EPS; /* max(denom, eps) */
    v3 w; w.x = t0 / denom; w.y = t1 / denom; w.z = t2 / denom;
    return w;
}

static v3 persp_bwd(v3 b, real z0, real z1, real z2, v3 g, real *gz0, real *gz1, real *gz2) {
    const real t0 = b.x * z1 * z2;
    const real t1 = z0 * b.y * z2;
    const real t2 = z0 * z1 * b.z;
    real denom = t0 + t1 + t2;
    if (!(denom > K_EPS)) denom = K_EPS;
    const real g_denom_top = -t0 * g.x - t1 * g.y - t2 * g.z;
    const real g_denom = g_denom_top / (denom * denom);
    const real gt0 = g_denom + g.x / denom;
    const real gt1 = g_denom + g.y / denom;
    const real gt2 = g_denom + g.z / denom;
    v3 gb; gb.x = gt0 * z1 * z2; gb.y = gt1 * z0 * z2; gb.z = gt2 * z0 * z1;
    *gz0 = gt1 * b.y * z2 + gt2 * b.z * z1;
    *gz1 = gt0 * b.x * z2 + gt2 * b.z * z0;
    *gz2 = gt0 * b.x * z1 + gt1 * b.y * z0;
    return gb;
}

/* SURVEY A.5 step 5 (lower clamp only, see header) */
static v3 clip_fwd(v3 b) {
    v3 w;
    w.x = b.x > 0 ? b.x : (real)0; w.y = b.y > 0 ? b.y : (real)0; w.z = b.z > 0 ? b.z : (real)0;
    real s = w.x + w.y + w.z;
    if (!(s > (real)1e-5)) s = (real)1e-5;
    w.x /= s; w.y /= s; w.z /= s;
    return w;
}

static v3 clip_bwd(v3 b, v3 g) {
    v3 w;
    w.x = b.x > 0 ? b.x : (real)0; w.y = b.y > 0 ? b.y : (real)0; w.z = b.z > 0 ? b.z : (real)0;
    real s = w.x + w.y + w.z;
    real gsc = 1;
    if (s < (real)1e-5) { gsc = 0; s = (real)1e-5; }
    const real cx = b.x < 0 ? (real)0 : (real)1, cy = b.y < 0 ? (real)0 : (real)1, cz = b.z < 0 ? (real)0 : (real)1;
    const real s2 = s * s;
    const real gsx = -w.x / s2 * gsc, gsy = -w.y / s2 * gsc, gsz = -w.z / s2 * gsc;
    const real common = g.x * gsx + g.y * gsy + g.z * gsz;
    v3 o;
    o.x = cx * (g.x / s + common);
    o.y = cy * (g.y / s + common);
    o.z = cz * (g.z / s + common);
    return o;
}

/* SURVEY A.5 step 7 */
static real point_line_dist(v2 p, v2 a, v2 b) {
    const real dx = b.x - a.x, dy = b.y - a.y;
    const real l2 = dx * dx + dy * dy;
    if (l2 <= K_EPS) return (p.x - b.x) * (p.x - b.x) + (p.y - b.y) * (p.y - b.y);
    const real t = (dx * (p.x - a.x) + dy * (p.y - a.y)) / l2;
    const real tt = t < 0 ? (real)0 : (t > 1 ? (real)1 : t);
    const real qx = a.x + tt * dx, qy = a.y + tt * dy;
    return (p.x - qx) * (p.x - qx) + (p.y - qy) * (p.y - qy);
}

static void point_line_dist_bwd(v2 p, v2 a, v2 b, real g, v2 *ga, v2 *gb) {
    const real dx = b.x - a.x, dy = b.y - a.y;
    const real t_bot = dx * dx + dy * dy;
    const real t_top = dx * (p.x - a.x) + dy * (p.y - a.y);
    const real t = t_top / t_bot;
    const real tt = t < 0 ? (real)0 : (t > 1 ? (real)1 : t);
    const real qx = ((real)1 - tt) * a.x + tt * b.x, qy = ((real)1 - tt) * a.y + tt * b.y;
    ga->x = g * ((real)1 - tt) * (real)2 * (qx - p.x); ga->y = g * ((real)1 - tt) * (real)2 * (qy - p.y);
    gb->x = g * tt * (real)2 * (qx - p.x);             gb->y = g * tt * (real)2 * (qy - p.y);
}

static real point_tri_dist(v2 p, v2 v0, v2 v1, v2 v2_) {
    const real e01 = point_line_dist(p, v0, v1);
    const real e02 = point_line_dist(p, v0, v2_);
    const real e12 = point_line_dist(p, v1, v2_);
    const real m = e01 < e02 ? e01 : e02;
    return m < e12 ? m : e12;
}

static void point_tri_dist_bwd(v2 p, v2 v0, v2 v1, v2 v2_, real g, v2 *g0, v2 *g1, v2 *g2) {
    const real e01 = point_line_dist(p, v0, v1);
    const real e02 = point_line_dist(p, v0, v2_);
    const real e12 = point_line_dist(p, v1, v2_);
    g0->x = g0->y = g1->x = g1->y = g2->x = g2->y = 0;
    if (e01 <= e02 && e01 <= e12) point_line_dist_bwd(p, v0, v1, g, g0, g1);
    else if (e02 <= e01 && e02 <= e12) point_line_dist_bwd(p, v0, v2_, g, g0, g2);
    else if (e12 <= e01 && e12 <= e02) point_line_dist_bwd(p, v1, v2_, g, g1, g2);
}

typedef struct { real pz; int64_t f; real dist, b0, b1, b2; } frag;

/* tuple order (pz, f, dist, b0, b1, b2): f is unique so (pz, f) decides */
static int frag_less(const frag *a, const frag *b) {
    if (a->pz < b->pz) return 1;
    if (b->pz < a->pz) return 0;
    return a->f < b->f;
}

static void frag_sort(frag *q, int n) { /* insertion sort == std::sort result for a total order */
    for (int i = 1; i < n; ++i) {
        frag t = q[i]; int j = i - 1;
        while (j >= 0 && frag_less(&t, &q[j])) { q[j + 1] = q[j]; --j; }
        q[j + 1] = t;
    }
}

/*
 * Forward. face_verts (F,3,3): x,y in NDC, z = view-space depth.
 * Outputs (N,H,W,K[,3]) pre-filled with -1 by this function.
 */
void FN(dbw_ref_rasterize_fwd)(const real *face_verts, const int64_t *first_idx, const int64_t *num_faces,
                               const int64_t *neighbor_idx, int N, int H, int W, int K, real blur_radius,
                               int perspective_correct, int clip_barycentric, int cull_backfaces,
                               int64_t *pix_to_face, real *zbuf, real *bary, real *dists, int n_threads) {
    const int64_t total = (int64_t)N * H * W * K;
    for (int64_t i = 0; i < total; ++i) { pix_to_face[i] = -1; zbuf[i] = -1; dists[i] = -1; }
    for (int64_t i = 0; i < total * 3; ++i) bary[i] = -1;
    const real margin = (real)sqrt((double)blur_radius);
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(dynamic, 4) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int n = 0; n < N; ++n) {
        for (int yi = 0; yi < H; ++yi) {
            const int64_t f0 = first_idx[n], f1 = f0 + num_faces[n];
            const real yf = pix_to_ndc(H - 1 - yi, H, W);
            frag q[MAX_K + 1];
            for (int xi = 0; xi < W; ++xi) {
                const real xf = pix_to_ndc(W - 1 - xi, W, H);
                v2 p; p.x = xf; p.y = yf;
                int qn = 0;
                for (int64_t f = f0; f < f1; ++f) {
                    const real *fv = face_verts + f * 9;
                    v2 a, b, c; a.x = fv[0]; a.y = fv[1]; b.x = fv[3]; b.y = fv[4]; c.x = fv[6]; c.y = fv[7];
                    const real z0 = fv[2], z1 = fv[5], z2 = fv[8];
                    real xmin = a.x < b.x ? a.x : b.x; xmin = xmin < c.x ? xmin : c.x;
                    real xmax = a.x > b.x ? a.x : b.x; xmax = xmax > c.x ? xmax : c.x;
                    real ymin = a.y < b.y ? a.y : b.y; ymin = ymin < c.y ? ymin : c.y;
                    real ymax = a.y > b.y ? a.y : b.y; ymax = ymax > c.y ? ymax : c.y;
                    real zmin = z0 < z1 ? z0 : z1; zmin = zmin < z2 ? zmin : z2;
                    const int outside = (xf < xmin - margin) || (xf > xmax + margin) ||
                                        (yf < ymin - margin) || (yf > ymax + margin);
                    if (outside || zmin < K_EPS) continue;
                    const real face_area = edge_fn(a, b, c);
                    if (face_area <= K_EPS && face_area >= -K_EPS) continue;
                    if (cull_backfaces && face_area < 0) continue;
                    const v3 bary0 = bary_fwd(p, a, b, c);
                    const v3 bp = perspective_correct ? persp_fwd(bary0, z0, z1, z2) : bary0;
                    const v3 bc = clip_barycentric ? clip_fwd(bp) : bp;
                    const real pz = bc.x * z0 + bc.y * z1 + bc.z * z2;
                    if (pz < 0) continue;
                    const real dist = point_tri_dist(p, a, b, c);
                    const int inside = bp.x > 0 && bp.y > 0 && bp.z > 0;
                    const real sdist = inside ? -dist : dist;
                    if (!inside && dist >= blur_radius) continue;
                    frag t; t.pz = pz; t.f = f; t.dist = sdist; t.b0 = bc.x; t.b1 = bc.y; t.b2 = bc.z;
                    const int64_t nb = neighbor_idx ? neighbor_idx[f] : -1;
                    int at = -1;
                    if (nb != -1) for (int i = 0; i < qn; ++i) if (q[i].f == nb) { at = i; break; }
                    if (at != -1) {
                        const real nd = q[at].dist < 0 ? -q[at].dist : q[at].dist;
                        if (dist < nd) q[at] = t;
                    } else {
                        q[qn++] = t;
                    }
                    frag_sort(q, qn);
                    if (qn > K) qn = K;
                }
                const int64_t base = (((int64_t)n * H + yi) * W + xi) * K;
                for (int i = 0; i < qn; ++i) {
                    pix_to_face[base + i] = q[i].f; zbuf[base + i] = q[i].pz; dists[base + i] = q[i].dist;
                    bary[(base + i) * 3 + 0] = q[i].b0; bary[(base + i) * 3 + 1] = q[i].b1; bary[(base + i) * 3 + 2] = q[i].b2;
                }
            }
        }
    }
}

/* Backward, SURVEY A.6. grad_face_verts (F,3,3) must be zeroed by the caller. Single-threaded
 * (deterministic accumulation order: n, y, x, k). */
void FN(dbw_ref_rasterize_bwd)(const real *face_verts, const int64_t *pix_to_face, const real *grad_zbuf,
                               const real *grad_bary, const real *grad_dists, int N, int H, int W, int K,
                               int perspective_correct, int clip_barycentric, real *grad_face_verts) {
    for (int n = 0; n < N; ++n)
        for (int yi = 0; yi < H; ++yi) {
            const real yf = pix_to_ndc(H - 1 - yi, H, W);
            for (int xi = 0; xi < W; ++xi) {
                const real xf = pix_to_ndc(W - 1 - xi, W, H);
                v2 p; p.x = xf; p.y = yf;
                for (int k = 0; k < K; ++k) {
                    const int64_t o = (((int64_t)n * H + yi) * W + xi) * K + k;
                    const int64_t f = pix_to_face[o];
                    if (f < 0) continue;
                    const real *fv = face_verts + f * 9;
                    v2 a, b, c; a.x = fv[0]; a.y = fv[1]; b.x = fv[3]; b.y = fv[4]; c.x = fv[6]; c.y = fv[7];
                    const real z0 = fv[2], z1 = fv[5], z2 = fv[8];
                    const real gd = grad_dists[o], gz = grad_zbuf[o];
                    v3 gb; gb.x = grad_bary[o * 3]; gb.y = grad_bary[o * 3 + 1]; gb.z = grad_bary[o * 3 + 2];
                    const v3 bary0 = bary_fwd(p, a, b, c);
                    const v3 bp = perspective_correct ? persp_fwd(bary0, z0, z1, z2) : bary0;
                    const v3 bc = clip_barycentric ? clip_fwd(bp) : bp;
                    const int inside = bp.x > 0 && bp.y > 0 && bp.z > 0;
                    const real sign = inside ? (real)-1 : (real)1;
                    v2 d0, d1, d2;
                    point_tri_dist_bwd(p, a, b, c, sign * gd, &d0, &d1, &d2);
                    v3 g; g.x = gb.x + gz * z0; g.y = gb.y + gz * z1; g.z = gb.z + gz * z2;
                    if (clip_barycentric) g = clip_bwd(bp, g);
                    real pz0 = 0, pz1 = 0, pz2 = 0;
                    if (perspective_correct) g = persp_bwd(bary0, z0, z1, z2, g, &pz0, &pz1, &pz2);
                    v2 b0g, b1g, b2g;
                    bary_bwd(p, a, b, c, g, &b0g, &b1g, &b2g);
                    real *o9 = grad_face_verts + f * 9;
                    o9[0] += b0g.x + d0.x; o9[1] += b0g.y + d0.y; o9[2] += gz * bc.x + pz0;
                    o9[3] += b1g.x + d1.x; o9[4] += b1g.y + d1.y; o9[5] += gz * bc.y + pz1;
                    o9[6] += b2g.x + d2.x; o9[7] += b2g.y + d2.y; o9[8] += gz * bc.z + pz2;
                }
            }
        }
}

/* Differentiable-in-double evaluation of (zbuf, bary, dist) for ONE (pixel, face) pair with the
 * selection frozen -- used by tests to finite-difference the backward above. */
void FN(dbw_ref_eval_pair)(const real *fv, real xf, real yf, int perspective_correct, int clip_barycentric,
                           real *out5 /* pz, b0, b1, b2, signed dist */) {
    v2 p, a, b, c; p.x = xf; p.y = yf;
    a.x = fv[0]; a.y = fv[1]; b.x = fv[3]; b.y = fv[4]; c.x = fv[6]; c.y = fv[7];
    const real z0 = fv[2], z1 = fv[5], z2 = fv[8];
    const v3 bary0 = bary_fwd(p, a, b, c);
    const v3 bp = perspective_correct ? persp_fwd(bary0, z0, z1, z2) : bary0;
    const v3 bc = clip_barycentric ? clip_fwd(bp) : bp;
    const int inside = bp.x > 0 && bp.y > 0 && bp.z > 0;
    const real dist = point_tri_dist(p, a, b, c);
    out5[0] = bc.x * z0 + bc.y * z1 + bc.z * z2;
    out5[1] = bc.x; out5[2] = bc.y; out5[3] = bc.z;
    out5[4] = inside ? -dist : dist;
}

real FN(dbw_ref_pix_to_ndc)(int i, int S1, int S2) { return pix_to_ndc(i, S1, S2); }
