// Camera transform and near-plane clipping arithmetic shared by the device kernels (hipcc, project_clip.hip) and the host (g++:
// tests/test_host_camera_math.py builds it into a checker-side shared object and holds it, without a GPU, bit for bit to
// oracle/oracle.py::transform_to_ndc + clip_faces, and its backward to autograd).  SURVEY.md A.2, A.4.
#pragma once
#include "raster_math.h"      // DBW_HD, f3

namespace dbw {

struct Cam {
    float R[9], T[3], K[16];
};

DBW_HD void load_cam(const float *R, const float *T, const float *Kmat, int b, Cam &c) {
#pragma unroll
    for (int i = 0; i < 9; ++i) c.R[i] = R[b * 9 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) c.T[i] = T[b * 3 + i];
#pragma unroll
    for (int i = 0; i < 16; ++i) c.K[i] = Kmat[i];
}

struct Proj {
    float vx, vy, vz;   // view space
    float px, py, pw;   // before the perspective divide
    float denom;
    f3 ndc;             // x,y NDC, z = view depth
};

// view depth of a world point: the one quantity near-plane clipping classifies a vertex by (clip_emit_count)
DBW_HD float view_z(const float *X, const Cam &c) { return X[0] * c.R[2] + X[1] * c.R[5] + X[2] * c.R[8] + c.T[2]; }

DBW_HD Proj project(const float *X, const Cam &c, float eps) {
    Proj o;
    const float x = X[0], y = X[1], z = X[2];
    o.vx = x * c.R[0] + y * c.R[3] + z * c.R[6] + c.T[0];
    o.vy = x * c.R[1] + y * c.R[4] + z * c.R[7] + c.T[1];
    o.vz = view_z(X, c);
    o.px = o.vx * c.K[0] + o.vy * c.K[1] + o.vz * c.K[2] + c.K[3];
    o.py = o.vx * c.K[4] + o.vy * c.K[5] + o.vz * c.K[6] + c.K[7];
    o.pw = o.vx * c.K[12] + o.vy * c.K[13] + o.vz * c.K[14] + c.K[15];
    const float sgn = o.pw > 0.f ? 1.f : (o.pw < 0.f ? -1.f : 1.f);
    const float ab = o.pw < 0.f ? -o.pw : o.pw;
    o.denom = sgn * (ab < eps ? eps : ab);
    o.ndc.x = o.px / o.denom;
    o.ndc.y = o.py / o.denom;
    o.ndc.z = o.vz;
    return o;
}

// intersection of segment pa->pb with z = c; w returned
DBW_HD f3 clip_point(f3 pa, f3 pb, float c, int persp, float &w) {
    w = (pa.z - c) / (pa.z - pb.z);
    const float omw = 1.f - w;
    f3 q;
    q.z = pa.z * omw + pb.z * w;
    if (persp) {
        q.x = ((pa.x * pa.z) * omw + (pb.x * pb.z) * w) / c;
        q.y = ((pa.y * pa.z) * omw + (pb.y * pb.z) * w) / c;
    } else {
        q.x = pa.x * omw + pb.x * w;
        q.y = pa.y * omw + pb.y * w;
    }
    return q;
}


// One face after the camera transform -> 0, 1 or 2 triangles: untouched (no vertex behind z = zc), dropped (all three),
// case 3 (two behind; p1 = the vertex in front -> (p4, p5, p1)) or case 4 (one behind; p1 = the vertex behind -> (p4, p2, p5) and
// (p5, p2, p3)).  code = i1 | kind << 2 (i1 = slot of p1 in the original face; kind 0 / 1 / 2 = which clipped triangle), -1 for an
// untouched face; w2, w3 = the (detached) interpolation weights of p4 on p1-p2 and p5 on p1-p3.
struct ClippedFace {
    int emit, code0, code1;
    float w2, w3;
    f3 t0[3], t1[3];
};
DBW_HD void clip_face(const f3 p[3], int zc_on, float zc, int persp, ClippedFace &o) {
    int nbh = 0, behind_mask = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (zc_on && p[i].z < zc) { ++nbh; behind_mask |= 1 << i; }
    o.emit = nbh == 0 ? 1 : (nbh == 3 ? 0 : (nbh == 2 ? 1 : 2));
    o.code0 = o.code1 = -1; o.w2 = o.w3 = 0.f;
    if (nbh == 0) { o.t0[0] = p[0]; o.t0[1] = p[1]; o.t0[2] = p[2]; return; }
    if (nbh == 3) return;
    // vertex roles rotated by i1 with selects (runtime indexing would put p[] in scratch memory on the device)
    const int i1 = nbh == 2 ? ((behind_mask == 6) ? 0 : (behind_mask == 5 ? 1 : 2)) : ((behind_mask == 1) ? 0 : (behind_mask == 2 ? 1 : 2));
    const f3 p1 = i1 == 0 ? p[0] : (i1 == 1 ? p[1] : p[2]), p2 = i1 == 0 ? p[1] : (i1 == 1 ? p[2] : p[0]),
             p3 = i1 == 0 ? p[2] : (i1 == 1 ? p[0] : p[1]);
    const f3 p4 = clip_point(p1, p2, zc, persp, o.w2), p5 = clip_point(p1, p3, zc, persp, o.w3);
    if (nbh == 2) {
        o.t0[0] = p4; o.t0[1] = p5; o.t0[2] = p1;
        o.code0 = i1 | (0 << 2);
    } else {
        o.t0[0] = p4; o.t0[1] = p2; o.t0[2] = p5;
        o.t1[0] = p5; o.t1[1] = p2; o.t1[2] = p3;
        o.code0 = i1 | (1 << 2); o.code1 = i1 | (2 << 2);
    }
}

// number of triangles clip_face emits for a face whose vertices have view depths z0, z1, z2 (its `emit`, without the rest)
DBW_HD int clip_emit_count(float z0, float z1, float z2, int zc_on, float zc) {
    const int nbh = (zc_on && z0 < zc ? 1 : 0) + (zc_on && z1 < zc ? 1 : 0) + (zc_on && z2 < zc ? 1 : 0);
    return nbh == 0 ? 1 : (nbh == 3 ? 0 : (nbh == 2 ? 1 : 2));
}

// d(ndc vertex)/d(world vertex): the world-space gradient of one vertex of one view
DBW_HD f3 vertex_bwd(const float *verts, int vi, const Cam &c, float eps, f3 g) {
    const Proj pr = project(verts + (long long)vi * 3, c, eps);
    const float gpx = g.x / pr.denom, gpy = g.y / pr.denom;
    const float gden = -(g.x * pr.px + g.y * pr.py) / (pr.denom * pr.denom);
    const float ab = pr.pw < 0.f ? -pr.pw : pr.pw;
    const float gpw = ab >= eps ? gden : 0.f;
    const float gvx = gpx * c.K[0] + gpy * c.K[4] + gpw * c.K[12];
    const float gvy = gpx * c.K[1] + gpy * c.K[5] + gpw * c.K[13];
    const float gvz = gpx * c.K[2] + gpy * c.K[6] + gpw * c.K[14] + g.z;
    f3 o;
    o.x = gvx * c.R[0] + gvy * c.R[1] + gvz * c.R[2];
    o.y = gvx * c.R[3] + gvy * c.R[4] + gvz * c.R[5];
    o.z = gvx * c.R[6] + gvy * c.R[7] + gvz * c.R[8];
    return o;
}

// grads of q = clip_point(pa, pb) (w detached) pushed to ga, gb
DBW_HD void clip_point_bwd(f3 pa, f3 pb, float c, int persp, float w, f3 gq, f3 &ga, f3 &gb) {
    const float omw = 1.f - w;
    ga.z += gq.z * omw; gb.z += gq.z * w;
    if (persp) {
        ga.x += gq.x * pa.z * omw / c; ga.y += gq.y * pa.z * omw / c;
        ga.z += (gq.x * pa.x + gq.y * pa.y) * omw / c;
        gb.x += gq.x * pb.z * w / c; gb.y += gq.y * pb.z * w / c;
        gb.z += (gq.x * pb.x + gq.y * pb.y) * w / c;
    } else {
        ga.x += gq.x * omw; ga.y += gq.y * omw;
        gb.x += gq.x * w; gb.y += gq.y * w;
    }
}


}  // namespace dbw
