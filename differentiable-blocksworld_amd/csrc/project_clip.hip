// Camera transform + near-plane clipping of one scene seen from B cameras, fully on device (no host sync).
//
// Replaces pytorch3d 0.7.1 MeshRasterizer.transform + clip_faces (+ the bookkeeping that
// convert_clipped_rasterization_to_original_faces needs), reached by the reference through
// src/model/renderer.py:92-94 (R,T,eps kwargs) with z_clip from renderer.py:35,46.  SURVEY.md A.2, A.4.
// The clipped face_verts are bit-exact to oracle/oracle.py::transform_to_ndc + clip_faces (same fp32 op order).
//
// One workgroup per view: faces are classified, an ordered block-wide scan (two ballots per wave: a face emits 0, 1
// or 2 triangles) assigns output slots in original face order, and each thread writes its own triangles.
#include "dbw_common.h"
#include "camera_math.h"
#include "shade_common.h"
#include "raster_bin.h"
#include "step_kernels.h"
#include "texture_body.h"
#include "../../include/dbw_hip.h"

using namespace dbw;

namespace {

constexpr int NT = 256;

__device__ __forceinline__ void store_tri(float *dst, f3 a, f3 b, f3 c) {
    dst[0] = a.x; dst[1] = a.y; dst[2] = a.z;
    dst[3] = b.x; dst[4] = b.y; dst[5] = b.z;
    dst[6] = c.x; dst[7] = c.y; dst[8] = c.z;
}

__global__ __launch_bounds__(NT) void project_clip_fwd_kernel(
    const float *__restrict__ verts, const int *__restrict__ faces, const float *__restrict__ R,
    const float *__restrict__ T, const float *__restrict__ Kmat, int V, int F, float eps, int zc_on, float zc,
    int persp, float *__restrict__ fvc, int *__restrict__ first_idx, int *__restrict__ num_faces,
    int *__restrict__ c2o, int *__restrict__ neighbor, int *__restrict__ code, float *__restrict__ cw) {
    __shared__ int s_wcnt[NT / DBW_WAVE];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    Cam cam;
    load_cam(R, T, Kmat, b, cam);
    const long long base_out = (long long)b * 2 * F;
    int running = 0;
    for (int f0 = 0; f0 < F; f0 += NT) {
        const int f = f0 + tid;
        ClippedFace cf;
        cf.emit = 0;
        if (f < F) {
            f3 p[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) p[i] = project(verts + (long long)faces[f * 3 + i] * 3, cam, eps).ndc;
            clip_face(p, zc_on, zc, persp, cf);
        }
        const int emit = cf.emit;
        const unsigned long long m1 = __ballot(emit >= 1), m2 = __ballot(emit == 2);
        const unsigned long long lower = (1ull << lane) - 1ull;
        const int prefix = __popcll(m1 & lower) + __popcll(m2 & lower);
        if (lane == 0) s_wcnt[wv] = __popcll(m1) + __popcll(m2);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NT / DBW_WAVE; ++w) { const int c = s_wcnt[w]; if (w < wv) woff += c; tot += c; }
        const int slot = running + woff + prefix;
        if (emit >= 1) {
            const long long o = base_out + slot;
            store_tri(fvc + o * 9, cf.t0[0], cf.t0[1], cf.t0[2]);
            c2o[o] = f; neighbor[o] = emit == 2 ? (int)(o + 1) : -1; code[o] = cf.code0; cw[o * 2] = cf.w2; cw[o * 2 + 1] = cf.w3;
            if (emit == 2) {
                store_tri(fvc + (o + 1) * 9, cf.t1[0], cf.t1[1], cf.t1[2]);
                c2o[o + 1] = f; neighbor[o + 1] = (int)o; code[o + 1] = cf.code1; cw[(o + 1) * 2] = cf.w2; cw[(o + 1) * 2 + 1] = cf.w3;
            }
        }
        running += tot;
        __syncthreads();
    }
    if (tid == 0) { first_idx[b] = (int)base_out; num_faces[b] = running; }
}

// Backward of project + clip.  One WAVE per clipped-face slot j, one LANE per view: the B views push their gradient of slot j
// onto the same three mesh vertices (unless clipping or culling shifted the slots of a view), so the wave sums the nine
// world-space components over its lanes in registers (DPP) and one lane issues 9 atomics -- instead of B x 9 atomics per face
// that all land on the same V x 3 addresses (measured: 68 us with them, 5 us without, at 49 views x 12.8 k slots).
// LDS_TABLE: meshes of a few thousand vertices (every scene of this path) keep a V x 3 accumulator in LDS.  When clipping shifts the
// slots of the views against each other (the ground plane crosses the near plane in every view), the lanes of a wave hold DIFFERENT
// faces and the register sum does not apply: their B x 9 contributions per slot used to go to the same few hundred global addresses
// (env pass: 395 k atomics on 730 addresses, 30 us); now they meet in LDS and each workgroup flushes every touched component once.
constexpr int BWD_SLOTS = 8;           // clipped-face slots per workgroup (LDS_TABLE)
template <bool LDS_TABLE>
__device__ __forceinline__ void project_clip_bwd_body(
    const float *__restrict__ verts, const int *__restrict__ faces, const float *__restrict__ R,
    const float *__restrict__ T, const float *__restrict__ Kmat, int B, int V, int F, float eps, float zc, int persp,
    const int *__restrict__ num_faces, const int *__restrict__ c2o, const int *__restrict__ code,
    const float *__restrict__ cw, const float *__restrict__ gfvc, float *gverts, float *s_acc, int blk) {
    const int lane = threadIdx.x & 63;
    if (LDS_TABLE) {
        for (int i = threadIdx.x; i < V * 3; i += NT) s_acc[i] = 0.f;
        __syncthreads();
    }
    constexpr int WAVES = NT / DBW_WAVE, ITERS = LDS_TABLE ? BWD_SLOTS / WAVES : 1;
    for (int it = 0; it < ITERS; ++it) {
    const int j = LDS_TABLE ? blk * BWD_SLOTS + it * WAVES + (threadIdx.x >> 6) : blk * WAVES + (threadIdx.x >> 6);
    if (j >= 2 * F) break;
    for (int b0 = 0; b0 < B; b0 += DBW_WAVE) {
        const int b = b0 + lane;
        bool act = b < B && j < num_faces[b];
        const long long o = (long long)(act ? b : 0) * 2 * F + j;
        float g[9];
        if (act) {
            bool any = false;
#pragma unroll
            for (int i = 0; i < 9; ++i) { g[i] = gfvc[o * 9 + i]; any |= (g[i] != 0.f); }
            act = any;
        }
        int f = -1, vi[3] = {0, 0, 0};
        float gw[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (act) {
            Cam cam;
            load_cam(R, T, Kmat, b, cam);
            f = c2o[o];
            const int cd = code[o];
            vi[0] = faces[f * 3]; vi[1] = faces[f * 3 + 1]; vi[2] = faces[f * 3 + 2];
            f3 gv[3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
            const f3 ga{g[0], g[1], g[2]}, gb{g[3], g[4], g[5]}, gc{g[6], g[7], g[8]};
            if (cd < 0) {
                gv[0] = ga; gv[1] = gb; gv[2] = gc;
            } else {
                const int i1 = cd & 3, kind = cd >> 2;
                // vertex roles rotated by i1 with selects (runtime indexing would put the arrays in scratch memory)
                const f3 q0 = project(verts + (long long)vi[0] * 3, cam, eps).ndc, q1 = project(verts + (long long)vi[1] * 3, cam, eps).ndc,
                         q2 = project(verts + (long long)vi[2] * 3, cam, eps).ndc;
                const f3 P1 = i1 == 0 ? q0 : (i1 == 1 ? q1 : q2), P2 = i1 == 0 ? q1 : (i1 == 1 ? q2 : q0),
                         P3 = i1 == 0 ? q2 : (i1 == 1 ? q0 : q1);
                const float w2 = cw[o * 2], w3 = cw[o * 2 + 1];
                f3 g1{0.f, 0.f, 0.f}, g2{0.f, 0.f, 0.f}, g3{0.f, 0.f, 0.f};
                if (kind == 0) {          // (p4, p5, p1)
                    clip_point_bwd(P1, P2, zc, persp, w2, ga, g1, g2);
                    clip_point_bwd(P1, P3, zc, persp, w3, gb, g1, g3);
                    g1.x += gc.x; g1.y += gc.y; g1.z += gc.z;
                } else if (kind == 1) {   // (p4, p2, p5)
                    clip_point_bwd(P1, P2, zc, persp, w2, ga, g1, g2);
                    g2.x += gb.x; g2.y += gb.y; g2.z += gb.z;
                    clip_point_bwd(P1, P3, zc, persp, w3, gc, g1, g3);
                } else {                  // (p5, p2, p3)
                    clip_point_bwd(P1, P3, zc, persp, w3, ga, g1, g3);
                    g2.x += gb.x; g2.y += gb.y; g2.z += gb.z;
                    g3.x += gc.x; g3.y += gc.y; g3.z += gc.z;
                }
                gv[0] = i1 == 0 ? g1 : (i1 == 1 ? g3 : g2);
                gv[1] = i1 == 0 ? g2 : (i1 == 1 ? g1 : g3);
                gv[2] = i1 == 0 ? g3 : (i1 == 1 ? g2 : g1);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (gv[i].x != 0.f || gv[i].y != 0.f || gv[i].z != 0.f) {
                    const f3 w = vertex_bwd(verts, vi[i], cam, eps, gv[i]);
                    gw[i * 3] = w.x; gw[i * 3 + 1] = w.y; gw[i * 3 + 2] = w.z;
                }
        }
        const unsigned long long am = __ballot(act);
        if (am == 0ull) continue;
        const int leader = __ffsll((long long)am) - 1;
        const int f0 = __shfl(f, leader, 64);
        if (__popcll(am) >= 4 && __ballot(act && f == f0) == am) {      // every view maps slot j to the same mesh face
            float s[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) s[c] = wave_sum_dpp(gw[c]);       // inactive lanes hold zeros
            if (lane == leader) {
#pragma unroll
                for (int c = 0; c < 9; ++c)
                    if (s[c] != 0.f) {
                        if (LDS_TABLE) atomicAdd(&s_acc[vi[c / 3] * 3 + (c % 3)], s[c]);
                        else unsafeAtomicAdd(gverts + (long long)vi[c / 3] * 3 + (c % 3), s[c]);
                    }
            }
        } else if (act) {
#pragma unroll
            for (int c = 0; c < 9; ++c)
                if (gw[c] != 0.f) {
                    if (LDS_TABLE) atomicAdd(&s_acc[vi[c / 3] * 3 + (c % 3)], gw[c]);
                    else unsafeAtomicAdd(gverts + (long long)vi[c / 3] * 3 + (c % 3), gw[c]);
                }
        }
    }
    }
    if (LDS_TABLE) {
        __syncthreads();
        for (int i = threadIdx.x; i < V * 3; i += NT) {
            const float v = s_acc[i];
            if (v != 0.f) unsafeAtomicAdd(gverts + i, v);
        }
    }
}

template <bool LDS_TABLE>
__global__ __launch_bounds__(NT) void project_clip_bwd_kernel(
    const float *__restrict__ verts, const int *__restrict__ faces, const float *__restrict__ R,
    const float *__restrict__ T, const float *__restrict__ Kmat, int B, int V, int F, float eps, float zc, int persp,
    const int *__restrict__ num_faces, const int *__restrict__ c2o, const int *__restrict__ code,
    const float *__restrict__ cw, const float *__restrict__ gfvc, float *__restrict__ gverts) {
    extern __shared__ float s_acc[];   // LDS_TABLE: V * 3
    project_clip_bwd_body<LDS_TABLE>(verts, faces, R, T, Kmat, B, V, F, eps, zc, persp, num_faces, c2o, code, cw, gfvc, gverts, s_acc, blockIdx.x);
}

// ---- training step: the blocks' projection backward next to the backward of their texture preparation (step_kernels.h) -----------------
template <bool LDS_TABLE>
__global__ __launch_bounds__(NT) void clip_bwd_tex_kernel(const ClipBwdArgs C, const dbw_texture_set t, int nclip) {
    extern __shared__ float s_acc[];   // LDS_TABLE: V * 3
    if ((int)blockIdx.x < nclip)
        project_clip_bwd_body<LDS_TABLE>(C.verts, C.faces, C.R, C.T, C.Kmat, C.B, C.V, C.F, C.eps, C.zc, C.persp, C.num_faces, C.c2o, C.code, C.cw, C.gfvc,
                                         C.gverts, s_acc, blockIdx.x);
    else
        texture_prep_bwd_body(t.texture, t.n, t.h, t.w, t.decim, t.grad_maps, t.grad_sig, t.grad_texture, (long long)blockIdx.x - nclip,
                              (long long)gridDim.x - nclip);
}

// ---- training step: camera transform + near-plane clipping + per-face raster records (+ shading records) of both scenes -----------------
// One workgroup per (chunk of 256 faces, view, scene).  The slots of a view's clipped faces are assigned in face order (a face emits 0,
// 1 or 2 triangles): a workgroup first COUNTS what the faces in front of its chunk emit -- a face's count follows from the view depths
// of its three vertices alone (clip_emit_count) -- then works on its own 256 faces exactly like project_clip_fwd_kernel does on its
// current iteration, and goes straight on to what face_setup_kernel (raster.hip) and shade_setup_kernel (render_fused.hip) compute from
// the values it holds in registers.  Same device functions, same bits.
// CACHED: the scene's vertices are projected ONCE per workgroup into LDS (V * 12 bytes) and the faces gather from there: a face used to
// start with two dependent global loads per vertex (index, then vertex), and so did every face in front of the chunk for the count --
// chains of memory round trips that were most of this kernel's time (it is launched for few, short workgroups: latency is what it costs).
template <bool CACHED>
__global__ __launch_bounds__(NT) void scene_setup_kernel(const SceneSetupArgs A) {
    __shared__ int s_wcnt[NT / DBW_WAVE];
    extern __shared__ float s_ndc[];            // CACHED: (V, 3) projected vertices of this view
    if (A.sync_flag && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
        __hip_atomic_store(A.sync_flag, A.sync_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const SceneGeom &G = A.sc[A.scene0 + blockIdx.z];
    const int chunk = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int F = G.F, nchunks = (F + NT - 1) / NT;
    if (chunk >= nchunks) return;
    if (G.hdr && chunk == 0 && b == 0)
        for (int i = tid; i < G.nhdr; i += NT) G.hdr[i] = 0;
    Cam cam;
    load_cam(A.R, A.T, A.Kmat, b, cam);
    const long long base_out = (long long)b * 2 * F;
    // this chunk's face indices and those of the faces in front of it are requested before anything waits
    const int f = chunk * NT + tid;
    int vi[3] = {0, 0, 0};
    if (f < F) { vi[0] = G.faces[f * 3]; vi[1] = G.faces[f * 3 + 1]; vi[2] = G.faces[f * 3 + 2]; }
    if (CACHED) {
        for (int v = tid; v < G.V; v += NT) {
            const f3 q = project(G.verts + (long long)v * 3, cam, G.cam_eps).ndc;
            s_ndc[v * 3] = q.x; s_ndc[v * 3 + 1] = q.y; s_ndc[v * 3 + 2] = q.z;
        }
        __syncthreads();
    }
    // triangles emitted by the faces in front of this chunk: a face's count follows from its vertices' view depths alone
    int before = 0;
    for (int g = tid; g < chunk * NT; g += NT) {
        const int i0 = G.faces[g * 3], i1 = G.faces[g * 3 + 1], i2 = G.faces[g * 3 + 2];
        float z0, z1, z2;
        if (CACHED) { z0 = s_ndc[i0 * 3 + 2]; z1 = s_ndc[i1 * 3 + 2]; z2 = s_ndc[i2 * 3 + 2]; }
        else { z0 = view_z(G.verts + (long long)i0 * 3, cam); z1 = view_z(G.verts + (long long)i1 * 3, cam); z2 = view_z(G.verts + (long long)i2 * 3, cam); }
        before += clip_emit_count(z0, z1, z2, G.zc_on, G.zc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    if (lane == 0) s_wcnt[wv] = before;
    __syncthreads();
    int running = 0;
#pragma unroll
    for (int w = 0; w < NT / DBW_WAVE; ++w) running += s_wcnt[w];
    __syncthreads();
    ClippedFace cf;
    cf.emit = 0;
    if (f < F) {
        f3 p[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (CACHED) p[i] = f3{s_ndc[vi[i] * 3], s_ndc[vi[i] * 3 + 1], s_ndc[vi[i] * 3 + 2]};
            else p[i] = project(G.verts + (long long)vi[i] * 3, cam, G.cam_eps).ndc;
        }
        clip_face(p, G.zc_on, G.zc, G.persp, cf);
    }
    const int emit = cf.emit;
    const unsigned long long m1 = __ballot(emit >= 1), m2 = __ballot(emit == 2);
    const unsigned long long lower = (1ull << lane) - 1ull;
    const int prefix = __popcll(m1 & lower) + __popcll(m2 & lower);
    if (lane == 0) s_wcnt[wv] = __popcll(m1) + __popcll(m2);
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / DBW_WAVE; ++w) { const int c = s_wcnt[w]; if (w < wv) woff += c; tot += c; }
    const int slot = running + woff + prefix;
    FaceRec *recs = (FaceRec *)G.recs;
    ShadeRec *srec = (ShadeRec *)G.srec;
#pragma unroll
    for (int t = 0; t < 2; ++t) {          // (unrolled: which triangle is a compile-time choice, nothing of `cf` is indexed at run time)
        if (t >= emit) break;
        const long long o = base_out + slot + t;
        const f3 a = t == 0 ? cf.t0[0] : cf.t1[0], bq = t == 0 ? cf.t0[1] : cf.t1[1], cq = t == 0 ? cf.t0[2] : cf.t1[2];
        const float p9[9] = {a.x, a.y, a.z, bq.x, bq.y, bq.z, cq.x, cq.y, cq.z};
        const int nbr = emit == 2 ? (int)(t == 0 ? o + 1 : o - 1) : -1, cd = t == 0 ? cf.code0 : cf.code1;
#pragma unroll
        for (int i = 0; i < 9; ++i) G.fvc[o * 9 + i] = p9[i];
        G.c2o[o] = f; G.neighbor[o] = nbr; G.code[o] = cd; G.cw[o * 2] = cf.w2; G.cw[o * 2 + 1] = cf.w3;
        FaceRec r;
        float box[4];
        make_face_rec(p9, G.margin, 0, nbr, r, box);
        G.bbox[o] = make_float4(box[0], box[1], box[2], box[3]);
        recs[o] = r;
        if (srec) {
            ShadeRec sr;
            const float *uv = G.face_uvs + (long long)f * 6;
#pragma unroll
            for (int i = 0; i < 6; ++i) sr.uv[i] = uv[i];
            sr.j = f; sr.cd = cd; sr.w2 = cf.w2; sr.w3 = cf.w3;
            sr.map = G.face_map[f];
            sr.fa = G.map_alpha ? G.map_alpha[sr.map] : 1.f;
            const int *md = G.map_desc + sr.map * 8;
            sr.off = md[0]; sr.hw = (md[1] << 16) | md[2]; sr.pads = (md[3] << 16) | md[4]; sr.sh = md[5];
            srec[o] = sr;
        }
    }
    if (chunk == nchunks - 1 && tid == 0) {
        G.first_idx[b] = (int)base_out;
        G.num_faces[b] = running + tot;
        if (srec && b == 0 && running + tot == 0) {      // record 0 is what an empty slot of the shading loop fetches through: keep it benign
            ShadeRec sr;
#pragma unroll
            for (int i = 0; i < 6; ++i) sr.uv[i] = 0.f;
            sr.j = 0; sr.cd = -1; sr.w2 = sr.w3 = 0.f; sr.map = 0; sr.fa = 0.f; sr.off = 0; sr.hw = (1 << 16) | 1; sr.pads = 0; sr.sh = 0;
            srec[0] = sr;
        }
    }
}

}  // namespace

int dbw::launch_clip_bwd_tex(const ClipBwdArgs &C, const dbw_texture_set &t, hipStream_t s) {
    DBW_REQUIRE(C.verts && C.faces && C.R && C.T && C.Kmat && C.num_faces && C.c2o && C.code && C.cw && C.gfvc && C.gverts, "null pointer");
    DBW_REQUIRE(C.B > 0 && C.V > 0 && C.F > 0, "bad size");
    DBW_REQUIRE(t.texture && t.grad_maps && t.grad_texture && t.n > 0 && t.h > 0 && t.w > 0 && t.decim >= 1, "bad texture set");
    const long long work = (long long)t.n * t.h * t.w * 3;
    long long ntex = (work + NT - 1) / NT;
    if (ntex > 4096) ntex = 4096;
    const bool table = (size_t)C.V * 3 * sizeof(float) <= 48 * 1024;
    const int nclip = table ? (2 * C.F + BWD_SLOTS - 1) / BWD_SLOTS : (2 * C.F + NT / DBW_WAVE - 1) / (NT / DBW_WAVE);
    if (table)
        hipLaunchKernelGGL(clip_bwd_tex_kernel<true>, dim3((unsigned)(nclip + ntex)), dim3(NT), (size_t)C.V * 3 * sizeof(float), s, C, t, nclip);
    else
        hipLaunchKernelGGL(clip_bwd_tex_kernel<false>, dim3((unsigned)(nclip + ntex)), dim3(NT), 0, s, C, t, nclip);
    return dbw_check_launch("clip_bwd_tex_kernel");
}

int dbw::launch_scene_setup(const SceneSetupArgs &A, hipStream_t s) {
    DBW_REQUIRE(A.R && A.T && A.Kmat && A.B > 0, "bad argument");
    int chunks = 0;
    DBW_REQUIRE(A.scene0 >= 0 && A.nscenes >= 1 && A.scene0 + A.nscenes <= 2, "bad scene range");
    for (int i = A.scene0; i < A.scene0 + A.nscenes; ++i) {
        const SceneGeom &G = A.sc[i];
        DBW_REQUIRE(G.verts && G.faces && G.fvc && G.first_idx && G.num_faces && G.c2o && G.neighbor && G.code && G.cw && G.bbox && G.recs, "null pointer");
        DBW_REQUIRE(G.V > 0 && G.F > 0 && (long long)A.B * 2 * G.F < 0x7fffffffLL && (!G.zc_on || G.zc > 0.f), "bad size / z_clip");
        DBW_REQUIRE(!G.srec || (G.face_uvs && G.face_map && G.map_desc), "null pointer");
        chunks = max(chunks, (G.F + NT - 1) / NT);
    }
    int vmax = 0;
    for (int i = A.scene0; i < A.scene0 + A.nscenes; ++i) vmax = max(vmax, A.sc[i].V);
    if ((size_t)vmax * 12 <= 40 * 1024)
        hipLaunchKernelGGL(scene_setup_kernel<true>, dim3((unsigned)chunks, (unsigned)A.B, (unsigned)A.nscenes), dim3(NT), (size_t)vmax * 12, s, A);
    else
        hipLaunchKernelGGL(scene_setup_kernel<false>, dim3((unsigned)chunks, (unsigned)A.B, (unsigned)A.nscenes), dim3(NT), 0, s, A);
    return dbw_check_launch("scene_setup_kernel");
}

extern "C" int dbw_project_clip_fwd(const float *verts_world, const int32_t *faces, const float *R, const float *T,
                                    const float *Kmat, int B, int V, int F, float eps, int z_clip_enabled,
                                    float z_clip, int perspective_correct, float *face_verts_c, int32_t *first_idx,
                                    int32_t *num_faces, int32_t *c2o, int32_t *neighbor, int32_t *clip_code,
                                    float *clip_w, dbw_stream_t stream) {
    DBW_REQUIRE(verts_world && faces && R && T && Kmat && face_verts_c && first_idx && num_faces && c2o && neighbor &&
                    clip_code && clip_w, "null pointer");
    DBW_REQUIRE(B >= 0 && V > 0 && F > 0, "bad size");
    DBW_REQUIRE((long long)B * 2 * F < 0x7fffffffLL, "B*2F overflows int32 face indices");
    DBW_REQUIRE(!z_clip_enabled || z_clip > 0.f, "z_clip must be > 0");
    if (B == 0) return DBW_OK;
    hipLaunchKernelGGL(project_clip_fwd_kernel, dim3(B), dim3(NT), 0, (hipStream_t)stream, verts_world, faces, R, T,
                       Kmat, V, F, eps, z_clip_enabled, z_clip, perspective_correct, face_verts_c, first_idx,
                       num_faces, c2o, neighbor, clip_code, clip_w);
    return dbw_check_launch("project_clip_fwd_kernel");
}

extern "C" int dbw_project_clip_bwd(const float *verts_world, const int32_t *faces, const float *R, const float *T,
                                    const float *Kmat, int B, int V, int F, float eps, float z_clip,
                                    int perspective_correct, const int32_t *num_faces, const int32_t *c2o,
                                    const int32_t *clip_code, const float *clip_w, const float *grad_face_verts_c,
                                    float *grad_verts_world, dbw_stream_t stream) {
    DBW_REQUIRE(verts_world && faces && R && T && Kmat && num_faces && c2o && clip_code && clip_w &&
                    grad_face_verts_c && grad_verts_world, "null pointer");
    DBW_REQUIRE(B >= 0 && V > 0 && F > 0, "bad size");
    if (B == 0) return DBW_OK;
    if ((size_t)V * 3 * sizeof(float) <= 48 * 1024)
        hipLaunchKernelGGL(project_clip_bwd_kernel<true>, dim3((2 * F + BWD_SLOTS - 1) / BWD_SLOTS), dim3(NT), (size_t)V * 3 * sizeof(float),
                           (hipStream_t)stream, verts_world, faces, R, T, Kmat, B, V, F, eps, z_clip, perspective_correct, num_faces, c2o,
                           clip_code, clip_w, grad_face_verts_c, grad_verts_world);
    else
        hipLaunchKernelGGL(project_clip_bwd_kernel<false>, dim3((2 * F + NT / DBW_WAVE - 1) / (NT / DBW_WAVE)), dim3(NT), 0, (hipStream_t)stream,
                           verts_world, faces, R, T, Kmat, B, V, F, eps, z_clip, perspective_correct, num_faces, c2o,
                           clip_code, clip_w, grad_face_verts_c, grad_verts_world);
    return dbw_check_launch("project_clip_bwd_kernel");
}
