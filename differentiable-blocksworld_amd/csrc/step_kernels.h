// Fused small kernels of the training step (train_step.hip enqueues them).  Each replaces a chain of dependent 5-20 us launches of the
// operator-level path, or runs two independent ones in one launch: at the reference's batch size (4 views, configs/dtu/default.yml:28) the
// step is as long as its chain of dependent launches, whatever they compute.  The kernels live next to the kernels they fuse (same device
// functions, same arithmetic, bit-identical results); this header is their host-side interface.  Internal: not part of the C ABI.
//
//   step_prologue     (model_ops.hip)     block opacities (+ their noise), blocks' vertices, ground vertices, clearing of the small
//                                         gradients [+ sigmoid / decimation of the three texture tensors when the bins do not take it]
//                                                                                                         [was 7 launches]
//   scene_setup       (project_clip.hip)  camera transform + near-plane clipping + per-face raster records + shading records of BOTH
//                                         scenes (env: sky + ground, fg: blocks)                          [was 5]
//   scene_bins        (raster.hip)        coarse bins of both scenes + per-tile face lists + (in extra slices of its grid, in the shadow
//                                         of the bins' latency) the texture preparation                   [was 3 + 1]
//   regularisers      (model_ops.hip)     parsimony + overlap (samples drawn in registers) + its finish   [was 4]
//   clip_bwd_tex      (project_clip.hip)  the blocks' projection backward next to the backward of their texture preparation [was 2]
//   blocks_tail       (model_ops.hip)     backward of the blocks' pose / shape (from the points the prologue kept) + of their opacities [was 2]
//   + the env layer inside the fg pass (render_fused.hip) and the cross-stream signals inside scene_setup / work_scatter / the uv backward
// (What is NOT fused, on purpose: a chain of dependent kernels on ONE stream enqueued from C runs without gaps -- measured -- so fusing
// dependent kernels buys nothing unless the fusion is parallel over the same index space, and a "last workgroup finishes the job" epilogue
// serialises what separate kernels run in parallel: the tails built that way were slower than the launches they replaced.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dbw_hip.h"

namespace dbw {
struct RasterWorkspace;

constexpr int STEP_MAX_SETS = 4;
struct StepTextureSets { dbw_texture_set s[STEP_MAX_SETS]; };

struct PrologueArgs {
    StepTextureSets tex; int nsets;                 // texture preparation (blockIdx.y < nsets)
    // block opacities (dbw.py:297-311): alpha = sigmoid(logit + noise_scale * noise), keep = sigmoid(logit) > thresh
    const float *alpha_logit, *noise;               // noise: caller's draw (n_blocks) or NULL -> Philox (seed, rng_step, stream 0)
    float noise_scale, thresh;
    int nb;
    float *alpha, *alpha_full;
    int *keep;
    unsigned long long seed, rng_step;
    // blocks' vertices (dbw.py:343-352)
    const float *sq_eps, *S, *R6, *T, *trig;
    int nv;
    float ratio, scale_min, S_world;
    const float *Rw, *Tw;
    float *blk_verts;
    float *sq_local;                                // (nb, nv, 9): block-frame point, d / d eps1, d / d eps2 of every vertex, kept for the tail kernel
    // ground (dbw.py:282-287)
    const float *ground_base; int ngv;
    const float *R6g, *Tg;
    float *ground_verts;
    // floats cleared by the launch (the accumulated pose / shape / opacity gradients)
    float *zero0; int nzero0;
};
int launch_step_prologue(const PrologueArgs &P, hipStream_t s);

// one scene of the set-up kernels
struct SceneGeom {
    const float *verts; const int *faces; int V, F;
    float cam_eps; int zc_on; float zc; int persp;
    // clipped faces (B, 2F, ...)
    float *fvc; int *first_idx, *num_faces, *c2o, *neighbor, *code; float *cw;
    // raster records
    float margin;                                   // sqrt(blur_radius)
    float4 *bbox; void *recs;                       // FaceRec
    int *hdr; int nhdr;                             // cell-list header to clear (or NULL)
    // shading records of the soft pass (NULL: none)
    void *srec; const float *face_uvs; const int *face_map, *map_desc; const float *map_alpha;    // one opacity per map, or NULL
};
struct SceneSetupArgs { SceneGeom sc[2]; const float *R, *T, *Kmat; int B; int scene0, nscenes;       // scenes [scene0, scene0 + nscenes) of sc
                        unsigned *sync_flag; unsigned sync_val; };     // (the first workgroup stores sync_val when it starts: ShadeArgs::sync_flag)
int launch_scene_setup(const SceneSetupArgs &A, hipStream_t s);

struct SceneBinsArgs {
    struct One {
        const float4 *bbox; const void *recs; const int *first_idx, *num_faces;
        int *list, *count; unsigned *mask;
        int cells; int2 *cell; int *pool; int pool_cap; int *hdr; int *rank;
        int *dom;                                   // per-tile dominant face out (raster_bin.h), or NULL
    } sc[2];
    int B, H, W, nx, ny;
    int scene0, nscenes;
    // the texture preparation of the step (sigmoid + decimation of tex_sets tensors) in the shadow of the bins: tex_z more slices of the
    // grid per tensor run it -- the prologue then only computes what the set-up waits for.  tex_sets == 0: off
    StepTextureSets tex; int tex_sets, tex_z;
};
int launch_scene_bins(const SceneBinsArgs &A, hipStream_t s);
// (sync_flag: the first workgroup stores sync_val when it starts -- "everything in front of this launch is complete", ShadeArgs::sync_flag)
int dbw_launch_work_scatter(const RasterWorkspace &L, int N, int H, int W, hipStream_t s, unsigned *sync_flag = nullptr, unsigned sync_val = 0);

struct RegulariserArgs {
    // overlap (dbw.py:389-405): scale = 0 -> off
    const float *u; int npts;                       // caller's samples (nb, npts, 3) or NULL -> Philox (seed, rng_step, stream 1)
    unsigned long long seed, rng_step;
    const float *sq_eps, *S, *R6, *T, *alpha_full;
    int nb;
    float ratio, scale_min, inv_temp, thresh, overlap_scale;
    // parsimony (dbw.py:373-377): scale = 0 -> off
    float pars_eps, pars_scale;
    float *loss_parsimony, *loss_overlap;
    float *g_sq_eps, *g_S, *g_R6, *g_T, *g_alpha_full;
    float *ws;                                      // nb * 18 floats, zero
    unsigned *ticket;                               // zero; left zero
};
int launch_regularisers(const RegulariserArgs &A, hipStream_t s);

// backward of the blocks' pose / shape from the gradient of their world vertices + backward of the block opacities: sq_blocks_bwd_kernel
// reading the block-frame points the prologue kept (no powf / logf) with block_alpha_bwd_kernel in the same launch, one workgroup per block
struct BlocksTailArgs {
    const float *sq_eps, *S, *R6, *T, *sq_local; const int *keep; int nb, nv; float scale_min, S_world; const float *Rw;
    const float *g_verts;
    float *g_sq_eps, *g_S, *g_R6, *g_T;
    const float *alpha, *g_alpha_parts, *g_alpha_full; int alpha_parts; float *g_logit;      // g_alpha_parts may be NULL (fine phase)
    const float *void_raised; float *void_flag;     // *void_flag = *void_raised by the launch's first thread (train_step.hip: sync_wait_kernel); or NULL
};
int launch_blocks_tail(const BlocksTailArgs &A, hipStream_t s);

// Tail of the fg chain: the backward of the projection + clipping of the blocks and the backward of their texture preparation do not
// depend on each other -- one launch, the grid split between them (project_clip.hip), instead of two dependent ones
struct ClipBwdArgs {
    const float *verts; const int *faces; const float *R, *T, *Kmat; int B, V, F; float eps, zc; int persp;
    const int *num_faces, *c2o, *code; const float *cw, *gfvc; float *gverts;
};
int launch_clip_bwd_tex(const ClipBwdArgs &C, const dbw_texture_set &tex, hipStream_t s);

// the env layer folded into the fg pass (render_fused.hip): the env scene's workspace (per-tile lists and shading records filled by the
// set-up kernels), its maps and background, and where its hard uv-fragments go (frag_layout 3: what the env backward reads)
struct EnvFoldHost {
    const RasterWorkspace *ws; const int *first_idx, *num_faces; const float *maps; float bg[3];
    int *p2f; float *uvj;
};
int render_fwd_fused_mse_fold(const float *face_verts_c, const int32_t *first_idx, const int32_t *num_faces, const int32_t *neighbor,
                              const int32_t *c2o, const int32_t *clip_code, const float *clip_w, int Fc_stride, const float *face_uvs,
                              const int32_t *face_map, const int32_t *map_desc, const float *maps, const float *faces_alpha, int alpha_len, int N,
                              int64_t F_total, int H, int W, int K, int F, float sigma, float blur_radius, int perspective_correct,
                              const float *background3, int32_t *pix_to_face, float *bary, float *dists, void *workspace, size_t workspace_bytes,
                              const float *target, float mse_scale, float *loss_partial, float *grad_fg, float *grad_env, const EnvFoldHost &fold,
                              float *rec_out, const float *grad_rec, hipStream_t stream);

// dbw_render_bwd_fused (shade_blend.hip) whose first workgroup also stores sync_val to *sync_flag when it starts (ShadeArgs::sync_flag; a
// kernel other than the specialised uv backward gets a one-thread launch in front of it instead).  sync_flag == NULL: exactly dbw_render_bwd_fused
int render_bwd_fused_signal(const int32_t *pix_to_face, const float *bary, const float *dists, const int32_t *c2o, const int32_t *clip_code,
                            const float *clip_w, int Fc_stride, const float *face_uvs, const int32_t *face_map, const int32_t *map_desc, const float *maps,
                            const float *faces_alpha, int alpha_len, int N, int H, int W, int K, int F, float sigma, const float *background3,
                            const float *grad_image, const float *face_verts_c, int perspective_correct, int detach_bary, float *grad_maps,
                            float *grad_faces_alpha, float *grad_face_verts_c, int lds_aggregate, int frag_layout, const int32_t *bin_base,
                            int32_t *bin_cursor, void *bin_records, int bin_cap, const uint32_t *bin_layout, int const_geometry_faces,
                            const float *grad_scale, int image_layout, dbw_stream_t stream, unsigned *sync_flag, unsigned sync_val);

}  // namespace dbw
