/*
 * dbw_hip.h -- C ABI of libdbw_hip.so: the MI355X (gfx950) differentiable superquadric render path.
 *
 * Drop-in boundary: these entry points are what the reference's Python would bind in place of the
 * third-party ops it reaches through src/model/renderer.py:53-54,92-94,226,236 (PyTorch3D `_C.rasterize_meshes`,
 * `_C.rasterize_meshes_backward`, `interpolate_face_attributes`, `grid_sample`, and the ~12 torch kernels of
 * `layered_rgb_blend`, renderer.py:241-273) and through src/model/dbw.py:250-408 (param -> mesh, losses).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (torch tensors on the host side), contiguous, fp32/int32;
 *  - no allocation, no host synchronisation, no state between calls; kernels are enqueued on `stream`
 *    (a hipStream_t passed as void*; NULL = the null stream);
 *  - return 0 on success, a negative DBW_ERR_* otherwise (never throws); dbw_last_error() gives the text;
 *  - "accumulate" outputs must be zeroed by the caller, all others are fully written;
 *  - empty fragment slots hold -1 in all four fragment tensors (PyTorch3D convention);
 *  - indices are int32 in this ABI; the Python boundary widens to int64 where PyTorch3D returns int64.
 */
#ifndef DBW_HIP_H
#define DBW_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifndef DBW_BIN_SUBCURSORS
#define DBW_BIN_SUBCURSORS 16   /* cursors (and record sub-ranges) per texture bin, see dbw_render_bwd_fused; the value a library was
                                 * built with: dbw_bin_subcursors() */
#endif

/* ABI revision of this header (dbw_abi_version() returns the value the library was built with).  2: image_layout argument of the fused
 * render entry points; 3: bin_layout; 4: dbw_train_step_* (the whole optimisation iteration behind one entry); 5: dbw_step_inputs.rng_step
 * (the random-number counter is the caller's step count, not the plan's), skip_flag of dbw_adam_step_groups, a cross-stream wait that
 * gives up voids its step and moves the plan to events instead of failing the next run (dbw_train_step_voided_runs); 6: dbw_lpips_head_*
 * (the head of the perceptual criterion); 7: the device-side test hooks (dbw_debug_divcheck / _model_math / _lane_merge) left the library,
 * dbw_debug_set_flags is per host thread, the void flag is written by every run from a word that only the host clears
 * (dbw_train_step_void_flag_offset), uv-fragment passes address their maps with 32-bit byte offsets (maps buffer < 2^30 floats) */
#define DBW_ABI_VERSION 7

#ifdef __cplusplus
extern "C" {
#endif

#define DBW_OK 0
#define DBW_ERR_INVALID (-1)     /* bad argument (null pointer, negative size, K out of range ...) */
#define DBW_ERR_UNSUPPORTED (-2) /* configuration not compiled in (faces_per_pixel > DBW_MAX_FACES_PER_PIXEL) */
#define DBW_ERR_LAUNCH (-3)      /* HIP launch failure */
#define DBW_MAX_FACES_PER_PIXEL 25

typedef void *dbw_stream_t;

int dbw_abi_version(void);
int dbw_bin_subcursors(void);      /* DBW_BIN_SUBCURSORS of this build: callers size bin_cursor and round bin_cap with it */
const char *dbw_last_error(void);
/* Ablation switches of the product kernels' alternative code paths, used by the parity tests (which run those paths against the oracle
 * too) and by tools/ (0 = product behaviour).  They belong to the CALLING HOST THREAD (thread-local, read when a launch is enqueued), like
 * the text of dbw_last_error: the library keeps no other state between calls.  bits 0-7: shading ablations, 16: no fragment stores,
 * 128: no coarse bins, 256: plain IEEE divisions in the rasteriser (instead of the shared-reciprocal div_fast, which is bit-identical
 * inside its guards), 512: no conservative tile-vs-edge culling in the binning, 4096: no per-tile face lists (every tile walks its coarse
 * bin); bits 16 and up: wall-clock ablations of the fused kernels for tools/diag (results are wrong by construction): 1 << 17 no record
 * stores, 1 << 18 no cursor atomics, 1 << 19 no record path in the binned backward.
 * (The device-side test hooks of earlier ABI versions -- dbw_debug_divcheck, dbw_debug_model_math, dbw_debug_lane_merge -- are no longer
 * part of this library: the same inline arithmetic is built into a checker-side library, tests/device_checks.hip.) */
void dbw_debug_set_flags(int flags);

/* ------------------------------------------------------------------------------------------------------------------
 * Camera transform + z-clipping of one scene seen from B cameras.
 * Replaces MeshRasterizer.transform + clip_faces (pytorch3d 0.7.1; called from renderer.py:92-94 with eps=1e-8,
 * z_clip from renderer.py:35,46).  SURVEY.md A.2, A.4.  No host sync (PyTorch3D's clip_faces syncs via .item()).
 *
 *  verts_world (V,3)  faces (F,3)  R (B,3,3) row-vector convention  T (B,3)  Kmat (4,4)
 * Outputs, per view b, at most 2F clipped faces stored at [b*2F, b*2F + num_faces[b]):
 *  face_verts_c (B,2F,3,3): x,y NDC, z view depth     first_idx (B) = b*2F      num_faces (B)
 *  c2o (B,2F): local original face id of each clipped face
 *  neighbor (B,2F): packed index of the sibling triangle of a split quad, else -1
 *  clip_code (B,2F): -1 = unclipped copy; else p1_index | (kind<<2), kind 0 = case-3 triangle (p4,p5,p1),
 *                    1 = case-4 t1 (p4,p2,p5), 2 = case-4 t2 (p5,p2,p3)
 *  clip_w (B,2F,2): interpolation weights (w2,w3) of the intersection points p4,p5
 * z_clip_enabled = 0 reproduces z_clip_value=None (plain copy).
 */
int dbw_project_clip_fwd(const float *verts_world, const int32_t *faces, const float *R, const float *T,
                         const float *Kmat, int B, int V, int F, float eps, int z_clip_enabled, float z_clip,
                         int perspective_correct, float *face_verts_c, int32_t *first_idx, int32_t *num_faces,
                         int32_t *c2o, int32_t *neighbor, int32_t *clip_code, float *clip_w, dbw_stream_t stream);

/* Backward of the above: grad_face_verts_c (B,2F,3,3) -> grad_verts_world (V,3) (accumulate, summed over views). */
int dbw_project_clip_bwd(const float *verts_world, const int32_t *faces, const float *R, const float *T,
                         const float *Kmat, int B, int V, int F, float eps, float z_clip, int perspective_correct,
                         const int32_t *num_faces, const int32_t *c2o, const int32_t *clip_code, const float *clip_w,
                         const float *grad_face_verts_c, float *grad_verts_world, dbw_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Rasteriser.  Mirrors pytorch3d `_C.rasterize_meshes` / `_C.rasterize_meshes_backward`;
 * canonical semantics = the CPU naive path, list kept sorted by (z, face index).
 *  face_verts (F_total,3,3)   first_idx/num_faces (N)   neighbor (F_total) or NULL
 *  pix_to_face i32 (N,H,W,K)  zbuf (N,H,W,K)  bary (N,H,W,K,3)  dists (N,H,W,K); zbuf may be NULL (not stored).
 *  workspace: dbw_rasterize_workspace_bytes(F_total) bytes of scratch, 128-byte aligned (per face a 16 B screen box and a 128 B
 *    record of the pixel-independent arithmetic, csrc/raster_math.h: FaceRec).  F_total must be below 2^27 - 1 (the per-pixel list
 *    packs depth | face id | payload slot into 64-bit keys).
 */
size_t dbw_rasterize_workspace_bytes(int64_t F_total);
/* Larger workspace that also holds the coarse bins of the two-level binning (per view and 64x64-pixel bin, the ordered list of
 * faces touching it): when dbw_rasterize_fwd / dbw_render_fwd_fused are given at least this many bytes, tiles scan the list
 * of their bin instead of every face of the view.  Results are identical either way. */
size_t dbw_rasterize_workspace_bytes_binned(int64_t F_total, int N, int H, int W);
int dbw_rasterize_fwd(const float *face_verts, const int32_t *first_idx, const int32_t *num_faces,
                      const int32_t *neighbor, int N, int64_t F_total, int H, int W, int K, float blur_radius,
                      int perspective_correct, int clip_barycentric_coords, int cull_backfaces, int32_t *pix_to_face,
                      float *zbuf, float *bary, float *dists, void *workspace, size_t workspace_bytes,
                      dbw_stream_t stream);
/* grad_zbuf / grad_bary / grad_dists may each be NULL (treated as zero). grad_face_verts: accumulate. */
int dbw_rasterize_bwd(const float *face_verts, const int32_t *pix_to_face, const float *grad_zbuf,
                      const float *grad_bary, const float *grad_dists, int N, int64_t F_total, int H, int W, int K,
                      int perspective_correct, int clip_barycentric_coords, float *grad_face_verts,
                      dbw_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Shading + layered blend, fused: barycentric conversion of clipped faces, UV interpolation, bilinear texture
 * sampling (TexturesUV.sample_textures: align_corners=True, border, v flipped) with the circular
 * u-padding of dbw.py:339-341 resolved by index arithmetic instead of a padded copy, and the front-to-back alpha
 * compositing of renderer.py:241-273 (clip_inside=True) with per-face learned opacity.
 *
 *  pix_to_face/bary/dists: fragments in CLIPPED face indexing (as produced by dbw_rasterize_fwd on the output of
 *    dbw_project_clip_fwd); c2o/clip_code/clip_w as above, or all NULL when the fragments already index original
 *    faces (then pix_to_face = b*F + j).
 *  face_uvs (F,3,2)   face_map (F) -> row of map_desc
 *  map_desc (M,8) int32: {offset in floats into maps, height h, width w (unpadded), pad_left, pad_right, shift, rows, 0};
 *                        rows: M in row 0 (0 = not given: allowed, the kernels then read every descriptor from memory), 0 elsewhere
 *  maps: flat fp32 buffer of RGB maps in [0,1], fewer than 2^31 floats in total (texels are addressed with the int32 offsets);
 *    map m is STORED as (h>>shift, w>>shift, 3): shift > 0 is a decimated map
 *    (avg_pool2d(2^shift) kept at cell resolution; the nearest upsampling of dbw.py:278,334 is the shift)
 *  faces_alpha: NULL, or |alpha_len| floats: alpha_len == F (per face, shared by all views), N*F (packed per view), or
 *    alpha_len = -M < 0: one opacity per texture map / mesh (indexed by face_map; the reference repeats each block's opacity over
 *    its faces, dbw.py:219 -- this form needs no repeated copy).  Its gradient buffer grad_faces_alpha then holds 64 partial
 *    sums per opacity (M x 64 floats, spread by face index so that a mesh's fragments do not all hit one address), to be added
 *    by the caller (dbw_block_alpha_bwd does, g_alpha_parts = 64).
 *  sigma: the blend's opacity from the signed squared distance d (renderer.py:252-258).  sigma > 0: exp(-max(d, 0) / sigma) (clip_inside =
 *    True, every shipped config); sigma == 0: the hard indicator [d <= 0]; sigma < 0: sigmoid(-d / |sigma|) (clip_inside = False) -- here and
 *    in every entry point that takes `sigma` except the training step (dbw_step_desc.sigma > 0) and the composite + MSE epilogue.
 *  background3: HOST pointer to 3 floats (blend background colour, renderer.py:32), NULL = black.
 *  image (N,4,H,W): premultiplied RGB + alpha (BCHW).
 */
int dbw_shade_blend_fwd(const int32_t *pix_to_face, const float *bary, const float *dists, const int32_t *c2o,
                        const int32_t *clip_code, const float *clip_w, int Fc_stride, const float *face_uvs,
                        const int32_t *face_map, const int32_t *map_desc, const float *maps,
                        const float *faces_alpha, int alpha_len, int N, int H, int W, int K, int F, float sigma,
                        const float *background3, float *image, dbw_stream_t stream);
/* grad_image (N,4,H,W).  Outputs: grad_maps (same layout as maps; accumulate), grad_faces_alpha (alpha_len;
 * accumulate; may be NULL), grad_dists (N,H,W,K; fully written; may be NULL), grad_bary (N,H,W,K,3 in CLIPPED
 * barycentrics; fully written; NULL = detach_bary, renderer.py:222-223).
 * lds_aggregate != 0: pre-aggregate texel/opacity gradients of each 16x16 tile in an LDS hash table before touching
 * memory (pays whenever neighbouring fragments share destination texels: magnified or decimated maps). */
int dbw_shade_blend_bwd(const int32_t *pix_to_face, const float *bary, const float *dists, const int32_t *c2o,
                        const int32_t *clip_code, const float *clip_w, int Fc_stride, const float *face_uvs,
                        const int32_t *face_map, const int32_t *map_desc, const float *maps,
                        const float *faces_alpha, int alpha_len, int N, int H, int W, int K, int F, float sigma,
                        const float *background3, const float *grad_image, float *grad_maps,
                        float *grad_faces_alpha, float *grad_dists, float *grad_bary, int lds_aggregate,
                        dbw_stream_t stream);

/* Fused forward of one render pass: dbw_rasterize_fwd (clip_barycentric_coords = 1, no culling, zbuf not stored) followed
 * by dbw_shade_blend_fwd in ONE kernel: the image is blended from the register-resident per-pixel lists; the fragments
 * (pix_to_face, bary, dists) are stored once for the backward pass and never re-read in the forward direction.
 * Arguments as in those two entry points.  frag_layout selects how the fragments are stored: 0 = (N,H,W,K[,3]) like
 * dbw_rasterize_fwd; 1 = internal 8x8-tile planar layout [N][ceil(H/8)][ceil(W/8)][K][64] (bary [..][K][3][64]), in which
 * every wave access is fully coalesced -- buffers must then hold N*ceil(H/8)*ceil(W/8)*K*64 (x3) elements and can only be
 * consumed by dbw_render_bwd_fused with the same frag_layout; 2 = layout 1 with EIGHT `bary` planes ([..][K][8][64]: the
 * buffer holds N*ceil(H/8)*ceil(W/8)*K*64*8 floats) holding (u, v, bitcast(face | map << 20), blend opacity, r, g, b of
 * the sampled texture colour, transmittance in front of the fragment) -- everything the blend needs already resolved --,
 * and the pix_to_face entry of a pixel's first layer = id | fragment count << 26; for passes whose barycentrics carry no
 * gradient (detach_bary): the backward then needs no per-face table gathers, no texel fetch and no front-to-back pass
 * (F < 2^20, F_total < 2^26, maps < 2^11, and the `maps` buffer holds fewer than 2^30 floats: the pass addresses its texels with 32-bit byte
 * offsets from `maps`); 3 = layout 1 whose three `bary` planes hold (u, v, bitcast(face | map << 20)) and whose
 * `dists` are not written, for the HARD single-layer pass (K == 1, sigma == 0, no faces_alpha; F < 2^20, maps < 2^11): a kept pixel
 * lies inside its face and its opacity is 1, so that is all the backward needs for the texture gradient, and it rebuilds the
 * barycentrics from the pixel position for the geometry gradient (dbw_render_bwd_fused then requires lds_aggregate != 0).
 * image_layout (this entry point, dbw_render_fwd_fused_mse and dbw_render_bwd_fused): layout of every image-shaped buffer of the
 * call -- `image`, `env_image`, `target`, `grad_fg`, `grad_env`, `grad_image`: 0 = (N,C,H,W) planes; 1 = 8x8-tile planar
 * [N][ceil(H/8)][ceil(W/8)][C][64] (C = 4, targets 3; the buffers then hold N*ceil(H/8)*ceil(W/8)*C*64 floats), the layout of the
 * fragments: a wave's access to one plane of its tile is one 256 B line instead of eight 32 B row pieces.  The training step keeps its
 * intermediate images (env image, the two gradient images) in that layout and tiles the targets once. */
int dbw_render_fwd_fused(const float *face_verts_c, const int32_t *first_idx, const int32_t *num_faces,
                         const int32_t *neighbor, const int32_t *c2o, const int32_t *clip_code, const float *clip_w,
                         int Fc_stride, const float *face_uvs, const int32_t *face_map, const int32_t *map_desc,
                         const float *maps, const float *faces_alpha, int alpha_len, int N, int64_t F_total, int H, int W,
                         int K, int F, float sigma, float blur_radius, int perspective_correct, const float *background3,
                         int32_t *pix_to_face, float *bary, float *dists, float *image, void *workspace,
                         size_t workspace_bytes, int frag_layout, int stage, int image_layout, dbw_stream_t stream);

/* The same forward for the training path's soft pass (uv-fragments, frag_layout 2, K > 1) with the decoupled composite and the MSE
 * (dbw.py:223,366-367) as its epilogue: instead of storing its image the pass composites it in registers over env_image (N,4,H,W:
 * the already rendered sky + ground pass), rec = fg_rgb * mask + (1 - mask) * env_rgb, compares with target (N,3,H,W) and stores
 *   loss_partial (N * ceil(H/8) * ceil(W/8)): the sum of squared differences of each 8x8 tile (loss = mse_scale * their sum),
 *   grad_fg (N,4,H,W), grad_env (N,4,H,W; alpha plane zero) = d(mse_scale * sum of squares) / d(fg image), / d(env image)
 * -- the gradients dbw_composite_mse would produce, without the composite kernel's two round trips of both images through HBM.
 * stage: 0 = the whole pass.  1 = only the per-face set-up (boxes, face and shading records, bins) into `workspace`: needs the
 * geometry, the tables and faces_alpha but neither env_image nor target (may be NULL), touches no output -- a caller can run it on
 * another stream while the pass that produces env_image is still rendering.  2 = the pass itself on a workspace that a stage-1 call
 * with the same arguments filled (the caller orders the two calls: same stream, or an event between them). */
int dbw_render_fwd_fused_mse(const float *face_verts_c, const int32_t *first_idx, const int32_t *num_faces,
                             const int32_t *neighbor, const int32_t *c2o, const int32_t *clip_code, const float *clip_w,
                             int Fc_stride, const float *face_uvs, const int32_t *face_map, const int32_t *map_desc,
                             const float *maps, const float *faces_alpha, int alpha_len, int N, int64_t F_total, int H, int W,
                             int K, int F, float sigma, float blur_radius, int perspective_correct, const float *background3,
                             int32_t *pix_to_face, float *bary, float *dists, void *workspace, size_t workspace_bytes,
                             const float *env_image, const float *target, float mse_scale, float *loss_partial,
                             float *grad_fg, float *grad_env, int stage, int image_layout, dbw_stream_t stream);

/* Fused backward of one render pass: dbw_shade_blend_bwd followed by dbw_rasterize_bwd (clip_barycentric_coords = 1,
 * grad_zbuf = 0) without the grad_dists / grad_bary round trip through memory.  Same inputs as dbw_shade_blend_bwd plus
 * face_verts_c (the rasteriser's input).  detach_bary != 0 reproduces renderer.py:222-223 (geometry gradient through
 * dists only).  grad_face_verts_c (B*2F,3,3): accumulate.
 * Texture-space binning (optional; used when lds_aggregate == 0, i.e. full-resolution maps under minification, where a
 * 16x16-pixel tile shares no texels but the whole batch hits every texel ~70 times): instead of 12 scattered atomics per
 * fragment, each fragment appends one 24 B record (32 B until ABI 6; the buffer is sized as before, 32 B per record) to the bin of the 32x32-texel tile its footprint starts in
 * (bin = bin_base[map] + tile_y * ceil(ws/32) + tile_x; bin_cursor (nbins * DBW_BIN_SUBCURSORS) zeroed by the caller: a bin's
 * record range is split into DBW_BIN_SUBCURSORS sub-ranges of bin_cap / DBW_BIN_SUBCURSORS records with one cursor each, because
 * returning atomics on one hot address serialise; bin_records
 * nbins*bin_cap*32 bytes); dbw_texbin_reduce then sums every bin in LDS and adds it to grad_maps.  Records that do not fit
 * (bin overflow, circular-wrap footprints) fall back to atomics, so the result is exact either way.  All NULL / 0 = off.
 * const_geometry_faces: the first that many faces of the scene (original indexing) have constant vertices -- the sky dome of the env
 * scene (dbw.py:74-76: a buffer, not a parameter) -- so nothing is propagated through their barycentrics and their rows of
 * grad_face_verts_c stay untouched; 0 = every face gets its geometry gradient.
 * bin_layout (device, (nbins * DBW_BIN_SUBCURSORS, 2) uint32; required with texture bins): {first record, capacity} of every sub-range
 * inside bin_records, from dbw_bin_layout -- equal shares (all-zero demand) or sized by demand: after a launch bin_cursor holds how many records every sub-range was
 * ASKED for, whether they fitted or not, so the next launch can give each what it needs out of the same total (ops.py does that:
 * with equal shares a large scene overflows its hot bins and 56 of 66 ms of config 5's backward were fallback atomics).  Both calls of
 * a pass get the same table; records are indexed with 32 bits (nbins * bin_cap < 2^32).
 * grad_scale: DEVICE scalar every value of grad_image is multiplied by (the upstream gradient of a loss node, so that no
 * elementwise pass over the image-sized gradient is needed), NULL = 1. */
int dbw_render_bwd_fused(const int32_t *pix_to_face, const float *bary, const float *dists, const int32_t *c2o,
                         const int32_t *clip_code, const float *clip_w, int Fc_stride, const float *face_uvs,
                         const int32_t *face_map, const int32_t *map_desc, const float *maps, const float *faces_alpha,
                         int alpha_len, int N, int H, int W, int K, int F, float sigma, const float *background3,
                         const float *grad_image, const float *face_verts_c, int perspective_correct, int detach_bary,
                         float *grad_maps, float *grad_faces_alpha, float *grad_face_verts_c, int lds_aggregate,
                         int frag_layout, const int32_t *bin_base, int32_t *bin_cursor, void *bin_records, int bin_cap,
                         const uint32_t *bin_layout, int const_geometry_faces, const float *grad_scale, int image_layout,
                         dbw_stream_t stream);
/* bin_info (nbins,4) int32 = {offset of the bin's map in floats, stored width, stored height, tile_y << 16 | tile_x}. */
int dbw_texbin_reduce(const int32_t *bin_info, const int32_t *bin_cursor, const void *bin_records, int bin_cap,
                      const uint32_t *bin_layout, int nbins, float *grad_maps, dbw_stream_t stream);
/* bin_layout by demand: asked (n = nbins * DBW_BIN_SUBCURSORS) = the bin_cursor of an earlier launch of the same pass -> layout (n, 2):
 * capacity_i = 1 + floor(max(asked_i, min_records) * 1.25 * scale), scale = (total_records - n) / sum: the capacities add up to at most
 * total_records (< 2^32; a caller passes nbins * bin_cap: the memory of the equal shares); first_i = sum of the capacities before i. */
int dbw_bin_layout(const int32_t *asked, int64_t n, double total_records, int min_records, uint32_t *layout, dbw_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Texture preparation: maps = sigmoid(texture) (dbw.py:273,288,306), optionally "decimated"
 * (avg_pool2d(d), dbw.py:276-278,331-334; the nearest upsampling by d is left to the sampler's `shift`).
 * n maps of (h,w,3).  maps_out (n, h/d, w/d, 3): what the renderer samples;  sig_out (n,h,w,3) (may be NULL when d == 1,
 * required when d > 1): undecimated sigmoid kept for the TV loss.
 * Backward: grad_texture = (grad_maps[cell] / d^2 + grad_sig) * s(1-s).  grad_sig may be NULL.
 */
int dbw_texture_prep_fwd(const float *texture, int n, int h, int w, int decim, float *maps_out, float *sig_out,
                         dbw_stream_t stream);
int dbw_texture_prep_bwd(const float *texture, int n, int h, int w, int decim, const float *grad_maps,
                         const float *grad_sig, float *grad_texture, dbw_stream_t stream);

/* The same three passes over SEVERAL texture tensors of different shapes in one launch each (a scene has three: the blocks' maps,
 * the sky dome, the ground; dbw.py:273-293) -- a launch is ~5 us, which is all these passes cost.  1 <= nsets <= 4; `sets` is host
 * memory, read during the call.  Per set: texture (n,h,w,3) logits; decim; forward: maps (out, cell resolution) and sig (out, full
 * resolution; required when decim > 1, else optional); TV (dbw_tv_l2sq_sets): sig (in), wrap_x, tv_scale, grad_sig_out (out or NULL),
 * every set adding into the one `loss`; backward: grad_maps (in), grad_sig (in or NULL), grad_texture (out). */
typedef struct dbw_texture_set {
    const float *texture;
    int n, h, w, decim;
    float *maps;
    float *sig;
    int wrap_x;
    float tv_scale;
    float *grad_sig_out;
    const float *grad_maps;
    const float *grad_sig;
    float *grad_texture;
} dbw_texture_set;
int dbw_texture_prep_fwd_sets(const dbw_texture_set *sets, int nsets, dbw_stream_t stream);
int dbw_texture_prep_bwd_sets(const dbw_texture_set *sets, int nsets, dbw_stream_t stream);
int dbw_tv_l2sq_sets(const dbw_texture_set *sets, int nsets, float *loss, dbw_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Superquadric deformation + posing of the K block meshes (dbw.py:297-311,343-344,348-352; superquadric.py:10-14):
 *   eps = sigmoid(sq_eps)*1.8+0.1; v = parametric_sq(eta,omega,eps)*ratio; S = exp(S)+scale_min;
 *   R = rotation_6d_to_matrix(R_6d); verts = ((v*S)@R + T) * S_world @ R_world + T_world
 * sq_eps (Kb,2) S (Kb,3) R6 (Kb,6) T (Kb,3); trig (4,Kb,nv) = cos(eta), sin(eta), cos(omega), sin(omega) of the
 * constant buffers sq_eta/sq_omega (dbw.py:86-87), tabulated once at init; R_world (3,3) T_world (3).
 * `keep` (Kb) int32 or NULL: blocks with keep==0 are skipped.  dense != 0: kept blocks are written densely in order
 * (verts (NB,nv,3), PyTorch3D packing of dbw.py:326); dense == 0: every block keeps its slot (verts (Kb,nv,3)) and a
 * skipped block collapses to a single point -- zero-area faces that the rasteriser drops -- which renders identically
 * without the host having to know how many blocks are alive (no device->host sync, graph-capturable).
 */
int dbw_sq_blocks_fwd(const float *sq_eps, const float *S, const float *R6, const float *T, const float *trig,
                      const int32_t *keep, int dense, int Kb, int nv, float ratio, float scale_min, float S_world,
                      const float *R_world, const float *T_world, float *verts, dbw_stream_t stream);
/* grad_verts (NB,nv,3) -> grads of sq_eps,S,R6,T (accumulate). */
int dbw_sq_blocks_bwd(const float *sq_eps, const float *S, const float *R6, const float *T, const float *trig,
                      const int32_t *keep, int dense, int Kb, int nv, float ratio, float scale_min, float S_world,
                      const float *R_world, const float *grad_verts, float *g_sq_eps, float *g_S, float *g_R6,
                      float *g_T, dbw_stream_t stream);
/* Generic posed mesh (ground plane, dbw.py:282-287): verts = ((base*1)@rot6d(R6)+T)*S_world@R_world+T_world. */
int dbw_posed_mesh_fwd(const float *base, int nv, const float *R6, const float *T, float S_world,
                       const float *R_world, const float *T_world, float *verts, dbw_stream_t stream);
int dbw_posed_mesh_bwd(const float *base, int nv, const float *R6, const float *T, float S_world,
                       const float *R_world, const float *grad_verts, float *g_R6, float *g_T, dbw_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Block opacities (dbw.py:297-311): alpha = sigmoid(alpha_logit + noise_scale*noise) (noise may be NULL);
 * keep = sigmoid(alpha_logit) > mask_threshold (the filter_transparent / kill_blocks mask; mask_threshold < 0: keep all);
 * alpha_full = alpha * keep.  All (Kb).  keep may be NULL.
 * bwd: g_logit = (g_alpha + keep * g_alpha_full) * alpha * (1 - alpha); g_alpha / g_alpha_full may be NULL.
 * g_alpha holds g_alpha_parts partial sums per block (Kb x g_alpha_parts; 1 = plain; 64 for the per-map opacity gradients of
 * dbw_render_bwd_fused, see faces_alpha). */
int dbw_block_alpha_fwd(const float *alpha_logit, const float *noise, float noise_scale, float mask_threshold, int Kb,
                        float *alpha, float *alpha_full, int32_t *keep, dbw_stream_t stream);
int dbw_block_alpha_bwd(const float *alpha, const int32_t *keep, const float *g_alpha, int g_alpha_parts,
                        const float *g_alpha_full, int Kb, float *g_logit, dbw_stream_t stream);
/* Parsimony (dbw.py:373-377, utils/pytorch.py:35 safe_pow): loss += scale * mean(clamp(x, eps)^0.5) over n values;
 * grad (n, may be NULL) accumulates d loss / d x (zero where x <= eps). */
int dbw_sqrt_mean(const float *x, int n, float eps, float scale, float *loss, float *grad, dbw_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Decoupled composite + MSE, forward and backward in one pass (dbw.py:223, 366-367):
 *   rec = fg_rgb*mask + (1-mask)*env_rgb ; loss_sum += scale * sum((imgs-rec)^2)
 *   grad_fg (N,4,H,W), grad_env (N,4,H,W; alpha plane zero) are d(s*sum)/d(.) with s = scale * (scale_dev ? *scale_dev : 1)
 *   (scale = weight/count from the host, scale_dev = the upstream gradient scalar living on the device).
 * fg (N,4,H,W) premultiplied RGB + mask, env (N,4,H,W), imgs (N,3,H,W), rec (N,3,H,W) or NULL,
 * loss_sum: 1 float, accumulate, may be NULL.  grad_fg/grad_env may both be NULL (forward only).  imgs may be NULL to
 * obtain `rec` alone.
 */
int dbw_composite_mse(const float *fg, const float *env, const float *imgs, int N, int H, int W, float scale,
                      const float *scale_dev, float *rec, float *loss_sum, float *grad_fg, float *grad_env,
                      dbw_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Regularisers, forward + gradient in one pass (dbw.py:373-405, loss.py:46).
 * TV (l2sq) of n maps (h,w,3): loss += scale * [ sum_k |m[:,:,x+1]-m[:,:,x]|^2 (wrap_x closes the seam, dbw.py:383)
 *   / (h*(w-1+wrap)) + sum |m[:,y+1]-m[:,y]|^2 / ((h-1)*w) ]  (sum over maps first = "each map receives same grad")
 * grad_maps (may be NULL): fully written.
 */
int dbw_tv_l2sq(const float *maps, int n, int h, int w, int wrap_x, float scale, float *loss, float *grad_maps,
                dbw_stream_t stream);
/* Overlap (dbw.py:389-405; implicit_sq superquadric.py:17-38 safe=True, as_sdf=2).  u (Kb,npts,3) uniform [0,1)
 * samples; sample p of block k is ((u*2-1)*ratio*S_k)@R_k+T_k (no grad), tested against EVERY block.
 * loss += scale*mean_p(clamp(sum_k sigmoid(-sdf_k/temperature)*alpha_k - n_blocks_thresh, 0)).
 * alpha (Kb) = _alpha_full.  Grads accumulate into g_sq_eps,g_S,g_R6,g_T,g_alpha.
 * workspace: Kb*17 floats, zeroed by the caller. */
int dbw_overlap_loss(const float *u, int npts, const float *sq_eps, const float *S, const float *R6, const float *T,
                     const float *alpha, int Kb, float ratio, float scale_min, float temperature,
                     float n_blocks_thresh, float scale, float *loss, float *g_sq_eps, float *g_S, float *g_R6,
                     float *g_T, float *g_alpha, float *workspace, dbw_stream_t stream);

/* Fused Adam over a flat fp32 buffer (optimizer.py:6-18 -> torch.optim.Adam defaults, no weight decay/amsgrad).
 * step is the 1-based step count; bias corrections computed on the host. */
int dbw_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                  float beta1, float beta2, float eps, int step, dbw_stream_t stream);
/* The same over parameter groups that are contiguous in ONE flat buffer and differ only in learning rate (optimizer.py:10-17: the
 * texture group vs the rest): group k covers [group_end[k-1], group_end[k]) (group_end[-1] = 0), 1 <= ngroups <= 4; host arrays.
 * zero_buf / zero_bytes (optional, NULL / 0): device scratch cleared by the same launch -- the zero-initialised work space of the NEXT
 * iteration, so that it does not have to open with a fill (16-byte aligned, a multiple of 16 bytes).
 * skip_flag (optional, NULL): one device float; when it is != 0 at launch time NOTHING is updated (parameters and moments keep their
 * values) and only zero_buf is cleared -- how a training step whose cross-stream wait gave up (dbw_step_desc.sync_events) is kept from
 * applying gradients that may be incomplete (dbw_train_step_void_flag_offset). */
int dbw_adam_step_groups(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, const int64_t *group_end,
                         const float *lr, int ngroups, float beta1, float beta2, float eps, int step, void *zero_buf,
                         int64_t zero_bytes, const float *skip_flag, dbw_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * The whole optimisation iteration of the training path behind ONE entry point (ABI 4).
 * Replaces one pass of src/trainer.py:137-147 -- optimizer.zero_grad(); loss = model(images, labels) (src/model/dbw.py:198-408: the
 * decoupled render of dbw.py:213-223 through src/model/renderer.py:84-98, compute_losses dbw.py:361-408 without the perceptual term);
 * loss['total'].backward(); optimizer.step() (Adam, two learning-rate groups, src/optimizer.py:9-15) -- for the configuration every
 * shipped config trains in (decouple_rendering, detach_bary, perspective cameras, MSE + parsimony + TV + overlap).
 *
 * Why an entry of its own: at the reference's batch size (4 views, configs/dtu/default.yml:28) the kernels of an iteration take ~0.1 ms
 * and 33 launches issued one by one from the host took 0.5 ms.  Here the host makes ONE call; the library enqueues ~18 launches on the
 * caller's two streams and one of its own (the dependent 5-20 us kernels of the operator-level path are fused into a prologue, one set-up kernel and one binning kernel
 * for both scenes, one regulariser kernel and one tail kernel per scene: csrc/step_kernels.h), draws the opacity noise and the overlap
 * samples from a counter-based generator inside those kernels, and leaves the loss values in host-visible memory with one copy.
 *
 * Ownership: every pointer of dbw_step_desc / dbw_step_inputs is caller-owned DEVICE memory that stays valid while the plan lives
 * (inputs: until the call's work has finished).  `workspace` (dbw_train_step_workspace_bytes bytes, 256-byte aligned) is caller-owned
 * scratch the plan carves everything else out of: clipped faces, raster workspaces, fragments, images, maps, gradient accumulators, the
 * records of the texture bins.  The plan is the one object of this ABI that holds state between calls: its events, the step counter of its
 * random numbers, and which of the two demand tables of the texture bins is current.  (The library also keeps ONE pair of lowest-priority
 * side streams per process and device, shared by all plans: HIP multiplexes streams onto a few hardware queues, a pair per plan ends up
 * sharing a queue with the caller's stream.)  Not thread-safe; one plan per
 * (model, training phase, device).
 */
typedef struct dbw_step_desc {
    /* ---- sizes ---- */
    int H, W, faces_per_pixel, max_views;       /* image size, K, the largest batch a run may bring */
    int n_blocks, block_nv, block_nf;           /* blocks, vertices and faces per block (42 / 80 for the icosphere-1 superquadric) */
    int n_sky_verts, n_ground_verts, n_sky_faces, n_ground_faces;
    int txt_size, env_txt_size;                 /* texture side of a block / of the sky and ground maps */
    int decim_env, decim_blocks;                /* decimation factor in force for the env maps / the blocks' maps (1 = none; dbw.py:276-278,331-334) */
    int coarse;                                 /* 1: coarse phase -- one learned opacity per block (+ noise); 0: fine -- opaque faces */
    /* ---- renderer constants (src/model/renderer.py:25-60) ---- */
    float sigma, blur_radius;                   /* of the soft pass */
    float z_clip, cam_eps;                      /* z_clip <= 0: no near-plane clipping */
    int perspective_correct;
    float bg_fg[3], bg_env[3];
    float S_world, ratio_block_scene, scale_min;
    float opacity_noise;                        /* std of the opacity noise (0 = none) */
    float mask_threshold;                       /* a block is kept while sigmoid(alpha_logit) > mask_threshold; < 0: all kept */
    /* ---- loss weights, phase factors and 1 / world_size folded in (dbw.py:361-408); 0 = term off ---- */
    float w_rgb, w_parsimony, w_tv_bkg, w_tv_blocks, w_tv_ground, w_overlap;
    int overlap_points;                         /* samples per block (dbw.py:33) */
    float overlap_temperature, overlap_n_blocks;
    /* ---- constant tables ---- */
    const float *R_world, *T_world, *Kmat;      /* (3,3), (3), (4,4) */
    const float *ground_base;                   /* (n_ground_verts, 3) */
    float *env_verts;                           /* (n_sky_verts + n_ground_verts, 3): the sky part filled by the caller, the ground part by the step */
    const int32_t *env_faces; const float *env_face_uvs; const int32_t *env_face_map, *env_map_desc;
    const float *trig;                          /* (4, n_blocks, block_nv), see dbw_sq_blocks_fwd */
    const int32_t *block_faces; const float *block_face_uvs; const int32_t *block_face_map, *block_map_desc;
    const int32_t *block_bin_base, *block_bin_info; int n_bins;       /* texture bins (decim_blocks == 1), see dbw_render_bwd_fused */
    /* ---- parameters and their gradients (the gradients: views of one flat buffer, see flat_*) ---- */
    const float *sq_eps, *S, *R6, *T, *alpha_logit, *R6_ground, *T_ground, *texture_bkg, *texture_ground, *textures;
    float *g_sq_eps, *g_S, *g_R6, *g_T, *g_alpha_logit, *g_R6_ground, *g_T_ground, *g_texture_bkg, *g_texture_ground, *g_textures;
    /* ---- Adam over the flat parameter buffer: group 0 = [0, group_end[0]) pose / shape / opacity, group 1 = textures ---- */
    float *flat_param, *flat_grad, *exp_avg, *exp_avg_sq;
    int64_t group_end[2];
    float *small_grads; int n_small_grads;      /* the accumulated (not fully written) gradients: cleared at the head of every run */
    /* ---- options ---- */
    int fuse;                                   /* bit 0: prologue, 1: scene set-up + bins, 2: regularisers, 3: blocks' tail, 4: the env layer inside the
                                                 * fg pass (no env pass, no env image; needs bit 1), 5: the blocks' projection backward and the backward of
                                                 * their texture preparation in one launch, 6: the texture preparation (forward) in the shadow of the
                                                 * bins' launch instead of the prologue's (needs bits 0, 1 and sync_events == 0); 0 = the operator-level kernels */
    int backward_order;                         /* 0: both backward kernels at once, 1: the fg kernel first and alone (data parallel) */
    int binned_concurrent;                      /* texture bins (which otherwise imply order 1): the env chain starts next to the fg kernel */
    int serial_setup_max_views;                 /* runs of up to this many views keep the blocks' set-up on stream_main (no cross-stream hop);
                                                 * larger ones run it next to the env pass on the env stream */
    uint64_t seed;                              /* of the step's random numbers: the same on every data-parallel rank */
    int sync_events;                            /* how the plan's streams wait for each other.  0 (default): through words in device memory -- the
                                                 * producing stream runs a one-thread kernel that stores a counter, the waiting stream a one-thread
                                                 * kernel that polls it (every wait is enqueued after its producer, so no ordering of the hardware
                                                 * queues can deadlock it; a poll that still gives up -- after 1 s of wall clock of which it has
                                                 * itself been running a good part: the producer's queue was held up that long -- VOIDS its
                                                 * step on the device: the step's Adam launch (and, through the
                                                 * summed flag, every data-parallel rank's) skips the update, and the plan's next run -- which sees
                                                 * a word in mapped host memory -- switches the plan to events for good and goes on:
                                                 * dbw_train_step_voided_runs, dbw_train_step_sync_timeouts).  Measured: an
                                                 * event costs the stream that records or waits for it 7-11 us before its next kernel and the
                                                 * waiting stream starts 12-26 us late; the two tiny kernels cost ~2 us and ~1 us.  The FIRST run
                                                 * of a plan always goes through events: it pays for everything lazy -- at BASELINE config 5 its
                                                 * first kernel starts 0.96 s late, behind the driver's clearing of the 40 GB just allocated).
                                                 * != 0: HIP events (hipEventRecord / hipStreamWaitEvent) */
    float tv_value_scale;                       /* 0 (= 1): factor on the REPORTED total-variation value only.  Data parallel with deferred texture
                                                 * gradients (dbw_step_inputs.defer_textures): the TV gradient is added on every rank AFTER the
                                                 * all-reduce, so the kernels take the full weight (w_tv_*) and the value a rank reports is scaled by
                                                 * 1 / world_size here, as the other view-independent terms are through their weights */
} dbw_step_desc;

typedef struct dbw_step_inputs {
    const float *imgs;                          /* targets: (B,3,H,W), or the 8x8-tile planar layout when imgs_tiled != 0 */
    int imgs_tiled;
    const float *R, *T;                         /* (B,3,3), (B,3) */
    int B;                                      /* 1 <= B <= max_views */
    double global_count;                        /* elements of the GLOBAL batch's images: the MSE is a mean over them (dbw.py:367) */
    const float *noise_override, *overlap_u_override;   /* caller's draws instead of the plan's (n_blocks), (n_blocks, overlap_points, 3); NULL */
    int with_adam;                              /* 0: stop in front of Adam (data parallel: the caller all-reduces flat_grad first) */
    int adam_step; float lr[2], beta1, beta2, adam_eps;
    int read_losses;                            /* != 0: copy the loss values to host memory (dbw_train_step_losses) */
    /* The perceptual term (src/model/dbw.py:369-371, loss.py:32-40: LPIPS on the composite, weight 0.1 in every shipped config) is a
     * third-party network outside this library; the step makes room for it: phase 1 runs everything up to and including the fg pass and
     * stores the composite `rec_out` (B,3,H,W); the caller evaluates its term on it and differentiates it; phase 2 runs the fg pass again
     * with `grad_rec` = d term / d rec (B,3,H,W) added to the MSE's gradient in front of the chain rule through the composite, then the
     * rest of the iteration.  phase 0 (default): the whole iteration in one call, no perceptual term.  Needs fuse bit 4. */
    int phase;
    float *rec_out;
    const float *grad_rec;
    int single_stream;                          /* != 0: everything in order on stream_main, no side streams (every kernel then runs alone on the GPU:
                                                 * what a per-kernel profile wants) */
    int arena_is_clean;                         /* != 0: the caller cleared the zero arena (dbw_train_step_offset 4 / 5) since the last run --
                                                 * it does when it runs Adam itself through dbw_adam_step_groups(zero_buf = the arena);
                                                 * otherwise a run that does not follow a run with_adam clears the arena with a fill of its own */
    uint64_t rng_step;                          /* counter of the step's random numbers (opacity noise, overlap samples: Philox keyed on
                                                 * (seed, rng_step, stream, index)): the caller's optimisation-step count -- the same on every
                                                 * data-parallel rank whatever plan, batch size or phase a rank's run uses, never repeated by a
                                                 * ragged last batch or the next phase, and part of what a checkpoint already holds */
    int defer_textures;                         /* != 0 (needs with_adam == 0): stop in front of the backward of the texture preparation too.  The
                                                 * gradient of the PREPARED maps (dbw_train_step_offset 13 .. 14: sigmoid + decimation are linear
                                                 * behind it) is then what a data-parallel caller sums over the ranks -- with 8x-decimated maps
                                                 * 1 / 64 of the texture gradient's bytes (0.15 instead of 9.4 MB at config 2) -- next to the 159
                                                 * small gradients in flat_grad; dbw_train_step_finish then runs that backward (adding the TV
                                                 * gradient, which every rank holds in full) and Adam */
} dbw_step_inputs;

typedef struct dbw_step_plan dbw_step_plan;
size_t dbw_train_step_workspace_bytes(const dbw_step_desc *desc);
/* returns NULL on error (dbw_last_error) */
dbw_step_plan *dbw_train_step_create(const dbw_step_desc *desc, void *workspace, size_t workspace_bytes);
void dbw_train_step_destroy(dbw_step_plan *plan);
/* Enqueues one iteration.  stream_main carries the critical chain -- set-up, the two passes, the fg backward and its tail, Adam (or,
 * with_adam == 0, everything the caller's all-reduce has to wait for) -- and is the only stream the caller has to order against.  The env
 * backward chain and the regularisers run next to it: on stream_side if the caller brings one, else (NULL) on the library's own
 * lowest-priority streams; dbw_step_inputs.single_stream: everything in order on stream_main.  On return nothing has been waited for. */
int dbw_train_step_run(dbw_step_plan *plan, const dbw_step_inputs *in, dbw_stream_t stream_main, dbw_stream_t stream_side);
/* Behind a run with defer_textures (and the caller's all-reduce of the map gradients and the small gradients, ordered on `stream`): the
 * backward of the texture preparation of all three tensors, then -- in->with_adam -- Adam, which clears the zero arena.  Reads of `in`:
 * with_adam, adam_step, lr, beta1, beta2, adam_eps */
int dbw_train_step_finish(dbw_step_plan *plan, const dbw_step_inputs *in, dbw_stream_t stream);
/* Blocks until the loss values of the last run with read_losses != 0 are in host memory: out5 = rgb, parsimony, tv, overlap, total */
int dbw_train_step_losses(dbw_step_plan *plan, float *out5);
/* Byte offset inside the workspace of one of the plan's buffers (tests, diagnostics, host-side views of per-step state):
 * which = 0 alpha (n_blocks), 1 alpha_full, 2 keep (int32), 3 loss values on the device (5 floats, valid after a run), 4 / 5 begin / end of
 * the zero arena (cleared by the plan's own Adam launch; a caller that runs Adam itself clears it), 6 grad of the fg image (tiled), 7 grad
 * of the env image, 8 env image, 9 the blocks' world vertices, 10 per-tile loss partials, 11 / 12 the env scene's hard uv-fragments (face ids;
 * u, v, face | map), 13 / 14 begin / end of the gradient of the prepared maps (blocks, sky, ground: one range inside the zero arena; what
 * a data-parallel caller all-reduces with defer_textures); -1 for an unknown name */
int64_t dbw_train_step_offset(const dbw_step_plan *plan, int which);
/* Makes `stream` wait until the blocks' texture gradient of the last run is final.  Data parallel: the caller reduces that slice -- 83 % of the gradient bytes -- on a stream of its
 * own while the rest of the step still runs. */
int dbw_train_step_wait_blocks_ready(dbw_step_plan *plan, dbw_stream_t stream);
/* Number of cross-stream waits of this plan that gave up (sync_events == 0; never in a healthy process) -- synchronises the device; < 0 on error */
int dbw_train_step_sync_timeouts(dbw_step_plan *plan);
/* Times this plan had a wait give up and voided the runs in flight (host-side, no synchronisation: counted once the NEXT run -- or this
 * call -- has seen the word in mapped host memory; the runs enqueued between the poll giving up and that moment are all voided and count as
 * one).  > 0: the plan runs on events (as if created with sync_events = 1). */
int dbw_train_step_voided_runs(const dbw_step_plan *plan);
/* Byte offset inside the workspace of the step's void flag: one float that every run WRITES behind the join of its streams (in front of its
 * Adam launch): != 0 when a wait of this plan has given up and the host has not handled it yet -- the plan's "a poll gave up" word is
 * sticky on the device (no launch clears it: a clear ordered on a stalled main stream would run after the polls that raised it), the head
 * of the dbw_train_step_run that sees the word in mapped host memory synchronises the device and clears it; every run enqueued in between
 * is voided.  A data-parallel caller sums the float over the ranks next to the gradients and hands it to dbw_adam_step_groups as skip_flag
 * (the plan's own Adam launches -- dbw_train_step_run with_adam, dbw_train_step_finish -- read it themselves), so that all ranks skip
 * together; the sum may be taken in place, the next run overwrites it. */
int64_t dbw_train_step_void_flag_offset(const dbw_step_plan *plan);
/* tests: the join of the plan's NEXT run polls for a value that never comes and gives up after 0.05 s */
int dbw_debug_train_step_force_timeout(dbw_step_plan *plan);
/* tests: the side streams' polls for the prologue of the plan's NEXT run give up after 0.02 s (with their real value: they only do give up
 * behind a main stream that is stalled for longer than that -- the case a void flag cleared at the head of the run used to lose) */
int dbw_debug_train_step_hasty_prologue_wait(dbw_step_plan *plan);
/* diagnostics: out3 = {which of the plan's cross-stream counters the FIRST poll that gave up was waiting on (0 prologue, 1 tile order, 2 fg
 * forward, 3 regularisers, 4 bin layout, 5 fg backward kernel, 6 blocks' textures, 7 env chain, 8 texture preparation; -1: no poll gave up),
 * the value it wanted, the value it last saw}.  Synchronises the device. */
int dbw_debug_train_step_last_timeout(dbw_step_plan *plan, int *out3);
/* ... and all twelve counters: seen12 as that poll saw them when it gave up (as they are now, if none did), asked12 the values the host has
 * enqueued stores for so far. */
int dbw_debug_train_step_counters(dbw_step_plan *plan, unsigned *seen12, unsigned *asked12);

/* Measurement aid (bench.py): on != 0 makes every following run record HIP timing events around its four big kernels, on the streams they
 * run on and with everything that shares the GPU with them in a real step running next to them (the events themselves cost each
 * stream a few microseconds per record: profiled runs are not the timed ones).  dbw_train_step_kernel_times waits for the last profiled run
 * and returns out4_ms = env pass, fg pass (+ composite + MSE), fg backward, env backward. */
int dbw_train_step_profile(dbw_step_plan *plan, int on);
int dbw_train_step_kernel_times(dbw_step_plan *plan, float *out4_ms);

/* ------------------------------------------------------------------------------------------------------------------
 * Head of the perceptual criterion (src/model/loss.py:32-40 -> lpips 0.1.4 `LPIPS(net='vgg')(.., normalize=True)`),
 * for ONE feature tap of the frozen VGG16.  The convolutions stay with the caller (MIOpen); this replaces what follows them:
 *   value[n] = mean over the HW pixels of sum_c lin_w[c] * (t[c] - f[c] / (sqrt(sum_c f[c]^2) + 1e-10))^2
 * feat (N, C, HW): the tap of the reconstruction.  target_unit (V, C, HW): the UNIT-NORMALISED tap of the target images -- row n, or row
 * view_ids[n] when view_ids is given (the features of the training views are constants, kept by the caller; an id outside [0, V) turns
 * that image's value and gradient into NaN).  lin_w (C).
 *   fwd: partial (N, dbw_lpips_head_blocks(N, HW)), fully written: value[n] = the sum of row n (summed by the caller: a fixed order)
 *   bwd: grad_value (N) -> grad_feat (N, C, HW), fully written; target and weights are constants (the network is frozen, loss.py:36-37)
 */
int dbw_lpips_head_blocks(int N, int HW);
int dbw_lpips_head_fwd(const float *feat, const float *target_unit, const int64_t *view_ids, const float *lin_w, int N, int V, int C, int HW,
                       float *partial, dbw_stream_t stream);
int dbw_lpips_head_bwd(const float *feat, const float *target_unit, const int64_t *view_ids, const float *lin_w, int N, int V, int C, int HW,
                       const float *grad_value, float *grad_feat, dbw_stream_t stream);

/* The two element-wise layers between the frozen network's convolutions, each one pass (torch: several kernels over the largest tensors):
 * dbw_bias_relu: y = max(x + bias[c], 0) for x (N, C, HW) behind a convolution run WITHOUT its bias (y == x allowed);
 * dbw_maxpool2_fwd / _bwd: 2x2, stride 2, floor mode over `planes` = N * C planes of (H, W) -> (H/2, W/2); the backward finds the window's
 * maximum again from x (the first one in row-major window order, as torch's forward picks it) and fully writes grad_x (N, C, H, W). */
int dbw_bias_relu(const float *x, const float *bias, int N, int C, int HW, float *y, dbw_stream_t stream);
int dbw_maxpool2_fwd(const float *x, int planes, int H, int W, float *y, dbw_stream_t stream);
int dbw_maxpool2_bwd(const float *x, const float *grad_y, int planes, int H, int W, float *grad_x, dbw_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DBW_HIP_H */
