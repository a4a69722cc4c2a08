"""torch.autograd wrappers over the C ABI (include/dbw_hip.h).  PyTorch is plumbing here: device memory, the current
HIP stream and autograd bookkeeping; every arithmetic step of the path is a kernel of libdbw_hip.so.

Operator-level mirrors of what the reference reaches in PyTorch3D:
  rasterize_meshes(...)            <-> pytorch3d.renderer.mesh.rasterize_meshes._C.rasterize_meshes (+ backward)
  render_scene(...)                <-> MeshRasterizer.transform + clip_faces + rasterize + TexturesUV.sample_textures
                                        + layered_rgb_blend (src/model/renderer.py:84-98,219-273)
"""
import ctypes
import math

import torch

from . import _lib

MAX_FACES_PER_PIXEL = 25
ALPHA_SPREAD = 64          # partial sums per per-map opacity gradient (csrc/shade_common.h: DBW_ALPHA_SPREAD)

def bin_subcursors():
    """Cursors per texture bin of the loaded library (include/dbw_hip.h: DBW_BIN_SUBCURSORS, dbw_bin_subcursors())."""
    lib = _lib.load()
    return int(lib.dbw_bin_subcursors()) if hasattr(lib, 'dbw_bin_subcursors') else 16


def __getattr__(name):             # ops.BIN_SUBCURSORS: the library's value, asked for on first use
    if name == 'BIN_SUBCURSORS':
        return bin_subcursors()
    raise AttributeError(name)

COARSE_BINS = True        # two-level face binning in the rasteriser (64x64-pixel coarse bins); False: every tile scans every face
TEXTURE_BINS = True       # full-resolution texel gradients: bin records by 32x32-texel tile and reduce in LDS (vs 12 atomics/fragment)
UV_FRAGMENTS = True       # detach_bary passes: the forward stores resolved (u, v, face|map) per fragment for the backward
TILED_FRAGMENTS = True    # fused path keeps its fragments in the 8x8-tile planar layout (coalesced); needs both FUSED_* = True
FUSED_FORWARD = True      # one kernel for raster + shade + blend (False: the two operator-level kernels)
FUSED_BACKWARD = True     # one kernel for blend-backward + rasteriser-backward (False: the two operator-level kernels)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class ZeroArena:
    """Per-step pool of zero-initialised scratch (gradient accumulators, cursors, workspaces).  An iteration used to issue ~30
    `fill` launches of a few microseconds each for them; with the arena enabled they are carved out of one buffer that
    `begin_step()` clears with a single launch.  Opt-in (ShardedTrainStep enables it): a tensor handed out is only valid until
    the next begin_step(), which is safe when the parameters' .grad buffers are preallocated (gradients are accumulated into
    them and nothing of the arena outlives the step) but not for callers that let autograd keep the returned gradient."""

    def __init__(self):
        self.enabled = False
        self.buf, self.off, self.want, self.clean = {}, {}, {}, {}

    def begin_step(self, device):
        if not self.enabled:
            return
        used, want = self.off.get(device, 0), self.want.get(device, 0)
        buf = self.buf.get(device)
        if buf is None or want > buf.numel():
            self.buf[device] = torch.zeros(max(int(want * 1.25), 1 << 20), dtype=torch.uint8, device=device)
        elif used and not self.clean.pop(device, False):
            buf[:used].zero_()
        self.clean.pop(device, None)
        self.off[device], self.want[device] = 0, 0

    def end_step(self, device):
        """-> the part of the arena this step dirtied (or None), for a caller that clears it itself at the END of the step (the fused Adam
        launch does, ops.adam_step_groups_(zero=...)): the next begin_step then issues no fill.  Nothing may take arena memory between
        this call and the next begin_step."""
        device = torch.device(device)
        buf, used = self.buf.get(device), self.off.get(device, 0)
        if not self.enabled or buf is None or not used or self.want.get(device, 0) > buf.numel():
            return None
        self.clean[device] = True
        return buf[:min((used + 15) // 16 * 16, buf.numel())]

    def zeros(self, shape, dtype, device):
        device = torch.device(device)
        if isinstance(shape, int):
            shape = (shape,)
        n = math.prod(shape) * torch.empty((), dtype=dtype).element_size()
        n_al = (n + 255) & ~255
        if self.enabled:
            self.want[device] = self.want.get(device, 0) + n_al
            buf, off = self.buf.get(device), self.off.get(device, 0)
            if buf is not None and off + n_al <= buf.numel():
                self.off[device] = off + n_al
                return buf[off:off + n].view(dtype).view(shape)
        return torch.zeros(shape, dtype=dtype, device=device)

    def zeros_like(self, t):
        return self.zeros(tuple(t.shape), t.dtype, t.device)


ARENA = ZeroArena()


def _bg_ptr(bg):
    """background colour: HOST pointer to 3 floats (include/dbw_hip.h) -- a ctypes array kept alive by the caller."""
    return 0 if bg is None else ctypes.cast(bg, ctypes.c_void_p).value


def make_bg(color):
    return (ctypes.c_float * 3)(*[float(c) for c in color])


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _chk(t, dtype, name):
    if not t.is_cuda:
        raise RuntimeError(f'{name} must live on the GPU: the dbw_amd render path has no CPU implementation')
    if t.dtype != dtype:
        raise TypeError(f'{name}: expected {dtype}, got {t.dtype}')
    return t.contiguous()


# ---------------------------------------------------------------------------------------------------------------------
# rasterize_meshes  (operator-level drop-in, int64 indices at the Python boundary like PyTorch3D)
# ---------------------------------------------------------------------------------------------------------------------
def _workspace_bytes(Ft, N, H, W):
    lib = _lib.load()
    return lib.dbw_rasterize_workspace_bytes_binned(Ft, N, H, W) if COARSE_BINS else lib.dbw_rasterize_workspace_bytes(Ft)


def _raster_fwd(face_verts, first, num, neighbor, N, H, W, K, blur, pc, cb, cull, need_zbuf=True):
    dev = face_verts.device
    Ft = face_verts.shape[0]
    ws_bytes = _workspace_bytes(Ft, N, H, W)
    ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
    p2f = torch.empty(N, H, W, K, dtype=torch.int32, device=dev)
    zbuf = torch.empty(N, H, W, K, dtype=torch.float32, device=dev) if need_zbuf else None
    bary = torch.empty(N, H, W, K, 3, dtype=torch.float32, device=dev)
    dists = torch.empty(N, H, W, K, dtype=torch.float32, device=dev)
    _lib.call('dbw_rasterize_fwd', _ptr(face_verts), _ptr(first), _ptr(num), _ptr(neighbor), N, Ft, H, W, K, float(blur),
              int(pc), int(cb), int(cull), _ptr(p2f), _ptr(zbuf), _ptr(bary), _ptr(dists), _ptr(ws), ws_bytes, _stream(face_verts))
    return p2f, zbuf, bary, dists


class _RasterizeMeshes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, face_verts, first, num, neighbor, image_size, blur, K, pc, cb, cull):
        H, W = image_size
        fv = _chk(face_verts.detach(), torch.float32, 'face_verts')
        p2f, zbuf, bary, dists = _raster_fwd(fv, first, num, neighbor, first.numel(), H, W, K, blur, pc, cb, cull)
        ctx.save_for_backward(fv, p2f)
        ctx.cfg = (pc, cb)
        p2f64 = p2f.long()
        ctx.mark_non_differentiable(p2f64)
        return p2f64, zbuf, bary, dists

    @staticmethod
    def backward(ctx, _g, g_zbuf, g_bary, g_dists):
        fv, p2f = ctx.saved_tensors
        pc, cb = ctx.cfg
        N, H, W, K = p2f.shape
        g = torch.zeros_like(fv)
        gz = None if g_zbuf is None else g_zbuf.contiguous()
        gb = None if g_bary is None else g_bary.contiguous()
        gd = None if g_dists is None else g_dists.contiguous()
        _lib.call('dbw_rasterize_bwd', _ptr(fv), _ptr(p2f), _ptr(gz), _ptr(gb), _ptr(gd), N, fv.shape[0], H, W, K, int(pc),
                  int(cb), _ptr(g), _stream(fv))
        return g, None, None, None, None, None, None, None, None, None


def rasterize_meshes(face_verts, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx, image_size, blur_radius,
                     faces_per_pixel, bin_size=0, max_faces_per_bin=0, perspective_correct=False, clip_barycentric_coords=False,
                     cull_backfaces=False):
    """Positional drop-in for PyTorch3D 0.7.1's `_C.rasterize_meshes` (pytorch3d/renderer/mesh/rasterize_meshes.py; reached by the
    reference through src/model/renderer.py:53-54,92-94): the same twelve arguments in the same order.
    face_verts (F,3,3) packed NDC faces of N meshes -> (pix_to_face int64, zbuf, bary_coords, dists), each (N,H,W,K[,3]), -1 where
    empty; gradients flow to face_verts through zbuf, bary_coords and dists (autograd calls the backward kernel; the stand-alone
    `rasterize_meshes_backward` below is the `_C.rasterize_meshes_backward` twin).
    bin_size / max_faces_per_bin select PyTorch3D's coarse-to-fine CUDA path and size its bins; they never change its result unless a
    bin overflows (which PyTorch3D reports as an error).  This rasteriser has its own two-level binning without capacity limits, so
    both are accepted (None or any int >= 0) and ignored: the result is always the naive rasterisation (bin_size = 0)."""
    if isinstance(image_size, int):
        image_size = (image_size, image_size)
    if bin_size is not None and bin_size < 0:
        raise ValueError('bin_size must be >= 0')
    if max_faces_per_bin is not None and max_faces_per_bin < 0:
        raise ValueError('max_faces_per_bin must be >= 0')
    if faces_per_pixel > MAX_FACES_PER_PIXEL:
        raise ValueError(f'faces_per_pixel={faces_per_pixel} > {MAX_FACES_PER_PIXEL}')
    first = _chk(mesh_to_face_first_idx.to(torch.int32), torch.int32, 'mesh_to_face_first_idx')
    num = _chk(num_faces_per_mesh.to(torch.int32), torch.int32, 'num_faces_per_mesh')
    nb = None if clipped_faces_neighbor_idx is None else _chk(clipped_faces_neighbor_idx.to(torch.int32), torch.int32, 'neighbor')
    return _RasterizeMeshes.apply(face_verts, first, num, nb, tuple(image_size), float(blur_radius), int(faces_per_pixel),
                                  bool(perspective_correct), bool(clip_barycentric_coords), bool(cull_backfaces))


def rasterize_meshes_backward(face_verts, pix_to_face, grad_zbuf, grad_bary, grad_dists, perspective_correct, clip_barycentric_coords):
    """Positional drop-in for PyTorch3D 0.7.1's `_C.rasterize_meshes_backward`: face_verts (F,3,3), pix_to_face (N,H,W,K) int64 (or
    int32), grad_zbuf / grad_dists (N,H,W,K), grad_bary (N,H,W,K,3) -> grad_face_verts (F,3,3)."""
    fv = _chk(face_verts.detach(), torch.float32, 'face_verts')
    p2f = _chk(pix_to_face.to(torch.int32), torch.int32, 'pix_to_face')
    N, H, W, K = p2f.shape
    gz, gb, gd = [_chk(t.detach(), torch.float32, n) for t, n in ((grad_zbuf, 'grad_zbuf'), (grad_bary, 'grad_bary'), (grad_dists, 'grad_dists'))]
    if gz.shape != p2f.shape or gd.shape != p2f.shape or gb.shape != p2f.shape + (3,):
        raise ValueError('gradient shapes must be (N,H,W,K), (N,H,W,K,3), (N,H,W,K)')
    g = torch.zeros_like(fv)
    _lib.call('dbw_rasterize_bwd', _ptr(fv), _ptr(p2f), _ptr(gz), _ptr(gb), _ptr(gd), N, fv.shape[0], H, W, K, int(bool(perspective_correct)),
              int(bool(clip_barycentric_coords)), _ptr(g), _stream(fv))
    return g


# ---------------------------------------------------------------------------------------------------------------------
# project + clip
# ---------------------------------------------------------------------------------------------------------------------
def project_clip(verts, faces_i32, R, T, Kmat, eps=1e-8, z_clip=0.001, perspective_correct=True):
    """-> dict of device tensors (no grad): face_verts (B,2F,3,3), first_idx, num_faces, c2o, neighbor, clip_code, clip_w."""
    dev = verts.device
    B, V, F_ = R.shape[0], verts.shape[0], faces_i32.shape[0]
    out = dict(face_verts=torch.empty(B, 2 * F_, 3, 3, dtype=torch.float32, device=dev),
               first_idx=torch.empty(B, dtype=torch.int32, device=dev), num_faces=torch.empty(B, dtype=torch.int32, device=dev),
               c2o=torch.empty(B, 2 * F_, dtype=torch.int32, device=dev), neighbor=torch.empty(B, 2 * F_, dtype=torch.int32, device=dev),
               clip_code=torch.empty(B, 2 * F_, dtype=torch.int32, device=dev),
               clip_w=torch.empty(B, 2 * F_, 2, dtype=torch.float32, device=dev))
    _lib.call('dbw_project_clip_fwd', _ptr(verts), _ptr(faces_i32), _ptr(R), _ptr(T), _ptr(Kmat), B, V, F_, float(eps),
              int(z_clip is not None), float(z_clip or 0.0), int(perspective_correct), _ptr(out['face_verts']),
              _ptr(out['first_idx']), _ptr(out['num_faces']), _ptr(out['c2o']), _ptr(out['neighbor']), _ptr(out['clip_code']),
              _ptr(out['clip_w']), _stream(verts))
    return out


def project_clip_bwd(verts, faces_i32, R, T, Kmat, cl, g_face_verts, eps=1e-8, z_clip=0.001, perspective_correct=True):
    B, V, F_ = R.shape[0], verts.shape[0], faces_i32.shape[0]
    g = ARENA.zeros_like(verts)
    _lib.call('dbw_project_clip_bwd', _ptr(verts), _ptr(faces_i32), _ptr(R), _ptr(T), _ptr(Kmat), B, V, F_, float(eps),
              float(z_clip or 0.0), int(perspective_correct), _ptr(cl['num_faces']), _ptr(cl['c2o']), _ptr(cl['clip_code']),
              _ptr(cl['clip_w']), _ptr(g_face_verts), _ptr(g), _stream(verts))
    return g


# ---------------------------------------------------------------------------------------------------------------------
# shade + blend on given fragments
# ---------------------------------------------------------------------------------------------------------------------
def _alpha_len(faces_alpha, map_desc, F_):
    """alpha_len of the C ABI: F or N*F for per-face opacities, -M for one opacity per texture map (include/dbw_hip.h)."""
    if faces_alpha is None:
        return 0
    n = faces_alpha.numel()
    return -n if (n == map_desc.shape[0] and n != F_) else n


def _shade_args(p2f, bary, dists, cl, face_uvs, face_map, map_desc, maps, faces_alpha, F_, sigma, bg, dims=None):
    N, H, W, K = p2f.shape if dims is None else dims      # dims: explicit (N,H,W,K) when the fragments use the tiled layout
    c2o, code, cw = (cl['c2o'], cl['clip_code'], cl['clip_w']) if cl is not None else (None, None, None)
    return (_ptr(p2f), _ptr(bary), _ptr(dists), _ptr(c2o), _ptr(code), _ptr(cw), 2 * F_, _ptr(face_uvs), _ptr(face_map),
            _ptr(map_desc), _ptr(maps), _ptr(faces_alpha), _alpha_len(faces_alpha, map_desc, F_), N, H, W, K, F_,
            float(sigma), _bg_ptr(bg))


def shade_blend_fwd(p2f, bary, dists, cl, face_uvs, face_map, map_desc, maps, faces_alpha, F_, sigma, bg):
    N, H, W, K = p2f.shape
    img = torch.empty(N, 4, H, W, dtype=torch.float32, device=p2f.device)
    _lib.call('dbw_shade_blend_fwd', *_shade_args(p2f, bary, dists, cl, face_uvs, face_map, map_desc, maps, faces_alpha, F_, sigma, bg),
              _ptr(img), _stream(p2f))
    return img


def shade_blend_bwd(p2f, bary, dists, cl, face_uvs, face_map, map_desc, maps, faces_alpha, F_, sigma, bg, g_img,
                    want_dists=True, want_bary=False, lds_aggregate=False):
    dev = p2f.device
    N, H, W, K = p2f.shape
    g_maps = torch.zeros_like(maps)
    g_alpha = torch.zeros_like(faces_alpha) if faces_alpha is not None else None
    g_dists = torch.empty(N, H, W, K, dtype=torch.float32, device=dev) if want_dists else None
    g_bary = torch.empty(N, H, W, K, 3, dtype=torch.float32, device=dev) if want_bary else None
    _lib.call('dbw_shade_blend_bwd', *_shade_args(p2f, bary, dists, cl, face_uvs, face_map, map_desc, maps, faces_alpha, F_, sigma, bg),
              _ptr(g_img), _ptr(g_maps), _ptr(g_alpha), _ptr(g_dists), _ptr(g_bary), int(lds_aggregate), _stream(p2f))
    return g_maps, g_alpha, g_dists, g_bary


# ---------------------------------------------------------------------------------------------------------------------
# The whole Renderer.forward as ONE autograd node: verts/maps/faces_alpha -> (B,4,H,W)
# ---------------------------------------------------------------------------------------------------------------------
class RenderCfg:
    __slots__ = ('H', 'W', 'K', 'sigma', 'blur', 'z_clip', 'persp', 'detach_bary', 'eps', 'F', 'lds_aggregate', 'texbins', 'const_faces', 'bin_demand')

    def __init__(self, H, W, K, sigma, z_clip, persp, detach_bary, F_, eps=1e-8, lds_aggregate=False, texbins=None, const_faces=0, clip_inside=True):
        self.lds_aggregate = lds_aggregate
        self.const_faces = int(const_faces)   # the first that many faces have constant vertices (sky dome): no geometry gradient for them
        self.texbins = texbins          # (bin_base, bin_info, nbins): texture-space binning of texel gradients when not aggregating
        self.bin_demand = None          # optional BinDemand of the caller (one per pass, kept across steps): record sub-ranges by demand
        # sigma as the library takes it: > 0 exp(-max(d, 0) / sigma), 0 hard, < 0 sigmoid(-d / |sigma|) = clip_inside False (renderer.py:257-258)
        self.H, self.W, self.K, self.sigma, self.z_clip, self.persp = H, W, K, float(sigma) if clip_inside else -float(sigma), z_clip, persp
        self.blur = math.log(1. / 1e-4 - 1.) * float(sigma)            # renderer.py:51
        self.detach_bary, self.eps, self.F = detach_bary, eps, F_


HARD_UV_FRAGMENTS = True  # hard single-layer passes (K = 1, sigma = 0) store (u, v, face|map) per pixel: frag_layout 3


def hard_layout(cfg, fa, map_desc):
    """frag_layout of a fused pass that keeps barycentric fragments by default: 3 for the hard single-layer pass whose backward goes
    through the LDS texel table (include/dbw_hip.h), else 1."""
    ok = (HARD_UV_FRAGMENTS and cfg.K == 1 and cfg.sigma == 0.0 and fa is None and cfg.lds_aggregate and cfg.F < (1 << 20)
          and map_desc.shape[0] < (1 << 11))
    return 3 if ok else 1


def tiled_image_empty(B, C, H, W, device):
    """Uninitialised image in the 8x8-tile planar layout of include/dbw_hip.h (image_layout 1): (B, ceil(H/8), ceil(W/8), C, 64)."""
    return torch.empty(B, (H + 7) // 8, (W + 7) // 8, C, 64, dtype=torch.float32, device=device)


def tile_image(img):
    """(B, C, H, W) -> the 8x8-tile planar layout (B, ceil(H/8), ceil(W/8), C, 64); rows / columns beyond the image are zero."""
    B, C, H, W = img.shape
    ty, tx = (H + 7) // 8, (W + 7) // 8
    if (H, W) != (ty * 8, tx * 8):
        img = torch.nn.functional.pad(img, (0, tx * 8 - W, 0, ty * 8 - H))
    return img.reshape(B, C, ty, 8, tx, 8).permute(0, 2, 4, 1, 3, 5).reshape(B, ty, tx, C, 64).contiguous()      # (reshape: imgs may be a strided view)


def untile_image(t, H, W):
    """Inverse of tile_image: (B, ty, tx, C, 64) -> (B, C, H, W)."""
    B, ty, tx, C, _ = t.shape
    return t.view(B, ty, tx, C, 8, 8).permute(0, 3, 1, 4, 2, 5).reshape(B, C, ty * 8, tx * 8)[:, :, :H, :W].contiguous()


def _render_fwd_fused(fvc, cl, B, cfg, face_uvs, face_map, map_desc, maps, fa, bg, tiled=None, stage=0, state=None, img_tiled=False):
    """stage 1: only the per-face set-up of the pass (needs no texture values) -> `state` for the stage-2 call that renders."""
    TILED_FRAGMENTS = globals()['TILED_FRAGMENTS'] if tiled is None else tiled
    dev = fvc.device
    Ft = fvc.shape[0]
    if stage == 2:
        ws, ws_bytes, out = state
    else:
        ws_bytes = _workspace_bytes(Ft, B, cfg.H, cfg.W)
        ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
        if TILED_FRAGMENTS:     # internal 8x8-tile planar layouts (include/dbw_hip.h: frag_layout 1, 2, 3)
            ty, tx = (cfg.H + 7) // 8, (cfg.W + 7) // 8
            p2f = torch.empty(B, ty, tx, cfg.K, 64, dtype=torch.int32, device=dev)
            bary = torch.empty(B, ty, tx, cfg.K, 8 if int(TILED_FRAGMENTS) == 2 else 3, 64, dtype=torch.float32, device=dev)
            dists = torch.empty(B, ty, tx, cfg.K, 64, dtype=torch.float32, device=dev)
        else:
            p2f = torch.empty(B, cfg.H, cfg.W, cfg.K, dtype=torch.int32, device=dev)
            bary = torch.empty(B, cfg.H, cfg.W, cfg.K, 3, dtype=torch.float32, device=dev)
            dists = torch.empty(B, cfg.H, cfg.W, cfg.K, dtype=torch.float32, device=dev)
        img = tiled_image_empty(B, 4, cfg.H, cfg.W, dev) if img_tiled else torch.empty(B, 4, cfg.H, cfg.W, dtype=torch.float32, device=dev)
        out = (p2f, bary, dists, img)
    p2f, bary, dists, img = out
    if int(TILED_FRAGMENTS) == 2 and maps.numel() >= (1 << 30):
        raise RuntimeError(f'uv-fragment passes address their texels with 32-bit byte offsets: the maps buffer ({maps.numel()} floats) must hold fewer than 2^30')
    _lib.call('dbw_render_fwd_fused', _ptr(fvc), _ptr(cl['first_idx']), _ptr(cl['num_faces']), _ptr(cl['neighbor']), _ptr(cl['c2o']),
              _ptr(cl['clip_code']), _ptr(cl['clip_w']), 2 * cfg.F, _ptr(face_uvs), _ptr(face_map), _ptr(map_desc), _ptr(maps), _ptr(fa),
              _alpha_len(fa, map_desc, cfg.F), B, Ft, cfg.H, cfg.W, cfg.K, cfg.F, float(cfg.sigma), float(cfg.blur), int(cfg.persp),
              _bg_ptr(bg), _ptr(p2f), _ptr(bary), _ptr(dists), _ptr(img), _ptr(ws), ws_bytes, int(TILED_FRAGMENTS), int(stage),
              int(img_tiled), _stream(fvc))   # frag_layout 0 / 1 / 2
    return (ws, ws_bytes, out) if stage == 1 else out


class _RenderScene(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, maps, faces_alpha, faces_i32, R, T, Kmat, face_uvs, face_map, map_desc, bg, cfg):
        verts_c = _chk(verts.detach(), torch.float32, 'verts')
        maps_c = _chk(maps.detach(), torch.float32, 'maps')
        fa = None if faces_alpha is None else _chk(faces_alpha.detach(), torch.float32, 'faces_alpha')
        B = R.shape[0]
        cl = project_clip(verts_c, faces_i32, R, T, Kmat, cfg.eps, cfg.z_clip, cfg.persp)
        fvc = cl['face_verts'].view(-1, 3, 3)
        ctx.tiled = int(FUSED_FORWARD and FUSED_BACKWARD and TILED_FRAGMENTS)
        if ctx.tiled and UV_FRAGMENTS and cfg.detach_bary and cfg.F < (1 << 20) and map_desc.shape[0] < (1 << 11):
            ctx.tiled = 2          # fragments carry (u, v, face|map) instead of barycentrics
        elif ctx.tiled:
            ctx.tiled = hard_layout(cfg, fa, map_desc)
        if FUSED_FORWARD:
            p2f, bary, dists, img = _render_fwd_fused(fvc, cl, B, cfg, face_uvs, face_map, map_desc, maps_c, fa, bg, ctx.tiled)
        else:
            p2f, _, bary, dists = _raster_fwd(fvc, cl['first_idx'], cl['num_faces'], cl['neighbor'].view(-1), B, cfg.H, cfg.W, cfg.K,
                                              cfg.blur, cfg.persp, True, False, need_zbuf=False)
            img = shade_blend_fwd(p2f, bary, dists, cl, face_uvs, face_map, map_desc, maps_c, fa, cfg.F, cfg.sigma, bg)
        ctx.cfg, ctx.cl = cfg, cl
        ctx.has_alpha = fa is not None
        ctx.bg = bg
        ctx.save_for_backward(verts_c, maps_c, fa if fa is not None else verts_c.new_empty(0), faces_i32, R, T, Kmat, face_uvs,
                              face_map, map_desc, p2f, bary, dists)
        return img

    @staticmethod
    def backward(ctx, g_img):
        verts, maps, fa, faces_i32, R, T, Kmat, face_uvs, face_map, map_desc, p2f, bary, dists = ctx.saved_tensors
        cfg, cl, bg = ctx.cfg, ctx.cl, ctx.bg
        fa = fa if ctx.has_alpha else None
        need_geom = ctx.needs_input_grad[0]
        want_dists = need_geom and cfg.sigma != 0
        want_bary = need_geom and not cfg.detach_bary
        if FUSED_BACKWARD and (ctx.tiled or (need_geom and (want_dists or want_bary))):
            g_maps, g_alpha, g_fvc = _fused_bwd(p2f, bary, dists, cl, face_uvs, face_map, map_desc, maps, fa, cfg, bg, ctx.tiled, g_img.contiguous(),
                                                R.shape[0], None)
            g_verts = project_clip_bwd(verts, faces_i32, R, T, Kmat, cl, g_fvc, cfg.eps, cfg.z_clip, cfg.persp) if need_geom else None
            return g_verts, g_maps, g_alpha, None, None, None, None, None, None, None, None, None
        if ctx.tiled:
            raise RuntimeError('tiled fragments can only be consumed by the fused backward')
        g_maps, g_alpha, g_dists, g_bary = shade_blend_bwd(p2f, bary, dists, cl, face_uvs, face_map, map_desc, maps, fa, cfg.F,
                                                           cfg.sigma, bg, g_img.contiguous(), want_dists, want_bary, cfg.lds_aggregate)
        g_verts = None
        if need_geom and (want_dists or want_bary):
            fvc = cl['face_verts'].view(-1, 3, 3)
            g_fvc = torch.zeros_like(fvc)
            B = R.shape[0]
            _lib.call('dbw_rasterize_bwd', _ptr(fvc), _ptr(p2f), 0, _ptr(g_bary), _ptr(g_dists), B, fvc.shape[0], cfg.H, cfg.W,
                      cfg.K, int(cfg.persp), 1, _ptr(g_fvc), _stream(fvc))
            g_verts = project_clip_bwd(verts, faces_i32, R, T, Kmat, cl, g_fvc, cfg.eps, cfg.z_clip, cfg.persp)
        elif need_geom:
            g_verts = torch.zeros_like(verts)
        return g_verts, g_maps, g_alpha, None, None, None, None, None, None, None, None, None


_UNIFORM_LAYOUTS = {}


def uniform_bin_layout(nbins, cap, device):
    """The layout table of equal shares (include/dbw_hip.h: bin_layout): nbins * cap records divided evenly over the sub-ranges;
    computed once per geometry (dbw_bin_layout on an all-zero demand) and kept."""
    device = torch.device(device)
    key = (nbins, cap, device)
    lay = _UNIFORM_LAYOUTS.get(key)
    if lay is None:
        n = nbins * bin_subcursors()
        lay = torch.zeros(n, 2, dtype=torch.int32, device=device)
        with torch.cuda.device(device):
            _lib.call('dbw_bin_layout', _ptr(torch.zeros(n, dtype=torch.int32, device=device)), n, float(nbins) * float(cap), 1, _ptr(lay),
                      torch.cuda.current_stream(device).cuda_stream)
        _UNIFORM_LAYOUTS[key] = lay
    return lay


class BinDemand:
    """Record sub-ranges of the texture bins sized by demand (include/dbw_hip.h: bin_layout, dbw_bin_layout).  After a backward pass
    bin_cursor holds how many records every sub-range was asked for; a caller that runs the same pass step after step
    (native_step.py) keeps one of these per pass, and the next launch divides the SAME total -- texbin_capacity(...) records per bin on
    average -- in proportion to that demand (+ 25 %, at least MIN records each) instead of in equal shares.  One small kernel on the
    device, no host read, the record buffer keeps its size.  With equal shares the hot bins of a large scene overflow into the atomic
    fallback: config 5's backward took 56 ms at full resolution, 9.6 ms with four times the memory.
    Two cursor arrays are kept and used in turn (this launch's cursors are the next launch's demand); they are never reallocated, so a
    captured hipGraph that replays one launch keeps reading a valid -- if frozen -- demand."""
    MIN = 64

    def __init__(self):
        self.key, self.cursors, self.layouts, self.turn, self.ready = None, None, None, 0, False

    def _fit(self, nbins, device):
        n = nbins * bin_subcursors()
        if self.key != (n, device):
            self.key = (n, device)
            self.cursors = [torch.zeros(n, dtype=torch.int32, device=device) for _ in range(2)]
            self.layouts = [torch.zeros(n, 2, dtype=torch.int32, device=device) for _ in range(2)]
            self.turn, self.ready = 0, False
        return n

    def prepare(self, nbins, cap, device):
        """Enqueue, on the current stream, the layout of the NEXT launch from the cursors of the previous one (call it early in the
        step, off the critical path; begin() does it itself otherwise)."""
        n = self._fit(nbins, device)
        if not self.ready:
            return None
        lay = self.layouts[self.turn]
        _lib.call('dbw_bin_layout', _ptr(self.cursors[1 - self.turn]), n, float(nbins) * float(cap), self.MIN, _ptr(lay),
                  torch.cuda.current_stream(device).cuda_stream)
        self.prepared = (nbins, cap, lay)
        return lay

    def begin(self, nbins, cap, device):
        """-> (zeroed cursors of this launch, layout or None for the first launch / after a change of geometry)."""
        self._fit(nbins, device)
        lay = None
        if self.ready:
            pre = getattr(self, 'prepared', None)
            lay = pre[2] if (pre is not None and pre[:2] == (nbins, cap) and pre[2] is self.layouts[self.turn]) else self.prepare(nbins, cap, device)
        self.prepared = None
        cur = self.cursors[self.turn]
        cur.zero_()
        self.turn, self.ready = 1 - self.turn, True
        return cur, lay


def _fused_bwd(p2f, bary, dists, cl, face_uvs, face_map, map_desc, maps, fa, cfg, bg, tiled, g_img, B, gscale, after_kernel=None, img_tiled=False,
               bin_demand=None):
    """dbw_render_bwd_fused (+ dbw_texbin_reduce when the texel gradients go through texture-space bins) of one pass.
    gscale: device scalar multiplying g_img inside the kernel (or None).  -> grad maps, grad faces_alpha (or None), grad face_verts_c.
    bin_demand: a BinDemand kept by the caller across steps -> the bins' record sub-ranges follow the previous step's demand."""
    fvc = cl['face_verts'].view(-1, 3, 3)
    per_map = fa is not None and _alpha_len(fa, map_desc, cfg.F) < 0        # then 64 partial sums per opacity (include/dbw_hip.h)
    g_maps = ARENA.zeros_like(maps)
    g_alpha = None if fa is None else (ARENA.zeros(fa.numel() * ALPHA_SPREAD, torch.float32, fa.device) if per_map else ARENA.zeros_like(fa))
    g_fvc = ARENA.zeros_like(fvc)
    bins = cfg.texbins if (TEXTURE_BINS and not cfg.lds_aggregate) else None
    bin_base = cursor = records = layout = None
    cap = 0
    if bins is not None and bins[2] > 0:
        bin_base, bin_info, nbins = bins
        cap = texbin_capacity(B, cfg.H, cfg.W, cfg.K, nbins)
        if bin_demand is None:
            bin_demand = cfg.bin_demand
        if bin_demand is not None:
            cursor, layout = bin_demand.begin(nbins, cap, fvc.device)
        else:
            cursor = ARENA.zeros(nbins * bin_subcursors(), torch.int32, fvc.device)
        if layout is None:
            layout = uniform_bin_layout(nbins, cap, fvc.device)         # (first launch, or a caller that keeps no demand: equal shares)
        records = torch.empty(nbins * cap * 8, dtype=torch.int32, device=fvc.device)
    _lib.call('dbw_render_bwd_fused', *_shade_args(p2f, bary, dists, cl, face_uvs, face_map, map_desc, maps, fa, cfg.F, cfg.sigma, bg,
                                                   (B, cfg.H, cfg.W, cfg.K)),
              _ptr(g_img), _ptr(fvc), int(cfg.persp), int(cfg.detach_bary), _ptr(g_maps), _ptr(g_alpha), _ptr(g_fvc),
              int(cfg.lds_aggregate), int(tiled), _ptr(bin_base), _ptr(cursor), _ptr(records), cap, _ptr(layout), int(cfg.const_faces),
              _ptr(gscale), int(img_tiled), _stream(fvc))
    if after_kernel is not None:
        after_kernel()            # (the big kernel is enqueued; the bin reduction, a low-occupancy kernel, may share the GPU with other work)
    if records is not None:
        _lib.call('dbw_texbin_reduce', _ptr(bin_info), _ptr(cursor), _ptr(records), cap, _ptr(layout), nbins, _ptr(g_maps), _stream(fvc))
    return g_maps, g_alpha, g_fvc


def render_fwd_fused_mse(cl, B, cfg, face_uvs, face_map, map_desc, maps, fa, bg, env_img, imgs, scale, stage=0, state=None, img_tiled=False):
    """dbw_render_fwd_fused_mse on the clipped faces `cl` of the fg scene (no grad bookkeeping): uv-fragments + per-tile sums of
    squared differences + d loss / d fg image, d loss / d env image.  -> p2f, bary, dists, part, g_fg, g_env.
    stage 1 (env_img / imgs may be None): only the per-face set-up, on the current stream -> `state` for the stage-2 call that
    renders (the caller orders the two calls)."""
    fvc = cl['face_verts'].view(-1, 3, 3)
    dev, Ft = fvc.device, fvc.shape[0]
    if stage == 2:
        ws, ws_bytes, out = state
    else:
        ws_bytes = _workspace_bytes(Ft, B, cfg.H, cfg.W)
        ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
        ty, tx = (cfg.H + 7) // 8, (cfg.W + 7) // 8
        p2f = torch.empty(B, ty, tx, cfg.K, 64, dtype=torch.int32, device=dev)
        bary = torch.empty(B, ty, tx, cfg.K, 8, 64, dtype=torch.float32, device=dev)
        dists = torch.empty(B, ty, tx, cfg.K, 64, dtype=torch.float32, device=dev)
        part = torch.empty(B * ty * tx, dtype=torch.float32, device=dev)
        g_fg = tiled_image_empty(B, 4, cfg.H, cfg.W, dev) if img_tiled else torch.empty(B, 4, cfg.H, cfg.W, dtype=torch.float32, device=dev)
        g_env = torch.empty_like(g_fg)
        out = (p2f, bary, dists, part, g_fg, g_env)
    p2f, bary, dists, part, g_fg, g_env = out
    if maps.numel() >= (1 << 30):
        raise RuntimeError(f'uv-fragment passes address their texels with 32-bit byte offsets: the maps buffer ({maps.numel()} floats) must hold fewer than 2^30')
    _lib.call('dbw_render_fwd_fused_mse', _ptr(fvc), _ptr(cl['first_idx']), _ptr(cl['num_faces']), _ptr(cl['neighbor']), _ptr(cl['c2o']),
              _ptr(cl['clip_code']), _ptr(cl['clip_w']), 2 * cfg.F, _ptr(face_uvs), _ptr(face_map), _ptr(map_desc), _ptr(maps), _ptr(fa),
              _alpha_len(fa, map_desc, cfg.F), B, Ft, cfg.H, cfg.W, cfg.K, cfg.F, float(cfg.sigma), float(cfg.blur), int(cfg.persp),
              _bg_ptr(bg), _ptr(p2f), _ptr(bary), _ptr(dists), _ptr(ws), ws_bytes, _ptr(env_img), _ptr(imgs), float(scale), _ptr(part),
              _ptr(g_fg), _ptr(g_env), int(stage), int(img_tiled), _stream(fvc))
    return (ws, ws_bytes, out) if stage == 1 else out


class _DecoupledRenderMSE(torch.autograd.Function):
    """The whole decoupled training render + reconstruction loss (dbw.py:213-223,366-367) as ONE autograd node:
    env pass (sky + ground, hard, 1 face per pixel) -> fg pass (blocks, soft, K faces per pixel) whose epilogue composites over the
    env image and takes the MSE against the targets on registers (dbw_render_fwd_fused_mse) -> scalar loss.  Neither the fg image
    nor the composite ever exists in memory; the two backward passes read the per-pixel loss gradients the forward left behind,
    scaled inside the kernels by the upstream gradient (a device scalar: no host sync, no elementwise pass)."""

    @staticmethod
    def forward(ctx, verts_e, maps_e, verts_f, maps_f, alpha, imgs, scale, R, T, Kmat, env_tab, fg_tab, cfg_e, cfg_f, bg_e, bg_f):
        ve, me = _chk(verts_e.detach(), torch.float32, 'env verts'), _chk(maps_e.detach(), torch.float32, 'env maps')
        vf, mf = _chk(verts_f.detach(), torch.float32, 'fg verts'), _chk(maps_f.detach(), torch.float32, 'fg maps')
        fa = None if alpha is None else _chk(alpha.detach(), torch.float32, 'faces_alpha')
        imgs = _chk(imgs, torch.float32, 'imgs')
        faces_e, uv_e, fmap_e, desc_e = env_tab
        faces_f, uv_f, fmap_f, desc_f = fg_tab
        B, dev = R.shape[0], ve.device
        # env pass: barycentric fragments (layout 1), image kept (the fg epilogue reads it)
        cl_e = project_clip(ve, faces_e, R, T, Kmat, cfg_e.eps, cfg_e.z_clip, cfg_e.persp)
        lay_e = ctx.lay_e = hard_layout(cfg_e, None, desc_e)
        p2f_e, bary_e, dists_e, img_e = _render_fwd_fused(cl_e['face_verts'].view(-1, 3, 3), cl_e, B, cfg_e, uv_e, fmap_e, desc_e, me, None, bg_e, lay_e)
        # fg pass with the composite + MSE epilogue
        cl_f = project_clip(vf, faces_f, R, T, Kmat, cfg_f.eps, cfg_f.z_clip, cfg_f.persp)
        p2f, bary, dists, part, g_fg, g_env = render_fwd_fused_mse(cl_f, B, cfg_f, uv_f, fmap_f, desc_f, mf, fa, bg_f, img_e, imgs, scale)
        ctx.save_for_backward(ve, me, vf, mf, fa if fa is not None else ve.new_empty(0), R, T, Kmat, p2f_e, bary_e, dists_e, p2f, bary, dists, g_fg, g_env)
        ctx.misc = (env_tab, fg_tab, cfg_e, cfg_f, bg_e, bg_f, cl_e, cl_f, fa is not None)
        return part.sum() * float(scale)

    @staticmethod
    def backward(ctx, go):
        ve, me, vf, mf, fa, R, T, Kmat, p2f_e, bary_e, dists_e, p2f, bary, dists, g_fg, g_env = ctx.saved_tensors
        env_tab, fg_tab, cfg_e, cfg_f, bg_e, bg_f, cl_e, cl_f, has_alpha = ctx.misc
        ctx.misc = None
        fa = fa if has_alpha else None
        faces_e, uv_e, fmap_e, desc_e = env_tab
        faces_f, uv_f, fmap_f, desc_f = fg_tab
        B = R.shape[0]
        gs = go.detach().to(torch.float32).reshape(1).contiguous()
        gm_f, ga, g_fvc_f = _fused_bwd(p2f, bary, dists, cl_f, uv_f, fmap_f, desc_f, mf, fa, cfg_f, bg_f, 2, g_fg, B, gs)
        gv_f = project_clip_bwd(vf, faces_f, R, T, Kmat, cl_f, g_fvc_f, cfg_f.eps, cfg_f.z_clip, cfg_f.persp) if ctx.needs_input_grad[2] else None
        gm_e, _, g_fvc_e = _fused_bwd(p2f_e, bary_e, dists_e, cl_e, uv_e, fmap_e, desc_e, me, None, cfg_e, bg_e, ctx.lay_e, g_env, B, gs)
        gv_e = project_clip_bwd(ve, faces_e, R, T, Kmat, cl_e, g_fvc_e, cfg_e.eps, cfg_e.z_clip, cfg_e.persp) if ctx.needs_input_grad[0] else None
        return (gv_e, gm_e, gv_f, gm_f, ga) + (None,) * 11


def render_decoupled_mse(env, fg, alpha, imgs, scale, R, T, Kmat, cfg_e, cfg_f, bg_e, bg_f):
    """env / fg: PackedScene-like (verts, maps, faces, face_uvs, face_map, map_desc); alpha: per-face opacities of fg or None;
    -> scale * sum((imgs - (fg_rgb * mask + (1 - mask) * env_rgb))^2) as a scalar tensor (scale = weight / element count)."""
    if cfg_f.K < 2 or not cfg_f.detach_bary or cfg_e.K != 1:
        raise ValueError('render_decoupled_mse: soft detach_bary fg pass over a hard single-layer env pass')
    return _DecoupledRenderMSE.apply(env.verts, env.maps, fg.verts, fg.maps, alpha, imgs, float(scale), R, T, Kmat,
                                     (env.faces, env.face_uvs, env.face_map, env.map_desc), (fg.faces, fg.face_uvs, fg.face_map, fg.map_desc),
                                     cfg_e, cfg_f, bg_e, bg_f)


def texbin_capacity(B, H, W, K, nbins):
    """Records per texture bin: room for max(K/2, 2) fragments per pixel spread evenly over the bins (a soft K-layer render
    fills ~20 % of its slots, a hard 1-layer render all of them; what does not fit falls back to atomics), at most 16 GiB
    of the 288 GB (config 5 -- 25 views of 1080x1920, K = 16 -- asks for 13 GB)."""
    import os
    cap = int(min(max(int(float(os.environ.get('DBW_CAP_SCALE', 1)) * B * H * W * max(K, 4)) // (2 * nbins), 256), (16 << 30) // (32 * nbins)))
    sub = bin_subcursors()
    return (cap + sub - 1) // sub * sub          # a bin's range = DBW_BIN_SUBCURSORS equal sub-ranges


def render_scene(verts, maps, faces_alpha, faces_i32, R, T, Kmat, face_uvs, face_map, map_desc, bg, cfg):
    """verts (V,3) world, maps flat fp32, faces_alpha None | (F,) | (B*F,) -> image (B,4,H,W)."""
    if cfg.K > MAX_FACES_PER_PIXEL:
        raise ValueError(f'faces_per_pixel={cfg.K} > {MAX_FACES_PER_PIXEL}')
    return _RenderScene.apply(verts, maps, faces_alpha, faces_i32, R, T, Kmat, face_uvs, face_map, map_desc, bg, cfg)


def render_fragments(verts, faces_i32, R, T, Kmat, cfg):
    """Debug/parity helper (no grad): the raw fragments of a render pass, in clipped indexing + the clip tables."""
    cl = project_clip(verts, faces_i32, R, T, Kmat, cfg.eps, cfg.z_clip, cfg.persp)
    fvc = cl['face_verts'].view(-1, 3, 3)
    p2f, zbuf, bary, dists = _raster_fwd(fvc, cl['first_idx'], cl['num_faces'], cl['neighbor'].view(-1), R.shape[0], cfg.H, cfg.W,
                                         cfg.K, cfg.blur, cfg.persp, True, False, need_zbuf=True)
    return cl, p2f, zbuf, bary, dists


# ---------------------------------------------------------------------------------------------------------------------
# texture preparation, param -> mesh, losses, optimiser
# ---------------------------------------------------------------------------------------------------------------------
class _TexturePrep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, texture, decim):
        tex = _chk(texture.detach(), torch.float32, 'texture')
        n, h, w, _ = tex.shape
        maps = torch.empty_like(tex) if decim == 1 else torch.empty(n, h // decim, w // decim, 3, dtype=tex.dtype, device=tex.device)
        sig = torch.empty_like(tex) if decim > 1 else None
        _lib.call('dbw_texture_prep_fwd', _ptr(tex), n, h, w, int(decim), _ptr(maps), _ptr(sig), _stream(tex))
        ctx.save_for_backward(tex)
        ctx.decim = decim
        if sig is None:
            sig = maps
            ctx.alias = True
        else:
            ctx.alias = False
        return maps, sig

    @staticmethod
    def backward(ctx, g_maps, g_sig):
        (tex,) = ctx.saved_tensors
        n, h, w, _ = tex.shape
        if ctx.alias:          # maps and sig are the same tensor: autograd hands two grads for it
            g = None
            for t in (g_maps, g_sig):
                if t is not None:
                    g = t if g is None else g + t
            g_maps, g_sig = g, None
        if g_maps is None:
            d = ctx.decim
            g_maps = torch.zeros(n, h // d, w // d, 3, dtype=tex.dtype, device=tex.device)
        out = torch.empty_like(tex)
        _lib.call('dbw_texture_prep_bwd', _ptr(tex), n, h, w, int(ctx.decim), _ptr(g_maps.contiguous()),
                  _ptr(None if g_sig is None else g_sig.contiguous()), _ptr(out), _stream(tex))
        return out, None


def texture_prep(texture, decim=1):
    """(n,h,w,3) logits -> (maps sampled by the renderer: (n,h/d,w/d,3) cell means when decim > 1 -- pair them with
    `shift = log2(decim)` in the map descriptor --, undecimated sigmoid for the TV loss)."""
    maps, sig = _TexturePrep.apply(texture, int(decim))
    return maps, sig


def sq_blocks(sq_eps, S, R6, T, trig, keep, nb, ratio, scale_min, S_world, R_world, T_world, dense=True):
    """-> world verts of the blocks.  dense=True: (nb,nv,3), only the kept blocks, packed (`nb` = their number, a host int);
    dense=False: (Kb,nv,3), skipped blocks collapsed to a point (no host knowledge of `keep` needed)."""
    if dense and keep is not None and nb == 0:
        return trig.new_empty(0, trig.shape[2], 3)
    if not dense:
        nb = trig.shape[1]
    return _SqBlocks.apply(sq_eps, S, R6, T, trig, keep, nb, (float(ratio), float(scale_min), float(S_world), R_world, T_world, bool(dense)))


class _SqBlocks(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sq_eps, S, R6, T, trig, keep, nb, consts):
        ratio, scale_min, S_world, Rw, Tw, dense = consts
        args = [_chk(t.detach(), torch.float32, 'pose param') for t in (sq_eps, S, R6, T)]
        Kb, nv = trig.shape[1], trig.shape[2]
        verts = torch.empty(nb, nv, 3, dtype=torch.float32, device=trig.device)
        _lib.call('dbw_sq_blocks_fwd', *[_ptr(a) for a in args], _ptr(trig), _ptr(keep), int(dense), Kb, nv, ratio, scale_min, S_world,
                  _ptr(Rw), _ptr(Tw), _ptr(verts), _stream(trig))
        ctx.save_for_backward(*args, trig, keep if keep is not None else trig.new_empty(0), Rw)
        ctx.consts = (ratio, scale_min, S_world, keep is not None, dense)
        return verts

    @staticmethod
    def backward(ctx, g_verts):
        sq_eps, S, R6, T, trig, keep, Rw = ctx.saved_tensors
        ratio, scale_min, S_world, has_keep, dense = ctx.consts
        keep = keep if has_keep else None
        Kb, nv = trig.shape[1], trig.shape[2]
        gs = [ARENA.zeros_like(t) for t in (sq_eps, S, R6, T)]
        _lib.call('dbw_sq_blocks_bwd', _ptr(sq_eps), _ptr(S), _ptr(R6), _ptr(T), _ptr(trig), _ptr(keep), int(dense), Kb, nv, ratio, scale_min,
                  S_world, _ptr(Rw), _ptr(g_verts.contiguous()), *[_ptr(g) for g in gs], _stream(trig))
        return gs[0], gs[1], gs[2], gs[3], None, None, None, None


class _PosedMesh(torch.autograd.Function):
    @staticmethod
    def forward(ctx, R6, T, base, S_world, Rw, Tw):
        r6, t = _chk(R6.detach(), torch.float32, 'R6'), _chk(T.detach(), torch.float32, 'T')
        nv = base.shape[0]
        verts = torch.empty(nv, 3, dtype=torch.float32, device=base.device)
        _lib.call('dbw_posed_mesh_fwd', _ptr(base), nv, _ptr(r6), _ptr(t), float(S_world), _ptr(Rw), _ptr(Tw), _ptr(verts), _stream(base))
        ctx.save_for_backward(r6, t, base, Rw)
        ctx.S_world = float(S_world)
        return verts

    @staticmethod
    def backward(ctx, g):
        r6, t, base, Rw = ctx.saved_tensors
        g6, gt = ARENA.zeros_like(r6), ARENA.zeros_like(t)
        _lib.call('dbw_posed_mesh_bwd', _ptr(base), base.shape[0], _ptr(r6), _ptr(t), ctx.S_world, _ptr(Rw), _ptr(g.contiguous()),
                  _ptr(g6), _ptr(gt), _stream(base))
        return g6, gt, None, None, None, None


def posed_mesh(R6, T, base, S_world, R_world, T_world):
    return _PosedMesh.apply(R6, T, base, S_world, R_world, T_world)


class _CompositeMSE(torch.autograd.Function):
    """loss = mean((imgs - (fg_rgb*mask + (1-mask)*env_rgb))^2) over `count` elements.  Forward: one streaming pass that only
    reduces; backward: one streaming pass that writes d/dfg and d/denv already scaled by the upstream gradient (read from
    device memory by the kernel -- no host sync, no extra elementwise multiply over the image-sized gradients)."""

    @staticmethod
    def forward(ctx, fg, env, imgs, count):
        N, _, H, W = fg.shape
        fg_c, env_c = _chk(fg.detach(), torch.float32, 'fg'), _chk(env.detach(), torch.float32, 'env')
        imgs = _chk(imgs, torch.float32, 'imgs')
        loss = torch.zeros(1, dtype=torch.float32, device=fg.device)
        _lib.call('dbw_composite_mse', _ptr(fg_c), _ptr(env_c), _ptr(imgs), N, H, W, 1.0 / count, 0, 0, _ptr(loss), 0, 0, _stream(fg))
        ctx.save_for_backward(fg_c, env_c, imgs)
        ctx.count = count
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        fg, env, imgs = ctx.saved_tensors
        N, _, H, W = fg.shape
        g_fg, g_env = torch.empty_like(fg), torch.empty_like(env)
        g = g.detach().to(torch.float32).reshape(1).contiguous()
        _lib.call('dbw_composite_mse', _ptr(fg), _ptr(env), _ptr(imgs), N, H, W, 1.0 / ctx.count, _ptr(g), 0, 0, _ptr(g_fg), _ptr(g_env),
                  _stream(fg))
        return g_fg, g_env, None, None


def composite_mse(fg, env, imgs, count=None):
    """fg (N,4,H,W), env (N,4,H,W), imgs (N,3,H,W) -> scalar MSE of the decoupled composite (dbw.py:223,366-367).
    `count` = number of elements of the GLOBAL batch (view-sharded data parallel), default this batch."""
    count = float(imgs.numel() if count is None else count)
    return _CompositeMSE.apply(fg, env, imgs, count)


def composite(fg, env):
    """rec = fg_rgb*mask + (1-mask)*env_rgb as an image (API-compat path for predict(); HIP forward, elementwise torch
    backward -- the training path uses composite_mse)."""
    return _Composite.apply(fg, env)


class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fg, env):
        N, _, H, W = fg.shape
        fg_c, env_c = _chk(fg.detach(), torch.float32, 'fg'), _chk(env.detach(), torch.float32, 'env')
        rec = torch.empty(N, 3, H, W, dtype=torch.float32, device=fg.device)
        _lib.call('dbw_composite_mse', _ptr(fg_c), _ptr(env_c), 0, N, H, W, 0.0, 0, _ptr(rec), 0, 0, 0, _stream(fg))
        ctx.save_for_backward(fg_c, env_c)
        return rec

    @staticmethod
    def backward(ctx, g):
        fg, env = ctx.saved_tensors
        mask = fg[:, 3:4]
        g_fg = torch.cat([g * mask, (g * (fg[:, :3] - env[:, :3])).sum(1, keepdim=True)], 1)
        g_env = torch.cat([g * (1 - mask), torch.zeros_like(mask)], 1)
        return g_fg, g_env


class _TV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, maps, wrap, scale):
        m = _chk(maps.detach(), torch.float32, 'maps')
        n, h, w, _ = m.shape
        loss = torch.zeros(1, dtype=torch.float32, device=m.device)
        g = torch.empty_like(m)
        _lib.call('dbw_tv_l2sq', _ptr(m), n, h, w, int(wrap), float(scale), _ptr(loss), _ptr(g), _stream(m))
        ctx.save_for_backward(g)
        return loss[0]

    @staticmethod
    def backward(ctx, go):
        (g,) = ctx.saved_tensors
        return g * go, None, None


def tv_l2sq(maps, wrap_x=False, scale=1.0):
    """sum-over-maps l2sq total variation (dbw.py:380-386, loss.py:46), forward + gradient in one kernel."""
    return _TV.apply(maps, bool(wrap_x), float(scale))


class _Overlap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sq_eps, S, R6, T, alpha, u, consts):
        ratio, scale_min, temp, thresh = consts
        args = [_chk(t.detach(), torch.float32, 'overlap param') for t in (sq_eps, S, R6, T, alpha)]
        Kb, npts = u.shape[0], u.shape[1]
        dev = u.device
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        gs = [torch.zeros_like(t) for t in args]
        ws = torch.zeros(Kb * 18, dtype=torch.float32, device=dev)
        _lib.call('dbw_overlap_loss', _ptr(u), npts, *[_ptr(a) for a in args], Kb, ratio, scale_min, temp, thresh, 1.0, _ptr(loss),
                  *[_ptr(g) for g in gs], _ptr(ws), _stream(u))
        ctx.save_for_backward(*gs)
        return loss[0]

    @staticmethod
    def backward(ctx, go):
        gs = ctx.saved_tensors
        return gs[0] * go, gs[1] * go, gs[2] * go, gs[3] * go, gs[4] * go, None, None


def overlap_loss(sq_eps, S, R6, T, alpha, u, ratio, scale_min, temperature=0.005, n_blocks=1.95):
    """dbw.py:389-405.  u (Kb,npts,3) uniform samples in [0,1)."""
    return _Overlap.apply(sq_eps, S, R6, T, alpha, _chk(u, torch.float32, 'u'), (float(ratio), float(scale_min), float(temperature), float(n_blocks)))


class _BlockAlpha(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logit, noise, noise_scale, thresh):
        lg = _chk(logit.detach(), torch.float32, 'alpha_logit')
        Kb = lg.numel()
        alpha, alpha_full = torch.empty_like(lg), torch.empty_like(lg)
        keep = torch.empty(Kb, dtype=torch.int32, device=lg.device)
        nz = None if noise is None else _chk(noise, torch.float32, 'noise')
        _lib.call('dbw_block_alpha_fwd', _ptr(lg), _ptr(nz), float(noise_scale), float(thresh), Kb, _ptr(alpha), _ptr(alpha_full),
                  _ptr(keep), _stream(lg))
        ctx.save_for_backward(alpha, keep)
        ctx.mark_non_differentiable(keep)
        return alpha, alpha_full, keep

    @staticmethod
    def backward(ctx, g_a, g_af, _):
        alpha, keep = ctx.saved_tensors
        g = torch.empty_like(alpha)
        _lib.call('dbw_block_alpha_bwd', _ptr(alpha), _ptr(keep), _ptr(None if g_a is None else g_a.contiguous()), 1,
                  _ptr(None if g_af is None else g_af.contiguous()), alpha.numel(), _ptr(g), _stream(alpha))
        return g, None, None, None


def block_alpha(alpha_logit, noise=None, noise_scale=0.0, mask_threshold=-1.0):
    """alpha = sigmoid(alpha_logit + noise_scale*noise); keep = sigmoid(alpha_logit) > mask_threshold (all ones when
    mask_threshold < 0); alpha_full = alpha * keep  (dbw.py:297-311) -> (alpha, alpha_full, keep int32) in one launch."""
    return _BlockAlpha.apply(alpha_logit, noise, float(noise_scale), float(mask_threshold))


class _FusedLosses(torch.autograd.Function):
    """[w_rgb * MSE(composite), parsimony, tv, overlap] as one autograd node: every term is a kernel that produces its value
    and its gradient in the same pass, the weights are folded into the kernels' scale arguments and all four values land in
    one 4-float tensor -- instead of ~45 scalar-sized torch launches (fills, weight multiplications, sums and their autograd
    mirrors).  A term whose weight is None / 0 is skipped (its slot stays 0 and it sends no gradient)."""

    @staticmethod
    def forward(ctx, fg, env, imgs, alpha_full, bkg_maps, blocks_maps, ground_maps, sq_eps, S, R6, T, u, cfg):
        dev = imgs.device
        out = torch.zeros(4, dtype=torch.float32, device=dev)
        ctx.has_rgb = fg is not None        # None: the reconstruction term comes from render_decoupled_mse, only the regularisers here
        saved = []
        if ctx.has_rgb:
            fg_c, env_c = _chk(fg.detach(), torch.float32, 'fg'), _chk(env.detach(), torch.float32, 'env')
            imgs = _chk(imgs, torch.float32, 'imgs')
            N, _, H, W = fg_c.shape
            ctx.rgb_scale = cfg['rgb'] / cfg['count']
            _lib.call('dbw_composite_mse', _ptr(fg_c), _ptr(env_c), _ptr(imgs), N, H, W, ctx.rgb_scale, 0, 0, _ptr(out), 0, 0, _stream(fg_c))
            saved = [fg_c, env_c, imgs]
        ctx.n_tv = 0
        g_alpha_p = g_alpha_o = None
        if cfg.get('parsimony') and alpha_full is not None:
            a = _chk(alpha_full.detach(), torch.float32, 'alpha')
            g_alpha_p = ARENA.zeros_like(a)
            _lib.call('dbw_sqrt_mean', _ptr(a), a.numel(), 1e-6, float(cfg['parsimony']), _ptr(out) + 4, _ptr(g_alpha_p), _stream(a))
        tv_grads = []
        for maps, wrap, scale in ((bkg_maps, False, cfg.get('tv')), (blocks_maps, True, cfg.get('tv')),
                                  (ground_maps, False, (cfg.get('tv') or 0.0) * cfg.get('tv_ground_factor', 1.0))):
            if scale and maps is not None:
                m = _chk(maps.detach(), torch.float32, 'maps')
                n, h, w, _ = m.shape
                g = torch.empty_like(m)
                _lib.call('dbw_tv_l2sq', _ptr(m), n, h, w, int(wrap), float(scale), _ptr(out) + 8, _ptr(g), _stream(m))
                tv_grads.append(g)
            else:
                tv_grads.append(None)
        ov_grads = [None] * 4
        if cfg.get('overlap') and u is not None:
            ratio, scale_min, temp, thresh = cfg['overlap_consts']
            args = [_chk(t.detach(), torch.float32, 'overlap param') for t in (sq_eps, S, R6, T, alpha_full)]
            Kb, npts = u.shape[0], u.shape[1]
            gs = [ARENA.zeros_like(t) for t in args]
            ws = ARENA.zeros(Kb * 18, torch.float32, dev)
            _lib.call('dbw_overlap_loss', _ptr(u), npts, *[_ptr(a) for a in args], Kb, ratio, scale_min, temp, thresh, float(cfg['overlap']),
                      _ptr(out) + 12, *[_ptr(g) for g in gs], _ptr(ws), _stream(u))
            ov_grads, g_alpha_o = gs[:4], gs[4]
        ctx.save_for_backward(*saved)
        ctx.grads = (g_alpha_p, tv_grads, ov_grads, g_alpha_o)
        ctx.count = cfg['count']
        return out

    @staticmethod
    def backward(ctx, go):
        g_alpha_p, tv_grads, ov_grads, g_alpha_o = ctx.grads
        ctx.grads = None
        go = go.detach().to(torch.float32).contiguous()
        g_fg = g_env = None
        if ctx.has_rgb:
            fg, env, imgs = ctx.saved_tensors
            N, _, H, W = fg.shape
        if ctx.has_rgb and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            g_fg, g_env = torch.empty_like(fg), torch.empty_like(env)
            _lib.call('dbw_composite_mse', _ptr(fg), _ptr(env), _ptr(imgs), N, H, W, ctx.rgb_scale, _ptr(go), 0, 0, _ptr(g_fg), _ptr(g_env),
                      _stream(fg))
        # the saved gradients assume an upstream gradient of 1 per slot: scale each group by its slot's gradient (one
        # multi-tensor launch per group)
        tv = [g for g in tv_grads if g is not None]
        if tv:
            torch._foreach_mul_(tv, go[2])
        ov = [g for g in list(ov_grads) + [g_alpha_o] if g is not None]
        if ov:
            torch._foreach_mul_(ov, go[3])
        g_alpha = g_alpha_o
        if g_alpha_p is not None:
            g_alpha = torch.addcmul(g_alpha_o, g_alpha_p, go[1]) if g_alpha_o is not None else g_alpha_p * go[1]
        return (g_fg, g_env, None, g_alpha, tv_grads[0], tv_grads[1], tv_grads[2], ov_grads[0], ov_grads[1], ov_grads[2], ov_grads[3],
                None, None)


def fused_losses(fg, env, imgs, alpha_full, bkg_maps, blocks_maps, ground_maps, sq_eps, S, R6, T, u, cfg):
    """-> (4,) tensor [rgb, parsimony, tv, overlap], each already multiplied by its weight (dbw.py:361-408).
    cfg: {'rgb': w, 'count': elements of the global batch, 'parsimony': w|None, 'tv': w|None, 'tv_ground_factor': f,
          'overlap': w|None, 'overlap_consts': (ratio, scale_min, temperature, n_blocks)}."""
    return _FusedLosses.apply(fg, env, imgs, alpha_full, bkg_maps, blocks_maps, ground_maps, sq_eps, S, R6, T, u, cfg)


def adam_step_groups_(param, grad, exp_avg, exp_avg_sq, group_end, lrs, step, betas=(0.9, 0.999), eps=1e-8, zero=None, skip=None):
    """One launch for parameter groups that lie back to back in one flat buffer and differ in learning rate (optimizer.py:10-17).
    zero: a uint8 device tensor the same launch clears (the zero arena of the next iteration, ZeroArena.end_step).
    skip: a one-float device tensor; != 0 at launch time -> nothing is updated, only `zero` is cleared (a voided C step, c_step.py)."""
    import ctypes
    n = len(lrs)
    ends = (ctypes.c_int64 * n)(*[int(e) for e in group_end])
    lr = (ctypes.c_float * n)(*[float(x) for x in lrs])
    zb = 0 if zero is None else (zero.numel() + 15) // 16 * 16
    _lib.call('dbw_adam_step_groups', _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), ends, lr, n, float(betas[0]),
              float(betas[1]), float(eps), int(step), _ptr(zero), zb, _ptr(skip), _stream(param))


def adam_step_(param, grad, exp_avg, exp_avg_sq, lr, step, betas=(0.9, 0.999), eps=1e-8):
    """In-place fused Adam on flat fp32 buffers (torch.optim.Adam defaults; optimizer.py:6-18)."""
    _lib.call('dbw_adam_step', _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), float(lr), float(betas[0]),
              float(betas[1]), float(eps), int(step), _stream(param))
