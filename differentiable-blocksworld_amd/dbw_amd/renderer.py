"""Renderer -- same constructor keywords, attributes and `forward` contract as the reference's
src/model/renderer.py:24-60,84-98 (`Renderer(img_size, **cfg.model.renderer)`; `forward(meshes, R, T,
viz_purpose=False, faces_alpha=...) -> (B,4,H,W)` BCHW, premultiplied RGB + alpha), with everything below it
(PyTorch3D MeshRenderer / MeshRasterizer / TexturesUV sampling / LayeredShader + layered_rgb_blend) replaced by the
HIP path in libdbw_hip.so.  Only the configuration the hot path uses is implemented:
perspective cameras with an explicit NDC K matrix, ambient white light, the 'raw' layered shader (clip_inside True or False).
Anything else raises NotImplementedError instead of silently rendering something different."""
from copy import deepcopy

import torch
from torch import nn
from torch.nn import functional as F

from . import ops
from .structures import Meshes, PackedScene

EPS = 1e-8          # renderer.py:20


class PerspectiveCameras:
    """Holder of the shared intrinsics.  K is None until update_cameras(K=...) (dbw.py:204-208)."""

    def __init__(self, K=None, device=None, **kwargs):
        self.K = None if K is None else torch.as_tensor(K, dtype=torch.float32).reshape(-1, 4, 4)[:1]
        if device is not None and self.K is not None:
            self.K = self.K.to(device)
        self.kwargs = kwargs

    def to(self, device):
        if self.K is not None:
            self.K = self.K.to(device)
        return self


class AmbientLights:
    ambient_color = ((1.0, 1.0, 1.0),)

    def to(self, device):
        return self


class Renderer(nn.Module):
    def __init__(self, img_size, **kwargs):
        super().__init__()
        self.img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        self._init_kwargs = deepcopy(kwargs)
        self.init_cameras(**kwargs.pop('cameras', {}))
        self.init_lights(**kwargs.pop('lights', {}))
        self.sigma = kwargs.pop('sigma', 1e-4)
        self.background_color = tuple(kwargs.pop('background_color', (0, 0, 0)))
        self.faces_per_pixel = kwargs.pop('faces_per_pixel', 25)
        p_correct = kwargs.pop('perspective_correct', None)
        self.z_clip = kwargs.pop('z_clip', None)
        kwargs.pop('debug', False)
        if not kwargs.pop('layered_shader', True):
            raise NotImplementedError('only the layered shader is on the hot path (renderer.py:39-43)')
        self.clip_inside = bool(kwargs.pop('clip_inside', True))          # False: sigmoid(-d / sigma) instead of exp(-max(d, 0) / sigma), renderer.py:257-258
        if kwargs.pop('shading_type', 'raw') != 'raw':
            raise NotImplementedError("only shading_type='raw' (phong/flat/gouraud are visualisation-only)")
        self.detach_bary = kwargs.pop('detach_bary', False)
        assert len(kwargs) == 0, kwargs
        # perspective_correct=None is inferred True for perspective cameras
        self.perspective_correct = True if p_correct is None else bool(p_correct)
        self._bg = ops.make_bg(self.background_color)

    # -- renderer.py:62-73
    def init_cameras(self, **kwargs):
        kwargs = deepcopy(kwargs)
        self.cam_name = kwargs.pop('name', 'fov')
        self.cam_kwargs = kwargs
        self.cameras = PerspectiveCameras(**kwargs)

    def init_lights(self, **kwargs):
        kwargs = deepcopy(kwargs)
        if kwargs.pop('name', 'ambient') != 'ambient':
            raise NotImplementedError('only ambient lights are on the hot path (directional = visualisation)')
        self.lights = AmbientLights()

    @property
    def init_kwargs(self):
        return deepcopy(self._init_kwargs)

    def to(self, device):
        super().to(device)
        self.cameras.to(device)
        return self

    # -- renderer.py:106-116
    def get_copy_cameras(self, **kwargs):
        merged = deepcopy(self.cam_kwargs)
        merged.update(kwargs)
        return PerspectiveCameras(**merged)

    def update_cameras(self, **kwargs):
        self.cameras = self.get_copy_cameras(**kwargs)

    def _cfg(self, n_faces, viz=False, lds_aggregate=False, texbins=None, const_faces=0):
        H, W = self.img_size
        if viz:   # exact anti-aliased rendering for visualisation (renderer.py:56-60): 4x res, sigma 0, 1 face per pixel
            return ops.RenderCfg(H * 4, W * 4, 1, 0.0, self.z_clip, self.perspective_correct, False, n_faces, EPS)
        cfg = ops.RenderCfg(H, W, self.faces_per_pixel, self.sigma, self.z_clip, self.perspective_correct, self.detach_bary,
                            n_faces, EPS, lds_aggregate, texbins, const_faces, clip_inside=self.clip_inside)
        if texbins is not None:         # the texture bins' record sub-ranges follow the demand of this renderer's previous backward (ops.BinDemand)
            if getattr(self, '_bin_demand', None) is None:
                self._bin_demand = ops.BinDemand()
            cfg.bin_demand = self._bin_demand
        return cfg

    def render_packed(self, scene, R, T, faces_alpha=None, viz_purpose=False, lds_aggregate=False):
        """scene: PackedScene shared by the len(R) views.  lds_aggregate: hint for the backward pass (pays when neighbouring
        pixels hit the same texels: magnified or decimated maps); results are identical either way."""
        if self.cam_name != 'perspective' or self.cameras.K is None:
            raise NotImplementedError('the HIP path needs perspective cameras with an explicit NDC K: call '
                                      'update_cameras(K=...) first (dbw.py:204-208)')
        Kmat = self.cameras.K[0].to(R.device).contiguous()
        R, T = R.float().contiguous(), T.float().contiguous()
        cfg = self._cfg(scene.faces.shape[0], viz_purpose, lds_aggregate, getattr(scene, 'texbins', None), getattr(scene, 'const_faces', 0))
        if viz_purpose:
            with torch.no_grad():
                img = ops.render_scene(scene.verts, scene.maps, None, scene.faces, R, T, Kmat, scene.face_uvs, scene.face_map,
                                       scene.map_desc, self._bg, cfg)
                return F.avg_pool2d(img, kernel_size=4, stride=4)
        return ops.render_scene(scene.verts, scene.maps, faces_alpha, scene.faces, R, T, Kmat, scene.face_uvs, scene.face_map,
                                scene.map_desc, self._bg, cfg)

    # -- renderer.py:134-175: wireframe overlays (visualisation) on the same rasteriser kernels
    def _as_scene(self, meshes):
        return PackedScene.from_meshes(meshes) if isinstance(meshes, Meshes) else meshes

    @torch.no_grad()
    def render_edges(self, meshes, R, T, image_size=None, linewidth=1, return_pix2face=False, faces_per_pixel=1):
        """(B,1,H,W) mask of the pixels that lie inside a face and closer than `linewidth` pixels to one of its edges
        (renderer.py:134-147): a hard rasterisation whose signed squared NDC distances are thresholded at
        (linewidth * 2 / min(image_size))^2; with return_pix2face also the (B,H,W) int64 packed face id of the nearest face."""
        if self.cam_name != 'perspective' or self.cameras.K is None:
            raise NotImplementedError('the HIP path needs perspective cameras with an explicit NDC K (dbw.py:204-208)')
        scene = self._as_scene(meshes)
        H, W = image_size or self.img_size
        cfg = ops.RenderCfg(H, W, int(faces_per_pixel), 0.0, self.z_clip, self.perspective_correct, False, scene.faces.shape[0], EPS)
        Kmat = self.cameras.K[0].to(R.device).contiguous()
        cl, p2f, _, _, dists = ops.render_fragments(scene.verts.detach(), scene.faces, R.float().contiguous(), T.float().contiguous(), Kmat, cfg)
        mask = (-dists < (linewidth * 2 / min(H, W)) ** 2).float()[:, None]       # B1HWK; empty slots hold -1: never an edge
        mask = mask.max(-1)[0]
        if not return_pix2face:
            return mask
        first = p2f[..., 0]
        B, F_ = R.shape[0], scene.faces.shape[0]
        orig = cl['c2o'].view(-1).long()[first.clamp(min=0).long()] + torch.arange(B, device=first.device).view(B, 1, 1) * F_
        return mask, torch.where(first >= 0, orig, torch.full_like(orig, -1))

    @torch.no_grad()
    def draw_edges(self, img, meshes, R=None, T=None, colors=None, linewidth=1, antialias=True):
        """img (B,3,H,W) with the wireframe of `meshes` painted on it (renderer.py:149-175); colors: one RGB triple, or one per
        packed face (B*F,3).  antialias: the mask is rasterised at 4x the resolution and average-pooled."""
        scene = self._as_scene(meshes)
        B = img.shape[0]
        dev = img.device
        if R is None:
            R = torch.eye(3, device=dev)[None].expand(B, -1, -1)
        if T is None:
            T = torch.zeros(1, 3, device=dev).expand(B, -1)
        colors = torch.as_tensor((1., 0., 0.) if colors is None else colors, dtype=torch.float32, device=dev)
        size = tuple(img.shape[-2:])
        if antialias:
            size, linewidth = (size[0] * 4, size[1] * 4), linewidth * 4
        mask, pix2face = self.render_edges(scene, R, T, image_size=size, linewidth=linewidth, return_pix2face=True)
        if colors.dim() == 2:
            face_img = colors[pix2face].permute(0, 3, 1, 2)                       # one colour per face (empty pixels: last row, masked)
        else:
            face_img = colors[None, :, None, None].expand(B, -1, *size)
        if antialias:
            mask, face_img = [F.avg_pool2d(t, kernel_size=4, stride=4) for t in (mask, face_img)]
        return img * (1 - mask) + mask * face_img

    def forward(self, meshes, R, T, viz_purpose=False, **kwargs):
        faces_alpha = kwargs.pop('faces_alpha', None)
        assert len(kwargs) == 0, kwargs
        if isinstance(meshes, Meshes):
            if len(meshes) != len(R) and len(meshes) != 1:
                raise ValueError(f'{len(meshes)} meshes for {len(R)} cameras')
            scene = PackedScene.from_meshes(meshes)
        else:
            scene = meshes
        return self.render_packed(scene, R, T, faces_alpha, viz_purpose)
