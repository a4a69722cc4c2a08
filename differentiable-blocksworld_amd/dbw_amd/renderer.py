"""Renderer -- same constructor keywords, attributes and `forward` contract as the reference's
src/model/renderer.py:24-60,84-98 (`Renderer(img_size, **cfg.model.renderer)`; `forward(meshes, R, T,
viz_purpose=False, faces_alpha=...) -> (B,4,H,W)` BCHW, premultiplied RGB + alpha), with everything below it
(PyTorch3D MeshRenderer / MeshRasterizer / TexturesUV sampling / LayeredShader + layered_rgb_blend) replaced by the
HIP path in libdbw_hip.so.  Only the configuration the hot path uses is implemented:
perspective cameras with an explicit NDC K matrix, ambient white light, the 'raw' layered shader with clip_inside.
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
        if not kwargs.pop('clip_inside', True):
            raise NotImplementedError('clip_inside=False')
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

    def _cfg(self, n_faces, viz=False, lds_aggregate=False, texbins=None):
        H, W = self.img_size
        if viz:   # exact anti-aliased rendering for visualisation (renderer.py:56-60): 4x res, sigma 0, 1 face per pixel
            return ops.RenderCfg(H * 4, W * 4, 1, 0.0, self.z_clip, self.perspective_correct, False, n_faces, EPS)
        return ops.RenderCfg(H, W, self.faces_per_pixel, self.sigma, self.z_clip, self.perspective_correct, self.detach_bary,
                             n_faces, EPS, lds_aggregate, texbins)

    def render_packed(self, scene, R, T, faces_alpha=None, viz_purpose=False, lds_aggregate=False):
        """scene: PackedScene shared by the len(R) views.  lds_aggregate: hint for the backward pass (pays when neighbouring
        pixels hit the same texels: magnified or decimated maps); results are identical either way."""
        if self.cam_name != 'perspective' or self.cameras.K is None:
            raise NotImplementedError('the HIP path needs perspective cameras with an explicit NDC K: call '
                                      'update_cameras(K=...) first (dbw.py:204-208)')
        Kmat = self.cameras.K[0].to(R.device).contiguous()
        R, T = R.float().contiguous(), T.float().contiguous()
        cfg = self._cfg(scene.faces.shape[0], viz_purpose, lds_aggregate, getattr(scene, 'texbins', None))
        if viz_purpose:
            with torch.no_grad():
                img = ops.render_scene(scene.verts, scene.maps, None, scene.faces, R, T, Kmat, scene.face_uvs, scene.face_map,
                                       scene.map_desc, self._bg, cfg)
                return F.avg_pool2d(img, kernel_size=4, stride=4)
        return ops.render_scene(scene.verts, scene.maps, faces_alpha, scene.faces, R, T, Kmat, scene.face_uvs, scene.face_map,
                                scene.map_desc, self._bg, cfg)

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
