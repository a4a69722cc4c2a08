"""Minimal mesh / UV-texture containers with the subset of the PyTorch3D `Meshes` / `TexturesUV` surface that
src/model/dbw.py and src/model/renderer.py touch (construction, len, extend, join_meshes_as_scene,
get_mesh_verts_faces, verts/faces accessors).  They carry no PyTorch3D code: a mesh batch here is always B copies of
ONE packed scene (that is the only thing the reference renders: `scene.extend(B)`, dbw.py:215,220,232), which is what
the HIP path exploits -- the scene's geometry and textures are stored once, never replicated per view.

SURVEY.md A.8: join order = list order; packed face id of face j of copy b = b*F + j."""
import torch


class TexturesUV:
    """maps: (N,H,W,3) tensor or list of (H_i,W_i,3) tensors; faces_uvs (N,F,3) / (F,3) int64; verts_uvs (N,V,2) / (V,2).
    `circular_pad=(left,right)`: the map is to be sampled as if circularly padded along u by that many texels
    (dbw.py:339-341) -- kept symbolic, the sampler wraps indices instead of materialising the padded copy."""

    def __init__(self, maps, faces_uvs, verts_uvs, align_corners=True, circular_pad=(0, 0)):
        if not align_corners:
            raise NotImplementedError('only align_corners=True (the reference setting, dbw.py:280,295,342)')
        self.maps = list(maps) if not isinstance(maps, torch.Tensor) else [maps[i] for i in range(maps.shape[0])]
        n = len(self.maps)
        faces_uvs = faces_uvs if faces_uvs.dim() == 3 else faces_uvs[None].expand(n, -1, -1)
        verts_uvs = verts_uvs if verts_uvs.dim() == 3 else verts_uvs[None].expand(n, -1, -1)
        self.faces_uvs, self.verts_uvs = faces_uvs, verts_uvs
        self.pads = [tuple(circular_pad)] * n
        self.align_corners = True

    def face_uv_table(self):
        """(sum F_i, 3, 2) uv of every face corner, (sum F_i,) map index."""
        uv = torch.cat([self.verts_uvs[i][self.faces_uvs[i]] for i in range(len(self.maps))], 0)
        fm = torch.cat([torch.full((self.faces_uvs[i].shape[0],), i, dtype=torch.int32, device=uv.device)
                        for i in range(len(self.maps))])
        return uv.contiguous().float(), fm


class Meshes:
    """verts (N,V,3) (or (V,3)), faces (N,F,3) int64: N meshes sharing V and F.  `copies` > 1 marks `extend(B)`."""

    def __init__(self, verts, faces, textures=None, copies=1):
        self._verts = verts if verts.dim() == 3 else verts[None]
        self._faces = faces if faces.dim() == 3 else faces[None]
        self.textures = textures
        self.copies = copies

    def __len__(self):
        return self._verts.shape[0] * self.copies

    @property
    def device(self):
        return self._verts.device

    def extend(self, B):
        if self._verts.shape[0] != 1:
            raise NotImplementedError('extend() is only used on a single joined scene (dbw.py:215,220,232)')
        return Meshes(self._verts, self._faces, self.textures, copies=self.copies * B)

    def get_mesh_verts_faces(self, i):
        n = self._verts.shape[0]
        return self._verts[i % n], self._faces[i % n]

    def verts_padded(self):
        return self._verts

    def faces_padded(self):
        return self._faces

    def verts_packed(self):
        return self._verts.reshape(-1, 3)

    def faces_packed(self):
        n, V = self._verts.shape[:2]
        off = (torch.arange(n, device=self._faces.device) * V)[:, None, None]
        return (self._faces + off).reshape(-1, 3)

    def num_faces_per_mesh(self):
        return torch.full((self._verts.shape[0],), self._faces.shape[1], dtype=torch.int64, device=self._faces.device)

    def to(self, device):
        tex = self.textures
        if tex is not None:
            t = TexturesUV.__new__(TexturesUV)
            t.maps = [m.to(device) for m in tex.maps]
            t.faces_uvs, t.verts_uvs, t.pads, t.align_corners = tex.faces_uvs.to(device), tex.verts_uvs.to(device), tex.pads, True
            tex = t
        return Meshes(self._verts.to(device), self._faces.to(device), tex, self.copies)


def join_meshes_as_scene(meshes):
    """One mesh out of a list of Meshes (each possibly a batch): verts concatenated, faces offset, maps listed in order."""
    if isinstance(meshes, Meshes):
        meshes = [meshes]
    verts, faces, maps, fuv, vuv_off, pads = [], [], [], [], 0, []
    face_uvs, face_map = [], []
    voff = 0
    for m in meshes:
        n, V = m._verts.shape[:2]
        verts.append(m._verts.reshape(-1, 3))
        faces.append(m.faces_packed() + voff)
        voff += n * V
        if m.textures is not None:
            uv, fm = m.textures.face_uv_table()
            face_uvs.append(uv)
            face_map.append(fm + len(maps))
            maps += m.textures.maps
            pads += m.textures.pads
    out = Meshes(torch.cat(verts, 0), torch.cat(faces, 0))
    if face_uvs:
        out.textures = _SceneTextures(maps, pads, torch.cat(face_uvs, 0), torch.cat(face_map, 0))
    return out


class _SceneTextures:
    """Textures of a joined scene: a list of maps + per-face (uv corners, map index).  No atlas is built: sampling each
    map directly is mathematically identical to PyTorch3D's packed atlas."""

    def __init__(self, maps, pads, face_uvs, face_map):
        self.maps, self.pads, self._face_uvs, self._face_map = maps, pads, face_uvs, face_map

    def face_uv_table(self):
        return self._face_uvs, self._face_map


class PackedScene:
    """What the kernels consume: verts (V,3) world [may require grad], faces int32 (F,3), face_uvs (F,3,2), face_map int32
    (F,), map_desc int32 (M,8) = {offset, h, w, pad_left, pad_right, shift, 0, 0} (map stored as (h>>shift, w>>shift, 3)),
    maps: flat fp32 [may require grad]."""

    def __init__(self, verts, faces_i32, face_uvs, face_map, map_desc, maps, texbins=None):
        self.verts, self.faces, self.face_uvs, self.face_map, self.map_desc, self.maps = verts, faces_i32, face_uvs, face_map, map_desc, maps
        self.texbins = texbins      # optional (bin_base, bin_info, nbins) from describe_bins
        self.const_faces = 0        # the first that many faces have constant vertices: the backward skips their geometry gradient

    @staticmethod
    def join(scenes):
        """One scene out of several (join_meshes_as_scene on packed scenes): vertex, face and map tables concatenated, face and
        map indices re-based; faces keep the order of `scenes`."""
        verts, faces, uvs, fmap, desc, maps = [], [], [], [], [], []
        v_off = m_off = f_off = 0
        for s in scenes:
            verts.append(s.verts)
            faces.append(s.faces + v_off)
            uvs.append(s.face_uvs)
            fmap.append(s.face_map + m_off)
            d = s.map_desc.clone()
            d[:, 0] += f_off
            d[:, 6] = 0                    # (the row count lives in row 0 of the joined table only)
            desc.append(d)
            maps.append(s.maps.reshape(-1))
            v_off += s.verts.shape[0]
            m_off += s.map_desc.shape[0]
            f_off += s.maps.numel()
        desc = torch.cat(desc, 0).contiguous()
        if desc.shape[0]:
            desc[0, 6] = desc.shape[0]
        return PackedScene(torch.cat(verts, 0), torch.cat(faces, 0).to(torch.int32).contiguous(), torch.cat(uvs, 0).contiguous(),
                           torch.cat(fmap, 0).to(torch.int32).contiguous(), desc, torch.cat(maps))

    @staticmethod
    def describe_maps(shapes, pads, device, shift=0):
        """shapes: full-resolution (h, w) of every map; shift: log2 of the decimation they are stored at."""
        rows, off = [], 0
        for (h, w), (pl, pr) in zip(shapes, pads):
            rows.append([off, h, w, pl, pr, shift, 0, 0])
            off += (h >> shift) * (w >> shift) * 3
        if rows:
            rows[0][6] = len(rows)        # the table's row count (include/dbw_hip.h: small tables are kept in LDS by the backward kernels)
        return torch.tensor(rows, dtype=torch.int32, device=device), off

    @staticmethod
    def describe_bins(shapes, device, shift=0, tile=32):
        """Texture-space bins (32x32 stored texels) for the binned texel-gradient reduction of the backward pass
        (include/dbw_hip.h: dbw_render_bwd_fused / dbw_texbin_reduce).  -> (bin_base (M,), bin_info (nbins,4), nbins)."""
        base, info, off = [], [], 0
        for h, w in shapes:
            hs, ws = h >> shift, w >> shift
            by, bx = (hs + tile - 1) // tile, (ws + tile - 1) // tile
            base.append(len(info))
            info += [[off, ws, hs, (ty << 16) | tx] for ty in range(by) for tx in range(bx)]
            off += hs * ws * 3
        return (torch.tensor(base, dtype=torch.int32, device=device), torch.tensor(info, dtype=torch.int32, device=device).reshape(-1, 4),
                len(info))

    @staticmethod
    def from_meshes(meshes):
        verts, faces = meshes.get_mesh_verts_faces(0) if meshes._verts.shape[0] == 1 else (meshes.verts_packed(), meshes.faces_packed())
        tex = meshes.textures
        if tex is None:
            raise ValueError('the layered shader needs UV textures')
        face_uvs, face_map = tex.face_uv_table()
        desc, _ = PackedScene.describe_maps([m.shape[:2] for m in tex.maps], tex.pads, verts.device)
        maps = torch.cat([m.reshape(-1) for m in tex.maps])
        return PackedScene(verts, faces.to(torch.int32).contiguous(), face_uvs.contiguous(), face_map.to(torch.int32).contiguous(), desc, maps)
