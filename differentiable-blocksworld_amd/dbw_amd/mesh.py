"""Init-time topology of the blocks-world scene (host side, runs once): icosphere, subdivision, plane, UV layouts and
rotation helpers.  Product-side counterparts of what the reference takes from PyTorch3D (`ico_sphere`,
`SubdivideMeshes`, `rotation_6d_to_matrix`, `random_rotations`: SURVEY.md A.9) and of src/utils/mesh.py:78-89,104-169,
210-211, src/model/tools.py:173-207.  Written independently of oracle/oracle.py;
tests/test_host_logic.py::test_topology_generators_agree_with_oracle checks the two agree exactly."""
import math

import numpy as np
import torch

# 12 vertices / 20 faces of the base icosahedron in PyTorch3D's listing order (4-decimal coordinates)
_A, _B = 0.5257, 0.8507
ICO_VERTS = np.array([[-_A, _B, 0], [_A, _B, 0], [-_A, -_B, 0], [_A, -_B, 0], [0, -_A, _B], [0, _A, _B], [0, -_A, -_B],
                      [0, _A, -_B], [_B, 0, -_A], [_B, 0, _A], [-_B, 0, -_A], [-_B, 0, _A]], dtype=np.float32)
ICO_FACES = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                      [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11],
                      [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)


def subdivide_mesh(verts, faces):
    """Loop-style 1->4 split: a midpoint vertex per unique edge, edges ranked by (min id, max id)."""
    V = verts.shape[0]
    f = faces.numpy()
    # edges opposite v0, v1, v2
    opp = np.stack([f[:, [1, 2]], f[:, [2, 0]], f[:, [0, 1]]], 0)                 # (3,F,2)
    key = np.sort(opp, axis=-1)
    code = key[..., 0] * V + key[..., 1]
    uniq, inv = np.unique(code.reshape(-1), return_inverse=True)
    e = torch.from_numpy(np.stack([uniq // V, uniq % V], 1))
    mid = verts[e].mean(dim=1)
    eid = torch.from_numpy(inv.reshape(3, -1)) + V                                 # (3,F) new vertex ids
    v0, v1, v2 = faces[:, 0], faces[:, 1], faces[:, 2]
    e0, e1, e2 = eid[0], eid[1], eid[2]
    new_faces = torch.cat([torch.stack([v0, e2, e1], 1), torch.stack([v1, e0, e2], 1), torch.stack([v2, e1, e0], 1),
                           torch.stack([e0, e1, e2], 1)], 0)
    return torch.cat([verts, mid], 0), new_faces


def ico_sphere(level=0):
    verts, faces = torch.from_numpy(ICO_VERTS.copy()), torch.from_numpy(ICO_FACES.copy())
    for _ in range(level):
        verts, faces = subdivide_mesh(verts, faces)
        verts = verts / verts.norm(p=2, dim=1, keepdim=True)
    return verts, faces


def get_icosphere(level=3, flip_faces=False):
    verts, faces = ico_sphere(level)
    if flip_faces:
        faces = faces.flip(1)
    return verts, faces


def get_plane():
    """primitives/plane.obj of the reference: unit square in the XZ plane, two triangles."""
    return (torch.tensor([[1., 0., -1.], [1., 0., 1.], [-1., 0., 1.], [-1., 0., -1.]]),
            torch.tensor([[3, 1, 0], [3, 2, 1]], dtype=torch.int64))


def point_to_uv_sphericalmap(X, eps=1e-7):
    r = torch.norm(X, dim=-1).clamp(min=eps)
    y = (X[..., 1] / r).clamp(-1 + eps, 1 - eps)
    v = torch.acos(-y) / np.pi
    u = (torch.atan2(X[..., 0], X[..., 2]) + np.pi) / (2 * np.pi)
    return torch.stack([u, v], dim=-1)


def get_icosphere_uvs(level=3, fix_continuity=False, fix_poles=False, eps=1e-8):
    """Spherical UV unwrap of the icosphere.  Faces straddling the u seam get private copies of their minority-side
    corners shifted by +-1 in u; faces touching a pole get a private pole corner whose u is the mean of the two other
    corners (same construction as src/utils/mesh.py:127-169)."""
    verts, faces = get_icosphere(level)
    uv = point_to_uv_sphericalmap(verts)
    faces = faces.clone()

    def append(uv, faces, sel_faces, corner_mask, new_uv):
        rows, cols = corner_mask.nonzero(as_tuple=True)
        ids = uv.shape[0] + torch.arange(rows.numel())
        faces[sel_faces[rows], cols] = ids
        return torch.cat([uv, new_uv], 0), faces

    if fix_continuity:
        fu = uv[faces][..., 0]                                                     # (F,3)
        span = torch.stack([fu[:, 1] - fu[:, 0], fu[:, 2] - fu[:, 1], fu[:, 0] - fu[:, 2]], 1).abs().amax(1)
        sel = (span > 0.5).nonzero(as_tuple=True)[0]
        u_sel, v_sel = uv[faces[sel]][..., 0], uv[faces[sel]][..., 1]
        side = torch.sign(u_sel - 0.5 + eps)
        major = side.sum(1, keepdim=True)
        minority = side != major
        moved = u_sel + major * minority
        uv, faces = append(uv, faces, sel, minority, torch.stack([moved[minority], v_sel[minority]], -1))
    if fix_poles:
        fv = uv[faces][..., 1]
        sel = ((fv.amax(1) > 0.99) | (fv.amin(1) < 0.01)).nonzero(as_tuple=True)[0]
        u_sel, v_sel = uv[faces[sel]][..., 0], uv[faces[sel]][..., 1]
        pole = (v_sel > 0.99) | (v_sel < 0.01)
        mean_u = ((1 - pole.float()) * u_sel).sum(1) / 2
        uv, faces = append(uv, faces, sel, pole, torch.stack([mean_u.repeat_interleave(pole.sum(1)), v_sel[pole]], -1))
    return faces, uv


def rotation_6d_to_matrix(d6):
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = torch.nn.functional.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), dim=-2)


def matrix_to_rotation_6d(M):
    return M[..., :2, :].clone().reshape(*M.shape[:-2], 6)


def random_rotations(n):
    """Uniform random rotations from normalised Gaussian quaternions (consumes torch.randn(n,4): same RNG draw as
    PyTorch3D so same-seed initialisation matches the reference, dbw.py:103)."""
    q = torch.randn(n, 4)
    q = q / torch.copysign((q * q).sum(1).sqrt(), q[:, 0])[:, None]
    r, i, j, k = q.unbind(-1)
    s = 2.0 / (q * q).sum(-1)
    M = torch.stack([1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r),
                     s * (i * j + k * r), 1 - s * (i * i + k * k), s * (j * k - i * r),
                     s * (i * k - j * r), s * (j * k + i * r), 1 - s * (i * i + j * j)], -1)
    return M.reshape(n, 3, 3)


def _axis_rotation(kind, deg):
    a = torch.tensor([float(deg)]) * math.pi / 180
    R = torch.eye(3)
    if kind == 'elev':       # about X, angle negated (tools.py:186-196)
        c, s = torch.cos(-a)[0], torch.sin(-a)[0]
        R[1, 1], R[1, 2], R[2, 1], R[2, 2] = c, s, -s, c
    elif kind == 'azim':     # about Y (tools.py:173-183)
        c, s = torch.cos(a)[0], torch.sin(a)[0]
        R[0, 0], R[0, 2], R[2, 0], R[2, 2] = c, s, -s, c
    else:                    # roll, about Z (tools.py:199-207)
        c, s = torch.cos(a)[0], torch.sin(a)[0]
        R[0, 0], R[0, 1], R[1, 0], R[1, 1] = c, s, -s, c
    return R


def world_rotation(elev, azim, roll):
    """R_world = elev @ azim @ roll (dbw.py:59)."""
    return _axis_rotation('elev', elev) @ _axis_rotation('azim', azim) @ _axis_rotation('roll', roll)


def look_at_view_transform(centers, at=(0., 0., 0.), up=(0., 1., 0.)):
    """PyTorch3D camera convention: R columns = camera axes in world coords, X_cam = X_world @ R + T."""
    C = torch.as_tensor(centers, dtype=torch.float32).reshape(-1, 3)
    at = torch.tensor(at, dtype=torch.float32).expand_as(C)
    up = torch.tensor(up, dtype=torch.float32).expand_as(C)
    z = torch.nn.functional.normalize(at - C, dim=-1)
    x = torch.nn.functional.normalize(torch.cross(up, z, dim=-1), dim=-1)
    y = torch.nn.functional.normalize(torch.cross(z, x, dim=-1), dim=-1)
    R = torch.stack([x, y, z], dim=-1)
    return R, -(C[:, None] @ R)[:, 0]


def synthetic_cameras(n_views, R_world=None, dist=2.8, f_ndc=4.82, elev_deg=30.0):
    """Synthetic DTU-like rig: a wobbling ring of cameras above the ground plane looking at the origin,
    shared NDC intrinsics K = [[f,0,px,0],[0,f,py,0],[0,0,0,1],[0,0,1,0]] with f ~ 2892 px / 600 (dtu.py:95-106).
    The ring lives in the model frame and is mapped to the world frame by R_world (dbw.py:59,264).
    -> R (V,3,3), T (V,3), K (V,4,4)."""
    az = torch.arange(n_views, dtype=torch.float32) * (2 * math.pi / n_views) + 0.1
    el = torch.full((n_views,), elev_deg * math.pi / 180) + 0.15 * torch.sin(3 * az)
    C = torch.stack([torch.cos(el) * torch.sin(az), torch.sin(el), torch.cos(el) * torch.cos(az)], -1) * dist
    up = torch.tensor([[0., 1., 0.]])
    if R_world is not None:
        C, up = C @ R_world.reshape(3, 3).cpu(), up @ R_world.reshape(3, 3).cpu()
    R, T = look_at_view_transform(C, up=tuple(up[0].tolist()))
    K = torch.tensor([[f_ndc, 0, 0, 0], [0, f_ndc, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=torch.float32)
    return R, T, K[None].repeat(n_views, 1, 1)


# ---------------------------------------------------------------------------------------------------------------------
# colour map of the visualisations (utils/plot.py:77-87: seaborn 'hls' palette of 21 colours behind gold, as a matplotlib
# LinearSegmentedColormap) restated with colorsys / numpy -- neither seaborn nor matplotlib is a dependency of the render path
# ---------------------------------------------------------------------------------------------------------------------
def get_fancy_cmap():
    import colorsys
    import numpy as np
    hues = np.linspace(0, 1, 22)[:-1] + 0.01                                  # seaborn.hls_palette(21, h=.01, l=.6, s=.65)
    hues = hues % 1
    hls = [colorsys.hls_to_rgb(float(h), 0.6, 0.65) for h in hues]
    colors = np.array([(1.0, 0.8431372549019608, 0.0)] + hls[3:] + hls[:2])    # gold first
    x = np.linspace(0, 1, len(colors))
    lut = np.stack([np.interp(np.linspace(0, 1, 256), x, colors[:, c]) for c in range(3)], -1)   # matplotlib's 256-entry table

    def cmap(values):
        if isinstance(values, torch.Tensor):
            values = values.detach().cpu().numpy()
        v = np.asarray(values, dtype=np.float64)
        idx = np.clip((v * 256).astype(np.int64), 0, 255)
        idx[v == 1.0] = 255
        return lut[idx]
    return cmap
