"""Camera ingest: OpenCV-style 3x4 projection matrices -> the (K 4x4 NDC, R row-vector, T)
convention the render path consumes.  Restates src/dataset/dtu.py:75-124 (`pytorch3d_KRT_from_proj`,
`opencv_KRT_from_proj`) with a NumPy RQ decomposition in place of `cv2.decomposeProjectionMatrix` (OpenCV is not
available here, so this row is checked by round trips and by projecting points both ways, tests/test_host_logic.py)."""
import numpy as np
import torch


def _rq3(M):
    """M = K @ R with K upper triangular (positive diagonal) and R a rotation."""
    P = np.flipud(np.eye(3))
    Q, U = np.linalg.qr((P @ M).T)
    K, R = P @ U.T @ P, P @ Q.T
    S = np.diag(np.sign(np.diag(K)))
    K, R = K @ S, S @ R
    if np.linalg.det(R) < 0:
        K, R = -K, -R
    return K, R


def opencv_KRT_from_proj(P):
    """dtu.py:118-124: K (4x4, K[2,2] = 1), R (camera-to-world rotation), T (camera centre in world coordinates)."""
    P = np.asarray(P, dtype=np.float64)[:3, :4]
    K_raw, R_w2c = _rq3(P[:, :3])
    c = -np.linalg.solve(P[:, :3], P[:, 3])                      # camera centre: P @ [c; 1] = 0
    K = np.eye(4, dtype=np.float32)
    K[:3, :3] = (K_raw / K_raw[2, 2]).astype(np.float32)
    return K, R_w2c.T.astype(np.float32), c.astype(np.float32)


def pytorch3d_KRT_from_proj(P, image_size):
    """dtu.py:75-115.  image_size = (H, W).  Returns K (4,4) NDC intrinsics [[fx,0,px,0],[0,fy,py,0],[0,0,0,1],[0,0,1,0]]
    (scale = min(W,H)/2, principal point measured from the image centre with flipped sign), R (3,3) and T (3,) such that
    X_cam = X_world @ R + T with +X left, +Y up, +Z into the screen."""
    K, R_c2w, C = map(torch.from_numpy, opencv_KRT_from_proj(P))
    R = R_c2w.T                                   # x_cam = R @ x_world + T
    T = -R @ C
    H, W = image_size
    wh = torch.tensor([float(W), float(H)])
    scale = wh.min() / 2.0
    c0 = wh / 2.0
    focal = torch.stack([K[0, 0], K[1, 1]]) / scale
    p0 = -(K[:2, 2] - c0) / scale
    Kp = torch.zeros(4, 4)
    Kp[0, 0], Kp[1, 1] = focal[0], focal[1]
    Kp[:2, 2] = p0
    Kp[2, 3] = Kp[3, 2] = 1.0
    Rp = R.clone().T                              # row-vector convention
    Tp = T.clone()
    Rp[:, :2] *= -1                               # OpenCV screen axes point the other way
    Tp[:2] *= -1
    return Kp, Rp, Tp


def load_idr_cameras(cam, image_size, n_views=None):
    """The IDR `cameras.npz` layout DTU / BlendedMVS scenes ship with (dtu.py:42-44): view i projects with
    `(world_mat_i @ scale_mat_i)[:3, :4]`.  `cam`: a path or an already opened mapping; `image_size` = (H, W) of the RAW images the
    matrices refer to (the NDC intrinsics are resolution independent, dtu.py:95-106).
    -> dict(K (N,4,4), R (N,3,3), T (N,3)) ready to be batched into `model(inp)` + scale_mat (4,4) of view 0 (the reference uses it
    to bring the ground-truth points into the normalised frame, dtu.py:47-50)."""
    if not hasattr(cam, 'keys'):
        cam = np.load(cam)
    if n_views is None:
        n_views = sum(1 for k in cam.keys() if k.startswith('world_mat_') and not k.startswith('world_mat_inv'))
    Ks, Rs, Ts = [], [], []
    for i in range(n_views):
        P = (np.asarray(cam[f'world_mat_{i}'], dtype=np.float64) @ np.asarray(cam[f'scale_mat_{i}'], dtype=np.float64))[:3, :4]
        K, R, T = pytorch3d_KRT_from_proj(P, image_size)
        Ks.append(K); Rs.append(R); Ts.append(T)
    return {'K': torch.stack(Ks), 'R': torch.stack(Rs), 'T': torch.stack(Ts),
            'scale_mat': torch.from_numpy(np.asarray(cam['scale_mat_0'], dtype=np.float32))}
