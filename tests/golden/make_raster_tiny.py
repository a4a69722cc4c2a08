"""Generates tests/golden/raster_tiny.npz: outputs of the ORACLE's rasteriser (oracle/raster_ref.c, forward + backward) and clipping
on three tiny scenes -- SURVEY.md 8(c) golden vector (7).  SELF-CONSISTENCY ONLY, parity unpinned against PyTorch3D (not installable
here): the fixture freezes the restatement, so that a later change of the oracle or of the HIP path that moves a single bit of these
outputs is noticed.  Run from the repo root:  python tests/golden/make_raster_tiny.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..', 'oracle'))
import oracle as O  # noqa: E402


def scenes():
    # 1. one front-facing triangle + a second one behind it, K larger than the number of faces
    fv1 = torch.tensor([[[-0.6, -0.5, 2.0], [0.7, -0.4, 2.5], [0.1, 0.8, 3.0]],
                        [[-0.9, 0.2, 4.0], [0.9, 0.3, 4.0], [0.0, -0.9, 4.0]]], dtype=torch.float32)
    yield 'front', fv1, None, dict(size=(12, 16), blur=2e-3, K=4)
    # 2. a triangle straddling the near plane in both ways (cases 3 and 4 of clip_faces) seen by two "views" (packed meshes)
    tri = torch.tensor([[[-0.5, -0.4, 0.4], [0.6, -0.3, -0.2], [0.0, 0.7, 1.5]],       # one vertex behind z = 0.1 -> quad -> 2 triangles
                        [[-0.7, 0.5, -0.3], [0.2, 0.6, -0.1], [-0.2, -0.6, 0.9]]], dtype=torch.float32)  # two behind -> 1 smaller triangle
    yield 'straddle', tri, 0.1, dict(size=(16, 12), blur=1e-3, K=3)
    # 3. ties: two coplanar faces sharing an edge and a duplicate face (same depth everywhere) -> order by face id
    q = torch.tensor([[[-0.5, -0.5, 2.0], [0.5, -0.5, 2.0], [0.5, 0.5, 2.0]],
                      [[-0.5, -0.5, 2.0], [0.5, 0.5, 2.0], [-0.5, 0.5, 2.0]],
                      [[-0.5, -0.5, 2.0], [0.5, -0.5, 2.0], [0.5, 0.5, 2.0]]], dtype=torch.float32)
    yield 'ties', q, None, dict(size=(10, 10), blur=0.0, K=2)


def main():
    out = {}
    gen = torch.Generator().manual_seed(0)
    for name, fv, zclip, cfg in scenes():
        first, num = torch.tensor([0]), torch.tensor([fv.shape[0]])
        nbr = None
        if zclip is not None:
            cl = O.clip_faces(fv, first, num, zclip, True)
            fv_r, first, num, nbr = cl['face_verts'], cl['first_idx'], cl['num_faces'], cl['neighbor']
            out[f'{name}/clipped'] = fv_r.numpy()
            out[f'{name}/neighbor'] = nbr.numpy()
            out[f'{name}/clipped_to_orig'] = cl['clipped_to_orig'].numpy()
        else:
            fv_r = fv
        p2f, zbuf, bary, dists = O.rasterize_fwd_raw(fv_r, first, num, nbr, cfg['size'], cfg['blur'], cfg['K'])
        gz, gb, gd = [torch.randn(t.shape, generator=gen) for t in (zbuf, bary, dists)]
        gfv = O.rasterize_bwd_raw(fv_r, p2f, gz, gb, gd)
        out.update({f'{name}/face_verts': fv.numpy(), f'{name}/zclip': np.float32(-1.0 if zclip is None else zclip),
                    f'{name}/size': np.array(cfg['size']), f'{name}/blur': np.float32(cfg['blur']), f'{name}/K': np.int64(cfg['K']),
                    f'{name}/p2f': p2f.numpy(), f'{name}/zbuf': zbuf.numpy(), f'{name}/bary': bary.numpy(), f'{name}/dists': dists.numpy(),
                    f'{name}/g_zbuf': gz.numpy(), f'{name}/g_bary': gb.numpy(), f'{name}/g_dists': gd.numpy(), f'{name}/g_face_verts': gfv.numpy()})
    np.savez_compressed(os.path.join(HERE, 'raster_tiny.npz'), **out)
    print({k: v.shape for k, v in out.items() if k.endswith('p2f')})


if __name__ == '__main__':
    main()
