"""Which scene states make the fg forward slow: 32 steps of the bench workload (bench.py's seed and learning rates), every step timed on its
own; for the slowest ones and one ordinary one the parameters of that step are put back and the fg pass's per-tile lists are read (needs a
-DDBW_DIAG build: dbw_debug_cell_layout): lengths, tiles walking their coarse bin, the faces with the largest screen boxes.
usage: DBW_HIP_LIB=tools/variants/<diag build>.so r06_spike.py [nsteps]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import _lib, ops
from dbw_amd.parallel import ShardedTrainStep
class A: pass
a = A(); a.views, a.H, a.W, a.blocks, a.fpp, a.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(a, dev)
model.set_cur_epoch(0); model.sync_free = True
step = ShardedTrainStep(model, lr=5e-3, lr_texture=5e-2, seed=227391)
snaps, ts = [], []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 32):
    snaps.append(step.params.flat.clone())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step(inp)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print('ms per step:', ' '.join('%.3f' % t for t in ts))
med = sorted(ts[3:])[len(ts[3:]) // 2]
slow = [i for i in range(3, len(ts)) if ts[i] > 1.15 * med]
look = slow[:4] + [len(ts) - 1]
lib = _lib.load()
B, H, W = a.views, a.H, a.W
tiles = ((H + 7) // 8) * ((W + 7) // 8)
for i in look:
    step.params.flat.copy_(snaps[i])
    with torch.no_grad():
        scene = model.build_blocks_scene(filter_transparent=False)
        alpha = model._alpha.detach().repeat_interleave(model.BNF).contiguous() if getattr(model, '_alpha', None) is not None else None
    r = model.renderer
    cfg = r._cfg(scene.faces.shape[0], lds_aggregate=True)
    Kmat = r.cameras.K[0].contiguous()
    cl = ops.project_clip(scene.verts.detach(), scene.faces, inp['R'], inp['T'], Kmat, cfg.eps, cfg.z_clip, cfg.persp)
    fvc = cl['face_verts'].view(-1, 3, 3)
    state = ops._render_fwd_fused(fvc, cl, B, cfg, scene.face_uvs, scene.face_map, scene.map_desc, scene.maps.detach(), alpha, r._bg, 2, stage=1)
    ws = state[0]
    off = (ctypes.c_ulonglong * 6)()
    lib.dbw_debug_cell_layout(ctypes.c_int64(fvc.shape[0]), B, H, W, off)
    wsb = ws.view(torch.uint8)
    torch.cuda.synchronize()
    cell = wsb[off[1]:off[1] + B * tiles * 8].view(torch.int32).view(B * tiles, 2).cpu().numpy()
    hdr = wsb[off[0]:off[0] + 4].view(torch.int32).cpu().numpy()
    cnt = cell[:, 1].astype(np.int64)
    occ = cnt[cnt > 0]
    print('step %d (%.3f ms): tiles walking their coarse bin %d, occupied %d, (tile, face) pairs %d, faces per occupied tile mean %.1f p99 %d max %d; pool used %d of %d'
          % (i, ts[i], (cnt < 0).sum(), len(occ), occ.sum(), occ.mean(), np.percentile(occ, 99), occ.max(), hdr[0], off[5]))
    pv = cnt.reshape(B, tiles)
    print('   pairs per view:', ' '.join(str(int(np.clip(v, 0, None).sum())) for v in pv), '| walking per view:', ' '.join(str(int((v < 0).sum())) for v in pv))
    # the faces the rasteriser evaluates with plain IEEE divisions (no REC_FAST, raster_math.h: make_face_rec), by the bound they miss, and
    # the tiles their screen boxes cover
    first, num = cl['first_idx'].cpu().numpy(), cl['num_faces'].cpu().numpy()
    rows = torch.cat([torch.arange(int(f), int(f) + int(n_), device=dev) for f, n_ in zip(first, num)])
    v = fvc[rows].float()
    x, y, z = v[:, :, 0], v[:, :, 1], v[:, :, 2]
    area = (x[:, 2] - x[:, 0]) * (y[:, 1] - y[:, 0]) - (y[:, 2] - y[:, 0]) * (x[:, 1] - x[:, 0])
    alive = (z.min(1).values >= 1e-8) & (area.abs() > 1e-8)
    cmax = torch.maximum(x.abs().max(1).values, y.abs().max(1).values)
    big_c, big_z, small_a, big_a = cmax > 1024, z.max(1).values > 1024, area.abs() < 9.5367432e-7, area.abs() > 8388608
    slow = alive & (big_c | big_z | small_a | big_a)
    xr, yr = 400.0 / 300.0, 1.0
    on = (x.max(1).values > -xr) & (x.min(1).values < xr) & (y.max(1).values > -yr) & (y.min(1).values < yr)
    tx = ((x.max(1).values.clamp(-xr, xr) - x.min(1).values.clamp(-xr, xr)) / (2 * xr) * 50).ceil().clamp(min=1)
    ty = ((y.max(1).values.clamp(-yr, yr) - y.min(1).values.clamp(-yr, yr)) / (2 * yr) * 38).ceil().clamp(min=1)
    sel = slow & on
    print('   faces listed %d, alive %d; evaluated with IEEE divisions and on screen: %d (coordinates > 2^10: %d, z > 2^10: %d, |area| < 2^-20: %d, |area| > 2^23: %d), '
          'tiles under their boxes %d of %d occupied' % (len(rows), int(alive.sum()), int(sel.sum()), int((sel & big_c).sum()), int((sel & big_z).sum()),
                                                         int((sel & small_a).sum()), int((sel & big_a).sum()), int((tx * ty)[sel].sum()), len(occ)))
    if int(sel.sum()):
        j = torch.nonzero(sel).flatten()[:6]
        for q in j:
            print('      face row %d: xy %s z %s area %.3g' % (int(rows[q]), [round(float(t), 2) for t in v[q, :, :2].flatten()], [round(float(t), 4) for t in z[q]], float(area[q])))
    for flags in (0, 1, 2, 1 << 14, 1 << 15):        # dbw_debug_set_flags: IEEE divisions everywhere, no tile culling, no layer loop, no insert
        lib.dbw_debug_set_flags(flags)
        kb = bench.kernel_breakdown(model, inp, reps=3)
        print('   flags %5d: operator-level kernels:' % flags, {k.replace('render_', '').replace('_fused', ''): round(v[0], 4) for k, v in kb.items()})
    lib.dbw_debug_set_flags(0)
    if hasattr(lib, 'dbw_debug_read_fwd_profile'):          # a -DDBW_PROFILE_FWD build: the event counts of the fg forward in this state
        buf = (ctypes.c_ulonglong * 16)()
        torch.cuda.synchronize(); lib.dbw_debug_read_fwd_profile(buf, 1)
        bench.kernel_breakdown(model, inp, reps=1); torch.cuda.synchronize()
        lib.dbw_debug_read_fwd_profile(buf, 1)
        v, L = list(buf), 3.0
        print('   per launch: tiles %.1f k (%.1f k with faces), (tile, face) pairs %.3f M (culled by the tile-vs-edge test %.3f M), with a pixel in the box %.3f M, (pixel, face) '
              'evaluations %.2f M, kept %.2f M, wave re-evaluations with IEEE divisions %.4f M; cycles: prologue %.1f%% binning %.1f%% face loop %.1f%% shading %.1f%%' %
              (v[9] / L / 1e3, v[10] / L / 1e3, v[4] / L / 1e6, v[11] / L / 1e6, v[5] / L / 1e6, v[6] / L / 1e6, v[7] / L / 1e6, v[8] / L / 1e6,
               100 * v[12] / v[3], 100 * v[0] / v[3], 100 * v[1] / v[3], 100 * v[2] / v[3]))
