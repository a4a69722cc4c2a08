"""The tiles of the fg forward that finish last in a slow scene state (r06_spike.py found the states), from a -DDBW_TILE_CLOCK -DDBW_DIAG build:
end stamps of every workgroup, the per-tile face lists of the latest ones.  usage: DBW_HIP_LIB=tools/variants/tclk.so r06_spike4.py slow_step fast_step"""
import ctypes, os, sys
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
slow_i, fast_i = int(sys.argv[1]), int(sys.argv[2])
# (the instrumented build renders without the folded env layer, which the one-call step needs: the states are made by a first process on the
# product library -- DBW_SNAP_SAVE=file -- and read by a second one on the instrumented library -- DBW_SNAP_LOAD=file)
snaps = {}
if os.environ.get('DBW_SNAP_LOAD'):
    snaps = {k: v.to(dev) for k, v in torch.load(os.environ['DBW_SNAP_LOAD']).items()}
    model(inp, None)          # (the opacities of a forward: what build_blocks_scene and the fg pass take)
else:
    for i in range(max(slow_i, fast_i) + 1):
        if i in (slow_i, fast_i): snaps[i] = step.params.flat.clone()
        step(inp)
    torch.cuda.synchronize()
    if os.environ.get('DBW_SNAP_SAVE'):
        torch.save({k: v.cpu() for k, v in snaps.items()}, os.environ['DBW_SNAP_SAVE'])
        sys.exit(0)
lib = _lib.load()
B, H, W = a.views, a.H, a.W
tx_, ty_ = (W + 7) // 8, (H + 7) // 8
tiles = tx_ * ty_
NB = 1 << 17
dbw_blocks = 8 * ((B * tiles + 7) // 8)
for name, i in (('fast', fast_i), ('slow', slow_i)):
    step.params.flat.copy_(snaps[i])
    with torch.no_grad():
        scene = model.build_blocks_scene(filter_transparent=False)
        alpha = model._alpha.detach().repeat_interleave(model.BNF).contiguous() if getattr(model, '_alpha', None) is not None else None
    r = model.renderer
    cfg = r._cfg(scene.faces.shape[0], lds_aggregate=True)
    Kmat = r.cameras.K[0].contiguous()
    cl = ops.project_clip(scene.verts.detach(), scene.faces, inp['R'], inp['T'], Kmat, cfg.eps, cfg.z_clip, cfg.persp)
    fvc = cl['face_verts'].view(-1, 3, 3)
    for _ in range(2):
        state = ops._render_fwd_fused(fvc, cl, B, cfg, scene.face_uvs, scene.face_map, scene.map_desc, scene.maps.detach(), alpha, r._bg, 2)
    torch.cuda.synchronize()
    buf = (ctypes.c_uint * (NB * 4))()
    lib.dbw_debug_read_tile_clock(buf, NB)
    t = np.frombuffer(buf, dtype=np.uint32).reshape(NB, 4)
    t = t[:dbw_blocks].copy()
    t = t[t[:, 2] != 0]
    hwid = t[:, 0] >> 16          # HW_REG_HW_ID: wave [3:0], SIMD [5:4], pipe [7:6], CU [11:8], shader array [12], shader engine [15:13]
    t[:, 0] &= 0xffff
    t0 = int(t[:, 3].astype(np.int64).min())
    end = t[:, 2].astype(np.int64) - t0
    dur = (t[:, 2].astype(np.int64) - t[:, 3].astype(np.int64))
    lo = np.argsort(dur)[-6:][::-1]
    print('   longest-running workgroups:', '; '.join('view %d tile (%d, %d) from %.1f to %.1f us' % (int(t[j, 0]), int(t[j, 1]) >> 16, int(t[j, 1]) & 0xffff,
                                                                                                  (int(t[j, 3]) - t0) / 100.0, end[j] / 100.0) for j in lo))
    order = np.argsort(end)
    print('%s state (step %d): %d workgroups; end stamps (us after the first to finish): p50 %.1f p90 %.1f p99 %.1f p99.9 %.1f last %.1f' %
          (name, i, len(t), *(np.percentile(end, q) / 100.0 for q in (50, 90, 99, 99.9)), end.max() / 100.0))
    ws = ops._render_fwd_fused(fvc, cl, B, cfg, scene.face_uvs, scene.face_map, scene.map_desc, scene.maps.detach(), alpha, r._bg, 2, stage=1)[0]
    off = (ctypes.c_ulonglong * 6)()
    lib.dbw_debug_cell_layout(ctypes.c_int64(fvc.shape[0]), B, H, W, off)
    wsb = ws.view(torch.uint8)
    torch.cuda.synchronize()
    cell = wsb[off[1]:off[1] + B * tiles * 8].view(torch.int32).view(B * tiles, 2).cpu().numpy()
    first = cl['first_idx'].cpu().numpy()
    # the launch order (work list): a permutation of every XCD segment's tiles?  where do its heavy tiles sit?
    work = wsb[off[2]:off[2] + B * tiles * 8].view(torch.int32).view(B * tiles, 2).cpu().numpy()
    hdr = wsb[off[0]:off[0] + 4 * 129].view(torch.int32).cpu().numpy()
    per = (B * tiles + 7) // 8
    for x in range(8):
        seg = work[x * per:min((x + 1) * per, B * tiles)]
        ids = (seg[:, 0].astype(np.int64) * ty_ + (seg[:, 1] >> 16)) * tx_ + (seg[:, 1] & 0xffff)
        cnts = cell[ids, 1]
        q = len(seg) // 8
        print('   segment %d: %d entries, %d distinct tiles, inside its own range %s; classes %s; mean list length by eighth of the order: %s; last finisher at %.1f us' %
              (x, len(seg), len(np.unique(ids)), bool(((ids >= x * per) & (ids < (x + 1) * per)).all()), list(hdr[1 + x * 16:1 + x * 16 + 10]),
               ' '.join('%.1f' % cnts[k * q:(k + 1) * q].mean() for k in range(8)),
               max([end[j] for j in range(len(t)) if x * per <= (int(t[j, 0]) * ty_ + (int(t[j, 1]) >> 16)) * tx_ + (int(t[j, 1]) & 0xffff) < (x + 1) * per] or [0]) / 100.0))
    tid = (t[:, 0].astype(np.int64) * ty_ + (t[:, 1] >> 16)) * tx_ + (t[:, 1] & 0xffff)
    for x in range(8):
        m = (tid >= x * per) & (tid < (x + 1) * per)
        e_, c_ = end[m] / 100.0, cell[tid[m], 1]
        late = e_ > 300
        d_ = dur[m] / 100.0
        st_ = (t[m, 3].astype(np.int64) - t0) / 100.0
        emp = c_ == 0
        print('   segment %d durations: empty tiles p50 %.1f p90 %.1f us, occupied p50 %.1f p90 %.1f max %.1f us; workgroups in flight at 50 / 100 / 200 us: %d %d %d' %
              (x, np.percentile(d_[emp], 50), np.percentile(d_[emp], 90), np.percentile(d_[~emp], 50), np.percentile(d_[~emp], 90), d_.max(),
               *(int(((st_ <= a0) & (e_ > a0)).sum()) for a0 in (50, 100, 200))))
        print('   segment %d end stamps: p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f us; finishing after 300 us: %d workgroups, list lengths mean %.1f max %d; '
              'completions per 50 us: %s' % (x, *(np.percentile(e_, q) for q in (10, 50, 90, 99)), e_.max(), int(late.sum()), c_[late].mean() if late.any() else 0,
                                             c_[late].max() if late.any() else 0, ' '.join(str(int(((e_ >= a0) & (e_ < a0 + 50)).sum())) for a0 in range(0, 800, 50))))
    m0 = (tid >= 0) & (tid < per)
    st0, en0, c0 = (t[m0, 3].astype(np.int64) - t0) / 100.0, end[m0] / 100.0, cell[tid[m0], 1]
    print('   segment 0, workgroups in flight every 10 us:', ' '.join(str(int(((st0 <= a0) & (en0 > a0)).sum())) for a0 in range(0, int(en0.max()) + 10, 10)))
    lastw = en0 > en0.max() - 30
    print('   segment 0, workgroups finishing in its last 30 us: %d, list lengths: %s; their run times p50 %.1f p90 %.1f max %.1f us' %
          (int(lastw.sum()), np.bincount(np.clip(c0[lastw], 0, 40)).tolist(), *(np.percentile((en0 - st0)[lastw], q) for q in (50, 90, 100))))
    for lo_, hi_ in ((0, 0), (1, 3), (4, 7), (8, 15), (16, 31), (32, 63), (64, 999)):
        mm = (c0 >= lo_) & (c0 <= hi_)
        if mm.any(): print('      lists of %d-%d faces: %d tiles, run time p50 %.1f p90 %.1f us, started p50 %.1f p90 %.1f us' %
                           (lo_, hi_, int(mm.sum()), *(np.percentile((en0 - st0)[mm], q) for q in (50, 90)), *(np.percentile(st0[mm], q) for q in (50, 90))))
    for x in range(8):          # where the hardware put the occupied and the empty tiles of a segment
        m = (tid >= x * per) & (tid < (x + 1) * per)
        oc = m & (cell[tid, 1] > 0)
        em = m & (cell[tid, 1] == 0)
        f = lambda sel, sh, bits: np.bincount((hwid[sel] >> sh) & ((1 << bits) - 1), minlength=1 << bits).tolist()
        print('   segment %d: occupied tiles by SIMD %s, by shader engine %s, by CU %s | empty tiles by SIMD %s, by shader engine %s' %
              (x, f(oc, 4, 2), f(oc, 13, 3), f(oc, 8, 4), f(em, 4, 2), f(em, 13, 3)))
    for j in order[-8:][::-1]:
        n, rc = int(t[j, 0]), int(t[j, 1])
        row, col = rc >> 16, rc & 0xffff
        L = (n * ty_ + row) * tx_ + col
        print('   view %2d tile (%2d, %2d): finished at %.1f us; list of %d faces' % (n, row, col, end[j] / 100.0, cell[L, 1]))
    j = lo[0]
    n, rc = int(t[j, 0]), int(t[j, 1]); row, col = rc >> 16, rc & 0xffff
    L = (n * ty_ + row) * tx_ + col
    cnt, o = int(cell[L, 1]), int(cell[L, 0])
    if cnt > 0:
        pool = wsb[off[4]:off[4] + 4 * (o + cnt)].view(torch.int32)[o:o + cnt].cpu().numpy() & 0xfffff
        v = fvc[torch.as_tensor(first[n] + pool, device=dev).long()].float().cpu().numpy()
        x, y, z = v[:, :, 0], v[:, :, 1], v[:, :, 2]
        area = (x[:, 2] - x[:, 0]) * (y[:, 1] - y[:, 0]) - (y[:, 2] - y[:, 0]) * (x[:, 1] - x[:, 0])
        print('   its faces (view-local index: |area|, z range):', ' '.join('%d:%.2g,%.2f-%.2f' % (f, abs(a_), zz.min(), zz.max()) for f, a_, zz in zip(pool, area, z)))
