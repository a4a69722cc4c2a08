import sys, time, torch
sys.path.insert(0, '.'); import bench
from dbw_amd import ops
dev = torch.device('cuda', 0)
def c4(tag):
    r = bench.measure_other(8, 576, 768, 20, 16, 256, dev, steps=20, warmup=3)
    ar = ops.ARENA.buf.get(dev)
    print('%-30s c4 %.4f ms/step; arena %.1f MB' % (tag, r['ms_per_step'], 0 if ar is None else ar.numel() / 2**20), flush=True)
def head(epoch):
    r = bench.measure_other(49, 300, 400, 10, 10, 256, dev, steps=20, warmup=5, epoch=epoch)
    ar = ops.ARENA.buf.get(dev)
    print('headline epoch %4d: %.4f ms/step; arena %.1f MB' % (epoch, r['ms_per_step'], 0 if ar is None else ar.numel() / 2**20), flush=True)
c4('fresh')
head(0); c4('after epoch 0')
head(800); c4('after epoch 800')
head(1600); c4('after epoch 1600')
ops.ARENA.buf.clear(); ops.ARENA.off.clear(); ops.ARENA.want.clear(); ops.ARENA.clean.clear()
c4('after arena reset')
