"""GPU helper: does a process get slower after it has run other workloads?  bench.measure_other on the headline workload, fresh, then
again after other configurations have been built, run and freed in the same process."""
import sys, time, torch
sys.path.insert(0, '.'); import bench
from dbw_amd import ops
dev = torch.device('cuda', 0)
def head(tag, **kw):
    r = bench.measure_other(49, 300, 400, 10, 10, 256, dev, steps=kw.pop('steps', 100), warmup=10, **kw)
    print('%-40s %.4f ms/step over %d steps; reserved %.2f GB' % (tag, r['ms_per_step'], r['steps'], torch.cuda.memory_reserved() / 2**30), flush=True)
head('fresh, 100 steps')
head('again, 100 steps')
head('200 steps unsynced', steps=200)
head('lr 0, 200 steps, >= 2 s', steps=200, lr_scale=0.0, min_seconds=2.0)
head('lr 0, 100 steps, >= 2 s', steps=100, lr_scale=0.0, min_seconds=2.0)
head('lr 1, 100 steps', steps=100)
r = bench.measure_other(4, 300, 400, 10, 10, 256, dev, steps=100, warmup=10, read_losses=True); print('batch4 reads', r['ms_per_step'], flush=True)
r = bench.measure_other(4, 300, 400, 10, 10, 256, dev, steps=100, warmup=10); print('batch4', r['ms_per_step'], flush=True)
head('after batch4')
r = bench.measure_other(8, 576, 768, 20, 16, 256, dev, steps=20, warmup=3); print('c4', r['ms_per_step'], flush=True)
head('after c4')
r = bench.measure_other(8, 576, 768, 20, 16, 256, dev, steps=20, warmup=3); print('c4 again', r['ms_per_step'], flush=True)
