"""GPU helper: the batch-4 extras of the bench line (eager / hipGraph, with / without host reads) with the regularisers enqueued
before / behind the fg pass."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd import native_step
dev = torch.device('cuda', 0)
orig = native_step.NativeStep.__init__
FLAG = [True]
def init(self, *a, **k):
    orig(self, *a, **k)
    self.regularisers_behind_fg = FLAG[0]
native_step.NativeStep.__init__ = init
for rep in range(2):
    for behind in (True, False):
        FLAG[0] = behind
        row = []
        for graph in (False, True):
            for reads in (True, False):
                r = bench.measure_other(4, 300, 400, 10, 10, 256, dev, steps=100, warmup=10, read_losses=reads, use_graph=graph)
                row.append('%s/%s %.3f' % ('graph' if graph else 'eager', 'reads' if reads else 'no reads', r['ms_per_step']))
        r = bench.measure_other(49, 300, 400, 10, 10, 256, dev, steps=20, warmup=5, lr_scale=0.0)
        print('behind' if behind else 'before', ' | '.join(row), '| 49 views frozen %.4f' % r['ms_per_step'])
