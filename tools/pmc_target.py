"""Workload for rocprofv3 --pmc runs: a few full training iterations of the bench config (eager launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch
import bench
from dbw_amd.parallel import ShardedTrainStep
class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
from dbw_amd import _lib
if os.environ.get('DBW_DEBUG_FLAGS'):
    _lib.load().dbw_debug_set_flags(int(os.environ['DBW_DEBUG_FLAGS']))
if os.environ.get('DBW_RENDER_VARIANT'):
    _lib.load().dbw_debug_set_render_variant(int(os.environ['DBW_RENDER_VARIANT']))
if os.environ.get('DBW_PMC_EMPTY'):          # every block far outside every view: all tiles of the fg pass are empty (env layer + epilogue only)
    with torch.no_grad():
        model.T.add_(100.0)
model.sync_free = True
model.set_cur_epoch(int(os.environ.get("DBW_EPOCH", "0")))
step = ShardedTrainStep(model, seed=1)
for _ in range(int(os.environ.get('DBW_STEPS', '3'))):
    step(inp)
torch.cuda.synchronize()
