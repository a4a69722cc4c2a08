"""Workload for rocprofv3 kernel traces of other configurations: a few training steps.  usage: trace_cfg.py views H W blocks fpp txt [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd.parallel import ShardedTrainStep
class A: pass
args = A(); args.views, args.H, args.W, args.blocks, args.fpp, args.txt = [int(x) for x in sys.argv[1:7]]
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(args, dev)
model.sync_free = True
model.set_cur_epoch(int(os.environ.get("DBW_EPOCH", "0")))
step = ShardedTrainStep(model, seed=int(os.environ.get('DBW_SEED', '1')))          # (bench.py: 227391)
reads = os.environ.get("DBW_READS", "0") != "0"          # the host reads the loss values of every step (src/trainer.py:143)
if step.cstep is not None:
    step.cstep.read_losses = reads
for _ in range(int(sys.argv[7]) if len(sys.argv) > 7 else 4):
    out = step(inp)
    if reads:
        out.host() if hasattr(out, 'host') else [float(v) for v in out.values()]
torch.cuda.synchronize()
