"""GPU helper: the fg pass with the folded env layer inside a step (HIP events of dbw_train_step_profile), for the library named by DBW_HIP_LIB.
Builds with -DDBW_FOLD_ABL=1/2/3 (tools/variants.sh) leave out the env layer's shading / evaluation / fragment stores: wrong images, right clocks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
from dbw_amd.parallel import ShardedTrainStep
class A: pass
a = A(); a.views, a.H, a.W, a.blocks, a.fpp, a.txt = int(sys.argv[1]) if len(sys.argv) > 1 else 49, 300, 400, 10, 10, 256
dev = torch.device('cuda', 0)
model, inp = bench.build_workload(a, dev)
model.sync_free = True
step = ShardedTrainStep(model, lr=0.0, lr_texture=0.0, seed=1)
for _ in range(5):
    step(inp)
kt = step.cstep.kernel_times(inp, reps=7)
print(os.environ.get('DBW_HIP_LIB', 'default'), {k: round(v, 4) for k, v in kt.items()} if isinstance(kt, dict) else kt)
