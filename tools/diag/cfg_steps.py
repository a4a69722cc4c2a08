"""GPU helper: ms per step of one GPU's share of BASELINE configs 4 and 5 (bench.measure_other) under the loaded library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
dev = torch.device('cuda', 0)
for name, cfg, steps in (('c4', (8, 576, 768, 20, 16, 256), 20), ('c5', (25, 1080, 1920, 50, 16, 512), 5)):
    for epoch in [int(x) for x in sys.argv[1:]] or (0, 800):
        r = bench.measure_other(*cfg, dev, steps=steps, warmup=3, epoch=epoch)
        print(name, 'epoch', epoch, '%.3f ms/step' % r['ms_per_step'])
