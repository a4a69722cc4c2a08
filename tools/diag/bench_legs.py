"""GPU helper: the extra legs of bench.py in one process, in the order given, reporting after each whether a poll of its C step gave up.
usage: bench_legs.py b4 b7 lbl perc sus c4 c5 c4f c5f"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'differentiable-blocksworld_amd'))
import torch, bench
dev = torch.device('cuda', 0)
LEGS = {'b4': lambda: bench.measure_other(4, 300, 400, 10, 10, 256, dev, steps=200, warmup=20, read_losses=True),
        'b7': lambda: bench.measure_other(7, 300, 400, 10, 10, 256, dev, steps=200, warmup=20, read_losses=True),
        'lbl': lambda: bench.measure_other(4, 300, 400, 10, 10, 256, dev, steps=100, warmup=10, read_losses=True, c_step=False),
        'perc': lambda: bench.measure_perceptual(dev),
        'sus': lambda: bench.measure_other(49, 300, 400, 10, 10, 256, dev, steps=200, warmup=10, lr_scale=0.0, min_seconds=2.0),
        'c4': lambda: bench.measure_other(8, 576, 768, 20, 16, 256, dev, steps=20, warmup=3),
        'c5': lambda: bench.measure_other(25, 1080, 1920, 50, 16, 512, dev, steps=5, warmup=2),
        'c4f': lambda: bench.measure_other(8, 576, 768, 20, 16, 256, dev, steps=20, warmup=3, epoch=800),
        'c5f': lambda: bench.measure_other(25, 1080, 1920, 50, 16, 512, dev, steps=5, warmup=3, epoch=800)}
for leg in sys.argv[1:]:
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        r = LEGS[leg]()
    gave_up = [str(x.message)[:60] for x in w if 'gave up' in str(x.message)]
    print(leg, 'ms_per_step %.4f' % r['ms_per_step'], 'GAVE UP' if gave_up else 'ok', flush=True)
