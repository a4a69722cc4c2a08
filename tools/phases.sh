#!/bin/bash
# bench the three training phases (coarse+decimated, coarse, fine)
for e in 0 800 1600; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --epoch $e 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('epoch', $e, round(d['value']), 'views/s', round(d['ms_per_step'], 3), 'ms', d['roofline']['all_kernels_ms'], d['roofline'].get('texbins'))"
done
