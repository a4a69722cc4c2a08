#!/bin/bash
for w in 5 5 50 200; do
python bench.py --steps 20 --warmup $w --no-cpu-baseline --no-phases --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('warmup', d['warmup'], 'ms/step', round(d['ms_per_step'],4), round(d['value']))"
done
python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-phases --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps 200 warmup', d['warmup'], 'ms/step', round(d['ms_per_step'],4), round(d['value']))"
