#!/bin/bash
# the whole bench.py line (driver protocol + every extra leg), with a readable digest.  usage: r06_bench.sh <tag> [bench args]
O=gpurun_out/r06/$1; shift; mkdir -p $O
timeout 1800 python bench.py "$@" > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - $O/bench.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print({k: d[k] for k in ('value', 'ms_per_step')})
print('roofline', {k: v for k, v in d['roofline'].items() if k not in ('counters', 'top_kernels')})
print('top', d['roofline'].get('top_kernels'))
print('phases', d.get('phases'))
for k in ('batch4', 'batch7', 'batch4_no_reads', 'perceptual', 'sustained'):
    print(k, d.get(k))
print('configs', json.dumps(d.get('configs'))[:1800])
print('cpu', d.get('cpu_baseline'))
PY
