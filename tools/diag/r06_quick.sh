#!/bin/bash
# round 6 quick GPU call: usage: r06_quick.sh <tag> "<pytest args or ->" <epoch> variant...   (variant = name of tools/variants/<name>.so, or "tree" = the in-tree build)
# prints, per variant (alternating, two rounds): ms/step of the driver protocol, kernels alone / in the step
tag=$1; tests=$2; epoch=$3; shift 3
O=gpurun_out/r06/$tag; mkdir -p $O; export TMPDIR=/tmp
if [ "$tests" != "-" ]; then timeout 1500 python -m pytest $tests -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log; fi
for round in 1 2; do
  for v in "$@"; do
    if [ $v = tree ]; then unset DBW_HIP_LIB; else export DBW_HIP_LIB=tools/variants/$v.so; fi
    timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-phases --no-extras --epoch $epoch > $O/bench_${v}_$round.json 2> $O/bench_${v}_$round.err
    python - $O/bench_${v}_$round.json $v <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    r = d['roofline']
    print('%-14s %.4f ms/step  alone %s  in step %s  frac %.3f' % (sys.argv[2], d['ms_per_step'], r.get('all_kernels_ms'), r.get('all_kernels_ms_in_step'), r['frac']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
  done
done
unset DBW_HIP_LIB
