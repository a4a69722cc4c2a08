#!/bin/bash
# per-kernel stats of a short bench run at one epoch for several library variants.  usage: r06_trace.sh <tag> <epoch> variant...
tag=$1; epoch=$2; shift 2
O=gpurun_out/r06/$tag; mkdir -p $O; export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = tree ]; then unset DBW_HIP_LIB; else export DBW_HIP_LIB=tools/variants/$v.so; fi
  timeout 600 rocprofv3 --kernel-trace -d $O/t_$v -o p --output-format csv -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-phases --no-extras --epoch $epoch > $O/trace_$v.log 2>&1
  csv=$(find $O/t_$v -name "*kernel_trace.csv" | head -1)
  echo "== $v"; grep -o '"ms_per_step": [0-9.]*' $O/trace_$v.log | head -1
  python tools/rocprof_csv_summary.py $csv $O/kernel_stats_${v}_epoch$epoch.txt "bench.py --steps 12 --epoch $epoch, variant $v (rocprofv3 --kernel-trace)" | head -12 | cut -c1-150
  python tools/step_sequence.py $csv > $O/step_sequence_${v}_epoch$epoch.txt 2>&1
  rm -rf $O/t_$v
done
unset DBW_HIP_LIB
