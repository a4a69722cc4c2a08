#!/bin/bash
# round-3 GPU call: kernel sequence of one training step (rocprofv3 --kernel-trace) at the given epochs
export TMPDIR=/tmp; O=gpurun_out/seq; mkdir -p $O
for e in "$@"; do
  rocprofv3 --kernel-trace -d $O/t$e -o p --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-phases --no-extras --epoch $e > $O/t$e.log 2>&1
  csv=$(find $O/t$e -name "*kernel_trace.csv" | head -1)
  python tools/step_sequence.py $csv > $O/step_sequence_epoch$e.txt 2>&1
  rm -rf $O/t$e
  cut -c1-130 $O/step_sequence_epoch$e.txt
done
