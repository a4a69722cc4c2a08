#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace) of a few training steps with / without the per-tile face lists
mkdir -p gpurun_out/c3; export TMPDIR=/tmp
for f in 0 4096; do
  DBW_DEBUG_FLAGS=$f DBW_STEPS=6 rocprofv3 --kernel-trace -d gpurun_out/c3/t$f -o p -- python tools/pmc_target.py > gpurun_out/c3/t$f.log 2>&1
  db=$(find gpurun_out/c3/t$f -name "*.db" | head -1)
  python tools/rocprof_summary.py $db gpurun_out/c3/stats_flags$f.txt "flags $f" | head -24
  rm -rf gpurun_out/c3/t$f
done
