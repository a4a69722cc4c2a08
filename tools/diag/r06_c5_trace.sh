#!/bin/bash
# kernel statistics + one steady-state step of config 5's share of one GPU (25 views of 1920x1080, 50 blocks, faces_per_pixel 16, 512^2 textures).
# usage: r06_c5_trace.sh <tag> [epoch]
O=gpurun_out/r06/$1; e=${2:-0}; mkdir -p $O; export TMPDIR=/tmp
DBW_EPOCH=$e timeout 900 rocprofv3 --kernel-trace -d $O/t -o p --output-format csv -- python tools/diag/trace_cfg.py 25 1080 1920 50 16 512 13 > $O/trace_c5_$e.log 2>&1
csv=$(find $O/t -name "*kernel_trace.csv" | head -1)
s=$([ "$e" = 0 ] && echo "" || echo "_epoch$e")
python tools/step_sequence.py $csv > $O/r06_step_sequence_c5$s.txt 2>&1
python tools/rocprof_csv_summary.py $csv $O/r06_kernel_stats_c5$s.txt "13 training steps of one GPU's share of BASELINE config 5 (25 views of 1920x1080, 50 blocks, faces_per_pixel 16, 512^2 textures), epoch $e (tools/diag/trace_cfg.py; rocprofv3 --kernel-trace)" > /dev/null
rm -rf $O/t; cut -c1-150 $O/r06_step_sequence_c5$s.txt
