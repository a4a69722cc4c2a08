#!/bin/bash
# One gpurun session of round 4 (everything lands under gpurun_out/r04/<tag>): usage: tools/gpu_call.sh <tag> <stage>...
# stages: cstep_tests | tests '<pytest args>' | all_tests | times [epoch] | trace <views> <epoch> | bench
O=gpurun_out/r04/$1; shift; mkdir -p $O; export TMPDIR=/tmp
while [ $# -gt 0 ]; do
  case $1 in
    cstep_tests) timeout 1500 python -m pytest tests/test_gpu_c_step.py -x -q > $O/cstep_tests.log 2>&1; tail -15 $O/cstep_tests.log;;
    tests) timeout 1500 python -m pytest $2 -x -q > $O/tests.log 2>&1; tail -40 $O/tests.log; shift;;
    all_tests) timeout 2400 python -m pytest tests -m gpu -q > $O/all_tests.log 2>&1; tail -15 $O/all_tests.log;;
    times) timeout 900 python tools/diag/cstep_times.py $2 > $O/times_$2.log 2>&1; cat $O/times_$2.log; shift;;
    trace) v=$2; e=$3; shift 2
      DBW_EPOCH=$e timeout 600 rocprofv3 --kernel-trace -d $O/t -o p --output-format csv -- python tools/diag/trace_cfg.py $v 300 400 10 10 256 14 > $O/trace_${v}_$e.log 2>&1
      csv=$(find $O/t -name "*kernel_trace.csv" | head -1)
      python tools/step_sequence.py $csv > $O/step_sequence_B${v}_epoch$e.txt 2>&1
      python tools/rocprof_csv_summary.py $csv $O/kernel_stats_B${v}_epoch$e.txt "14 training steps, $v views of 400x300, 10 blocks, faces_per_pixel 10, 256^2 textures, epoch $e (tools/diag/trace_cfg.py; rocprofv3 --kernel-trace)" > /dev/null
      rm -rf $O/t; cat $O/step_sequence_B${v}_epoch$e.txt;;
    ablate) timeout 900 python tools/ablate.py $2 > $O/ablate.log 2>&1; cat $O/ablate.log; shift;;
    bench) timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cut -c1-1500 $O/bench.json;;
  esac
  shift
done
