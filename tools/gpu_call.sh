#!/bin/bash
# One gpurun session of round 6 (everything lands under gpurun_out/r06/<tag>): usage: tools/gpu_call.sh <tag> <stage>...
# stages: cstep_tests | tests '<pytest args>' | all_tests | times [epoch] | trace <views> <epoch> | ablate '<epoch flags...>' | profiles | bench
O=gpurun_out/r06/$1; shift; mkdir -p $O; export TMPDIR=/tmp
while [ $# -gt 0 ]; do
  case $1 in
    cstep_tests) timeout 1500 python -m pytest tests/test_gpu_c_step.py -x -q > $O/cstep_tests.log 2>&1; tail -15 $O/cstep_tests.log;;
    tests) timeout 1500 python -m pytest $2 -x -q > $O/tests.log 2>&1; tail -40 $O/tests.log; shift;;
    all_tests) timeout 2400 python -m pytest tests -m gpu -q > $O/all_tests.log 2>&1; tail -15 $O/all_tests.log;;
    times) timeout 900 python tools/diag/cstep_times.py $2 > $O/times_${2%% *}.log 2>&1; cat $O/times_${2%% *}.log; shift;;
    trace) v=$2; e=$3; shift 2
      DBW_EPOCH=$e timeout 600 rocprofv3 --kernel-trace -d $O/t -o p --output-format csv -- python tools/diag/trace_cfg.py $v 300 400 10 10 256 14 > $O/trace_${v}_$e.log 2>&1   # (DBW_READS=1 in the environment: with host reads)
      csv=$(find $O/t -name "*kernel_trace.csv" | head -1)
      python tools/step_sequence.py $csv > $O/step_sequence_B${v}_epoch$e.txt 2>&1
      python tools/rocprof_csv_summary.py $csv $O/kernel_stats_B${v}_epoch$e.txt "14 training steps, $v views of 400x300, 10 blocks, faces_per_pixel 10, 256^2 textures, epoch $e (tools/diag/trace_cfg.py; rocprofv3 --kernel-trace)" > /dev/null
      rm -rf $O/t; cat $O/step_sequence_B${v}_epoch$e.txt;;
    ablate) timeout 900 python tools/ablate.py $2 > $O/ablate.log 2>&1; cat $O/ablate.log; shift;;
    profiles)   # the evidence kept under profiles/ (copied from gpurun_out/r06/<tag>/ by hand): kernel stats + step sequences of the three phases,
                # every kernel alone (--no-overlap), batch 4, PMC counters with the byte-counter calibration, the bench line
      for e in 0 800 1600; do
        timeout 900 rocprofv3 --kernel-trace -d $O/t -o p --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-phases --no-extras --epoch $e > $O/trace_$e.log 2>&1
        csv=$(find $O/t -name "*kernel_trace.csv" | head -1)
        python tools/rocprof_csv_summary.py $csv $O/r06_kernel_stats_epoch$e.txt "python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-phases --no-extras --epoch $e (rocprofv3 --kernel-trace)" > /dev/null
        python tools/step_sequence.py $csv > $O/r06_step_sequence_epoch$e.txt 2>&1
        rm -rf $O/t
      done
      timeout 900 rocprofv3 --kernel-trace -d $O/t -o p --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-phases --no-extras --no-overlap > $O/trace_alone.log 2>&1
      csv=$(find $O/t -name "*kernel_trace.csv" | head -1)
      python tools/rocprof_csv_summary.py $csv $O/r06_kernel_stats_epoch0_alone.txt "python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-phases --no-extras --no-overlap: everything in order on ONE stream, every kernel alone on the GPU (rocprofv3 --kernel-trace)" > /dev/null
      python tools/step_sequence.py $csv > $O/r06_step_sequence_epoch0_alone.txt 2>&1
      rm -rf $O/t
      DBW_EPOCH=0 timeout 600 rocprofv3 --kernel-trace -d $O/t -o p --output-format csv -- python tools/diag/trace_cfg.py 4 300 400 10 10 256 14 > $O/trace_b4.log 2>&1
      csv=$(find $O/t -name "*kernel_trace.csv" | head -1)
      python tools/step_sequence.py $csv > $O/r06_step_sequence_batch4.txt 2>&1
      python tools/rocprof_csv_summary.py $csv $O/r06_kernel_stats_batch4.txt "14 training steps of 4 views (configs/dtu/default.yml:28) of 400x300, 10 blocks, faces_per_pixel 10 (tools/diag/trace_cfg.py; rocprofv3 --kernel-trace)" > /dev/null
      rm -rf $O/t
      bash tools/pmc_sq.sh $O/pmc 0 > $O/r06_pmc_sq_counters.txt 2>&1
      cp $O/pmc/bench_counters.json $O/r06_pmc_counters.json
      rm -rf $O/pmc/g1 $O/pmc/g2 $O/pmc/g3 $O/pmc/g4 $O/pmc/calib_FETCH_SIZE $O/pmc/calib_WRITE_SIZE
      tail -5 $O/r06_pmc_sq_counters.txt; head -24 $O/r06_step_sequence_epoch0.txt;;
    bench) timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cut -c1-1500 $O/bench.json;;
  esac
  shift
done
