#!/bin/bash
# round-3 evidence: per-kernel rocprofv3 stats of the three training phases and of configs 4 / 5, one step's kernel sequence, the bench
# line, SQ / HBM counters (separate --pmc passes), cycle accounting of the fused forward.  Run on the GPU box; copies go to profiles/.
O=gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp
# (the cycle-accounting build must be of the same sources as the library: rebuild it when it is older)
[ tools/variants/fprof.so -nt differentiable-blocksworld_amd/dbw_amd/libdbw_hip.so ] || tools/variants.sh fprof "-DDBW_PROFILE_FWD" > /dev/null 2>&1
for e in 0 800 1600; do
  rocprofv3 --kernel-trace -d $O/t$e -o p --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-phases --no-extras --epoch $e > $O/t$e.log 2>&1
  csv=$(find $O/t$e -name "*kernel_trace.csv" | head -1)
  python tools/rocprof_csv_summary.py $csv $O/r03_kernel_stats_epoch$e.txt "python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-phases --no-extras --epoch $e (rocprofv3 --kernel-trace)" > /dev/null
  python tools/step_sequence.py $csv > $O/r03_step_sequence_epoch$e.txt 2>&1
  rm -rf $O/t$e
done
for c in "8 576 768 20 16 256 c4" "25 1080 1920 50 16 512 c5"; do
  set -- $c
  rocprofv3 --kernel-trace -d $O/t$7 -o p --output-format csv -- python tools/diag/trace_cfg.py $1 $2 $3 $4 $5 $6 6 > $O/t$7.log 2>&1
  csv=$(find $O/t$7 -name "*kernel_trace.csv" | head -1)
  python tools/rocprof_csv_summary.py $csv $O/r03_kernel_stats_$7.txt "6 training steps of BASELINE config ${7#c} (per-GPU share: $1 views of ${3}x${2}, $4 blocks, faces_per_pixel $5, ${6}^2 textures), tools/diag/trace_cfg.py (rocprofv3 --kernel-trace)" > /dev/null
  rm -rf $O/t$7
done
bash tools/pmc_sq.sh $O/pmc 0 > $O/r03_pmc_sq_counters.txt 2>&1
cp $O/pmc/bench_counters.json $O/r03_pmc_counters.json
rm -rf $O/pmc/g1 $O/pmc/g2 $O/pmc/g3 $O/pmc/g4
DBW_HIP_LIB=tools/variants/fprof.so python tools/fwd_cycles.py 0 > $O/r03_fwd_cycle_accounting.txt 2>&1
DBW_HIP_LIB=tools/variants/fprof.so python tools/fwd_timeline.py 0 > $O/r03_fwd_timeline.txt 2>&1
python bench.py > $O/r03_bench.json 2> $O/r03_bench.err
tail -2 $O/r03_bench.err
head -12 $O/r03_kernel_stats_epoch0.txt
python -c "
import json; d=json.load(open('$O/r03_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['all_kernels_ms']); print(d['phases']['schedule_weighted']); print({k: d[k]['ms_per_step'] for k in ('batch4','batch4_graph','sustained')}, {k: v['ms_per_step'] for k, v in d['configs'].items()}); print(d['cpu_baseline'])"
