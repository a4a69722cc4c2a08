# usage: trace_call.sh <tag> <views> [env assignments...]: kernel trace of 14 steps -> step sequence of one steady-state step
O=gpurun_out/r05/$1; mkdir -p $O; export TMPDIR=/tmp; v=$2; shift 2
for e in "$@"; do export $e; done
DBW_EPOCH=${DBW_EPOCH:-0} DBW_READS=1 timeout 600 rocprofv3 --kernel-trace -d $O/t -o p --output-format csv -- python tools/diag/trace_cfg.py $v 300 400 10 10 256 14 > $O/trace.log 2>&1
csv=$(find $O/t -name "*kernel_trace.csv" | head -1)
python tools/step_sequence.py $csv > $O/step_sequence.txt 2>&1
rm -rf $O/t; cat $O/step_sequence.txt
