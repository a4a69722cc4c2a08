# usage: trace_py.sh <tag> <script> [args...]: rocprofv3 kernel trace of a python workload -> step sequence of one steady-state step
O=gpurun_out/r05/$1; mkdir -p $O; export TMPDIR=/tmp; shift
timeout 600 rocprofv3 --kernel-trace -d $O/t -o p --output-format csv -- python "$@" > $O/trace.log 2>&1
csv=$(find $O/t -name "*kernel_trace.csv" | head -1)
python tools/step_sequence.py $csv > $O/step_sequence.txt 2>&1
cp $csv $O/kernel_trace.csv; rm -rf $O/t; wc -l $O/step_sequence.txt
