# usage: trace_scratch.sh <tag> <script> [args...]: rocprofv3 kernel + scratch-memory trace; prints the scratch events
O=gpurun_out/r05/$1; mkdir -p $O; export TMPDIR=/tmp; shift
timeout 600 rocprofv3 --kernel-trace --scratch-memory-trace -d $O/t -o p --output-format csv -- python "$@" > $O/trace.log 2>&1
ls $O/t/
f=$(find $O/t -name "*scratch_memory_trace.csv" | head -1)
wc -l $f; head -3 $f; tail -40 $f | cut -c1-200
cp $f $O/scratch.csv; cp $(find $O/t -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv; rm -rf $O/t
