O=gpurun_out/r05/$1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_c_step.py -x -q -k "split or equals_native" > $O/tests.log 2>&1; tail -5 $O/tests.log
shift
for cfg in "$@"; do echo "slices $cfg"; DBW_SLICES=$cfg timeout 300 python tools/diag/cstep_times.py 0 4 7 c127s1 2>&1 | grep epoch; done
timeout 300 python tools/diag/cstep_times.py 0 4 7 c127s0 2>&1 | grep epoch
