O=gpurun_out/r05/c3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_c_step.py -x -q -k "split or equals_native" > $O/tests.log 2>&1; tail -5 $O/tests.log
for c in 1 2 3 4; do echo "classes $c"; DBW_SPLIT_CLASSES=$c timeout 300 python tools/diag/cstep_times.py 0 4 7 c127s1 2>&1 | grep epoch; done
