export DBW_SLICES=8,6,4,3,2,2,2,2,1
timeout 600 python tools/diag/cstep_times.py 0 4 7 12 16 24 c127s1 c127s0 2>&1 | grep epoch
