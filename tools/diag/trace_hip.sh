# usage: trace_hip.sh <tag> <script> [args...]: rocprofv3 kernel + HIP API trace; lists the HIP calls that took longer than 100 us
O=gpurun_out/r05/$1; mkdir -p $O; export TMPDIR=/tmp; shift
timeout 600 rocprofv3 --kernel-trace --hip-trace -d $O/t -o p --output-format csv -- python "$@" > $O/trace.log 2>&1
ls $O/t/* | head
api=$(find $O/t -name "*hip_api_trace.csv" | head -1); k=$(find $O/t -name "*kernel_trace.csv" | head -1)
python - "$api" "$k" > $O/hip_long_calls.txt <<'PY'
import csv, sys
api = list(csv.DictReader(open(sys.argv[1], encoding='utf-8', errors='replace')))
ker = sorted(csv.DictReader(open(sys.argv[2], encoding='utf-8', errors='replace')), key=lambda r: int(r['Start_Timestamp']))
adam = [r for r in ker if 'adam_' in r['Kernel_Name']]
t0, t1 = int(adam[9]['End_Timestamp']), int(adam[11]['End_Timestamp'])
print('two steady-state steps: host HIP calls > 100 us, and every call inside the window of the first step_prologue stall')
pro = [r for r in ker if 'step_prologue' in r['Kernel_Name'] and t0 <= int(r['Start_Timestamp']) <= t1]
for r in api:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s < t0 or s > t1: continue
    if e - s > 100000: print(f'{(s - t0) / 1e3:10.1f} us  {(e - s) / 1e3:9.1f} us  {r["Function"]}')
if pro:
    ps, pe = int(pro[0]['Start_Timestamp']), int(pro[0]['End_Timestamp'])
    print(f'prologue kernel on the GPU: {(ps - t0) / 1e3:.1f} .. {(pe - t0) / 1e3:.1f} us; HIP calls that overlap it:')
    for r in api:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if e >= ps - 200000 and s <= pe: print(f'{(s - t0) / 1e3:10.1f} us  {(e - s) / 1e3:9.1f} us  {r["Function"]}')
PY
rm -rf $O/t; head -150 $O/hip_long_calls.txt
