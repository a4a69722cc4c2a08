import json,sys,subprocess,os
out=subprocess.run([sys.executable,'bench.py','--steps','10','--warmup','3','--no-cpu-baseline','--no-phases','--no-extras'],capture_output=True,text=True)
d=json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
print(os.environ.get('DBW_HIP_LIB','default'), d['ms_per_step'], d['roofline']['all_kernels_ms'], d['roofline']['all_kernels_ms_in_step'])
