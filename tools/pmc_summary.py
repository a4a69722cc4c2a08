"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) of tools/pmc_target.py into the per-kernel HBM
traffic table kept under profiles/ (and read by bench.py for roofline.traffic).
usage: pmc_summary.py fetch.db write.db out.json"""
import json, sqlite3, sys
P = 300 * 400
ALG = {10: (20 * P * 10 + 16 * P) * 49, 1: (20 * P * 1 + 16 * P) * 49}


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, counter_value from pmc_events where counter_name = ? order by dispatch_id", (counter,)).fetchall()
    out = {}
    for name, v in rows:
        out.setdefault(name, []).append(v)
    return out


def label(name):
    if 'render_fwd_kernel<10' in name: return 'render_fwd_kernel<10> (fg pass)', 10
    if 'render_fwd_kernel<1,' in name: return 'render_fwd_kernel<1> (env pass)', 1
    return None, None


fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
res = {'_how': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE (one run) and --pmc WRITE_SIZE (a second, separate run) -- python '
               'tools/pmc_target.py (3 eager iterations of the bench config: 49 views, 300x400, K=10); counters are KB per dispatch; '
               'median of the dispatches; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE doubled as MI355X_MICROARCH.md '
               'section HBM prescribes for gfx950; WRITE_SIZE used as reported (uncalibrated per the guide).'}
med = lambda xs: sorted(xs)[len(xs) // 2]
for name in fetch:
    lab, k = label(name)
    if lab:
        f, w = med(fetch[name]), med(write.get(name, [0]))
        res[lab] = {'fetch_kb_raw': round(f), 'write_kb': round(w), 'algorithmic_bytes': ALG[k], 'hbm_bytes': int((2 * f + w) * 1024)}
    elif 'shade_blend_bwd_kernel' in name:
        single = name.replace(' ', '').find('<true,false,true>') >= 0          # the K = 1 instantiation is the env pass
        lab, k = ('shade_blend_bwd_kernel<fused> (env pass)', 1) if single else ('shade_blend_bwd_kernel<fused> (fg pass)', 10)
        f, w = med(fetch[name]), med(write.get(name, [0]))
        res[lab] = {'fetch_kb_raw': round(f), 'write_kb': round(w), 'algorithmic_bytes': ALG[k], 'hbm_bytes': int((2 * f + w) * 1024)}
    elif any(t in name for t in ('texbin_reduce', 'composite_mse', 'coarse_bin')):
        f, w = med(fetch[name]), med(write.get(name, [0]))
        res[name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].strip()] = {'fetch_kb_raw': round(f), 'write_kb': round(w), 'hbm_bytes': int((2 * f + w) * 1024)}
json.dump(res, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(res, indent=1))
