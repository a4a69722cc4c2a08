"""Per-kernel medians of the counters collected by tools/pmc_sq.sh (values in millions per launch, summed over the instances rocprofv3
reports; FETCH_SIZE / WRITE_SIZE in MB with the gfx950 FETCH x2 correction of MI355X_MICROARCH.md applied to hbm_mb)."""
import csv, glob, os, sys, json
out = sys.argv[1]
vals = {}
for f in glob.glob(os.path.join(out, 'g*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        n = n.split('(')[0].strip()
        if not any(k in n for k in ('render_fwd', 'render_bwd', 'shade_blend_bwd', 'composite', 'texbin', 'coarse_bin', 'cell_bin', 'work_scatter', 'face_setup', 'shade_setup', 'project_clip', 'env', 'scene_', 'step_', 'blocks_tail', 'regularisers')):
            continue
        vals.setdefault(n, {}).setdefault(r['Counter_Name'], {}).setdefault(r['Dispatch_Id'], 0.0)
        vals[n][r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
med = lambda xs: sorted(xs)[len(xs) // 2]
# calibration of the byte counters on this access pattern (tools/ubench/plane_rw.hip under the same counters: calib/ next to the groups):
# known bytes / reported bytes for 4 B/lane plane reads and writes
calib = {'fetch': 2.0, 'write': 1.0, 'source': 'MI355X_MICROARCH.md (x2 for 16 B/lane streaming reads; 4 B/lane uncalibrated)'}
KNOWN = {'plane_read_kernel': ('FETCH_SIZE', 2 * 1024 ** 3), 'plane_write_kernel': ('WRITE_SIZE', 1024 ** 3)}
got = {}
for f in glob.glob(os.path.join(out, 'calib*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].strip()
        if n in KNOWN and r['Counter_Name'] == KNOWN[n][0]:
            got.setdefault(n, {}).setdefault(r['Dispatch_Id'], 0.0)
            got[n][r['Dispatch_Id']] += float(r['Counter_Value'])
if len(got) == 2:
    rep = {n: med(list(d.values())) * 1024 for n, d in got.items()}            # the counters are in KB
    calib = {'fetch': KNOWN['plane_read_kernel'][1] / rep['plane_read_kernel'], 'write': KNOWN['plane_write_kernel'][1] / rep['plane_write_kernel'],
             'source': 'tools/ubench/plane_rw.hip: 2 GiB read / 1 GiB written 4 B per lane under the same counters (known bytes / reported bytes)'}
print('byte-counter calibration:', calib)
res = {}
for n, cs in sorted(vals.items()):
    row = {}
    for c, d in sorted(cs.items()):
        v = med(list(d.values()))
        row[c] = round(v / 1024, 2) if c in ('FETCH_SIZE', 'WRITE_SIZE') else round(v / 1e6, 3)
    if 'FETCH_SIZE' in row and 'WRITE_SIZE' in row:
        row['hbm_mb'] = round(calib['fetch'] * row['FETCH_SIZE'] + calib['write'] * row['WRITE_SIZE'], 1)
        row['hbm_mb_guide_x2'] = round(2 * row['FETCH_SIZE'] + row['WRITE_SIZE'], 1)
    if 'SQ_THREAD_CYCLES_VALU' in row and 'SQ_ACTIVE_INST_VALU' in row and row['SQ_ACTIVE_INST_VALU']:
        row['valu_lane_utilisation'] = round(row['SQ_THREAD_CYCLES_VALU'] / (64 * row['SQ_ACTIVE_INST_VALU']), 3)
    res[n] = row
    print(n[:90]); print('   ', row)
# kernel durations of the same runs (kernel trace of pass 1) -> share of the SIMD time the VALU is busy
dur = {}
for f in glob.glob(os.path.join(out, 'g1', '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].strip()
        dur.setdefault(n, []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
CLK, SIMDS = 2.4e9, 1024
for n, row in res.items():
    if n in dur and 'SQ_ACTIVE_INST_VALU' in row:
        d = med(dur[n]) * 1e-9
        row['avg_us_under_counters'] = round(d * 1e6, 1)
        row['valu_busy_frac'] = round(row['SQ_ACTIVE_INST_VALU'] * 1e6 * 4 / (d * CLK * SIMDS), 3)
        if 'SQ_WAVE_CYCLES' in row:
            row['waves_per_simd'] = round(row['SQ_WAVE_CYCLES'] * 1e6 * 4 / (d * CLK * SIMDS), 2)
json.dump(res, open(os.path.join(out, 'summary.json'), 'w'), indent=1)
# the file bench.py reads: keyed by its kernel labels
LABEL = {'render_fwd_kernel<10, 8, 8, 2, true>': 'render_fwd_fused K=10 (fg pass)', 'render_fwd_kernel<10, 8, 8, 2, true, true>': 'render_fwd_fused K=10 (fg pass)',
         'render_fwd_kernel<10, 8, 8, 2, true, false>': 'render_fwd_fused K=10 (fg pass)', 'render_bwd_uv_kernel<false>': 'render_bwd_fused K=10 (fg pass)',
         'render_bwd_uv_kernel<true>': 'render_bwd_fused K=10 (fg pass)', 'render_bwd_uv_kernel': 'render_bwd_fused K=10 (fg pass)',
         'render_fwd_kernel<1, 16, 16, 2, false>': 'render_fwd_fused K=1 (env pass)', 'render_fwd_kernel<1, 16, 16, 1, false>': 'render_fwd_fused K=1 (env pass)',
         'render_bwd_hard_kernel': 'render_bwd_fused K=1 (env pass)',
         'shade_blend_bwd_kernel<true, false, true>': 'render_bwd_fused K=1 (env pass)', 'texbin_reduce_kernel': 'texbin_reduce_kernel (fg pass)'}
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench as _bench
bench = {'_csrc_sha16': _bench.csrc_sha16(),
         '_calibration': calib,
         '_how': 'rocprofv3 --kernel-trace --pmc <group> -- python tools/pmc_target.py, one run per counter group (tools/pmc_sq.sh); medians over the '
                 'dispatches; hbm_bytes = (fetch x FETCH_SIZE + write x WRITE_SIZE) KB with the factors of _calibration: the guide prescribes x2 for '
                 'FETCH_SIZE on 16 B/lane streaming reads and calls other widths uncalibrated -- these kernels read and write planes 4 B per lane, so '
                 'the factors are measured on exactly that pattern (tools/ubench/plane_rw.hip, known byte counts, same counters, same session); '
                 'valu_busy_frac = SQ_ACTIVE_INST_VALU * 4 / (kernel duration * 2.4 GHz * 1024 SIMDs); valu_lane_utilisation = '
                 'SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)'}
for n, lab in LABEL.items():
    if n in res:
        r = res[n]
        bench[lab] = {'hbm_bytes': int(r.get('hbm_mb', 0) * 1024 * 1024), 'valu_busy_frac': r.get('valu_busy_frac'), 'valu_lane_utilisation': r.get('valu_lane_utilisation'),
                      'waves_per_simd': r.get('waves_per_simd'), 'insts_valu_M': r.get('SQ_INSTS_VALU'), 'insts_lds_M': r.get('SQ_INSTS_LDS'),
                      'wait_any_share_of_wave_time': round(r['SQ_WAIT_ANY'] / r['SQ_WAVE_CYCLES'], 3) if 'SQ_WAIT_ANY' in r and r.get('SQ_WAVE_CYCLES') else None}
json.dump(bench, open(os.path.join(out, 'bench_counters.json'), 'w'), indent=1)
