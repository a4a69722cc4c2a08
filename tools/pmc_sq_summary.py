"""Per-kernel medians of the counters collected by tools/pmc_sq.sh (values in millions per launch, summed over the instances rocprofv3
reports; FETCH_SIZE / WRITE_SIZE in MB with the gfx950 FETCH x2 correction of MI355X_MICROARCH.md applied to hbm_mb)."""
import csv, glob, os, sys, json
out = sys.argv[1]
vals = {}
for f in glob.glob(os.path.join(out, 'g*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        n = n.split('(')[0].strip()
        if not any(k in n for k in ('render_fwd', 'shade_blend_bwd', 'composite', 'texbin', 'coarse_bin', 'face_setup', 'project_clip', 'env')):
            continue
        vals.setdefault(n, {}).setdefault(r['Counter_Name'], {}).setdefault(r['Dispatch_Id'], 0.0)
        vals[n][r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
med = lambda xs: sorted(xs)[len(xs) // 2]
res = {}
for n, cs in sorted(vals.items()):
    row = {}
    for c, d in sorted(cs.items()):
        v = med(list(d.values()))
        row[c] = round(v / 1024, 2) if c in ('FETCH_SIZE', 'WRITE_SIZE') else round(v / 1e6, 3)
    if 'FETCH_SIZE' in row and 'WRITE_SIZE' in row:
        row['hbm_mb'] = round(2 * row['FETCH_SIZE'] + row['WRITE_SIZE'], 1)
    if 'SQ_THREAD_CYCLES_VALU' in row and 'SQ_ACTIVE_INST_VALU' in row and row['SQ_ACTIVE_INST_VALU']:
        row['valu_lane_utilisation'] = round(row['SQ_THREAD_CYCLES_VALU'] / (64 * row['SQ_ACTIVE_INST_VALU']), 3)
    res[n] = row
    print(n[:90]); print('   ', row)
json.dump(res, open(os.path.join(out, 'summary.json'), 'w'), indent=1)
