#!/bin/bash
mkdir -p gpurun_out/c23
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "graph_replay" > gpurun_out/c23/pytest.txt 2>&1
tail -15 gpurun_out/c23/pytest.txt
python - <<'PY' 2>&1 | grep -v amdgpu
import sys, torch
sys.path.insert(0, '.'); import bench
dev = torch.device('cuda', 0)
for g in (False, True):
    for reads in (False, True):
        r = bench.measure_other(4, 300, 400, 10, 10, 256, dev, steps=100, warmup=10, read_losses=reads, use_graph=g)
        print('batch 4, graph', g, 'reads', reads, round(r['ms_per_step'], 4), 'ms/step', round(r['views_per_s']), 'views/s', flush=True)
for g in (False, True):
    r = bench.measure_other(49, 300, 400, 10, 10, 256, dev, steps=50, warmup=10, use_graph=g)
    print('batch 49, graph', g, round(r['ms_per_step'], 4), flush=True)
PY
