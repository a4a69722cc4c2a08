#!/bin/bash
mkdir -p gpurun_out/c16
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tile_face_lists or per_tile" > gpurun_out/c16/pytest.txt 2>&1
tail -3 gpurun_out/c16/pytest.txt
timeout 600 python tools/diag/sustained.py 0.0 100 > gpurun_out/c16/sustained0.txt 2>&1; cat gpurun_out/c16/sustained0.txt | cut -c1-200
timeout 600 python tools/diag/sustained.py 0.0 1 > gpurun_out/c16/sustained0_sync1.txt 2>&1; tail -8 gpurun_out/c16/sustained0_sync1.txt | cut -c1-200
python - <<'PY'
import sys, time, torch
sys.path.insert(0, '.'); import bench
dev = torch.device('cuda', 0)
for cfg in ((8, 576, 768, 20, 16, 256, 20, 3), (25, 1080, 1920, 50, 16, 512, 5, 2)):
    r = bench.measure_other(*cfg[:6], dev, steps=cfg[6], warmup=cfg[7])
    print(r)
PY
