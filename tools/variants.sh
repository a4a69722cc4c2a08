#!/bin/bash
# Tuning helper: build libdbw_hip.so variants with extra -D flags into tools/variants/ (shipped to the GPU box, git-ignored).
#   tools/variants.sh name1 "-DA=1 -DB=2" name2 "-DC=3" ...
# then on the GPU box:  DBW_HIP_LIB=tools/variants/name1.so python tools/ablate.py 0 0
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants
CS=differentiable-blocksworld_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-gpu-flush-denormals-to-zero -Wno-unused-function -DDBW_DIAG"   # (DBW_DIAG: the tools-only exports -- hard-pass tile shapes, the cell-table layout)
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  ( objs=""; for f in util raster project_clip shade_blend render_fused texture model_ops train_step lpips_head; do
      rm -f /tmp/var_${name}_$f.o; /opt/rocm/bin/hipcc $FLAGS $defs -c $CS/$f.hip -o /tmp/var_${name}_$f.o & objs="$objs /tmp/var_${name}_$f.o"; done; wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o tools/variants/$name.so; echo built $name ) &
done
wait
