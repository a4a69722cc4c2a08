#!/bin/bash
# Register / scratch / LDS usage of every kernel, compiled with the flags of build.py (device code only).
# Note: `-save-temps` changes the device pipeline and reports different (higher) VGPR counts -- do not use it for this.
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-gpu-flush-denormals-to-zero"
for f in differentiable-blocksworld_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -S --cuda-device-only -o /tmp/_regs.s "$f" 2>/dev/null || { echo "compile failed: $f"; continue; }
  grep -E "^\s+\.(name|vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size):" /tmp/_regs.s | paste - - - - - |
    awk -v F="$(basename $f)" '{printf "%-18s lds %6s  scratch %4s  sgpr %3s  vgpr %3s  %s\n", F, $2, $6, $8, $10, $4}' | sed 's/_ZN12_GLOBAL__N_1[0-9]*//' | cut -c1-150
done
