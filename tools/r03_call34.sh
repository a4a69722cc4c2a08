#!/bin/bash
for v in product ht7 ht8 ht9; do
  if [ $v = product ]; then unset DBW_HIP_LIB; else export DBW_HIP_LIB=tools/variants/$v.so; fi
  timeout 300 python tools/diag/ab_kernels.py 800 0:0 2>/dev/null | tail -1 | sed "s/^/$v e800 /" | cut -c1-260
  timeout 300 python tools/diag/ab_kernels.py 0 0:0 2>/dev/null | tail -1 | sed "s/^/$v e0 /" | cut -c1-200
done
