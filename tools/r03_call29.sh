#!/bin/bash
for i in 1 2; do
for v in product nt; do
  if [ $v = product ]; then unset DBW_HIP_LIB; else export DBW_HIP_LIB=tools/variants/$v.so; fi
  timeout 300 python tools/diag/ab_kernels.py 0 0:0 2>/dev/null | tail -1 | sed "s/^/$v /" | cut -c1-200
done; done
