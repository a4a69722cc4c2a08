#!/bin/bash
mkdir -p gpurun_out/c11
timeout 600 python tools/diag/work_orders.py 0 > gpurun_out/c11/orders.txt 2>&1; tail -26 gpurun_out/c11/orders.txt
