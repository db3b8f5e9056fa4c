#!/bin/bash
mkdir -p gpurun_out
mv variants_tmp/lib_v2.so variants_tmp/timing.so
timeout -k 10 400 python tools/ab.py 100000 sw 2>&1 | tee gpurun_out/r03_ab_sw4.txt
ECCKD_LIB=$PWD/variants_tmp/timing.so timeout -k 10 200 python tools/sys_timing.py 2>&1 | tee gpurun_out/r03_sys_timing3.txt
