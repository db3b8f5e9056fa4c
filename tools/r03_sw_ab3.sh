#!/bin/bash
# round 3, late: the shortwave solver's sweeps -- tests, same-box A/B, time line of a step
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -x -q -m gpu -k "sw or systolic or shortwave" 2>&1 | tail -5 | tee gpurun_out/r03_sw_tests.txt &&
timeout -k 10 500 python tools/ab.py 100000 sw 2>&1 | tee gpurun_out/r03_ab_sw16.txt &&
ECCKD_LIB=$PWD/variants_tmp/timing.so timeout -k 10 200 python tools/sys_timing.py 2>&1 | tee gpurun_out/r03_sys_timing7.txt
