#!/bin/bash
# round 3: fp64 gas optics over the float32 image of the tables in LDS (gas_slab_f32): v1 default rows (R = 8 with the float32
# image), v2 capped at R = 3 (cost of the widening alone), v3 two 256-thread blocks per CU (80 KB each), v4 R = 5
mkdir -p gpurun_out
ECCKD_AB_OPTSETS="gas_slab_f32=0;gas_slab_f32=1" timeout -k 10 900 python tools/ab_gas.py 1000000 f64 2>&1 | tee gpurun_out/r03_ab_gas_slab32.txt
for o in 0 1; do ECCKD_AB_OPTS=gas_slab_f32=$o timeout -k 10 200 python tools/bench_gas_optics_spread.py 2>&1 | tee -a gpurun_out/r03_spread_slab32.txt; done
