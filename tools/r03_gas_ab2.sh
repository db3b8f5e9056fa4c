#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r03_gputests6.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r03_gputests6.log | cut -c1-200
for o in 0 2; do ECCKD_AB_OPTS=gas_slab_f32=$o timeout -k 10 200 python tools/bench_gas_optics_spread.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_spread_auto.txt; done
rm -f variants_tmp/lib_v[234].so
ECCKD_AB_OPTSETS="gas_slab_f32=0;gas_slab_f32=2" timeout -k 10 600 python tools/ab_gas.py 1000000 f64 2>&1 | tee gpurun_out/r03_ab_gas_auto.txt
