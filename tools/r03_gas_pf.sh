ECCKD_AB_OPTSETS="gas_slab_f32=0" timeout -k 10 500 python tools/ab_gas.py 1000000 f64 2>&1 | tee gpurun_out/r03_ab_gas_pfdepth.txt
timeout -k 10 300 python tools/ab_gas.py 1000000 f32 2>&1 | tee -a gpurun_out/r03_ab_gas_pfdepth.txt
