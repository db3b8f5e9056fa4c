# same-box A/B of gas_fused_kernel variants built by tools/build_variants.sh (negative results of round 3)
mkdir -p gpurun_out
ECCKD_AB_OPTSETS="gas_slab_f32=0" timeout -k 10 500 python tools/ab_gas.py 1000000 f64 2>&1 | tee gpurun_out/r03_ab_gas_stage.txt
ECCKD_AB_OPTSETS="gas_slab_f32=0" timeout -k 10 300 python tools/ab_gas.py 100000 f64 2>&1 | tee -a gpurun_out/r03_ab_gas_stage.txt
