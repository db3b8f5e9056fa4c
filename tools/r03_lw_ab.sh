#!/bin/bash
# round 3, late: rte_lw per-cell instruction diet (lean exp, frame-less division, 32-bit lane offsets) -- checks, tests, same-box A/B
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -w -o /tmp/check_lw_div tools/check_lw_div.hip && /tmp/check_lw_div 2>&1 | tee gpurun_out/r03_check_lw_div.txt &&
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -k "rte_lw or lw_solver or fused_lw or full_size or lw_flux or inc_flux or byband or single_precision or rfmip" 2>&1 | tail -5 | tee gpurun_out/r03_lw_tests.txt &&
timeout -k 10 600 python tools/ab.py 1000000 lw 2>&1 | tee gpurun_out/r03_ab_lw1.txt
