#!/bin/bash
# Round-3 measurement set on the MI355X box (run through gpurun):   bash tools/measure_round3.sh <tag>
# Writes gpurun_out/<tag>_*; copy what is to be judged to profiles/ (tools/README.md).
#   default bench line; rocprofv3 kernel stats of the headline command; HBM traffic passes (FETCH_SIZE / WRITE_SIZE, one
#   counter per pass) of the LW pair at 1e6 columns and of the SW pair at 1e5; SQ / LDS passes of the SW pair.
set -e
tag=$1
export TMPDIR=/tmp
o=gpurun_out
P="--cpu-seconds 0 --no-side"
python bench.py --steps 10 --warmup 2 > $o/${tag}_bench.json 2> $o/${tag}_bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof -- python3 bench.py --steps 10 --warmup 2 $P > $o/${tag}_bench_prof.json 2> $o/${tag}_bench_prof.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $o/${tag}_pmc_$c -- python3 bench.py --steps 2 --warmup 1 $P > /dev/null 2> $o/${tag}_pmc_$c.err
done
python tools/pmc_summary.py "synthetic 1000000 columns x 60 layers x 32 g-points, LW fsck-tol0.0161, fp64" 1000000 $o/${tag}_pmc_FETCH_SIZE $o/${tag}_pmc_WRITE_SIZE > $o/${tag}_hbm_traffic.json
echo "lw passes done"
S="--mode sw --ncol 100000"
python bench.py $S --steps 10 --warmup 2 > $o/${tag}_bench_sw.json 2> $o/${tag}_bench_sw.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof_sw -- python3 bench.py $S --steps 10 --warmup 2 > /dev/null 2> $o/${tag}_prof_sw.err
# (counter passes with --no-side: the API pair only -- the fused path's gas-optics launches carry the same kernel name)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $o/${tag}_pmcsw_$c -- python3 bench.py $S --no-side --steps 2 --warmup 1 > /dev/null 2> $o/${tag}_pmcsw_$c.err
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $o/${tag}_pmcsw_sq1 -- python3 bench.py $S --no-side --steps 2 --warmup 1 > /dev/null 2> $o/${tag}_pmcsw_sq1.err
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $o/${tag}_pmcsw_sq2 -- python3 bench.py $S --no-side --steps 2 --warmup 1 > /dev/null 2> $o/${tag}_pmcsw_sq2.err
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $o/${tag}_pmcsw_lds -- python3 bench.py $S --no-side --steps 2 --warmup 1 > /dev/null 2> $o/${tag}_pmcsw_lds.err || echo "lds pass failed"
python tools/pmc_summary.py "synthetic 100000 columns x 60 layers x 27 g-points, SW wide-tol0.05, fp64" 100000 $o/${tag}_pmcsw_FETCH_SIZE $o/${tag}_pmcsw_WRITE_SIZE $o/${tag}_pmcsw_sq1 $o/${tag}_pmcsw_sq2 $o/${tag}_pmcsw_lds > $o/${tag}_pmc_sw.json
echo "sw passes done"
find $o -name "*agent_info.csv" -path "*${tag}_*" -delete 2>/dev/null || true
ls $o/${tag}_prof/*/ 2>/dev/null | head
