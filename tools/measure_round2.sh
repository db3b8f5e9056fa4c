#!/bin/bash
# Round-2 measurement set on the MI355X box (run through gpurun):   bash tools/measure_round2.sh <tag> [quick]
# Writes gpurun_out/<tag>_*; copy what is to be judged to profiles/ (tools/README.md).
#   headline bench line (+ side measurements), rocprofv3 kernel stats of the same command, HBM traffic passes
#   (FETCH_SIZE / WRITE_SIZE, one counter per pass), SQ passes, the same for the SW pair, memory-side stall counters
#   (one per pass, 1e5 columns, bounded by timeout), secondary bench lines.
set -e
tag=$1
export TMPDIR=/tmp
o=gpurun_out
P="--cpu-seconds 0 --no-side"
if [ "$2" != "extra" ]; then
python bench.py --steps 10 --warmup 2 > $o/${tag}_bench.json 2> $o/${tag}_bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof -- python3 bench.py --steps 10 --warmup 2 $P > $o/${tag}_bench_prof.json 2> $o/${tag}_bench_prof.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $o/${tag}_pmc_$c -- python3 bench.py --steps 2 --warmup 1 $P > /dev/null 2> $o/${tag}_pmc_$c.err
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $o/${tag}_pmc_sq1 -- python3 bench.py --ncol 300000 --steps 2 --warmup 1 $P > /dev/null 2> $o/${tag}_pmc_sq1.err
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $o/${tag}_pmc_sq2 -- python3 bench.py --ncol 300000 --steps 2 --warmup 1 $P > /dev/null 2> $o/${tag}_pmc_sq2.err
python tools/pmc_summary.py "synthetic 1000000 columns x 60 layers x 32 g-points, LW fsck-tol0.0161, fp64" 1000000 $o/${tag}_pmc_FETCH_SIZE $o/${tag}_pmc_WRITE_SIZE > $o/${tag}_hbm_traffic.json
python tools/pmc_summary.py "synthetic 300000 columns x 60 layers x 32 g-points, LW fsck-tol0.0161, fp64" 300000 $o/${tag}_pmc_sq1 $o/${tag}_pmc_sq2 > $o/${tag}_pmc_sq.json
echo "lw passes done"
# ---- SW pair (BASELINE configs[2], 1e5 columns) ----
S="--mode sw --ncol 100000"
python bench.py $S --steps 10 --warmup 2 --cpu-seconds 0 > $o/${tag}_bench_sw.json 2> $o/${tag}_bench_sw.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof_sw -- python3 bench.py $S --steps 10 --warmup 2 --cpu-seconds 0 > /dev/null 2> $o/${tag}_prof_sw.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $o/${tag}_pmcsw_$c -- python3 bench.py $S --steps 2 --warmup 1 --cpu-seconds 0 > /dev/null 2> $o/${tag}_pmcsw_$c.err
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $o/${tag}_pmcsw_sq1 -- python3 bench.py $S --steps 2 --warmup 1 --cpu-seconds 0 > /dev/null 2> $o/${tag}_pmcsw_sq1.err
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $o/${tag}_pmcsw_sq2 -- python3 bench.py $S --steps 2 --warmup 1 --cpu-seconds 0 > /dev/null 2> $o/${tag}_pmcsw_sq2.err
python tools/pmc_summary.py "synthetic 100000 columns x 60 layers x 27 g-points, SW wide-tol0.05, fp64" 100000 $o/${tag}_pmcsw_FETCH_SIZE $o/${tag}_pmcsw_WRITE_SIZE $o/${tag}_pmcsw_sq1 $o/${tag}_pmcsw_sq2 > $o/${tag}_pmc_sw.json
echo "sw passes done"
fi
if [ "$2" != "quick" ]; then
  # ---- memory-side stall counters: ONE counter per pass (the *_sum metrics expand to one hardware counter per TCC
  # channel / TCP instance: several of them in a pass is what rocprofv3 refused in round 1), bounded by timeout ----
  for c in TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr TCC_BUSY_avr TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE; do
    timeout -k 10 150 rocprofv3 --pmc $c --output-format csv -d $o/${tag}_stall_$c -- python3 bench.py --ncol 100000 --steps 1 --warmup 1 $P > /dev/null 2> $o/${tag}_stall_$c.err || echo "stall pass $c failed or timed out" >> $o/${tag}_stall_failed.txt
    echo "stall pass $c done"
  done
  python tools/pmc_summary.py "synthetic 100000 columns x 60 layers x 32 g-points, LW fsck-tol0.0161, fp64, one counter per pass" 100000 $o/${tag}_stall_* > $o/${tag}_pmc_stall.json
  python bench.py --dtype f32 --steps 10 --warmup 2 $P > $o/${tag}_bench_f32.json 2> $o/${tag}_bench_f32.err
  python bench.py --lut rrtmgp --steps 10 --warmup 2 $P > $o/${tag}_bench_36g.json 2> $o/${tag}_bench_36g.err
  python bench.py --lut rrtmgp --dtype f32 --steps 10 --warmup 2 $P > $o/${tag}_bench_36g_f32.json 2> $o/${tag}_bench_36g_f32.err
  python bench.py --arithmetic reference --steps 5 --warmup 2 $P > $o/${tag}_bench_refmode.json 2> $o/${tag}_bench_refmode.err
fi
find $o -name "*agent_info.csv" -path "*${tag}_*" -delete 2>/dev/null || true
python - <<PY
import json, glob
for f in sorted(glob.glob("$o/${tag}_bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["dtype"], round(d["value"]), "ms %.2f" % d["ms_per_step"], {k: round(v["avg_ms"], 3) for k, v in d["kernels"].items()}, "frac %.3f" % d["roofline_pipeline"]["frac"])
    except Exception as e:
        print(f, "unreadable", e)
PY
