#!/bin/bash
# One measurement set on the MI355X box (run through gpurun):  tools/measure_round.sh <tag>
# Writes gpurun_out/<tag>_*: the headline bench line, rocprofv3 kernel stats of the same command, the two
# HBM-traffic PMC passes, and the secondary lines (SW, fp32, 36-g LW).  Copy what is to be judged to profiles/.
set -e
tag=$1
export TMPDIR=/tmp
o=gpurun_out
python bench.py --steps 10 --warmup 2 > $o/${tag}_bench.json 2> $o/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-side > $o/${tag}_bench_prof.json 2> $o/${tag}_bench_prof.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/${tag}_pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-side > /dev/null 2> $o/${tag}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/${tag}_pmc_write -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-side > /dev/null 2> $o/${tag}_pmc_write.err
python tools/pmc_traffic.py $o/${tag}_pmc_fetch $o/${tag}_pmc_write "synthetic 1000000 columns x 60 layers x 32 g-points per GPU, LW fsck-tol0.0161" > $o/${tag}_hbm_traffic.json
python bench.py --mode sw --ncol 100000 --steps 10 --warmup 2 --cpu-seconds 0 > $o/${tag}_bench_sw.json 2> $o/${tag}_bench_sw.err
python bench.py --dtype f32 --steps 10 --warmup 2 --cpu-seconds 0 > $o/${tag}_bench_f32.json 2> $o/${tag}_bench_f32.err
python bench.py --lut rrtmgp --steps 10 --warmup 2 --cpu-seconds 0 > $o/${tag}_bench_36g.json 2> $o/${tag}_bench_36g.err
python bench.py --ncol 100000 --steps 10 --warmup 2 --cpu-seconds 0 > $o/${tag}_bench_1e5.json 2> $o/${tag}_bench_1e5.err
rm -f $o/${tag}_pmc_fetch/*/*agent_info.csv $o/${tag}_pmc_write/*/*agent_info.csv
python - <<PY
import json, glob
for f in sorted(glob.glob("$o/${tag}_bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["dtype"], round(d["value"]), "ms %.2f" % d["ms_per_step"], {k: round(v["avg_ms"], 3) for k, v in d["kernels"].items()}, "frac %.3f" % d["roofline_pipeline"]["frac"], d.get("cpu_baseline") and round(d["cpu_baseline"]["value"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
