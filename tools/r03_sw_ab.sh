#!/bin/bash
# round 3: layer-systolic shortwave solver against the two-pass kernel (same box), then the shortwave GPU tests
mkdir -p gpurun_out
for opt in 0 1; do
  timeout -k 10 300 python bench.py --mode sw --ncol 100000 --steps 10 --warmup 2 --solver-option sw_solver=$opt > gpurun_out/r03_sw_solver$opt.json 2> gpurun_out/r03_sw_solver$opt.err
  rc=$?; echo "sw_solver=$opt rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03_sw_solver$opt.json").read().strip().splitlines()[-1])
    print("  ms/step %.3f value %.0f" % (d["ms_per_step"], d["value"]), {k: round(v["avg_ms"],3) for k,v in d["kernels"].items()}, "dflux %.2e" % d["check_max_abs_flux_diff_vs_oracle_Wm2"])
except Exception as e:
    print("  no json:", e); print(open("gpurun_out/r03_sw_solver$opt.err").read()[-1500:])
PY
done
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "sw or SW or shortwave" > gpurun_out/r03_sw_tests.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r03_sw_tests.log
