#!/usr/bin/env python3
"""A/B of library variants built by tools/build_variants.sh: runs bench.py once per variant in a
fresh process (one GPU, same box) and prints the per-kernel HIP-event times."""
import glob
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ncol = sys.argv[1] if len(sys.argv) > 1 else "300000"
mode = sys.argv[2] if len(sys.argv) > 2 else "lw"
dtype = sys.argv[3] if len(sys.argv) > 3 else "f64"
libs = sorted(glob.glob(os.path.join(root, "variants_tmp", "lib_v*.so")))
for rep in range(2):
    for lib in libs:
        env = dict(os.environ, ECCKD_LIB=lib)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--ncol", ncol, "--steps", "5",
                              "--warmup", "2", "--cpu-seconds", "0", "--mode", mode, "--dtype", dtype] + (["--no-side"] if mode == "lw" else []), env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            kk = d["kernels"] if "kernels" in d else {k: {"avg_ms": v} for k, v in d["kernels_avg_ms"].items()}
            print(os.path.basename(lib), "ms/step %.3f" % d["ms_per_step"],
                  {k: round(v["avg_ms"], 3) for k, v in kk.items()},
                  "dflux %.1e" % d["check_max_abs_flux_diff_vs_oracle_Wm2"], flush=True)
        except Exception as e:
            print(os.path.basename(lib), "FAILED", e, out.stderr[-500:], flush=True)
