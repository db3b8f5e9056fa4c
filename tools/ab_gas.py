#!/usr/bin/env python3
"""A/B of the gas-optics kernel alone over the library variants of tools/build_variants.sh: per variant (fresh
process, same box) the HIP-event time of the fused longwave gas optics and of the tau-only mode, with the
well-mixed gases given per column (BASELINE's inputs) and as scalars (the merged slot).
    python tools/ab_gas.py [ncol] [dtype]      (parent)"""
import glob
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(ncol, dtype):
    sys.path.insert(0, root)
    import torch
    import bench
    import rte_ecckd_amd as pkg
    L = pkg.lib()
    for item in filter(None, os.environ.get("ECCKD_AB_OPTS", "").split(",")):     # e.g. gas_slab_f32=1
        name, _, value = item.partition("=")
        pkg.set_solver_option(name, float(value))
    dev = torch.device("cuda:0")
    k = pkg.GasOpticsEcckd()
    assert k.load(os.path.join(root, "data", "ecckd-1.2_lw_ckd-definition_climate_fsck-tol0.0161.nc"), device=0) == ""
    tdt = torch.float64 if dtype == "f64" else torch.float32
    case = bench.LwCase(pkg, k, ncol, 0, dev, tdt, k.get_press_min())
    res = {}
    for label, gc in (("percol", case.gc), ("scalars", case.gc_scalars)):
        def full():
            e = k.gas_optics(None, case.plev, case.tlay, case.percol["tsfc"], gc, case.op, case.src, tlev=case.tlev)
            assert e == "", e
        def tau():
            e = k.gas_optics_tau(case.plev, case.tlay, gc, case.op)
            assert e == "", e
        for name, fn in (("lw", full),) + ((("tau", tau),) if dtype == "f64" else ()):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            L.ecckd_prof_enable(1)
            for _ in range(6):
                fn()
            torch.cuda.synchronize()
            L.ecckd_prof_enable(0)
            r = bench.prof_report(L)
            res[label + "_" + name] = round(sum(v[0] for v in r.values()), 3)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), sys.argv[3])
        sys.exit(0)
    ncol = sys.argv[1] if len(sys.argv) > 1 else "1000000"
    dtype = sys.argv[2] if len(sys.argv) > 2 else "f64"
    libs = sorted(glob.glob(os.path.join(root, "variants_tmp", "lib_v*.so")))
    optsets = os.environ.get("ECCKD_AB_OPTSETS", "").split(";")      # e.g. "gas_slab_f32=0;gas_slab_f32=1": every variant with each
    for rep in range(2):
        for lib in libs:
            for opts in optsets:
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", ncol, dtype],
                                     env=dict(os.environ, ECCKD_LIB=lib, ECCKD_AB_OPTS=opts), capture_output=True, text=True)
                print(os.path.basename(lib), opts, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "FAILED " + out.stderr[-600:], flush=True)
