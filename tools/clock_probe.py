#!/usr/bin/env python3
"""Samples rocm-smi (clocks, power) while a kernel loop runs, to see whether a kernel mix is power-limited.
    python tools/clock_probe.py [seconds]"""
import os
import subprocess
import sys
import threading
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch  # noqa: E402
import bench  # noqa: E402
import rte_ecckd_amd as pkg  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
dev = torch.device("cuda:0")
k = pkg.GasOpticsEcckd()
assert k.load(os.path.join(root, "data", "ecckd-1.2_lw_ckd-definition_climate_fsck-tol0.0161.nc"), device=0) == ""
case = bench.LwCase(pkg, k, 1000000, 0, dev, torch.float64, k.get_press_min())
pkg.set_solver_option("gas_split_streams", 0)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        import json
        d = json.loads(out)
        c = d[sorted(d)[0]]
        keep = {kk: v for kk, v in c.items() if any(s in kk.lower() for s in ("sclk", "mclk", "fclk", "power"))}
        return keep
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def loop(name, fn):
    stop = []
    samples = []

    def sampler():
        while not stop:
            samples.append(smi())
            time.sleep(0.3)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        n += 10
    dt = (time.perf_counter() - t0) / n * 1e3
    stop.append(1)
    th.join()
    print(name, "%.3f ms/call" % dt, flush=True)
    for smp in samples[1:-1][:6]:
        print("   ", smp, flush=True)


print("idle", smi(), flush=True)
loop("gas_optics (fused)", lambda: k.gas_optics(None, case.plev, case.tlay, case.percol["tsfc"], case.gc, case.op, case.src, tlev=case.tlev))
loop("tau only", lambda: k.gas_optics_tau(case.plev, case.tlay, case.gc, case.op))
loop("planck only", lambda: k.planck_sources(case.tlay, case.percol["tsfc"], case.src, tlev=case.tlev))
loop("rte_lw", lambda: pkg.rte_lw(case.op, True, case.src, case.emis, case.fl, n_gauss_angles=1))
