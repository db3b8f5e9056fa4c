#!/usr/bin/env python3
"""LW gas_optics alone against the spread of surface pressure inside a block of columns: the fused kernel
stages a slab of R pressure rows per 4096-column segment; lanes outside it take the slow (table-from-L2) path.
Usage: python tools/bench_gas_optics_spread.py [ncol]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rte_ecckd_amd as pkg   # noqa: E402
from rte_ecckd_amd import synthetic   # noqa: E402

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
for item in filter(None, os.environ.get("ECCKD_AB_OPTS", "").split(",")):     # e.g. gas_slab_f32=1
    name, _, value = item.partition("=")
    pkg.set_solver_option(name, float(value))
    print("option", name, value)
nlay = 60
dev = torch.device("cuda:0")
k = pkg.GasOpticsEcckd()
assert k.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data",
                           "ecckd-1.2_lw_ckd-definition_climate_fsck-tol0.0161.nc"), device=0) == ""
ng = k.get_ngpt()
base = synthetic.columns(0, ncol, k.get_press_min())
eta = ((np.arange(nlay + 1, dtype=np.float64)) / nlay) ** 2
ptop = base["plev"][0, 0]
rng = np.random.default_rng(1)
cases = {
    "synthetic default: ps = 95-103 kPa, random per column": None,
    "random ps = 85-103 kPa": 85000 + 18000 * rng.random(ncol),
    "random ps = 50-103 kPa (mountains, shuffled columns)": 50000 + 53000 * rng.random(ncol),
    "smooth ps = 50-103 kPa (period 20000 columns)": 76500 + 26500 * np.sin(2 * np.pi * np.arange(ncol) / 20000.0),
}
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for name, ps in cases.items():
    cols = dict(base)
    if ps is not None:
        cols["plev"] = ptop + (ps[None, :] - ptop) * eta[:, None]
    gc = pkg.GasConcs(synthetic.GAS_ORDER)
    for n in synthetic.GAS_ORDER:
        v = cols[n]
        if np.isscalar(v):
            gc.set_vmr(n, float(v))
        elif v.ndim == 1:
            gc.set_vmr_column(n, t(v))
        else:
            gc.set_vmr(n, t(v))
    plev, tlay, tlev, tsfc = t(cols["plev"]), t(cols["tlay"]), t(cols["tlev"]), t(cols["tsfc"])
    op = pkg.OpticalProps1scl(); op.alloc_1scl(ncol, nlay, k, like=plev)
    src = pkg.SourceFuncLW(); src.alloc(ncol, nlay, k, like=plev)
    for _ in range(2):
        assert k.gas_optics(None, plev, tlay, tsfc, gc, op, src, tlev=tlev) == ""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        k.gas_optics(None, plev, tlay, tsfc, gc, op, src, tlev=tlev)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("%-58s %.3f ms  %6.0f Mcell/s" % (name, dt * 1e3, ncol * nlay * ng / dt / 1e6), flush=True)
