#!/usr/bin/env python3
"""Shortwave gas optics alone at 27 g-points (the wide-tol0.05 model: the g-point count is not a multiple of the chunk
of four, so the kernel instantiation carries per-g-point bounds checks) against the same tables padded to 28 g-points
(full chunks): what the bounds checks cost.   python tools/sw_gas_ng_probe.py [ncol]"""
import os
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "oracle"))
import bench  # noqa: E402
import oracle  # noqa: E402
import rte_ecckd_amd as pkg  # noqa: E402
from rte_ecckd_amd import synthetic  # noqa: E402

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
path = os.path.join(root, "data", "ecckd-1.2_sw_ckd-definition_climate_wide-tol0.05.nc")
m = oracle.CkdModel(path)
L = pkg.lib()
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cols = synthetic.columns(0, ncol, float(np.exp(m.log_pressure[0])) * (1 + 2.3e-16), shortwave=True)
names = ["co2", "ch4", "n2o", "o2", "h2o", "o3"]
for pad in (0, 1, 5):
    ng = m.ng + pad
    ext = lambda a: np.concatenate([a] + [a[..., -1:]] * pad, axis=-1) if pad else a
    tabs = [dict(name=n, code=tb["code"], composite_only=tb["composite_only"], mole_fraction=tb.get("mole_fraction"),
                 reference_mole_fraction=tb["reference_mole_fraction"], coefficient=ext(tb["coefficient"])) for n, tb in zip(m.gas, m.tables)]
    k = pkg.GasOpticsEcckd()
    err = k.init_from_tables(m.log_pressure, m.temperature, tabs, solar=(ext(m.solar_irradiance), ext(m.rayleigh)), device=0)
    assert err == "", err
    gc = pkg.GasConcs(names)
    for n in names:
        v = cols[n] if n in cols else 0.209
        if np.isscalar(v):
            gc.set_vmr(n, float(v))
        elif v.ndim == 1:
            gc.set_vmr_column(n, t(v))
        else:
            gc.set_vmr(n, t(v))
    plev, tlay = t(cols["plev"]), t(cols["tlay"])
    op = pkg.OpticalProps2str(); op.alloc_2str(ncol, 60, k, like=plev)
    toa = torch.empty((ng, ncol), dtype=torch.float64, device=dev)
    for _ in range(2):
        assert k.gas_optics(None, plev, tlay, gc, op, toa) == ""
    torch.cuda.synchronize()
    L.ecckd_prof_enable(1)
    for _ in range(8):
        k.gas_optics(None, plev, tlay, gc, op, toa)
    torch.cuda.synchronize()
    L.ecckd_prof_enable(0)
    r = bench.prof_report(L)
    ms = r["tau"][0]
    print("ng = %d: gas optics %.3f ms, %.2f ns per 1000 cells, plan %s" % (ng, ms, ms * 1e6 / (ncol * 60 * ng) * 1e3,
                                                                            k.plan(ncol, 60, names)), flush=True)
