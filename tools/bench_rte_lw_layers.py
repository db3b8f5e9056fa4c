#!/usr/bin/env python3
"""rte_lw alone at several layer counts (random optical properties, 32 g-points, fp64): which solver variant a
layer count takes and what it costs.  Usage: python tools/bench_rte_lw_layers.py [ncol]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rte_ecckd_amd as pkg   # noqa: E402

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
ng = 32
dev = torch.device("cuda:0")
for nlay in (32, 47, 60, 64, 72, 91, 96, 100, 137):
    g = torch.Generator(device=dev); g.manual_seed(nlay)
    r = lambda *shape: torch.rand(*shape, dtype=torch.float64, device=dev, generator=g)
    op = pkg.OpticalProps1scl(); op.tau = r(ng, nlay, ncol); op.band2gpt = np.array([[1, ng]], dtype=np.int32)
    src = pkg.SourceFuncLW()
    src.lay_source, src.lev_source_inc, src.lev_source_dec, src.sfc_source = (r(ng, nlay, ncol) + 1, r(ng, nlay, ncol) + 1,
                                                                              r(ng, nlay, ncol) + 1, r(ng, ncol) + 1)
    emis = torch.full((ncol, 1), 0.98, dtype=torch.float64, device=dev)
    fl = pkg.FluxesBroadband(torch.empty((nlay + 1, ncol), dtype=torch.float64, device=dev),
                             torch.empty((nlay + 1, ncol), dtype=torch.float64, device=dev))
    for _ in range(2):
        assert pkg.rte_lw(op, True, src, emis, fl) == ""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        pkg.rte_lw(op, True, src, emis, fl)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    cells = ncol * nlay * ng
    print("nlay %3d  %.3f ms  %6.0f Mcell/s  %.2f TB/s (32 B/cell)" % (nlay, dt * 1e3, cells / dt / 1e6, cells * 32 / dt / 1e12), flush=True)
    del op, src, fl
