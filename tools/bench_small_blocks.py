#!/usr/bin/env python3
"""Small column blocks (the reference driver uses blocks of ONE column, ecckd_rfmip_lw.F90:39): gas_optics +
rte_lw, direct calls against replaying a captured HIP graph, and the same with whole-tile solver waves only
(lw_tail_split = 0).  Finding: 68 us for 1-64 columns however it is launched (258 us without the tail split: the
serial depth of one wave of rte_lw, 16 g-point groups x 60 layers x two sweeps); blocks should hold >= 32k columns.
Usage: python tools/bench_small_blocks.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rte_ecckd_amd as pkg   # noqa: E402
from rte_ecckd_amd import synthetic   # noqa: E402

dev = torch.device("cuda:0")
k = pkg.GasOpticsEcckd()
assert k.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data",
                           "ecckd-1.2_lw_ckd-definition_climate_fsck-tol0.0161.nc"), device=0) == ""
nlay, ng = 60, k.get_ngpt()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for ncol in (1, 64, 512, 4096, 16384):
    cols = synthetic.columns(0, ncol, k.get_press_min())
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
    emis = t(cols["sfc_emis"][:, None])
    op = pkg.OpticalProps1scl(); op.alloc_1scl(ncol, nlay, k, like=plev)
    src = pkg.SourceFuncLW(); src.alloc(ncol, nlay, k, like=plev)
    fl = pkg.FluxesBroadband(torch.zeros((nlay + 1, ncol), dtype=torch.float64, device=dev),
                             torch.zeros((nlay + 1, ncol), dtype=torch.float64, device=dev))

    def step():
        k.gas_optics(None, plev, tlay, tsfc, gc, op, src, tlev=tlev)
        pkg.rte_lw(op, True, src, emis, fl)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):   # the stream of the warm-up call: its scratch block exists (tail split)
        step()
    reps = 200
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize(); direct = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        graph.replay()
    torch.cuda.synchronize(); replay = (time.perf_counter() - t0) / reps
    pkg.set_solver_option("lw_tail_split", 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize(); whole = (time.perf_counter() - t0) / reps
    pkg.set_solver_option("lw_tail_split", 1)
    print("ncol %5d: direct calls (Python mirror) %7.1f us/step, graph replay %7.1f us/step  (%.1f Mcell/s replayed); direct calls with lw_tail_split = 0: %7.1f us/step"
          % (ncol, direct * 1e6, replay * 1e6, ncol * nlay * ng / replay / 1e6, whole * 1e6), flush=True)
