#!/usr/bin/env python3
"""Soak of the layer-systolic shortwave solver's hand-off: the same call repeated many times at several sizes (tail-split
units, partial last tile, more tiles than CUs), every result compared bit for bit with the first one.  A lost hand-off
would show as NaN fluxes (the waits are bounded) or as different bits.
    python tools/soak_rte_sw.py [repeats]"""
import os
import sys

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import torch  # noqa: E402
import rte_ecckd_amd as pkg  # noqa: E402
from test_gpu_round3 import sw_inputs, run_sw  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
gpu = torch.device("cuda:0")
for arith in (pkg.FAST, pkg.REFERENCE_ORDER):
    pkg.set_arithmetic(arith)
    for ncol, nlay, ng in ((37, 60, 27), (5000, 60, 27), (40000, 60, 27), (3000, 47, 9), (70000, 60, 5)):
        inp = sw_inputs(np.random.default_rng(ncol), ncol, nlay, ng, g_zero=(ncol % 2 == 0))
        first = run_sw(pkg, gpu, inp, True)
        assert all(np.all(np.isfinite(x)) for x in first)
        bad = 0
        for r in range(reps):
            out = run_sw(pkg, gpu, inp, True)
            bad += not all(np.array_equal(a, b) for a, b in zip(out, first))
        print("arithmetic %d, %6d columns x %d layers x %2d g-points: %d repeats, %d differ" % (arith, ncol, nlay, ng, reps, bad), flush=True)
        assert bad == 0
print("soak ok")
