#!/usr/bin/env python3
"""Per-wave time line of one g-point step of the layer-systolic shortwave solver (variant build with
-DECCKD_SYS_TIMING; ECCKD_LIB points at it): s_memtime stamps of block 0, ticks relative to the earliest stamp.
    ECCKD_LIB=variants_tmp/lib_v1.so python tools/sys_timing.py [ncol]"""
import ctypes as C
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
bench.sw_measure(ncol, 3, 1)
import rte_ecckd_amd as pkg  # noqa: E402
L = pkg.lib()
buf = (C.c_longlong * (12 * 8))()
rc = L.ecckd_debug_sys_times(buf)
t = [[buf[w * 8 + i] for i in range(8)] for w in range(12)]
t0 = min(x for r in t for x in r[:7] if x)
print("rc", rc, "ticks (s_memtime, 100 MHz = 10 ns per tick)" )
print("wave   P_start    P_end  loads_issued  U_token   U_done->D_wait  D_token   D_done  U_published")
for w, r in enumerate(t):
    print("%4d " % w + " ".join("%9d" % (x - t0) for x in r[:8]))
