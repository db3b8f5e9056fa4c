#!/usr/bin/env python3
"""HBM bytes per launch of the product kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <workload string> > profiles/rNN_hbm_traffic.json

Counters are in KiB.  gfx950 correction (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts a
128-byte request as 64 bytes, so reads are doubled; this is calibrated on rte_lw_kernel, whose read set is known
exactly.  WRITE_SIZE is taken as is."""
import csv
import re
import glob
import json
import sys

def per_kernel(d, counter):
    acc = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            for key in ("gas_fused_kernel", "rte_lw_kernel", "rte_sw_kernel", "tau_kernel", "planck_kernel", "rte_lw_split_kernel"):   # noqa: E501
                if key in name:
                    if key == "gas_fused_kernel" and "<float" in name:
                        key = "gas_fused_kernel_f32"
                    if key == "rte_lw_kernel" and re.search(r"rte_lw_kernel<[^>]*, true>", name):
                        key = "rte_lw_kernel_shared"   # last template argument: SHARED level sources
                    acc.setdefault(key, []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}

fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
names = {"gas_fused_kernel": "gas_lw_fused", "rte_lw_kernel": "rte_lw", "rte_sw_kernel": "rte_sw",
         "gas_fused_kernel_f32": "gas_lw_fused_f32", "rte_lw_kernel_shared": "rte_lw_shared_levels", "tau_kernel": "tau", "planck_kernel": "planck"}
out = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing) -- python3 bench.py "
               "--steps 2 --warmup 1 --cpu-seconds 0; mean over the dispatches of each kernel; counters are in KiB; "
               "reads = 2 * FETCH_SIZE * 1024 (gfx950 counts 128-B requests as 64 B; calibrated on rte_lw_kernel, "
               "whose read set is 4 arrays of ncol*60*32 doubles + sfc_source + emis), writes = WRITE_SIZE * 1024 "
               "(tools/pmc_traffic.py)",
       "workload": sys.argv[3], "kernel_sha": None, "kernels": {}}
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
out["kernel_sha"] = bench.kernel_source_sha()   # bench.py quotes this file only for the build it was measured on
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    out["kernels"][names[k]] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024}
print(json.dumps(out, indent=1))
