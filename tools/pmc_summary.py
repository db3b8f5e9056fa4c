#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters over one or more pass directories.

usage: pmc_summary.py <workload string> <ncol> <dir> [<dir> ...]  > profiles/rNN_pmc_xx.json

Kernels are keyed by a short name: gas_fused_kernel instantiations by their MODE template argument (tau / gas_lw_fused /
gas_sw), the solvers by their kernel name.  Counters are chip-wide sums per dispatch as rocprofv3 reports them (SQ *_CYCLES /
ACTIVE / WAIT counters in quad-cycles: MI355X_MICROARCH.md).  The file records the kernel-source hash of the build it was
measured on (bench.kernel_source_sha) -- bench.py quotes it only for that build."""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def short(name):
    m = re.search(r"gas_fused_kernel<(\w+), \d+, \d+, \w+, \w+, (\d)(?:, (\w+))?(?:, \d+)?>", name)
    if m:
        base = {"0": "tau", "1": "gas_lw_fused", "2": "gas_sw"}[m.group(2)]
        if m.group(1) == "double" and m.group(3) == "float":      # fp64 over the float32 image of the tables ("gas_slab_f32"):
            return base + "_slab32"                               # with the option on auto this is the launch that returns at once
        return base + ("_f32" if m.group(1) == "float" else "")
    m = re.search(r"rte_lw_kernel<(\w+), \d+, \d+, \w+, \w+, (\w+), \w+(?:, \w+)?>", name)
    if m:
        return "rte_lw" + ("_f32" if m.group(1) == "float" else "") + ("_shared_levels" if m.group(2) == "true" else "")
    m = re.search(r"rte_lw_split_kernel<\d+, \d+, \d+, \w+, \w+, (\w+), \d>", name)
    if m:
        return "rte_lw_fused" if m.group(1) == "true" else "rte_lw_split"
    m = re.search(r"rte_sw_sys_kernel<(\w+), \w+, \w+, (\w+), \w+>", name)
    if m:       # (the 4th template argument: ecckd_sw_fluxes' form, which reads the total optical depth only)
        return "rte_sw" + ("_f32" if m.group(1) == "float" else "") + ("_fused" if m.group(2) == "true" else "")
    if "rte_sw_kernel" in name:
        return "rte_sw_two_pass"
    for key in ("tau_kernel", "planck_kernel", "toa_src_kernel", "lw_gpt_kernel", "sw_gpt_kernel"):
        if key in name:
            return key.replace("_kernel", "")
    return None


def collect(dirs):
    acc = {}
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                if k:
                    acc.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


if __name__ == "__main__":
    out = {"workload": sys.argv[1], "ncol": int(sys.argv[2]), "kernel_sha": bench.kernel_source_sha(),
           "_how": "rocprofv3 --pmc <one group per pass> --output-format csv -- python3 bench.py ... --no-side "
                   "--cpu-seconds 0 (no tracing); mean over the dispatches of each kernel; tools/pmc_summary.py",
           "kernels": collect(sys.argv[3:])}
    for k, cs in out["kernels"].items():   # HBM bytes per launch, gfx950 correction (MI355X_MICROARCH.md, HBM section)
        if "FETCH_SIZE" in cs or "WRITE_SIZE" in cs:
            cs["hbm_bytes_per_launch"] = 2 * cs.get("FETCH_SIZE", 0.0) * 1024 + cs.get("WRITE_SIZE", 0.0) * 1024
    print(json.dumps(out, indent=1))
