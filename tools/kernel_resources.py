#!/usr/bin/env python3
"""Register / LDS / spill figures of every kernel in the built library, read from the code objects' own
metadata (no GPU needed):  python tools/kernel_resources.py [lib.so] [--json]

The .so carries one clang offload bundle per object file in its .hip_fatbin section; each bundle holds a gfx950
ELF whose .note section is the msgpack-encoded AMDGPU metadata (llvm-readelf --notes prints it)."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    """gfx950 ELF images inside the library."""
    blob = open(path, "rb").read()
    out = []
    pos = blob.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "gfx950" in triple and size > 0:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(MAGIC, pos + 1)
    if blob.find(b"CCOB") >= 0 and not out:
        raise RuntimeError("compressed offload bundle: build with --no-offload-compress")
    return out


def kernels(path=None):
    """{demangled kernel name: dict(vgpr, agpr, sgpr, spill_vgpr, spill_sgpr, scratch_bytes, lds_bytes, max_flat_wg)}"""
    path = path or os.path.join(ROOT, "rte-ecckd_amd", "librte_ecckd_hip.so")
    res = {}
    for img in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
        cur = {}
        for line in txt.splitlines():
            m = re.match(r"\s*-?\s*\.([a-z_]+):\s*(.*)$", line)
            if not m:
                continue
            key, val = m.group(1), m.group(2).strip()
            if key == "agpr_count" and cur.get("name"):   # first key of a kernel record (keys are sorted)
                cur = {}
            cur[key] = val
            if key == "wavefront_size":                    # last key of a kernel record
                name = cur.get("name", "?")
                res[name] = dict(vgpr=int(cur.get("vgpr_count", 0)), agpr=int(cur.get("agpr_count", 0)),
                                 sgpr=int(cur.get("sgpr_count", 0)), spill_vgpr=int(cur.get("vgpr_spill_count", 0)),
                                 spill_sgpr=int(cur.get("sgpr_spill_count", 0)),
                                 scratch_bytes=int(cur.get("private_segment_fixed_size", 0)),
                                 lds_bytes=int(cur.get("group_segment_fixed_size", 0)),
                                 max_flat_wg=int(cur.get("max_flat_workgroup_size", 0)))
                cur = {}
    names = list(res)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return {d: res[n] for n, d in zip(names, dem)}


def waves_per_simd(k):
    """allocation granule 8 registers, 512 per lane per SIMD (MI355X_MICROARCH.md, Register files)"""
    alloc = -(-(max(k["vgpr"], 1)) // 8) * 8
    return min(8, 512 // alloc)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ks = kernels(args[0] if args else None)
    if "--json" in sys.argv:
        print(json.dumps(ks, indent=1))
    else:
        for n in sorted(ks):
            k = ks[n]
            short = re.sub(r"^void ecckd::\(anonymous namespace\)::", "", n)
            short = re.sub(r"\(.*\)$", "", short)
            print("%-78s vgpr %3d (+agpr %3d) sgpr %3d spill v%d/s%d scratch %4d B  waves/SIMD %d" %
                  (short[:78], k["vgpr"], k["agpr"], k["sgpr"], k["spill_vgpr"], k["spill_sgpr"], k["scratch_bytes"], waves_per_simd(k)))
