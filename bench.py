#!/usr/bin/env python3
"""Headline benchmark: LW gas_optics + rte_lw throughput in Mcol*lay*gpt/s.

One "step" = one pass of the hot path (ecckd_gas_optics_lw then ecckd_rte_lw, 1 quadrature
angle, top_at_1, fp64) over `--ncol` synthetic columns x 60 layers x 32 g-points PER GPU, all
inputs and the four (ncol,nlay,ngpt) intermediates resident in HBM.  Columns are independent, so
N GPUs = N column ranges (rank r generates columns [r*ncol, (r+1)*ncol)), no data-path
collective: weak scaling.  See DESIGN.md "Measurement".

  python bench.py --gpus 1 --steps 10 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
LW_FILE = os.path.join(ROOT, "data", "ecckd-1.2_lw_ckd-definition_climate_fsck-tol0.0161.nc")
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
NLAY = 60


def algorithmic_bytes_per_column(ng, nlay=NLAY):
    """SURVEY.md §8(d) accounting at the API boundary (fp64), split per kernel.
    tau kernel:    writes tau (8 B/cell); reads plev(61) tlay(60) h2o(60) o3(60) + 5 per-column vmr
    planck kernel: writes lay_source, lev_source_inc, lev_source_dec (24 B/cell) + sfc_source;
                   reads tlay(60) tlev(61) tsfc(1)
    gas_lw_fused:  the two above in one launch (fast arithmetic mode)
    rte_lw kernel: reads the four 3-D arrays (32 B/cell) + sfc_source + emis; writes 2x61 fluxes"""
    cells = nlay * ng
    tau = 8 * cells + 8 * ((nlay + 1) + nlay + 2 * nlay + 5)
    planck = 24 * cells + 8 * ng + 8 * (nlay + (nlay + 1) + 1)
    rte = 32 * cells + 8 * ng + 8 + 8 * 2 * (nlay + 1)
    return {"tau": tau, "planck": planck, "rte_lw": rte, "gas_lw_fused": tau + planck}


def columns_per_gpu(gpus, ncol_arg):
    """Columns per GPU when --ncol is not given: 1e6 (the north_star target size, weak scaling) for 1, 2, 4 GPUs;
    with 8 GPUs the job is BASELINE configs[3]: 1e7 columns, column-range sharded 1.25e6 per rank."""
    if ncol_arg is not None:
        return int(ncol_arg)
    return 1250000 if gpus == 8 else 1000000


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def kernel_source_sha():
    """sha256 (16 hex digits) of the kernel sources: ties a committed PMC traffic file to the build it measured."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rte-ecckd_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith(".hip") or f in ("kernels.hpp", "wave_pair.hpp", "sw_two_stream.hpp", "sw_two_stream_body.inc", "lw_layer.hpp"):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(name, key):
    """HBM bytes per launch of kernel `key` from the committed rocprofv3 PMC passes profiles/<name> -- only if
    that file was measured on THIS build of the kernels (kernel_sha), else None."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", name)))
        if tj.get("kernel_sha") != kernel_source_sha():
            return None, "profiles/%s was measured on another build of the kernels (kernel_sha differs): not quoted" % name
        return tj["kernels"][key]["hbm_bytes_per_launch"], ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                                                            "passes of this workload and build, gfx950-corrected)" % name)
    except Exception as e:   # no file, no such kernel
        return None, "no committed PMC pass for this workload (%s)" % type(e).__name__


FP64_VECTOR_PEAK_TFLOPS = 78.6    # MI355X fp64 vector peak: 256 CUs x 4 SIMDs x 16 FMA lanes/clk x 2 flop x 2.4 GHz
                                   # (half the 157.3 TFLOP/s fp32 vector figure of MI355X_MICROARCH.md)


def committed_counters(name, key, ncol):
    """Per-launch PMC counters of kernel `key` from profiles/<name> (tools/pmc_summary.py), scaled from the column count
    they were measured at to `ncol` -- only if the file was measured on THIS build of the kernels, else None."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", name)))
        if tj.get("kernel_sha") != kernel_source_sha():
            return None, "profiles/%s was measured on another build of the kernels (kernel_sha differs): not quoted" % name
        f = ncol / float(tj["ncol"])
        return {c: v * f for c, v in tj["kernels"][key].items()}, "profiles/%s (rocprofv3 --pmc, separate passes of this build)" % name
    except Exception as e:
        return None, "no committed PMC pass for this workload (%s)" % type(e).__name__


def cgroup_cpu_quota():
    """CPUs the cgroup of this process may use (cpu.max of cgroup v2, cfs quota of v1), or None if unlimited / unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / p if q > 0 else None
    except Exception:
        return None


def cpu_baseline(args, press_min):
    """The CPU restatement (oracle/, 'port' of the reference Fortran: same per-gas passes and temporaries as
    src/gas_optics_ecckd.f90:117-240,370 plus the RTE LW recurrences) timed on this host's cores over bounded
    samples of the same synthetic columns.  Legs:
      value          all CPUs the process may use (cgroup quota, else the visible hardware threads): column blocks spread
                     over that many OpenMP threads, every thread re-using one arena of temporaries for all its blocks;
                     block size = the best of a short sweep over 8 / 64 / 256
      one_thread     1 thread, block = 1 (what the reference driver does: ecckd_rfmip_lw.F90:39) and block = 1000
    and the ratio of the two, so that a poor all-cores figure shows."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    from rte_ecckd_amd import synthetic
    m = oracle.CkdModel(LW_FILE)
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = cgroup_cpu_quota()
    # threads = the CPUs this process may actually use: the GPU boxes show all 256 hardware threads of the host but grant
    # a cgroup quota of 16 CPUs per GPU -- 256 threads on that quota time-slice (round 2 did that: 10x one thread), and
    # pinning them (OMP_PROC_BIND=close) made it worse (1.2x one thread, measured in round 3): threads float, one per CPU
    cores = max(1, min(visible, int(quota + 0.5))) if quota else visible

    def timed(n, block, nthreads):
        cols = synthetic.columns(0, n, press_min)
        items = synthetic.gas_items(cols)
        t0 = time.perf_counter()
        oracle.lw_pipeline(m, cols["plev"], cols["tlay"], cols["tlev"], cols["tsfc"], items, cols["sfc_emis"],
                           block=block, nthreads=nthreads)
        return max(time.perf_counter() - t0, 1e-6)

    def leg(block, nthreads, seconds, probe_n):
        dt = timed(probe_n, block, nthreads)
        n = int(min(args.ncol or 1000000, max(probe_n, probe_n * seconds / dt)))
        n -= n % (block * nthreads) or 0
        n = max(n, block * nthreads)
        dt = timed(n, block, nthreads)
        return n, dt, n * NLAY * m.ng / dt / 1e6

    # short sweep over the block size at all cores (second call of each: threads and pages are warm)
    sweep = {}
    blocks = (args.cpu_block,) if args.cpu_block else (8, 64, 256)
    for blk in blocks:
        n0 = blk * cores * 2
        timed(n0, blk, cores)
        sweep[blk] = n0 * NLAY * m.ng / timed(n0, blk, cores) / 1e6
    block = max(sweep, key=sweep.get)
    n, dt, rate = leg(block, cores, args.cpu_seconds, block * cores * 2)
    out = {"value": rate, "unit": "Mcol*lay*gpt/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
           "nproc": os.cpu_count(), "block": block, "block_sweep_Mcell_s": {str(k): v for k, v in sweep.items()},
           "hardware_threads_visible": visible, "cgroup_cpu_quota": quota,
           "omp": {"OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_PLACES": os.environ.get("OMP_PLACES")},
           "sample": "%d synthetic columns x %d layers x %d g-points, column blocks of %d over %d OpenMP "
                     "threads, %.1f s" % (n, NLAY, m.ng, block, cores, dt)}
    legs = {}
    for blk in (1, 1000):
        n1, dt1, r1 = leg(blk, 1, args.cpu_seconds / 3.0, max(blk, 8))
        legs["block_%d" % blk] = {"value": r1, "unit": "Mcol*lay*gpt/s", "cores": 1,
                                  "sample": "%d columns, blocks of %d, 1 thread, %.1f s" % (n1, blk, dt1)}
    out["one_thread"] = legs
    best1 = max(v["value"] for v in legs.values())
    out["scaling_vs_one_thread"] = rate / best1
    out["scaling_note"] = ("all-CPUs rate / best 1-thread rate with %d threads (%d hardware threads visible, cgroup quota %s CPUs; "
                           "the port streams ~380 B per cell through memory as the reference does)" % (cores, visible, quota))
    return out


def fortran_device_resident(pkg, lw_file, n, press_min, ng):
    """The reference-language path at the GPU's rate: the Fortran driver (rte-ecckd_amd/fortran/ecckd_driver.F90, the
    block loop of ecckd_rfmip_lw.F90:107-136 over the drop-in module) with the device-resident twins of optical_props /
    source (ECCKD_MIXED: the atmosphere goes in, the fluxes come out, tau and the sources stay in HBM), all columns in
    one block, against the same driver with host containers (ECCKD_HOST).  Wall time of the loop, best of 3, PCIe
    transfers included -- a PCIe-bound figure, reported beside the headline, never as it."""
    import struct
    import subprocess
    import tempfile
    from rte_ecckd_amd import synthetic
    drv = pkg.FORTRAN_DRIVER
    if not os.path.exists(drv):
        return {"skipped": "no Fortran driver binary (amdflang missing at build time)"}
    cols = synthetic.columns(0, n, press_min)
    names = synthetic.GAS_ORDER
    nlay = NLAY
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(inp, "wb") as f:
            f.write(struct.pack("<iii", n, nlay, len(names)))
            for nm in names:
                f.write(nm.encode().ljust(32, b" "))
            for key in ("plev", "tlev", "tlay", "tsfc", "sfc_emis"):
                f.write(np.ascontiguousarray(cols[key], dtype="<f8").tobytes())
            for nm in names:
                v = cols[nm]
                full = np.broadcast_to(np.float64(v) if np.isscalar(v) else np.asarray(v, dtype=np.float64), (nlay, n))
                f.write(np.ascontiguousarray(full, dtype="<f8").tobytes())
        res = {}
        for label, dev in (("device_resident", "1"), ("host_arrays", "0")):
            nn = n if dev == "1" else min(n, 20000)
            r = subprocess.run([drv, "lw", lw_file, inp, outp, str(nn), "1", dev, "3"], capture_output=True, text=True)
            if r.returncode != 0:
                return {"error": r.stderr[-300:]}
            secs = None
            for line in r.stderr.splitlines():
                if "loop_seconds" in line:
                    secs = float(line.split()[-1])
            blocks = (n + nn - 1) // nn
            res[label] = {"value": n * nlay * ng / secs / 1e6, "unit": "Mcol*lay*gpt/s", "loop_seconds": secs,
                          "ncol": n, "block": nn, "blocks": blocks}
        a = np.fromfile(outp, dtype="<f8")
    res["note"] = ("Fortran ecckd%gas_optics + rte_lw through the type-bound API; device_resident: mo_ecckd_device twins, "
                   "about 5 KB per column in and 1 KB out over PCIe, pageable host memory; host_arrays: the reference's "
                   "calling convention, 64 B per cell out and back in")
    return res


WELL_MIXED = dict(co2=420e-6, ch4=1.9e-6, n2o=3.3e-7, cfc11=2.3e-10, cfc12=5.2e-10)


class LwCase:
    """Device-resident inputs, intermediates and outputs of one LW gas_optics + rte_lw workload (columns
    c0 .. c0+ncol-1 of the counter-based synthetic generator) and the step that runs it."""

    def __init__(self, pkg, k, ncol, c0, dev, tdt, press_min):
        import torch
        from rte_ecckd_amd import synthetic
        self.pkg, self.k, self.ncol = pkg, k, ncol
        nlay = NLAY
        kw = dict(dtype=tdt, device=dev)
        self.plev = torch.empty((nlay + 1, ncol), **kw)
        self.tlev = torch.empty((nlay + 1, ncol), **kw)
        self.tlay = torch.empty((nlay, ncol), **kw)
        h2o = torch.empty((nlay, ncol), **kw)
        o3 = torch.empty((nlay, ncol), **kw)
        self.percol = {n: torch.empty((ncol,), **kw) for n in ("tsfc", "sfc_emis", "co2", "ch4", "n2o", "cfc11", "cfc12")}
        chunk = 100000
        for s0 in range(0, ncol, chunk):      # generated in chunks on the host
            n = min(chunk, ncol - s0)
            cols = synthetic.columns(c0 + s0, n, press_min)
            for dst, key in ((self.plev, "plev"), (self.tlev, "tlev"), (self.tlay, "tlay"), (h2o, "h2o"), (o3, "o3")):
                dst[:, s0:s0 + n] = torch.from_numpy(cols[key]).to(dev).to(tdt)
            for key, dst in self.percol.items():
                dst[s0:s0 + n] = torch.from_numpy(cols[key]).to(dev).to(tdt)
        self.gc = pkg.GasConcs(synthetic.GAS_ORDER)
        # RFMIP's description of the same gases (mo_rfmip_io.F90: <gas>_GM): one number per well-mixed gas for the call
        self.gc_scalars = pkg.GasConcs(synthetic.GAS_ORDER)
        for name in synthetic.GAS_ORDER:
            if name in ("h2o", "o3"):
                self.gc.set_vmr(name, h2o if name == "h2o" else o3)
                self.gc_scalars.set_vmr(name, h2o if name == "h2o" else o3)
            elif name in self.percol:
                self.gc.set_vmr_column(name, self.percol[name])
                self.gc_scalars.set_vmr(name, WELL_MIXED[name])
            else:
                self.gc.set_vmr(name, 0.209 if name == "o2" else 0.0)
                self.gc_scalars.set_vmr(name, 0.209 if name == "o2" else 0.0)
        self.emis = self.percol["sfc_emis"].reshape(ncol, 1).expand(ncol, k.get_nband()).contiguous()
        self.op = pkg.OpticalProps1scl()
        self.op.alloc_1scl(ncol, nlay, k, like=self.plev)
        self.src = pkg.SourceFuncLW()
        self.src.alloc(ncol, nlay, k, like=self.plev)
        self.fl = pkg.FluxesBroadband(torch.empty((nlay + 1, ncol), **kw), torch.empty((nlay + 1, ncol), **kw))

    def step(self, shared_levels=False):
        e = self.k.gas_optics(None, self.plev, self.tlay, self.percol["tsfc"], self.gc, self.op, self.src, tlev=self.tlev)
        e = e or self.pkg.rte_lw(self.op, True, self.src, self.emis, self.fl, n_gauss_angles=1, shared_levels=shared_levels)
        if e:
            raise SystemExit(e)

    def timed(self, steps, warmup):
        import torch
        for _ in range(max(warmup, 1)):
            self.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps


def prof_report(L):
    names = C.create_string_buffer(8 * 32)
    ms = (C.c_double * 8)()
    cnt = (C.c_longlong * 8)()
    nk = L.ecckd_prof_report(8, names, ms, cnt)
    return {names.raw[i * 32:(i + 1) * 32].split(b"\0")[0].decode(): (ms[i] / max(cnt[i], 1), int(cnt[i])) for i in range(nk)}


def lw_measure(pkg, lw_file, ncol, dtype, steps, warmup, device=0):
    """One LW gas_optics + rte_lw workload beside the headline (another table and / or precision): value, ms per step,
    per-kernel HIP-event times, fraction of the HBM roofline at the API-boundary accounting of the precision, and the
    largest broadband-flux difference of 64 columns against the fp64 CPU oracle."""
    import torch
    from rte_ecckd_amd import synthetic
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    L = pkg.lib()
    k = pkg.GasOpticsEcckd()
    err = k.load(lw_file, device=device)
    if err:
        raise SystemExit(err)
    ng, nlay = k.get_ngpt(), NLAY
    press_min = k.get_press_min()
    tdt = torch.float32 if dtype == "f32" else torch.float64
    case = LwCase(pkg, k, ncol, 0, torch.device("cuda", device), tdt, press_min)
    for _ in range(max(warmup, 1)):
        case.step()
    torch.cuda.synchronize()
    L.ecckd_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        case.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    L.ecckd_prof_enable(0)
    kern = prof_report(L)
    bpc = algorithmic_bytes_per_column(ng)
    total_b = (bpc["tau"] + bpc["planck"] + bpc["rte_lw"]) * ncol // (2 if dtype == "f32" else 1)
    m = oracle.CkdModel(lw_file)
    cols = synthetic.columns(0, 64, press_min)
    tau, lay, inc, dec, sfc, _ = oracle.gas_optics_int(m, cols["plev"], cols["tlay"], cols["tsfc"], synthetic.gas_items(cols), cols["tlev"])
    fu, fd = oracle.rte_lw(tau, lay, inc, dec, np.repeat(cols["sfc_emis"][None, :], ng, 0), sfc)
    dflux = max(float(np.max(np.abs(case.fl.flux_up[:, :64].double().cpu().numpy() - fu))),
                float(np.max(np.abs(case.fl.flux_dn[:, :64].double().cpu().numpy() - fd))))
    del case
    torch.cuda.empty_cache()
    return {"workload": "synthetic %d columns x %d layers x %d g-points, LW %s, gas_optics + rte_lw (1 angle), %s"
                        % (ncol, nlay, ng, os.path.basename(lw_file)[36:-3], "fp64" if dtype == "f64" else "fp32"),
            "value": ncol * nlay * ng / dt / 1e6, "unit": "Mcol*lay*gpt/s", "ms_per_step": dt * 1e3, "dtype": dtype,
            "kernels_avg_ms": {n: v[0] for n, v in kern.items()},
            "frac_of_hbm_roofline": total_b / dt / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_cell": total_b / ncol / (nlay * ng),
            "max_abs_flux_diff_vs_fp64_oracle_Wm2": dflux}


def self_launch(ngpus):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU as a CHILD process group
    (`python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>`), pass its output through and
    return its exit code.  Nothing in this parent touches the GPU (no torch import, no HIP call): the ranks
    initialise their own devices.  Launched through torch.distributed.run directly (RANK set) this is never reached."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:   # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in child.stdout:      # rank 0's JSON line (and anything else the ranks print) as it comes
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = child.wait()
    if rc != 0:
        raise SystemExit(rc)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ncol", type=int, default=None,
                    help="synthetic columns per GPU (default: 1e6, the north_star target size; with --gpus 8: 1.25e6 = "
                         "BASELINE configs[3], 1e7 columns sharded over 8 GPUs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget of the all-cores leg (0 = skip)")
    ap.add_argument("--host-sample", type=int, default=20000,
                    help="columns of the PCIe-inclusive side measurement through the ECCKD_HOST memory space "
                         "(host arrays in, host arrays out; 0 = skip).  Reported beside the headline, never as it.")
    ap.add_argument("--cpu-block", type=int, default=0,
                    help="columns per block in the all-cores CPU baseline leg (0: the best of a sweep over 8, 64, 256)")
    ap.add_argument("--lut", choices=["fsck", "rrtmgp"], default="fsck",
                    help="LW table: fsck-tol0.0161 (32 g, headline) or rrtmgp-tol0.061 (36 g, 16 bands; BASELINE configs[4])")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64",
                    help="f64 = headline; f32 = single-precision flavour of the LW path (BASELINE configs[4] sweep)")
    ap.add_argument("--arithmetic", choices=["fast", "reference"], default="fast",
                    help="gas-optics arithmetic mode (ecckd_set_arithmetic): fast = fused kernel (headline); reference = "
                         "per-gas kernels in the reference's expression order")
    ap.add_argument("--mode", choices=["lw", "sw"], default="lw",
                    help="lw = headline metric; sw = secondary line (BASELINE configs[2]: gas_optics + rte_sw)")
    ap.add_argument("--no-side", action="store_true", help="skip every side measurement (profiling passes)")
    ap.add_argument("--solver-option", action="append", default=[], metavar="NAME=VALUE",
                    help="ecckd_set_solver_option before the run (A/B of implementation choices, e.g. lw_tail_split=0)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a single-GPU box: every rank uses cuda:0 and the process group is gloo "
                         "(RCCL cannot put two ranks on one device).  Exercises the rank/column-range/barrier/MAX/gather "
                         "logic; the number it prints is NOT a scaling measurement and says so.")
    ap.add_argument("--fortran-sample", type=int, default=200000,
                    help="columns of the side measurement through the Fortran type-bound API in device-resident mode "
                         "(ecckd_driver: host arrays in, fluxes out, tau and sources stay in HBM; 0 = skip)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        return self_launch(args.gpus)
    if args.mode == "sw":
        if args.ncol is None:
            args.ncol = 100000
        return main_sw(args)

    import torch
    import torch.distributed as dist
    import rte_ecckd_amd as pkg
    from rte_ecckd_amd import synthetic

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched through torch.distributed.run" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1 or "RANK" in os.environ   # launched through torch.distributed.run
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cpu_coll = distributed and args.rehearse_on_one_gpu   # gloo: collectives on host tensors

    lw_file = LW_FILE if args.lut == "fsck" else LW_FILE.replace("fsck-tol0.0161", "rrtmgp-tol0.061")
    L = pkg.lib()
    pkg.set_arithmetic(pkg.FAST if args.arithmetic == "fast" else pkg.REFERENCE_ORDER)
    for item in args.solver_option:
        name, _, value = item.partition("=")
        pkg.set_solver_option(name, float(value))
    k = pkg.GasOpticsEcckd()
    err = k.load(lw_file, device=local_rank)
    if err:
        raise SystemExit(err)
    ncol = columns_per_gpu(world, args.ncol)
    args.ncol = ncol
    ng, nlay = k.get_ngpt(), NLAY
    press_min = k.get_press_min()
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    f64 = dict(dtype=tdt, device=dev)

    # rank r holds columns [r*ncol, (r+1)*ncol) of the generator: column-range sharding, no data-path collective
    case = LwCase(pkg, k, ncol, rank * ncol, dev, tdt, press_min)
    fl, op, src, emis = case.fl, case.op, case.src, case.emis
    step = case.step

    def barrier():
        if distributed:
            if cpu_coll:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    L.ecckd_prof_enable(1)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    t_rank = time.perf_counter() - t0      # this rank's own time, before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    L.ecckd_prof_enable(0)

    # per-kernel HIP-event durations recorded inside the timed region
    kern = prof_report(L)

    rank_ms = [t_rank / args.steps * 1e3]
    if distributed:
        cdev = torch.device("cpu") if cpu_coll else dev
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([t_rank / args.steps * 1e3], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rank_ms = [float(x.item()) for x in allr]

    if rank == 0:
        cells_per_gpu = ncol * nlay * ng
        value = world * cells_per_gpu * args.steps / elapsed / 1e6
        ms_per_step = elapsed / args.steps * 1e3
        bpc = algorithmic_bytes_per_column(ng)
        if args.dtype == "f32":
            bpc = {kk: vv // 2 for kk, vv in bpc.items()}
            bpc["gas_lw_fused_f32"] = bpc["gas_lw_fused"]
        per_kernel = {}
        for name, (avg_ms, n) in kern.items():
            b = bpc.get(name, 0) * ncol
            per_kernel[name] = {"avg_ms": avg_ms, "launches": n, "alg_bytes_per_launch": b,
                                "GBps": b / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else None}
        dom = max(kern, key=lambda n: kern[n][0]) if kern else None
        roofline = None
        if dom:
            traffic, tsrc = (None, "only quoted for the 1e6-column fp64 fsck workload")
            if ncol == 1000000 and args.dtype == "f64" and args.lut == "fsck" and args.arithmetic == "fast":
                traffic, tsrc = committed_traffic("r03_hbm_traffic.json", dom)
            ach = per_kernel[dom]["GBps"]
            roofline = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                        "avg_launch_ms": per_kernel[dom]["avg_ms"],
                        "alg_bytes_per_launch": per_kernel[dom]["alg_bytes_per_launch"]}
        total_b = (bpc["tau"] + bpc["planck"] + bpc["rte_lw"]) * ncol
        pipe = total_b / (ms_per_step * 1e-3) / 1e9
        # spot check of the timed configuration against the CPU oracle (64 columns)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle
        m = oracle.CkdModel(lw_file)
        cols = synthetic.columns(0, 64, press_min)
        tau, lay, inc, dec, sfc, _ = oracle.gas_optics_int(m, cols["plev"], cols["tlay"], cols["tsfc"],
                                                           synthetic.gas_items(cols), cols["tlev"])
        fu, fd = oracle.rte_lw(tau, lay, inc, dec, np.repeat(cols["sfc_emis"][None, :], ng, 0), sfc)
        dflux = max(float(np.max(np.abs(fl.flux_up[:, :64].double().cpu().numpy() - fu))),
                    float(np.max(np.abs(fl.flux_dn[:, :64].double().cpu().numpy() - fd))))
        if world == 8 and ncol == 1250000:
            what = "BASELINE configs[3]: 1e7 synthetic columns, column-range sharded 1.25e6 per GPU over 8 GPUs"
        elif ncol == 1000000:
            what = "north_star target size, 1e6 columns per GPU (configs[1] is the same workload at 1e5 columns: key configs_1)"
        elif ncol == 100000:
            what = "BASELINE configs[1] size"
        else:
            what = "custom size"
        out = {
            "metric": "Mcol*lay*gpt/s LW gas_optics+rte_lw", "value": value, "unit": "Mcol*lay*gpt/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": ("synthetic %d columns x %d layers x %d g-points per GPU, LW " % (ncol, nlay, ng)) + os.path.basename(lw_file)[36:-3] + ", "
                                   "gas_optics + rte_lw (1 angle), " + ("fp64" if args.dtype == "f64" else "fp32") + ", inputs and intermediates HBM-resident; " + what,
                       "ncol_per_gpu": ncol, "ncol_total": ncol * world, "nlay": nlay, "ngpt": ng,
                       "parallelism": "column-range x%d, no collective" % world + (
                           " -- REHEARSAL: all ranks on one GPU over gloo, not a scaling measurement" if args.rehearse_on_one_gpu else ""),
                       "arithmetic": args.arithmetic, "solver_options": pkg.solver_options()},
            "per_rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms), "ranks": rank_ms},
            "roofline": roofline,
            "roofline_pipeline": {"bound": "hbm", "achieved": pipe, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": pipe / HBM_PEAK_GBS, "alg_bytes_per_cell": total_b / ncol / (nlay * ng),
                                  "note": "all kernels of one step, per GPU"},
            "kernels": per_kernel,
            "check_max_abs_flux_diff_vs_oracle_Wm2": dflux,
        }
        side = world == 1 and not args.no_side
        if args.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, press_min)
        else:
            out["cpu_baseline"] = None
        # Side measurement (not the headline): the solver told that ecckd's level sources hold one value per level
        # (ecckd_rte_lw_shared_levels): 24 instead of 32 B/cell read, bit-identical fluxes.
        out["rte_lw_shared_levels"] = None
        if side and args.dtype == "f64":
            ref_up = fl.flux_up.clone()
            for _ in range(2):
                e = pkg.rte_lw(op, True, src, emis, fl, n_gauss_angles=1, shared_levels=True)
                if e:
                    raise SystemExit(e)
            torch.cuda.synchronize()
            identical = bool(torch.equal(ref_up, fl.flux_up))
            t0 = time.perf_counter()
            for _ in range(args.steps):
                pkg.rte_lw(op, True, src, emis, fl, n_gauss_angles=1, shared_levels=True)
            torch.cuda.synchronize()
            ms_sh = (time.perf_counter() - t0) / args.steps * 1e3
            ms_gas = sum(v["avg_ms"] for n, v in per_kernel.items() if n != "rte_lw")
            out["rte_lw_shared_levels"] = {
                "rte_lw_ms": ms_sh, "pipeline_ms": ms_gas + ms_sh,
                "pipeline_value": cells_per_gpu / ((ms_gas + ms_sh) * 1e-3) / 1e6, "unit": "Mcol*lay*gpt/s",
                "fluxes_bit_identical_to_generic_solver": identical,
                "note": "opt-in entry point; the headline value uses the generic ecckd_rte_lw, which reads both level arrays"}
        # Side measurements on the same resident inputs: (a) the layer-split longwave solver behind the same API; (b) the
        # fused longwave path -- gas optics writes tau only, the solver recomputes the Planck sources (16 instead of
        # 64 B/cell between the kernels; bound by fp64 issue, not HBM: reported apart from the API-boundary roofline).
        out["lw_solver_variants"] = None
        out["well_mixed_scalars"] = None
        out["fused_lw"] = None
        if side and args.dtype == "f64" and args.arithmetic == "fast":
            saved = pkg.get_solver_option("lw_solver")
            var = {}
            for label, sel, seg in (("register_resident", 0, 10), ("layer_split_seg10", 1, 10), ("layer_split_seg12", 1, 12),
                                    ("layer_split_seg15", 1, 15)):
                pkg.set_solver_option("lw_solver", sel); pkg.set_solver_option("lw_split_seg", seg)
                for _ in range(2):
                    pkg.rte_lw(op, True, src, emis, fl, n_gauss_angles=1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    pkg.rte_lw(op, True, src, emis, fl, n_gauss_angles=1)
                torch.cuda.synchronize()
                var[label] = {"rte_lw_ms": (time.perf_counter() - t0) / args.steps * 1e3}
            pkg.set_solver_option("lw_solver", saved); pkg.set_solver_option("lw_split_seg", 10)
            out["lw_solver_variants"] = var
            # (c) the same columns with the well-mixed gases described as RFMIP describes them -- one number per gas for
            # the call instead of one per column: the gas-optics kernel folds them and the composite into one table
            # ("gas_merge_scalars"); h2o and o3 stay profiles.  Not the headline (BASELINE's inputs vary them per column).
            def sstep():
                e = k.gas_optics(None, case.plev, case.tlay, case.percol["tsfc"], case.gc_scalars, op, src, tlev=case.tlev)
                e = e or pkg.rte_lw(op, True, src, emis, fl, n_gauss_angles=1)
                if e:
                    raise SystemExit(e)
            for _ in range(max(args.warmup, 1)):
                sstep()
            torch.cuda.synchronize()
            L.ecckd_prof_enable(1)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                sstep()
            torch.cuda.synchronize()
            ms_s = (time.perf_counter() - t0) / args.steps * 1e3
            L.ecckd_prof_enable(0)
            ks = prof_report(L)
            pl = k.plan(ncol, nlay, synthetic.GAS_ORDER, scalar_gases=list(WELL_MIXED) + ["o2", "no2"])
            out["well_mixed_scalars"] = {
                "value": cells_per_gpu / (ms_s * 1e-3) / 1e6, "unit": "Mcol*lay*gpt/s", "ms_per_step": ms_s,
                "frac_of_hbm_roofline": total_b / (ms_s * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "kernels_avg_ms": {n: v[0] for n, v in ks.items()}, "gases_merged": pl["merged"], "slots": pl["slots"],
                "note": "co2, ch4, n2o, cfc11, cfc12 as one number each (RFMIP's *_GM), o2 0.209; same kernels, same API"}
            case.step()          # the headline's arrays back in op / src for what follows
            if nlay == 60:
                ref_up, ref_dn = fl.flux_up.clone(), fl.flux_dn.clone()
                tsfc_d = case.percol["tsfc"]

                def fstep():
                    e = k.lw_fluxes(case.plev, case.tlay, tsfc_d, case.tlev, case.gc, True, emis, fl, n_gauss_angles=1)
                    if e:
                        raise SystemExit(e)
                del src          # the fused path needs no source arrays: free 46 GB before tau scratch is taken
                case.src = None
                torch.cuda.empty_cache()
                for _ in range(max(args.warmup, 1)):
                    fstep()
                torch.cuda.synchronize()
                L.ecckd_prof_enable(1)
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    fstep()
                torch.cuda.synchronize()
                ms_f = (time.perf_counter() - t0) / args.steps * 1e3
                L.ecckd_prof_enable(0)
                kf = prof_report(L)
                dmax = max(float((fl.flux_up - ref_up).abs().max()), float((fl.flux_dn - ref_dn).abs().max()))
                out["fused_lw"] = {
                    "value": cells_per_gpu / (ms_f * 1e-3) / 1e6, "unit": "Mcol*lay*gpt/s", "ms_per_step": ms_f,
                    "speedup_vs_headline": ms_per_step / ms_f, "kernels_avg_ms": {n: v[0] for n, v in kf.items()},
                    "hbm_bytes_per_cell_between_kernels": 16, "bound": "fp64 VALU issue (not HBM): not quoted against the HBM roofline",
                    "max_abs_flux_diff_vs_api_path_Wm2": dmax,
                    "note": "ecckd_lw_fluxes: ecckd_gas_optics_lw_tau + ecckd_rte_lw_fused; same inputs, same fluxes"}
                # the fused path on the RFMIP-style gas description (side measurement (c) above)
                def fsstep():
                    e = k.lw_fluxes(case.plev, case.tlay, tsfc_d, case.tlev, case.gc_scalars, True, emis, fl, n_gauss_angles=1)
                    if e:
                        raise SystemExit(e)
                for _ in range(max(args.warmup, 1)):
                    fsstep()
                torch.cuda.synchronize()
                L.ecckd_prof_enable(1)
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    fsstep()
                torch.cuda.synchronize()
                ms_fs = (time.perf_counter() - t0) / args.steps * 1e3
                L.ecckd_prof_enable(0)
                kfs = prof_report(L)
                out["fused_lw"]["well_mixed_scalars"] = {
                    "value": cells_per_gpu / (ms_fs * 1e-3) / 1e6, "ms_per_step": ms_fs,
                    "speedup_vs_headline": ms_per_step / ms_fs, "kernels_avg_ms": {n: v[0] for n, v in kfs.items()}}
                src = None
                pkg.release_scratch(local_rank)
        del case, fl, op, src, emis
        torch.cuda.empty_cache()
        # BASELINE configs[1] (the same workload at 1e5 columns) and one rank's shard of configs[3] (1e7 columns over
        # 8 GPUs = 1.25e6 per rank; the LAST rank's columns) beside the 1e6-column headline
        out["configs_1"] = None
        out["configs_3_shard"] = None
        if side and ncol == 1000000 and args.lut == "fsck":
            for key, n1, c0, label in (("configs_1", 100000, 0, "synthetic 100000 columns x %d layers x %d g-points (BASELINE configs[1])"),
                                       ("configs_3_shard", 1250000, 7 * 1250000,
                                        "columns 8750000..9999999 of 1e7 x %d layers x %d g-points: the last rank's shard of BASELINE "
                                        "configs[3] (1e7 columns over 8 GPUs), run on one GPU")):
                c1 = LwCase(pkg, k, n1, c0, dev, tdt, press_min)
                dt1 = c1.timed(args.steps, args.warmup)
                out[key] = {"workload": label % (nlay, ng), "value": n1 * nlay * ng / dt1 / 1e6, "unit": "Mcol*lay*gpt/s",
                            "ms_per_step": dt1 * 1e3, "frac_of_hbm_roofline": (total_b / ncol) * n1 / dt1 / 1e9 / HBM_PEAK_GBS}
                del c1
                torch.cuda.empty_cache()
        # BASELINE configs[2] (SW pair, 1e5 columns) and configs[4] (the present higher-g-count LW table, 36 g-points, 1e6
        # columns, fp64 and fp32 with the flux difference of each against the fp64 oracle) in the same driver-run line
        out["configs_2"] = None
        out["configs_4"] = None
        if side and ncol == 1000000 and args.lut == "fsck" and args.dtype == "f64" and args.arithmetic == "fast":
            sw = sw_measure(100000, args.steps, args.warmup)
            out["configs_2"] = {k2: sw[k2] for k2 in ("value", "unit", "ms_per_step", "roofline", "roofline_fp64_valu", "roofline_pipeline",
                                                      "kernels", "fused_sw", "check_max_abs_flux_diff_vs_oracle_Wm2")}
            out["configs_2"]["workload"] = sw["config"]["workload"]
            sw32 = sw_measure(100000, args.steps, args.warmup, fused=False, dtype="f32")     # the same pair in single precision
            out["configs_2"]["f32"] = {k2: sw32[k2] for k2 in ("value", "unit", "ms_per_step", "roofline_pipeline", "kernels",
                                                               "check_max_abs_flux_diff_vs_oracle_Wm2")}
            torch.cuda.empty_cache()
            out["configs_4"] = {dt4: lw_measure(pkg, LW_FILE.replace("fsck-tol0.0161", "rrtmgp-tol0.061"), 1000000, dt4, args.steps,
                                                args.warmup, local_rank) for dt4 in ("f64", "f32")}
            out["configs_4"]["note"] = ("LW rrtmgp-tol0.061 (36 g-points, 16 bands): the higher-g-count table present in the reference "
                                        "tree (BASELINE names rrtmgp-tol0.0161, listed in the reference's .MISSING_LARGE_BLOBS)")
        out["host_memspace"] = None
        if args.host_sample > 0 and side and args.dtype == "f64":
            # The reference's calling convention: host arrays in and out (ECCKD_HOST).  Every call stages its
            # arguments over PCIe, so this is the PCIe-inclusive rate of the same two calls.
            n = min(args.host_sample, ncol)
            hc = synthetic.columns(0, n, press_min)
            hgc = pkg.GasConcs(synthetic.GAS_ORDER)
            for name in synthetic.GAS_ORDER:
                v = hc[name]
                if np.isscalar(v):
                    hgc.set_vmr(name, float(v))
                elif v.ndim == 1:
                    hgc.set_vmr_column(name, v)
                else:
                    hgc.set_vmr(name, v)
            hop = pkg.OpticalProps1scl(); hop.alloc_1scl(n, nlay, k)
            hsrc = pkg.SourceFuncLW(); hsrc.alloc(n, nlay, k)
            hfl = pkg.FluxesBroadband(np.empty((nlay + 1, n)), np.empty((nlay + 1, n)))
            hemis = np.ascontiguousarray(np.repeat(hc["sfc_emis"][:, None], k.get_nband(), 1))
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                e = k.gas_optics(None, hc["plev"], hc["tlay"], hc["tsfc"], hgc, hop, hsrc, tlev=hc["tlev"])
                e = e or pkg.rte_lw(hop, True, hsrc, hemis, hfl, n_gauss_angles=1)
                dt = time.perf_counter() - t0
                if e:
                    raise SystemExit(e)
                best = dt if best is None else min(best, dt)
            out["host_memspace"] = {"value": n * nlay * ng / best / 1e6, "unit": "Mcol*lay*gpt/s", "ncol": n,
                                    "note": "ECCKD_HOST: pageable host arrays staged over PCIe by every call (64 B/cell "
                                            "of intermediates out and back in); best of 3"}
        out["fortran_device_resident"] = None
        if side and args.dtype == "f64" and args.lut == "fsck" and args.fortran_sample > 0:
            out["fortran_device_resident"] = fortran_device_resident(pkg, lw_file, min(args.fortran_sample, ncol), press_min, ng)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


def main_sw(args):
    print(json.dumps(sw_measure(args.ncol, args.steps, args.warmup, args.solver_option, fused=not args.no_side, dtype=args.dtype)), flush=True)


def sw_measure(ncol, steps, warmup, solver_option=(), fused=True, dtype="f64"):
    """Secondary line: SW wide-tol0.05 (27 g-points), gas_optics (tau, ssa, g) + rte_sw two-stream, fp64 (or fp32),
    single GPU.  Algorithmic bytes: tau, ssa, g written once and read once = 48 B/cell (+ per-column terms)."""
    import types
    args = types.SimpleNamespace(ncol=ncol, steps=steps, warmup=warmup, solver_option=list(solver_option))
    import torch
    import rte_ecckd_amd as pkg
    from rte_ecckd_amd import synthetic
    sw_file = os.path.join(ROOT, "data", "ecckd-1.2_sw_ckd-definition_climate_wide-tol0.05.nc")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L = pkg.lib()
    for item in args.solver_option:
        name, _, value = item.partition("=")
        pkg.set_solver_option(name, float(value))
    k = pkg.GasOpticsEcckd()
    err = k.load(sw_file, device=0)
    if err:
        raise SystemExit(err)
    ng, ncol, nlay = k.get_ngpt(), args.ncol, NLAY
    tdt = torch.float32 if dtype == "f32" else torch.float64
    esz = 4 if dtype == "f32" else 8
    f64 = dict(dtype=tdt, device=dev)      # (the working precision of this run)
    plev = torch.empty((nlay + 1, ncol), **f64)
    tlay = torch.empty((nlay, ncol), **f64)
    h2o = torch.empty((nlay, ncol), **f64)
    o3 = torch.empty((nlay, ncol), **f64)
    percol = {n: torch.empty((ncol,), **f64) for n in ("co2", "ch4", "n2o", "mu0", "albedo")}
    for c0 in range(0, ncol, 100000):
        n = min(100000, ncol - c0)
        cols = synthetic.columns(c0, n, k.get_press_min(), shortwave=True)
        for dst, key in ((plev, "plev"), (tlay, "tlay"), (h2o, "h2o"), (o3, "o3")):
            dst[:, c0:c0 + n] = torch.from_numpy(cols[key]).to(dev).to(tdt)
        for key, dst in percol.items():
            dst[c0:c0 + n] = torch.from_numpy(cols[key]).to(dev).to(tdt)
    names = ["co2", "ch4", "n2o", "o2", "h2o", "o3"]
    gc = pkg.GasConcs(names)
    for name in names:
        if name in ("h2o", "o3"):
            gc.set_vmr(name, h2o if name == "h2o" else o3)
        elif name in percol:
            gc.set_vmr_column(name, percol[name])
        else:
            gc.set_vmr(name, 0.209)
    op = pkg.OpticalProps2str()
    op.alloc_2str(ncol, nlay, k, like=plev)
    toa = torch.empty((ng, ncol), **f64)
    nband = k.get_nband()
    alb = percol["albedo"].reshape(ncol, 1).expand(ncol, nband).contiguous()
    fl = pkg.FluxesBroadband(torch.empty((nlay + 1, ncol), **f64), torch.empty((nlay + 1, ncol), **f64))

    def step():
        e = k.gas_optics(None, plev, tlay, gc, op, toa)
        if e:
            raise SystemExit(e)
        e = pkg.rte_sw(op, True, percol["mu0"], toa, alb, alb, fl)
        if e:
            raise SystemExit(e)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    L.ecckd_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    L.ecckd_prof_enable(0)
    names_b = C.create_string_buffer(8 * 32)
    ms = (C.c_double * 8)()
    cnt = (C.c_longlong * 8)()
    nk = L.ecckd_prof_report(8, names_b, ms, cnt)
    kern = {names_b.raw[i * 32:(i + 1) * 32].split(b"\0")[0].decode(): ms[i] / max(cnt[i], 1) for i in range(nk)}
    cells = ncol * nlay * ng
    ms_per_step = elapsed / args.steps * 1e3
    alg = (48.0 * cells + 8.0 * ncol * ((nlay + 1) + 3 * nlay + 5 + 2 * ng + 2 * (nlay + 1))) * esz / 8
    # per kernel: gas optics writes tau, ssa, g (24 B/cell) + toa_src; the solver reads them
    # (24 B/cell) + toa, mu0, albedos and writes 2 x 61 fluxes.
    alg_k = {"tau": (24.0 * cells + 8.0 * ncol * ((nlay + 1) + 3 * nlay + 5 + ng)) * esz / 8,
             "rte_sw": (24.0 * cells + 8.0 * ncol * (ng + 3 + 2 * (nlay + 1))) * esz / 8}
    kernels = {n: {"avg_ms": v, "alg_bytes_per_launch": alg_k.get(n), "GBps": alg_k[n] / (v * 1e-3) / 1e9 if n in alg_k and v > 0 else None}
               for n, v in kern.items()}
    kern = {n: v for n, v in kern.items() if n in alg_k}
    dom = max(kern, key=kern.get)
    # HBM roofline of the dominant kernel, and beside it the roofline of what actually binds rte_sw: fp64 vector
    # arithmetic.  Counters (HBM bytes, fp64 instruction counts) come from the committed rocprofv3 --pmc passes of this build.
    # side measurement on the same resident inputs: the fused shortwave path (ecckd_sw_fluxes: 16 instead of 48 B per cell
    # between the kernels; the solver is bound by its serial sweeps, not by HBM: reported apart from the API roofline)
    ref_up, ref_dn = fl.flux_up.clone(), fl.flux_dn.clone()

    def fstep():
        e = k.sw_fluxes(plev, tlay, gc, True, percol["mu0"], alb, alb, fl)
        if e:
            raise SystemExit(e)
    fused_sw = None
    if fused:
        for _ in range(max(args.warmup, 1)):
            fstep()
        torch.cuda.synchronize()
        L.ecckd_prof_enable(1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fstep()
        torch.cuda.synchronize()
        ms_f = (time.perf_counter() - t0) / args.steps * 1e3
        L.ecckd_prof_enable(0)
        nk = L.ecckd_prof_report(8, names_b, ms, cnt)
        kf = {names_b.raw[i * 32:(i + 1) * 32].split(b"\0")[0].decode(): ms[i] / max(cnt[i], 1) for i in range(nk)}
    if fused:
        fused_sw = {"value": cells / (ms_f * 1e-3) / 1e6, "unit": "Mcol*lay*gpt/s", "ms_per_step": ms_f,
                "speedup_vs_api_pair": ms_per_step / ms_f, "kernels_avg_ms": kf, "hbm_bytes_per_cell_between_kernels": 16,
                "max_abs_flux_diff_vs_api_path_Wm2": max(float((fl.flux_up - ref_up).abs().max()), float((fl.flux_dn - ref_dn).abs().max())),
                "note": "ecckd_sw_fluxes: gas optics writes the total optical depth only, the solver derives ssa, g = 0 and the "
                        "incoming beam from plev and the model's tables (src/gas_optics_ecckd.f90:455-472); same inputs, same fluxes"}
    ctr, csrc = committed_counters("r03_pmc_sw.json", dom, ncol) if dtype == "f64" else (None, "counter passes were taken in fp64 only")
    roofline = {"kernel": dom, "bound": "hbm", "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (kernels[dom]["GBps"] or 0.0) / HBM_PEAK_GBS,
                "traffic": ctr.get("hbm_bytes_per_launch") if ctr else None, "traffic_source": csrc,
                "avg_launch_ms": kernels[dom]["avg_ms"], "alg_bytes_per_launch": kernels[dom]["alg_bytes_per_launch"],
                "note": "rte_sw (layer-systolic solver) is bound by the serial sweeps of the adding method -- one wave at a time "
                        "holds a tile's token and issues one fp64 instruction per ~10 clocks -- not by HBM: DESIGN.md 5.4"}
    valu_roof = None
    if ctr and "SQ_INSTS_VALU_FMA_F64" in ctr:
        secs = kernels[dom]["avg_ms"] * 1e-3
        flops = 64.0 * (2.0 * ctr["SQ_INSTS_VALU_FMA_F64"] + ctr.get("SQ_INSTS_VALU_MUL_F64", 0.0) + ctr.get("SQ_INSTS_VALU_ADD_F64", 0.0)
                        + ctr.get("SQ_INSTS_VALU_TRANS_F64", 0.0))
        valu_roof = {"kernel": dom, "bound": "fp64-valu", "achieved": flops / secs / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": flops / secs / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                     "valu_instructions_per_cell": ctr.get("SQ_INSTS_VALU", 0.0) * 64.0 / cells if ctr.get("SQ_INSTS_VALU") else None,
                     "valu_issue_frac_of_measured_peak": (ctr.get("SQ_INSTS_VALU", 0.0) / secs) / (256 * 0.86 * 2.4e9) if ctr.get("SQ_INSTS_VALU") else None,
                     "note": "fp64 FMA (2 flop), MUL, ADD, transcendental wave-instructions x 64 lanes from SQ_INSTS_VALU_*_F64; "
                             "measured sustained fp64 issue on this chip: 0.86 wave-instructions/clk/CU (tools/ubench.hip)",
                     "source": csrc}
    # spot check against the CPU oracle
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    m = oracle.CkdModel(sw_file)
    cols = synthetic.columns(0, 64, k.get_press_min(), shortwave=True)
    items = [(n, np.atleast_1d(np.asarray(cols[n], dtype=np.float64)), 0 if np.isscalar(cols[n]) else 1,
              0 if (np.isscalar(cols[n]) or cols[n].ndim == 1) else 64) for n in names]
    tau, ssa, g, toa_o, _ = oracle.gas_optics_ext(m, cols["plev"], cols["tlay"], items)
    a2 = np.repeat(cols["albedo"][None], ng, 0)
    fu, fd, _ = oracle.rte_sw(tau, ssa, g, cols["mu0"], toa_o, a2, a2)
    dflux = max(float(np.max(np.abs(fl.flux_up[:, :64].double().cpu().numpy() - fu))),
                float(np.max(np.abs(fl.flux_dn[:, :64].double().cpu().numpy() - fd))))
    return ({
        "metric": "Mcol*lay*gpt/s SW gas_optics+rte_sw", "value": cells * args.steps / elapsed / 1e6,
        "unit": "Mcol*lay*gpt/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": "synthetic %d columns x %d layers x %d g-points, SW wide-tol0.05, gas_optics + rte_sw "
                               "two-stream, %s (BASELINE configs[2]%s)" % (ncol, nlay, ng, "fp64" if dtype == "f64" else "fp32",
                                                                             "" if dtype == "f64" else " in single precision"),
                   "solver_options": pkg.solver_options()},
        "roofline": roofline, "roofline_fp64_valu": valu_roof,
        "roofline_pipeline": {"bound": "hbm", "achieved": alg / (ms_per_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "kernels": kernels, "fused_sw": fused_sw, "check_max_abs_flux_diff_vs_oracle_Wm2": dflux, "cpu_baseline": None})


if __name__ == "__main__":
    main()
