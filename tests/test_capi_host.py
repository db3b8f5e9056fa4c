"""CPU tests of the host side: the C ABI library loads and exports everything the header
declares, load() (= load_and_init) agrees with the Python restatement of the reference's loader,
error behaviour, and -- the important one -- nothing computes without a GPU."""
import ctypes as C
import os

import numpy as np
import pytest

import __graft_entry__ as entry
from conftest import LW_FSCK, LW_RRTMGP, SW_WIDE
from rte_ecckd_amd import synthetic


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    syms = entry.exported_symbols()
    assert "ecckd_gas_optics_lw" in syms and "ecckd_rte_sw" in syms and len(syms) >= 28
    for s in syms:
        assert hasattr(L, s), s
    assert b"gfx950" in L.ecckd_build_info()
    # the second library: RTE-RRTMGP's kernel-level bind(C) names (include/rte_kernels_hip.h)
    K = C.CDLL(pkg.RTE_KERNELS_LIB)
    names = entry.rte_kernel_symbols()
    assert names == ["lw_solver_noscat_GaussQuad", "net_broadband_precalc", "sum_broadband", "sw_solver_2stream"]
    for s in names:
        assert hasattr(K, s), s


@pytest.mark.parametrize("path", [LW_FSCK, LW_RRTMGP, SW_WIDE])
def test_load_matches_reference_loader_restatement(pkg, oracle_mod, path):
    """Own CDF-1 reader + load_and_init (C++) vs scipy + the Python restatement."""
    k = pkg.GasOpticsEcckd()
    assert k.load(path, device=-1) == ""          # host-only model: no GPU needed for getters
    m = oracle_mod.CkdModel(path)
    assert k.get_ngpt() == m.ng
    assert k.get_ngas() == m.num_gases and k.get_gases() == m.gas
    assert k.get_nband() == m.band2gpt.shape[0]
    assert np.array_equal(k.get_band2gpt(), m.band2gpt)
    assert np.array_equal(k.get_band_lims_wavenumber(), m.band_lims_wvn)
    assert k.source_is_internal() == (not m.shortwave) and k.source_is_external() == m.shortwave
    assert k.get_press_min() == float(np.exp(m.log_pressure[0]))
    assert k.get_press_max() == float(np.exp(m.log_pressure[-1]))
    assert k.get_temp_min() == float(m.temperature.min()) and k.get_temp_max() == float(m.temperature.max())
    if m.shortwave:
        assert k.get_total_solar_irradiance() == m.total_solar_irradiance
    k.finalize()


def test_load_errors(pkg, tmp_path):
    k = pkg.GasOpticsEcckd()
    assert "can't open file" in k.load(str(tmp_path / "nope.nc"), device=-1)
    bad = tmp_path / "bad.nc"
    bad.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    assert "not a netCDF-3" in k.load(str(bad), device=-1)
    trunc = tmp_path / "trunc.nc"
    trunc.write_bytes(open(LW_FSCK, "rb").read()[:3000])
    assert k.load(str(trunc), device=-1) != ""


def test_cdf2_and_small_model_roundtrip(pkg, oracle_mod, tmp_path):
    """A tiny ecCKD-style file written as 64-bit-offset netCDF (CDF-2) goes through the same loader."""
    from scipy.io import netcdf_file
    p = str(tmp_path / "tiny.nc")
    f = netcdf_file(p, "w", version=2)
    for n, d in (("temperature", 3), ("pressure", 4), ("g_point", 2), ("temperature_planck", 5),
                 ("wavenumber", 3), ("band", 1), ("x_mole_fraction", 2)):
        f.createDimension(n, d)
    v = f.createVariable("pressure", "f4", ("pressure",)); v[:] = [10., 100., 1000., 10000.]
    v = f.createVariable("temperature", "f4", ("temperature", "pressure")); v[:] = np.arange(12).reshape(3, 4) + 200
    v = f.createVariable("temperature_planck", "f4", ("temperature_planck",)); v[:] = [100, 150, 200, 250, 300]
    v = f.createVariable("planck_function", "f4", ("temperature_planck", "g_point")); v[:] = np.arange(10).reshape(5, 2)
    v = f.createVariable("gpoint_fraction", "f4", ("g_point", "wavenumber")); v[:] = 0.5
    v = f.createVariable("wavenumber1_band", "f4", ("band",)); v[:] = [0.]
    v = f.createVariable("wavenumber2_band", "f4", ("band",)); v[:] = [3000.]
    v = f.createVariable("band_number", "i2", ("g_point",)); v[:] = [0, 0]
    v = f.createVariable("x_mole_fraction", "f4", ("x_mole_fraction",)); v[:] = [1e-6, 1e-5]
    v = f.createVariable("x_molar_absorption_coeff", "f4", ("x_mole_fraction", "temperature", "pressure", "g_point"))
    v[:] = np.arange(48).reshape(2, 3, 4, 2)
    v = f.createVariable("y_conc_dependence_code", "i2", ()); v.data[...] = 3
    v = f.createVariable("y_reference_mole_fraction", "f4", ()); v.data[...] = 4e-7
    v = f.createVariable("y_molar_absorption_coeff", "f4", ("temperature", "pressure", "g_point")); v[:] = 1.0
    f.constituent_id = "x yy"       # "yy" does not exist ...
    f.close()
    k = pkg.GasOpticsEcckd()
    assert "yy" in k.load(p, device=-1)        # ... so the loader reports the missing variable
    f = netcdf_file(p, "a"); f.constituent_id = "x y z"; f.close()   # last single-char token is dropped
    assert open(p, "rb").read(4) == b"CDF\x02"
    assert k.load(p, device=-1) == ""
    assert k.get_gases() == ["x", "y"] and k.get_ngpt() == 2 and k.get_nband() == 1
    assert oracle_mod.CkdModel(p).gas == ["x", "y"]


def _write_tiny(path, np_=4, nt=3, ng=2, ntp=5, nband=1, ngb=None, planck_g=None, t_shape=None, mf=(1e-6, 1e-5),
                band_number=None, p_values=None):
    """A tiny ecCKD-style file with knobs for every extent the kernels index (test_malformed_tables...)."""
    from scipy.io import netcdf_file
    ngb = ng if ngb is None else ngb
    planck_g = ng if planck_g is None else planck_g
    f = netcdf_file(path, "w")
    dims = dict(temperature=nt, pressure=np_, g_point=ng, temperature_planck=ntp, wavenumber=3, x_mole_fraction=len(mf),
                gb=ngb, pg=planck_g)
    if nband > 0:
        dims["band"] = nband
    for n, d in dims.items():
        f.createDimension(n, d)
    if t_shape is not None:
        f.createDimension("tt", t_shape[0]); f.createDimension("tp", t_shape[1])
    v = f.createVariable("pressure", "f4", ("pressure",))
    v[:] = p_values if p_values is not None else 10.0 ** (1 + np.arange(np_))
    tdims = ("tt", "tp") if t_shape is not None else ("temperature", "pressure")
    v = f.createVariable("temperature", "f4", tdims)
    shp = t_shape if t_shape is not None else (nt, np_)
    v[:] = 200 + 20 * np.arange(shp[0])[:, None] + np.arange(shp[1])[None, :]
    v = f.createVariable("temperature_planck", "f4", ("temperature_planck",)); v[:] = 100 + 50 * np.arange(ntp)
    v = f.createVariable("planck_function", "f4", ("temperature_planck", "pg")); v[:] = 1.0
    v = f.createVariable("gpoint_fraction", "f4", ("g_point", "wavenumber")); v[:] = 0.5
    if nband > 0:
        v = f.createVariable("wavenumber1_band", "f4", ("band",)); v[:] = np.arange(nband) * 100.
        v = f.createVariable("wavenumber2_band", "f4", ("band",)); v[:] = (np.arange(nband) + 1) * 100.
    v = f.createVariable("band_number", "i2", ("gb",)); v[:] = band_number if band_number is not None else np.zeros(ngb)
    v = f.createVariable("x_mole_fraction", "f4", ("x_mole_fraction",)); v[:] = mf
    v = f.createVariable("x_molar_absorption_coeff", "f4", ("x_mole_fraction", "temperature", "pressure", "g_point")); v[:] = 1.0
    f.constituent_id = "x z"     # (the single-character last token is dropped, as in the reference)
    f.close()


def test_malformed_tables_are_load_errors(pkg, tmp_path):
    """Extents the kernels index without a bounds check are validated at load / finalize time:
    a short or inconsistent table is an error message, never an out-of-bounds read on the host or the device."""
    k = pkg.GasOpticsEcckd()
    p = str(tmp_path / "m.nc")
    _write_tiny(p)
    assert k.load(p, device=-1) == ""                      # the well-formed form of the same file loads
    cases = [(dict(np_=1), "pressure grid"), (dict(nt=1), "temperature grid"), (dict(ntp=1), "temperature_planck"),
             (dict(planck_g=1), "planck_function"), (dict(ngb=1), "band_number"), (dict(ngb=3), "band_number"),
             (dict(t_shape=(3, 5)), "temperature"), (dict(mf=(1e-5, 1e-6)), "mole fractions"),
             (dict(mf=(0.0, 1e-6)), "mole fractions"), (dict(nband=2, band_number=[0, 0]), "band"),
             (dict(band_number=[0, 5]), "band"), (dict(p_values=[10., 100., -1., 1e4]), "pressure")]
    for knobs, word in cases:
        _write_tiny(p, **knobs)
        err = k.load(p, device=-1)
        assert err != "" and word in err, (knobs, err)
    # the builder route goes through the same checks at finalize
    lp = np.log([10., 100., 1000.]); T = 200. + np.arange(6).reshape(2, 3)
    coef = np.ones((2, 3, 2))
    bad = k.init_from_tables(lp, T, [dict(name="x", code=1, coefficient=coef)], planck=(np.array([100., 90.]), np.ones((2, 2))), device=-1)
    assert "temperature_planck" in bad
    ok = k.init_from_tables(lp, T, [dict(name="x", code=1, coefficient=coef)], planck=(np.array([100., 200.]), np.ones((2, 2))), device=-1)
    assert ok == ""


def test_no_gpu_means_error_not_fallback(pkg):
    """On a machine without a GPU every compute entry point must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this is the no-GPU behaviour test")
    k = pkg.GasOpticsEcckd()
    err = k.load(LW_FSCK, device=0)
    assert "no HIP device" in err and "no CPU fallback" in err
    assert k.load(LW_FSCK, device=-1) == ""
    nlay, ncol, ng = 60, 4, 32
    gc = pkg.GasConcs(["h2o"]); gc.set_vmr("h2o", 1e-3)
    op = pkg.OpticalProps1scl(); op.alloc_1scl(ncol, nlay, k)
    src = pkg.SourceFuncLW(); src.alloc(ncol, nlay, k)
    op.tau[:] = -7.0
    err = k.gas_optics(None, np.full((nlay + 1, ncol), 1e4), np.full((nlay, ncol), 250.), np.full(ncol, 280.),
                       gc, op, src, tlev=np.full((nlay + 1, ncol), 250.))
    assert "no CPU fallback" in err
    assert np.all(op.tau == -7.0)             # untouched: nothing was computed anywhere
    fl = pkg.FluxesBroadband(np.zeros((nlay + 1, ncol)), np.zeros((nlay + 1, ncol)))
    assert "no HIP device" in pkg.rte_lw(op, True, src, np.ones((ncol, 1)), fl)


def test_round3_entry_points_without_a_gpu(pkg):
    """The entry points and switches added in round 3 on a host-only model / without a GPU: the fused shortwave path and the
    single-precision shortwave pair fail loudly (no CPU fallback), the new options check their values, the tail-scratch
    queries answer 0 where no device can be asked for its size."""
    import torch
    from conftest import SW_WIDE
    k = pkg.GasOpticsEcckd()
    assert k.load(SW_WIDE, device=-1) == ""
    nlay, ncol, ng = 60, 4, k.get_ngpt()
    gc = pkg.GasConcs(["h2o"]); gc.set_vmr("h2o", 1e-3)
    for dt in (np.float64, np.float32):
        fl = pkg.FluxesBroadband(np.zeros((nlay + 1, ncol), dtype=dt), np.zeros((nlay + 1, ncol), dtype=dt))
        msg = k.sw_fluxes(np.full((nlay + 1, ncol), 1e4, dtype=dt), np.full((nlay, ncol), 250., dtype=dt), gc, True,
                          np.full(ncol, 0.5, dtype=dt), np.full((ncol, k.get_nband()), 0.1, dtype=dt),
                          np.full((ncol, k.get_nband()), 0.1, dtype=dt), fl)
        assert "no CPU fallback" in msg and np.all(fl.flux_up == 0)
        op = pkg.OpticalProps2str(); op.alloc_2str(ncol, nlay, k, like=np.empty(0, dtype=dt))
        assert op.tau.dtype == dt
        if not torch.cuda.is_available():
            msg = pkg.rte_sw(op, True, np.full(ncol, 0.5, dtype=dt), np.ones((ng, ncol), dtype=dt),
                             np.full((ncol, k.get_nband()), 0.1, dtype=dt), np.full((ncol, k.get_nband()), 0.1, dtype=dt), fl)
            assert "no HIP device" in msg
    for name, bad in (("gas_slab_f32", 3), ("sw_solver", 2)):
        with pytest.raises(ValueError):
            pkg.set_solver_option(name, bad)
    assert pkg.get_solver_option("gas_slab_f32") == 2 and pkg.get_solver_option("sw_solver") == 0
    assert set(("sw_solver", "gas_slab_f32")) <= set(pkg.solver_options())
    if not torch.cuda.is_available():
        assert pkg.rte_sw_tail_scratch_bytes(1000, 60, 27) == 0 and pkg.rte_lw_tail_scratch_bytes(100000, 60, 32) == 0


def test_launch_plan_host_logic(pkg, oracle_mod, monkeypatch):
    """ecckd_gas_optics_plan: the host-side decisions of gas_optics (fused or not, slab rows, Planck
    window, pass splitting, grid) on host-only models -- no GPU, nothing launched."""
    gases = synthetic.GAS_ORDER
    k32 = pkg.GasOpticsEcckd(); assert k32.load(LW_FSCK, device=-1) == ""
    p = k32.plan(1000000, 60, gases)
    assert p["passes"] == 1 and p["fused"] == 1 and p["planck_fused"] == 1
    assert p["planck_rows"] == 231                      # whole Planck table next to the slab
    assert p["slab_rows"] >= 3 and p["lds_bytes"] <= 160 * 1024 and p["g_chunk"] == 8
    assert (p["col_chunks"] * 60) % 256 == 0            # full rounds of the 256 CUs
    small = k32.plan(100, 60, gases)
    assert small["col_chunks"] == 1
    # every gas an array: seven bilinear slots, nothing merged; RFMIP's description (well-mixed gases as scalars): the
    # five of them and the composite share one slot next to o3, and more pressure rows fit
    assert p["slots"] == 7 and p["merged"] == 0
    pm = k32.plan(1000000, 60, gases, scalar_gases=["co2", "ch4", "n2o", "o2", "cfc11", "cfc12"])
    assert pm["slots"] == 2 and pm["merged"] == 6 and pm["slab_rows"] > p["slab_rows"] and pm["g_chunk"] == 8
    assert k32.plan(1000, 60, gases, scalar_gases=["co2"])["merged"] == 2        # co2 + the composite (o2 is an array: 1.0 * table)
    pkg.set_solver_option("gas_merge_scalars", 0)
    try:
        assert k32.plan(1000, 60, gases, scalar_gases=["co2", "ch4"])["merged"] == 0
    finally:
        pkg.set_solver_option("gas_merge_scalars", 1)
    # fp32 halves the table bytes: more pressure rows fit
    assert k32.plan(1000000, 60, gases, single_precision=True)["slab_rows"] > p["slab_rows"]
    # the 36-g file: the whole table does not fit next to 3 rows -> a window of it
    k36 = pkg.GasOpticsEcckd(); assert k36.load(LW_RRTMGP, device=-1) == ""
    q = k36.plan(1000000, 60, gases)
    assert q["fused"] == 1 and q["planck_fused"] == 1 and 16 <= q["planck_rows"] < 231 and q["slab_rows"] >= 3
    assert q["g_chunk"] == 4                            # 36 is a multiple of 4, not of 8
    monkeypatch.setenv("ECCKD_PLANCK_WINDOW", "16")
    assert k32.plan(1000, 60, gases)["planck_rows"] == 16
    monkeypatch.delenv("ECCKD_PLANCK_WINDOW")
    # shortwave: fused tau + Rayleigh epilogue, no Planck
    ksw = pkg.GasOpticsEcckd(); assert ksw.load(SW_WIDE, device=-1) == ""
    r = ksw.plan(100000, 60, gases)
    assert r["fused"] == 1 and r["planck_fused"] == 0 and r["planck_rows"] == 0 and r["g_chunk"] == 4
    # reference-order arithmetic: the per-gas kernels, nothing fused; single precision is refused there
    pkg.set_arithmetic(pkg.REFERENCE_ORDER)
    try:
        assert k32.plan(1000, 60, gases)["fused"] == 0
        with pytest.raises(RuntimeError, match="single precision"):
            k32.plan(1000, 60, gases, single_precision=True)
    finally:
        pkg.set_arithmetic(pkg.FAST)
    # unknown gases are skipped, an empty list is still one (empty) pass; oversize calls are refused
    assert k32.plan(1000, 60, ["no2", "xyz"])["passes"] == 1
    with pytest.raises(RuntimeError, match="split the column range"):
        k32.plan(9_000_000, 60, gases)
    # two look_up_table gases cannot share a pass
    m = oracle_mod.CkdModel(LW_FSCK)
    tabs = [dict(name=n, code=t["code"], composite_only=0, mole_fraction=t["mole_fraction"],
                 reference_mole_fraction=t["reference_mole_fraction"],
                 coefficient=t["coefficient"] if t["code"] == 2 else t["coefficient"][0])
            for n, t in zip(m.gas[:3], m.tables[:3])]
    tabs.append(dict(name="h2o_b", code=2, composite_only=0, mole_fraction=m.tables[0]["mole_fraction"] * 0.5,
                     reference_mole_fraction=0.0, coefficient=m.tables[0]["coefficient"] * 0.25))
    k2 = pkg.GasOpticsEcckd()
    assert k2.init_from_tables(m.log_pressure, m.temperature, tabs,
                               planck=(m.temperature_planck, m.planck_function), device=-1) == ""
    two = k2.plan(5000, 60, [t["name"] for t in tabs])
    assert two["passes"] == 2 and two["planck_fused"] == 1


def test_gas_concs_mirror(pkg):
    gc = pkg.GasConcs()
    assert gc.init(["H2O ", "co2"]) == ""
    assert gc.get_gas_names() == ["h2o", "co2"] and gc.get_num_gases() == 2
    assert "name not provided" in gc.set_vmr("o3", 1e-6)
    assert "should be >= 0" in gc.set_vmr("co2", -1.0)
    assert gc.set_vmr("co2", 4e-4) == ""
    with pytest.raises(KeyError):
        gc.entries(3, 2)                       # h2o never set -> get_vmr error
    assert gc.set_vmr("h2o", np.full((2, 3), 1e-3)) == ""
    e = gc.entries(3, 2)
    assert e[0][2:4] == (1, 3) and e[1][1] is None and e[1][4] == 4e-4
    assert "duplicate" in gc.init(["a", "a"])
    k = pkg.GasOpticsEcckd()
    assert k.load(LW_FSCK, device=-1) == ""
    g2 = pkg.GasConcs(["h2o", "co2"]); g2.set_vmr("co2", 4e-4)
    op = pkg.OpticalProps1scl(); op.alloc_1scl(3, 2, k)
    src = pkg.SourceFuncLW(); src.alloc(3, 2, k)
    msg = k.gas_optics(None, np.ones((3, 3)), np.ones((2, 3)), np.ones(3), g2, op, src, tlev=np.ones((3, 3)))
    assert msg == "ty_gas_concs%get_vmr; gas h2o not found"     # get_vmr error is passed through (:351-354)


def test_bench_default_shard_size_for_8_gpus():
    """bench.py --gpus 8 runs BASELINE configs[3]: 1e7 columns in total, 1.25e6 per rank; fewer GPUs weak-scale the
    1e6-column headline."""
    import bench
    assert bench.columns_per_gpu(8, None) == 1250000 and bench.columns_per_gpu(1, None) == 1000000
    assert bench.columns_per_gpu(2, None) == 1000000 and bench.columns_per_gpu(4, 300000) == 300000


def test_code_object_resources(pkg):
    """Register allocation of the built gfx950 code objects, read from their own metadata (tools/kernel_resources.py; no
    GPU needed): the instantiations the BASELINE workloads take keep everything in registers."""
    import re
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import kernel_resources
    ks = kernel_resources.kernels(pkg.LIB_PATH)
    assert len(ks) > 60
    seen = set()
    for name, k in ks.items():
        m = re.search(r"(gas_fused_kernel|rte_lw_kernel|rte_lw_split_kernel|rte_sw_kernel|tau_kernel)<([^>]*)>", name)
        if not m:
            continue
        kind, targs = m.group(1), [a.strip() for a in m.group(2).split(",")]
        seen.add(kind)
        if kind == "gas_fused_kernel" and targs[7] == "512":
            assert targs[5] == "1", name                  # the longwave mode keeps two waves per SIMD
        if kind == "gas_fused_kernel" and targs[3] == "true" and targs[7] == "512":
            # FULL: g-point count a multiple of the chunk -- both longwave tables (32, 36 g-points), 5 / 7 / 10 gas slots,
            # with or without per-g-point clamping, every mode and precision
            assert k["spill_vgpr"] == 0, name
        if kind == "gas_fused_kernel" and targs[7] != "512":
            # shortwave and tau-only modes: blocks of 768 threads, three waves per SIMD under 168 VGPRs; the shape the
            # 27-g-point file takes (5 slots, chunks of 4, ragged last chunk) spills 6 registers outside the loops in fp64
            assert targs[5] in ("0", "2") and targs[7] == "768", name
            assert kernel_resources.waves_per_simd(k) == 3 and k["spill_vgpr"] <= 32, name
            if targs[0] == "float" and targs[4] == "false":       # (tables without negative entries: all ecCKD files)
                assert k["spill_vgpr"] == 0, name
            if targs[0] == "double" and targs[1:6] == ["4", "5", "false", "false", "2"]:
                assert k["spill_vgpr"] <= 8, name
        if kind == "rte_lw_kernel" or kind == "rte_sw_kernel":
            assert k["spill_vgpr"] == 0, name
        if kind == "rte_lw_split_kernel" and targs[0] in ("10", "12") and targs[5] == "false":
            assert k["spill_vgpr"] == 0, name
        if kind == "rte_lw_split_kernel" and targs[5] == "true":      # Planck-recomputing form: two waves per SIMD, no spill
            assert targs[0] == "15" and targs[6] == "2" and k["spill_vgpr"] == 0, name
    assert seen == {"gas_fused_kernel", "rte_lw_kernel", "rte_lw_split_kernel", "rte_sw_kernel", "tau_kernel"}
    # the headline instantiation, over the fp64 slab and over the float32 image of the tables ("gas_slab_f32")
    head = [k for n, k in ks.items() if "gas_fused_kernel<double, 8, 7, true, false, 1, double, 512>" in n]
    assert len(head) == 1 and head[0]["vgpr"] <= 256 and kernel_resources.waves_per_simd(head[0]) == 2
    head32 = [k for n, k in ks.items() if "gas_fused_kernel<double, 8, 7, true, false, 1, float, 512>" in n]
    assert len(head32) == 1 and head32[0]["spill_vgpr"] == 0 and kernel_resources.waves_per_simd(head32[0]) == 2
    sys_sw = [k for n, k in ks.items() if "rte_sw_sys_kernel<double, true, false, false, true>" in n]
    assert len(sys_sw) == 1 and kernel_resources.waves_per_simd(sys_sw[0]) == 3     # 12 waves per block, one block per CU
    sw = [k for n, k in ks.items() if "rte_sw_kernel<16, true, true, false>" in n]
    assert len(sw) == 1 and kernel_resources.waves_per_simd(sw[0]) == 3       # 143 VGPRs: 12 waves per CU
