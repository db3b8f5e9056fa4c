"""GPU parity tests added in round 2 (all through the C ABI, against the CPU oracle or against properties):

  * node-identity known answers from the reference-held LUT files, on the HIP path;
  * a NaN in one column stays in that column;
  * incident-flux boundary condition of rte_lw and the solver version switches, HIP vs oracle;
  * every ECCKD_DEVICE solver call is asynchronous: rte_sw and rte_lw with 137 layers captured in HIP graphs,
    two streams running concurrently;
  * BASELINE configs[4]: the 36-g-point table in fp32 against the fp64 oracle (tolerance sweep);
  * BASELINE configs[3]: the 1.25e6-column per-rank shard of the 1e7-column job on one GPU.

Tolerances as in test_gpu_parity.py: Planck sources bit-identical, tau 1e-12 relative, fluxes 1e-9 W m-2
(north_star: 1e-6)."""
import numpy as np
import pytest

import helpers
from conftest import LW_FSCK, LW_RRTMGP, SW_WIDE
from rte_ecckd_amd import synthetic

pytestmark = pytest.mark.gpu
TAU_RTOL = 1e-12
FLUX_ATOL = 1e-9


@pytest.fixture(scope="module")
def lw(pkg, gpu, oracle_mod):
    k = pkg.GasOpticsEcckd()
    assert k.load(LW_FSCK, device=0) == ""
    return k, oracle_mod.CkdModel(LW_FSCK)


@pytest.fixture(params=["fast", "reference_order"])
def arithmetic(request, pkg):
    pkg.set_arithmetic(pkg.FAST if request.param == "fast" else pkg.REFERENCE_ORDER)
    yield request.param
    pkg.set_arithmetic(pkg.FAST)


@pytest.fixture(autouse=True)
def default_options(pkg):
    pkg.reset_solver_options()
    yield
    pkg.reset_solver_options()


def T(gpu):
    import torch
    return lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


def lw_objects(pkg, gpu, tau, lay, inc, dec, sfc, b2g=None):
    import torch
    t = T(gpu)
    ng, nlay, ncol = tau.shape
    op = pkg.OpticalProps1scl(); op.tau = t(tau)
    op.band2gpt = np.array([[1, ng]], dtype=np.int32) if b2g is None else b2g
    src = pkg.SourceFuncLW()
    src.lay_source, src.lev_source_inc, src.lev_source_dec, src.sfc_source = t(lay), t(inc), t(dec), t(sfc)
    fl = pkg.FluxesBroadband(torch.zeros((nlay + 1, ncol), dtype=torch.float64, device=gpu),
                             torch.zeros((nlay + 1, ncol), dtype=torch.float64, device=gpu))
    return op, src, fl


# ------------------------------------------------------------------------------------------------
# node identities -- "parity unpinned" still holds: these tie the HIP path to numbers
# the reference's own data files hold, not to a run of the reference code
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", [LW_FSCK, LW_RRTMGP])
def test_planck_sources_on_table_nodes(pkg, gpu, oracle_mod, path, arithmetic):
    k = pkg.GasOpticsEcckd()
    assert k.load(path, device=0) == ""
    m = oracle_mod.CkdModel(path)
    cols, want = helpers.planck_node_case(m)
    err, tau, lay, inc, dec, sfc = helpers.run_lw_gas_optics(pkg, k, cols, gpu, names=[])
    assert err == ""
    assert np.array_equal(sfc, want)
    for a in (lay, inc, dec):
        assert np.array_equal(a, np.repeat(want[:, None, :], 3, 1))
    assert np.all(tau == 0)      # empty gas list: tau is zeroed (:346)


def test_tau_on_table_nodes(pkg, gpu, oracle_mod, arithmetic):
    """tau == weight * coefficient(:,ip,it) on exact table nodes.  Reference-order arithmetic: bit for bit
    (weights are exactly 1 and 0); fast arithmetic multiplies the weights out first -- still exact here."""
    import torch
    t = T(gpu)
    k = pkg.GasOpticsEcckd()
    assert k.load(LW_FSCK, device=0) == ""
    m = oracle_mod.CkdModel(LW_FSCK)
    for name, cols, item, want in helpers.tau_node_cases(m):
        err, tau = helpers.run_lw_gas_optics(pkg, k, cols, gpu, names=[name], overrides={name: float(item[1][0])})[:2]
        assert err == ""
        if name == "h2o":    # the device log() of the first mole-fraction node may sit one ulp off the host's
            assert helpers.max_rel(tau, want) < 1e-13, name
        else:
            assert np.array_equal(tau, want), name
    ks = pkg.GasOpticsEcckd()
    assert ks.load(SW_WIDE, device=0) == ""
    ms = oracle_mod.CkdModel(SW_WIDE)
    for name, cols, item, want in helpers.tau_node_cases(ms):
        ncol = cols["plev"].shape[1]
        gc = pkg.GasConcs([name]); gc.set_vmr(name, float(item[1][0]))
        op = pkg.OpticalProps2str(); op.alloc_2str(ncol, 1, ks, like=t(np.zeros(1)))
        toa = torch.empty((ms.ng, ncol), dtype=torch.float64, device=gpu)
        assert ks.gas_optics(None, t(cols["plev"]), t(cols["tlay"]), gc, op, toa) == ""
        ray = helpers.GLOBAL_WEIGHT * (cols["plev"][1] - cols["plev"][0])[None, None, :] * ms.rayleigh[:, None, None]
        if name == "h2o":
            assert helpers.max_rel(op.tau.cpu().numpy(), want + ray) < 1e-13, name
        else:
            assert np.array_equal(op.tau.cpu().numpy(), want + ray), name
        assert np.array_equal(toa.cpu().numpy(), np.repeat(ms.solar_irradiance[:, None], ncol, 1))


def test_nan_stays_in_its_own_column(pkg, gpu, oracle_mod, lw, arithmetic):
    """a NaN in one column's inputs (pressure, temperature, a gas array -- or the one element the
    unused gas slots used to read) must poison that column only, and no column may be left unwritten."""
    k, m = lw
    ncol = 700
    cols = synthetic.columns(11, ncol, k.get_press_min())
    cols = {n: (v.copy() if isinstance(v, np.ndarray) else v) for n, v in cols.items()}
    bad = {0: "plev", 65: "plev", 130: "tlay", 257: "h2o", 300: "co2", 699: "plev"}
    cols["plev"][7, 0] = np.nan          # (level 8 of column 0: the element every dummy slot load used to read at j = 7)
    cols["plev"][:, 65] = np.nan
    cols["tlay"][3, 130] = np.nan
    cols["h2o"][40, 257] = np.nan
    cols["co2"][300] = np.nan
    cols["plev"][60, 699] = np.nan
    err, tau, lay, inc, dec, sfc = helpers.run_lw_gas_optics(pkg, k, cols, gpu)
    assert err == ""
    otau, olay, oinc, odec, osfc, _ = oracle_mod.gas_optics_int(m, cols["plev"], cols["tlay"], cols["tsfc"],
                                                               helpers.oracle_gas_items(cols), cols["tlev"])
    good = np.ones(ncol, dtype=bool)
    good[list(bad)] = False
    assert np.all(np.isfinite(tau[..., good])) and np.all(np.isfinite(lay[..., good]))
    assert helpers.max_rel(tau[..., good], otau[..., good]) < TAU_RTOL
    assert np.array_equal(lay[..., good], olay[..., good]) and np.array_equal(inc[..., good], oinc[..., good])
    assert np.array_equal(np.isnan(tau), np.isnan(otau))           # NaN exactly where the oracle has it
    assert np.array_equal(np.isnan(lay), np.isnan(olay))
    ok = ~np.isnan(otau)
    assert helpers.max_rel(tau[ok], otau[ok]) < TAU_RTOL


def test_unknown_gas_needs_no_concentration(pkg, gpu, oracle_mod, lw):
    """The reference only consults get_vmr for gases of the k-distribution (:348-364): a gas_desc naming an
    unknown gas that was never set is not an error."""
    k, m = lw
    cols = synthetic.columns(0, 70, k.get_press_min())
    t = T(gpu)
    gc = pkg.GasConcs(["co2", "xyz", "h2o"])
    assert gc.set_vmr("co2", 4e-4) == "" and gc.set_vmr("h2o", t(cols["h2o"])) == ""     # "xyz" never set
    op = pkg.OpticalProps1scl(); op.alloc_1scl(70, 60, k, like=t(np.zeros(1)))
    src = pkg.SourceFuncLW(); src.alloc(70, 60, k, like=t(np.zeros(1)))
    assert k.gas_optics(None, t(cols["plev"]), t(cols["tlay"]), t(cols["tsfc"]), gc, op, src, tlev=t(cols["tlev"])) == ""
    otau = oracle_mod.gas_optics_int(m, cols["plev"], cols["tlay"], cols["tsfc"],
                                     [("co2", np.array([4e-4]), 0, 0), ("h2o", cols["h2o"], 1, 70)], cols["tlev"])[0]
    assert helpers.max_rel(op.tau.cpu().numpy(), otau) < TAU_RTOL
    gc2 = pkg.GasConcs(["co2", "h2o"]); gc2.set_vmr("co2", 4e-4)                          # a KNOWN gas that was never set
    assert "h2o" in k.gas_optics(None, t(cols["plev"]), t(cols["tlay"]), t(cols["tsfc"]), gc2, op, src, tlev=t(cols["tlev"]))


# ------------------------------------------------------------------------------------------------
# incident flux + solver switches
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nlay,top_at_1,nmus", [(60, True, 1), (60, False, 3), (33, True, 2), (137, False, 1)])
def test_rte_lw_incident_flux(pkg, gpu, oracle_mod, nlay, top_at_1, nmus):
    rng = np.random.default_rng(nlay)
    ng, ncol = 6, 130
    tau = rng.uniform(0, 2, (ng, nlay, ncol)) * rng.choice([1e-9, 1e-3, 1.0], size=(ng, nlay, ncol))
    lay, inc, dec = (rng.uniform(1, 9, (ng, nlay, ncol)) for _ in range(3))
    sfc = rng.uniform(1, 9, (ng, ncol)); emis = rng.uniform(0.7, 1.0, (ncol, 1))
    incf = rng.uniform(0, 30, (ng, ncol))
    t = T(gpu)
    op, src, fl = lw_objects(pkg, gpu, tau, lay, inc, dec, sfc)
    emis_gpt = np.repeat(emis.T, ng, 0)
    for iso in (0, 1):
        pkg.set_solver_option("lw_inc_flux_isotropic", iso)
        assert pkg.rte_lw(op, top_at_1, src, t(emis), fl, n_gauss_angles=nmus, inc_flux=t(incf)) == ""
        fu, fd = oracle_mod.rte_lw(tau, lay, inc, dec, emis_gpt, sfc, top_at_1=top_at_1, nmus=nmus, inc_flux=incf,
                                   options=oracle_mod.solver_options(lw_inc_flux_isotropic=iso))
        assert np.max(np.abs(fl.flux_up.cpu().numpy() - fu)) < FLUX_ATOL
        assert np.max(np.abs(fl.flux_dn.cpu().numpy() - fd)) < FLUX_ATOL
    top = 0 if top_at_1 else nlay
    assert np.allclose(fl.flux_dn.cpu().numpy()[top], incf.sum(0), rtol=1e-9)        # isotropic: the flux comes back
    # host memory space: the same numbers; NULL inc_flux: the plain solver
    op_h = pkg.OpticalProps1scl(); op_h.tau = tau; op_h.band2gpt = op.band2gpt
    s_h = pkg.SourceFuncLW(); s_h.lay_source, s_h.lev_source_inc, s_h.lev_source_dec, s_h.sfc_source = lay, inc, dec, sfc
    fl_h = pkg.FluxesBroadband(np.empty((nlay + 1, ncol)), np.empty((nlay + 1, ncol)))
    assert pkg.rte_lw(op_h, top_at_1, s_h, np.ascontiguousarray(emis), fl_h, n_gauss_angles=nmus, inc_flux=incf) == ""
    assert np.array_equal(fl_h.flux_dn, fl.flux_dn.cpu().numpy())
    # analytic: transparent column, F_dn = sum_g inc_flux at every level
    z = np.zeros_like(tau)
    opz, srcz, flz = lw_objects(pkg, gpu, z, z, z, z, np.zeros((ng, ncol)))
    pkg.reset_solver_options()
    assert pkg.rte_lw(opz, top_at_1, srcz, t(np.ones((ncol, 1))), flz, n_gauss_angles=1, inc_flux=t(incf)) == ""
    assert np.allclose(flz.flux_dn.cpu().numpy(), incf.sum(0)[None, :], rtol=1e-14)
    assert np.all(flz.flux_up.cpu().numpy() == 0)


def test_lw_solver_switches(pkg, gpu, oracle_mod):
    rng = np.random.default_rng(5)
    ng, nlay, ncol = 5, 60, 200
    tau = rng.uniform(0, 1, (ng, nlay, ncol)) * rng.choice([1e-9, 1e-5, 1e-3, 1.0], size=(ng, nlay, ncol))
    lay, inc, dec = (rng.uniform(1, 9, (ng, nlay, ncol)) for _ in range(3))
    sfc = rng.uniform(1, 9, (ng, ncol)); emis = rng.uniform(0.7, 1.0, (ncol, 1))
    t = T(gpu)
    op, src, fl = lw_objects(pkg, gpu, tau, lay, inc, dec, sfc)
    emis_gpt = np.repeat(emis.T, ng, 0)
    outs = []
    for thresh, terms in ((0.0, 2), (1.220703125e-4, 3), (1e-2, 2), (1e-2, 3)):
        pkg.set_solver_option("lw_tau_thresh", thresh); pkg.set_solver_option("lw_series_terms", terms)
        assert pkg.get_solver_option("lw_series_terms") == terms
        assert pkg.rte_lw(op, True, src, t(emis), fl) == ""
        kw = dict(lw_series_terms=terms)
        if thresh > 0:
            kw["lw_tau_thresh"] = thresh
        fu, fd = oracle_mod.rte_lw(tau, lay, inc, dec, emis_gpt, sfc, options=oracle_mod.solver_options(**kw))
        assert np.max(np.abs(fl.flux_up.cpu().numpy() - fu)) < FLUX_ATOL
        assert np.max(np.abs(fl.flux_dn.cpu().numpy() - fd)) < FLUX_ATOL
        outs.append(fl.flux_dn.cpu().numpy().copy())
    assert not np.array_equal(outs[2], outs[3])         # the switch does switch
    with pytest.raises(ValueError):
        pkg.set_solver_option("lw_series_terms", 4)
    with pytest.raises(ValueError):
        pkg.set_solver_option("no_such_option", 1)


def test_sw_solver_switches(pkg, gpu, oracle_mod, arithmetic):
    import torch
    rng = np.random.default_rng(6)
    ng, nlay, ncol = 7, 60, 300
    tau = rng.uniform(0.001, 3.0, (ng, nlay, ncol)); ssa = rng.uniform(0.0, 0.999999, (ng, nlay, ncol))
    g = rng.uniform(0, 0.9, (ng, nlay, ncol))
    ssa[:, :, :40] = 1.0; g[:, :, :40] = 0.0             # conservative scattering: k*k hits the floor
    mu0 = rng.uniform(0.05, 1.0, ncol); toa = rng.uniform(10, 100, (ng, ncol))
    alb = rng.uniform(0.05, 0.4, (ncol, 1))
    t = T(gpu)
    op = pkg.OpticalProps2str(); op.tau, op.ssa, op.g = t(tau), t(ssa), t(g)
    op.band2gpt = np.array([[1, ng]], dtype=np.int32)
    fl = pkg.FluxesBroadband(*(torch.empty((nlay + 1, ncol), dtype=torch.float64, device=gpu) for _ in range(3)))
    a_gpt = np.repeat(alb.T, ng, 0)
    res = {}
    for clamp, kfl in ((0, 1e-12), (1, 1e-12), (0, 2.2e-12), (1, 1e-6)):
        pkg.set_solver_option("sw_dir_clamp", clamp); pkg.set_solver_option("sw_k_floor", kfl)
        assert pkg.rte_sw(op, True, t(mu0), t(toa), t(alb), t(alb), fl) == ""
        fu, fd, fdir = oracle_mod.rte_sw(tau, ssa, g, mu0, toa, a_gpt, a_gpt,
                                         options=oracle_mod.solver_options(sw_dir_clamp=clamp, sw_k_floor=kfl))
        scale = max(1.0, float(np.max(np.abs(fu))))
        assert np.max(np.abs(fl.flux_up.cpu().numpy() - fu)) < FLUX_ATOL * scale
        assert np.max(np.abs(fl.flux_dn.cpu().numpy() - fd)) < FLUX_ATOL * scale
        assert np.max(np.abs(fl.flux_dn_dir.cpu().numpy() - fdir)) < FLUX_ATOL * scale
        res[(clamp, kfl)] = fl.flux_up.cpu().numpy().copy()
    assert not np.array_equal(res[(0, 1e-12)], res[(1, 1e-12)])


# ------------------------------------------------------------------------------------------------
# ECCKD_DEVICE is asynchronous for every solver
# ------------------------------------------------------------------------------------------------
def _sw_case(pkg, gpu, rng, ncol, nlay=60, ng=9):
    import torch
    t = T(gpu)
    tau = rng.uniform(0.001, 2.0, (ng, nlay, ncol)); ssa = rng.uniform(0.0, 0.99, (ng, nlay, ncol)); g = rng.uniform(0, 0.8, (ng, nlay, ncol))
    op = pkg.OpticalProps2str(); op.tau, op.ssa, op.g = t(tau), t(ssa), t(g)
    op.band2gpt = np.array([[1, ng]], dtype=np.int32)
    mu0 = t(rng.uniform(0.1, 1.0, ncol)); toa = t(rng.uniform(10, 100, (ng, ncol))); alb = t(rng.uniform(0.05, 0.4, (ncol, 1)))
    fl = pkg.FluxesBroadband(*(torch.zeros((nlay + 1, ncol), dtype=torch.float64, device=gpu) for _ in range(3)))
    return op, mu0, toa, alb, fl


def test_rte_sw_and_deep_rte_lw_in_hip_graphs(pkg, gpu):
    """rte_sw with the two-pass kernel (sw_solver = 1: always uses a scratch ring; the layer-systolic default needs none,
    tests/test_gpu_round3.py) and rte_lw with 137 layers (scratch ring beyond 96 layers) captured in
    HIP graphs after one warm-up call on the capturing stream; a capture that would have to allocate fails with
    a message instead of invalidating the capture silently; a caller-owned scratch buffer needs no warm-up."""
    import torch
    pkg.set_solver_option("sw_solver", 1)
    try:
        _graphs_body(pkg, gpu)
    finally:
        pkg.set_solver_option("sw_solver", 0)


def _graphs_body(pkg, gpu):
    import torch
    rng = np.random.default_rng(8)
    ncol = 512
    op, mu0, toa, alb, fl = _sw_case(pkg, gpu, rng, ncol)
    nlay = 137
    tau = rng.uniform(0, 1, (5, nlay, ncol)); lay, inc, dec = (rng.uniform(1, 9, (5, nlay, ncol)) for _ in range(3))
    sfc = rng.uniform(1, 9, (5, ncol))
    op_l, src_l, fl_l = lw_objects(pkg, gpu, tau, lay, inc, dec, sfc)
    emis = T(gpu)(np.full((ncol, 1), 0.97))

    def step():
        assert pkg.rte_sw(op, True, mu0, toa, alb, alb, fl) == ""
        assert pkg.rte_lw(op_l, True, src_l, emis, fl_l) == ""

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()                                   # warm-up on the stream that will be captured: its scratch block exists now
    torch.cuda.synchronize()
    ref = (fl.flux_up.clone(), fl_l.flux_up.clone())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        step()
    fl.flux_up.zero_(); fl_l.flux_up.zero_()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(fl.flux_up, ref[0]) and torch.equal(fl_l.flux_up, ref[1])
    # a stream that has never run the solver: capturing fails cleanly, with the remedy in the message
    fresh = torch.cuda.Stream()
    g2 = torch.cuda.CUDAGraph()
    msg = None
    try:
        with torch.cuda.graph(g2, stream=fresh):
            msg = pkg.rte_sw(op, True, mu0, toa, alb, alb, fl)
    except RuntimeError:
        pass                                     # torch may refuse to end an empty capture; the message is what matters
    assert msg is not None and "captured" in msg and "ecckd_set_stream_scratch" in msg
    # caller-owned scratch: no warm-up needed
    fresh2 = torch.cuda.Stream()
    need = max(pkg.rte_sw_scratch_bytes(ncol, 60, 9), pkg.rte_lw_scratch_bytes(ncol, nlay, 5))
    assert need > 0 and pkg.rte_lw_scratch_bytes(ncol, 60, 32) == 0
    buf = torch.empty(need, dtype=torch.uint8, device=gpu)
    pkg.set_stream_scratch(buf, stream=fresh2)
    g3 = torch.cuda.CUDAGraph()
    fl.flux_up.zero_(); fl_l.flux_up.zero_()
    torch.cuda.synchronize()
    with torch.cuda.graph(g3, stream=fresh2):
        step()
    g3.replay()
    torch.cuda.synchronize()
    assert torch.equal(fl.flux_up, ref[0]) and torch.equal(fl_l.flux_up, ref[1])
    small = torch.empty(1024, dtype=torch.uint8, device=gpu)
    pkg.set_stream_scratch(small, stream=fresh2)
    with torch.cuda.stream(fresh2):
        assert "too small" in pkg.rte_sw(op, True, mu0, toa, alb, alb, fl)
    pkg.set_stream_scratch(None, stream=fresh2)
    del g3, graph
    pkg.release_scratch(0)


def test_two_streams_run_the_solvers_concurrently(pkg, gpu, oracle_mod):
    """Calls on different streams use different scratch blocks: interleaved launches of different problems on two
    streams give the same fluxes as running each alone (a shared ring would be overwritten mid-flight)."""
    import torch
    rng = np.random.default_rng(9)
    cases = [_sw_case(pkg, gpu, rng, 40000), _sw_case(pkg, gpu, rng, 37000)]
    alone = []
    for op, mu0, toa, alb, fl in cases:
        assert pkg.rte_sw(op, True, mu0, toa, alb, alb, fl) == ""
        torch.cuda.synchronize()
        alone.append(fl.flux_up.clone())
        fl.flux_up.zero_()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for rep in range(3):
        for s, (op, mu0, toa, alb, fl) in zip(streams, cases):
            with torch.cuda.stream(s):
                assert pkg.rte_sw(op, True, mu0, toa, alb, alb, fl) == ""
    torch.cuda.synchronize()
    for want, (_, _, _, _, fl) in zip(alone, cases):
        assert torch.equal(fl.flux_up, want)
    # spot check against the oracle
    op, mu0, toa, alb, fl = cases[1]
    sl = slice(0, 64)
    a = np.repeat(alb.cpu().numpy()[sl].T, 9, 0)
    fu, _, _ = oracle_mod.rte_sw(op.tau[..., sl].cpu().numpy(), op.ssa[..., sl].cpu().numpy(), op.g[..., sl].cpu().numpy(),
                                 mu0[sl].cpu().numpy(), toa[:, sl].cpu().numpy(), a, a)
    assert np.max(np.abs(fl.flux_up[:, sl].cpu().numpy() - fu)) < FLUX_ATOL


# ------------------------------------------------------------------------------------------------
# BASELINE configs[4]: 36-g-point table, fp32 vs fp64 tolerance sweep
# ------------------------------------------------------------------------------------------------
def test_lw_36g_fp32_against_fp64_oracle(pkg, gpu, oracle_mod):
    """LW rrtmgp-tol0.061 (36 g-points, 16 bands; the higher-g file present in the reference: SURVEY section 8(c) 'Data
    gap'), 1000 columns: the fp64 path against the fp64 oracle at the fp64 bars, the fp32 path (float arrays: a host
    built with wp = real32; the Planck window is active because the 36-g table does not fit LDS whole in fp64 and is
    staged whole in fp32) against the same oracle at the stated single-precision bars:
        tau 2e-5 relative (where tau > 1e-6 max), sources 2e-6 relative, fluxes 2e-3 W m-2."""
    import torch
    k = pkg.GasOpticsEcckd()
    assert k.load(LW_RRTMGP, device=0) == ""
    m = oracle_mod.CkdModel(LW_RRTMGP)
    ncol, nlay, ng = 1000, 60, 36
    assert k.get_ngpt() == ng and k.get_nband() == 16
    cols = synthetic.columns(17, ncol, k.get_press_min())
    emis2 = np.repeat(cols["sfc_emis"][:, None], 16, 1) * np.linspace(0.9, 1.0, 16)[None, :]
    emis_gpt = np.ascontiguousarray(emis2[:, m.gpt2band - 1].T)
    report = {}
    for name, npdt, tdt in (("f64", np.float64, torch.float64), ("f32", np.float32, torch.float32)):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=npdt)).to(gpu)
        gc = pkg.GasConcs(synthetic.GAS_ORDER)
        for n in synthetic.GAS_ORDER:
            v = cols[n]
            if np.isscalar(v):
                gc.set_vmr(n, float(v))
            elif v.ndim == 1:
                gc.set_vmr_column(n, t(v))
            else:
                gc.set_vmr(n, t(v))
        plev = t(cols["plev"])
        op = pkg.OpticalProps1scl(); op.alloc_1scl(ncol, nlay, k, like=plev)
        src = pkg.SourceFuncLW(); src.alloc(ncol, nlay, k, like=plev)
        assert k.gas_optics(None, plev, t(cols["tlay"]), t(cols["tsfc"]), gc, op, src, tlev=t(cols["tlev"])) == ""
        fl = pkg.FluxesBroadband(torch.empty((nlay + 1, ncol), dtype=tdt, device=gpu), torch.empty((nlay + 1, ncol), dtype=tdt, device=gpu))
        assert pkg.rte_lw(op, True, src, t(emis2), fl) == ""
        # the oracle sees the inputs as this precision rounds them, and computes in double
        r = lambda a: np.ascontiguousarray(np.asarray(a, dtype=npdt), dtype=np.float64)
        c = {n: (r(v) if isinstance(v, np.ndarray) else v) for n, v in cols.items()}
        tau, lay, inc, dec, sfc, _ = oracle_mod.gas_optics_int(m, c["plev"], c["tlay"], c["tsfc"], helpers.oracle_gas_items(c), c["tlev"])
        fu, fd = oracle_mod.rte_lw(tau, lay, inc, dec, r(emis_gpt), sfc)
        gt = op.tau.cpu().numpy().astype(np.float64)
        big = tau > 1e-6 * tau.max()
        report[name] = dict(tau=float(np.max(np.abs(gt - tau)[big] / tau[big])),
                            src=max(helpers.max_rel(src.lay_source.cpu().numpy(), lay), helpers.max_rel(src.lev_source_inc.cpu().numpy(), inc),
                                    helpers.max_rel(src.sfc_source.cpu().numpy(), sfc)),
                            flux=max(float(np.max(np.abs(fl.flux_up.cpu().numpy() - fu))), float(np.max(np.abs(fl.flux_dn.cpu().numpy() - fd)))))
    print("configs[4] sweep, 36 g-points:", report)
    assert report["f64"]["tau"] < TAU_RTOL and report["f64"]["src"] == 0.0 and report["f64"]["flux"] < FLUX_ATOL
    assert report["f32"]["tau"] < 2e-5 and report["f32"]["src"] < 2e-6 and report["f32"]["flux"] < 2e-3
    plan = k.plan(ncol, nlay, synthetic.GAS_ORDER, single_precision=True)
    assert plan["fused"] == 1 and plan["planck_fused"] == 1


# ------------------------------------------------------------------------------------------------
# BASELINE configs[3]: one rank's shard of the 1e7-column job
# ------------------------------------------------------------------------------------------------
def test_configs3_shard_properties(pkg, gpu, oracle_mod, lw):
    """1e7 columns sharded over 8 GPUs = 1.25e6 columns per rank (77 GB of intermediates).  The LAST rank's shard
    (columns 8 750 000 .. 9 999 999 of the counter-based generator) on one GPU: oracle spot checks at both ends and
    in the middle, then properties the full size allows -- determinism, exact linearity of rte_lw in the sources,
    zero incident flux, and the shard boundary: its first columns equal the last columns of a run that straddles
    the boundary (column-range sharding is exact)."""
    import torch
    k, m = lw
    ncol, nlay, ng = 1250000, 60, 32
    c0 = 7 * ncol
    f64 = dict(dtype=torch.float64, device=gpu)
    d = {n: torch.empty((nlay + 1 if n in ("plev", "tlev") else nlay, ncol), **f64) for n in ("plev", "tlev", "tlay", "h2o", "o3")}
    pc = {n: torch.empty((ncol,), **f64) for n in ("tsfc", "sfc_emis", "co2", "ch4", "n2o", "cfc11", "cfc12")}
    keep = {}
    for s0 in range(0, ncol, 125000):
        cols = synthetic.columns(c0 + s0, 125000, k.get_press_min())
        for n in d:
            d[n][:, s0:s0 + 125000] = torch.from_numpy(cols[n]).to(gpu)
        for n in pc:
            pc[n][s0:s0 + 125000] = torch.from_numpy(cols[n]).to(gpu)
        if s0 in (0, 625000, ncol - 125000):
            keep[s0] = cols
    gc = pkg.GasConcs(synthetic.GAS_ORDER)
    for n in synthetic.GAS_ORDER:
        if n in d:
            gc.set_vmr(n, d[n])
        elif n in pc:
            gc.set_vmr_column(n, pc[n])
        else:
            gc.set_vmr(n, 0.209 if n == "o2" else 0.0)
    op = pkg.OpticalProps1scl(); op.alloc_1scl(ncol, nlay, k, like=d["plev"])
    src = pkg.SourceFuncLW(); src.alloc(ncol, nlay, k, like=d["plev"])
    fl = pkg.FluxesBroadband(torch.empty((nlay + 1, ncol), **f64), torch.empty((nlay + 1, ncol), **f64))
    emis = pc["sfc_emis"].reshape(ncol, 1)

    def run():
        assert k.gas_optics(None, d["plev"], d["tlay"], pc["tsfc"], gc, op, src, tlev=d["tlev"]) == ""
        assert pkg.rte_lw(op, True, src, emis, fl) == ""
        torch.cuda.synchronize()

    run()
    for s0, sl in ((0, slice(0, 96)), (625000, slice(625000, 625096)), (ncol - 125000, slice(ncol - 96, ncol))):
        cols = keep[s0]
        lo = sl.start - s0
        sub = {n: (np.ascontiguousarray(v[..., lo:lo + 96]) if isinstance(v, np.ndarray) else v) for n, v in cols.items()}
        otau, olay, oinc, odec, osfc, _ = oracle_mod.gas_optics_int(m, sub["plev"], sub["tlay"], sub["tsfc"], synthetic.gas_items(sub), sub["tlev"])
        fu, fd = oracle_mod.rte_lw(otau, olay, oinc, odec, np.repeat(sub["sfc_emis"][None], ng, 0), osfc)
        assert helpers.max_rel(op.tau[..., sl].cpu().numpy(), otau) < TAU_RTOL
        assert np.array_equal(src.lay_source[..., sl].cpu().numpy(), olay)
        assert np.array_equal(src.lev_source_dec[..., sl].cpu().numpy(), odec)
        assert np.max(np.abs(fl.flux_up[:, sl].cpu().numpy() - fu)) < FLUX_ATOL
        assert np.max(np.abs(fl.flux_dn[:, sl].cpu().numpy() - fd)) < FLUX_ATOL
    assert bool(torch.all(fl.flux_dn[0] == 0))
    assert bool(torch.all(torch.isfinite(fl.flux_up))) and bool(torch.all(fl.flux_up > 0))
    ref_up, ref_dn = fl.flux_up.clone(), fl.flux_dn.clone()
    tau_sum = op.tau.sum(dtype=torch.float64).item()
    # determinism
    fl.flux_up.zero_(); fl.flux_dn.zero_()
    run()
    assert torch.equal(fl.flux_up, ref_up) and torch.equal(fl.flux_dn, ref_dn) and op.tau.sum(dtype=torch.float64).item() == tau_sum
    # linearity of rte_lw in the sources: x2 is exact in binary floating point
    for a in (src.lay_source, src.lev_source_inc, src.lev_source_dec, src.sfc_source):
        a.mul_(2.0)
    assert pkg.rte_lw(op, True, src, emis, fl) == ""
    torch.cuda.synchronize()
    assert torch.equal(fl.flux_up, 2.0 * ref_up) and torch.equal(fl.flux_dn, 2.0 * ref_dn)
    # shard boundary: a 4096-column run that straddles it (2048 columns of rank 6, 2048 of rank 7)
    n2 = 4096
    cb = synthetic.columns(c0 - 2048, n2, k.get_press_min())
    err, tau_b, lay_b, inc_b, dec_b, sfc_b = helpers.run_lw_gas_optics(pkg, k, cb, gpu)
    assert err == ""
    assert np.array_equal(tau_b[..., 2048:], op.tau[..., :2048].cpu().numpy())


# ------------------------------------------------------------------------------------------------
# layer-split longwave solver and the fused longwave path
# ------------------------------------------------------------------------------------------------
@pytest.fixture
def split_solver(pkg):
    saved = pkg.get_solver_option("lw_solver"), pkg.get_solver_option("lw_split_seg")
    yield
    pkg.set_solver_option("lw_solver", saved[0]); pkg.set_solver_option("lw_split_seg", saved[1])


@pytest.mark.parametrize("seg", [10, 12, 15])
@pytest.mark.parametrize("top_at_1,nmus", [(True, 1), (False, 3)])
def test_layer_split_solver_vs_oracle(pkg, gpu, oracle_mod, lw, split_solver, seg, top_at_1, nmus):
    """kernels_rte_lw_split.hip (the waves of a block take 10-15 layers each and exchange affine segment composites)
    against the oracle, and against the register-resident solver: same fluxes to 1e-9 W m-2 (the intensities entering a
    segment are composed in another association); shared-levels form bit-identical to its generic form; incident flux;
    ragged column counts."""
    import torch
    k, m = lw
    t = T(gpu)
    ncol = 1000 + seg                                    # not a multiple of the 32-column tile
    cols = synthetic.columns(500, ncol, k.get_press_min())
    err, tau, lay, inc, dec, sfc = helpers.run_lw_gas_optics(pkg, k, cols, gpu)
    assert err == ""
    if not top_at_1:
        f = lambda a: np.ascontiguousarray(a[:, ::-1, :])
        tau, lay, inc, dec = f(tau), f(lay), f(dec), f(inc)      # bottom-up storage: inc/dec swap roles
    op, src, fl = lw_objects(pkg, gpu, tau, lay, inc, dec, sfc)
    emis = cols["sfc_emis"][:, None]
    incf = np.random.default_rng(seg).uniform(0, 20, (32, ncol))
    fu, fd = oracle_mod.rte_lw(tau, lay, inc, dec, np.repeat(emis.T, 32, 0), sfc, top_at_1=top_at_1, nmus=nmus)
    fu_i, fd_i = oracle_mod.rte_lw(tau, lay, inc, dec, np.repeat(emis.T, 32, 0), sfc, top_at_1=top_at_1, nmus=nmus, inc_flux=incf)
    pkg.set_solver_option("lw_solver", 0)
    assert pkg.rte_lw(op, top_at_1, src, t(emis), fl, n_gauss_angles=nmus) == ""
    classic = fl.flux_up.cpu().numpy().copy()
    pkg.set_solver_option("lw_solver", 1); pkg.set_solver_option("lw_split_seg", seg)
    fl.flux_up.zero_(); fl.flux_dn.zero_()
    assert pkg.rte_lw(op, top_at_1, src, t(emis), fl, n_gauss_angles=nmus) == ""
    gu, gd = fl.flux_up.cpu().numpy().copy(), fl.flux_dn.cpu().numpy().copy()
    assert np.max(np.abs(gu - fu)) < FLUX_ATOL and np.max(np.abs(gd - fd)) < FLUX_ATOL
    assert np.max(np.abs(gu - classic)) < FLUX_ATOL
    assert pkg.rte_lw(op, top_at_1, src, t(emis), fl, n_gauss_angles=nmus, shared_levels=True) == ""
    assert np.array_equal(fl.flux_up.cpu().numpy(), gu) and np.array_equal(fl.flux_dn.cpu().numpy(), gd)
    assert pkg.rte_lw(op, top_at_1, src, t(emis), fl, n_gauss_angles=nmus, inc_flux=t(incf)) == ""
    assert np.max(np.abs(fl.flux_up.cpu().numpy() - fu_i)) < FLUX_ATOL and np.max(np.abs(fl.flux_dn.cpu().numpy() - fd_i)) < FLUX_ATOL


@pytest.mark.parametrize("seg", [10, 15])
def test_fused_lw_path_vs_oracle(pkg, gpu, oracle_mod, lw, split_solver, seg):
    """Fused longwave path: ecckd_gas_optics_lw_tau (tau only) + ecckd_rte_lw_fused (Planck sources recomputed in the
    solver from tlay / tlev / tsfc), and ecckd_lw_fluxes (both, tau in library scratch; device and host arrays) against
    the oracle's gas_optics_int + rte_lw at the fp64 bar (1e-9 W m-2), including columns outside the Planck table
    (below 120 K: the (T/T1) B(:,1) branch; above 350 K: extrapolation), incident flux, 3 angles, both orientations."""
    import torch
    k, m = lw
    t = T(gpu)
    pkg.set_solver_option("lw_split_seg", seg)
    ncol, nlay, ng = 777, 60, 32
    cols = synthetic.columns(123, ncol, k.get_press_min())
    cols = {n: (v.copy() if isinstance(v, np.ndarray) else v) for n, v in cols.items()}
    cols["tlev"][:, 20:24] = 100.0; cols["tlay"][:, 20:24] = 100.0; cols["tsfc"][20:24] = 110.0
    cols["tlev"][:, 24:28] = 400.0; cols["tlay"][:, 24:28] = 400.0; cols["tsfc"][24:28] = 360.0
    otau, olay, oinc, odec, osfc, _ = oracle_mod.gas_optics_int(m, cols["plev"], cols["tlay"], cols["tsfc"],
                                                               helpers.oracle_gas_items(cols), cols["tlev"])
    emis = cols["sfc_emis"][:, None]
    gc = helpers.product_gas_concs(pkg, cols, t)
    plev, tlay, tlev, tsfc = t(cols["plev"]), t(cols["tlay"]), t(cols["tlev"]), t(cols["tsfc"])
    op = pkg.OpticalProps1scl(); op.alloc_1scl(ncol, nlay, k, like=plev)
    fl = pkg.FluxesBroadband(torch.zeros((nlay + 1, ncol), dtype=torch.float64, device=gpu),
                             torch.zeros((nlay + 1, ncol), dtype=torch.float64, device=gpu))
    assert k.gas_optics_tau(plev, tlay, gc, op) == ""
    assert helpers.max_rel(op.tau.cpu().numpy(), otau) < TAU_RTOL
    incf = np.random.default_rng(3).uniform(0, 20, (ng, ncol))
    for nmus, inc in ((1, None), (3, incf)):
        fu, fd = oracle_mod.rte_lw(otau, olay, oinc, odec, np.repeat(emis.T, ng, 0), osfc, nmus=nmus, inc_flux=inc)
        ok = np.isfinite(fu)
        assert k.rte_lw_fused(op, True, tlay, tlev, tsfc, t(emis), fl, n_gauss_angles=nmus, inc_flux=None if inc is None else t(inc)) == ""
        assert np.max(np.abs(fl.flux_up.cpu().numpy() - fu)[ok]) < FLUX_ATOL and np.max(np.abs(fl.flux_dn.cpu().numpy() - fd)[ok]) < FLUX_ATOL
        fl.flux_up.zero_(); fl.flux_dn.zero_()
        assert k.lw_fluxes(plev, tlay, tsfc, tlev, gc, True, t(emis), fl, n_gauss_angles=nmus, inc_flux=None if inc is None else t(inc)) == ""
        assert np.max(np.abs(fl.flux_up.cpu().numpy() - fu)[ok]) < FLUX_ATOL and np.max(np.abs(fl.flux_dn.cpu().numpy() - fd)[ok]) < FLUX_ATOL
        dev_up = fl.flux_up.cpu().numpy().copy()
        # host arrays through the same entry point
        hgc = helpers.product_gas_concs(pkg, cols)
        hfl = pkg.FluxesBroadband(np.zeros((nlay + 1, ncol)), np.zeros((nlay + 1, ncol)))
        assert k.lw_fluxes(cols["plev"], cols["tlay"], cols["tsfc"], cols["tlev"], hgc, True, np.ascontiguousarray(emis), hfl,
                           n_gauss_angles=nmus, inc_flux=inc) == ""
        assert np.array_equal(hfl.flux_up, dev_up)
    # bottom-up storage, solver only: the reference's gas optics takes the pressure difference plev(:,j+1) - plev(:,j)
    # as the layer mass (:143), i.e. it assumes the top at index 1 itself, so only the solver has another orientation
    f2 = lambda a: np.ascontiguousarray(a[::-1])
    op.tau = t(np.ascontiguousarray(otau[:, ::-1, :]))
    fl.flux_up.zero_()
    assert k.rte_lw_fused(op, False, t(f2(cols["tlay"])), t(f2(cols["tlev"])), tsfc, t(emis), fl) == ""
    fu, fd = oracle_mod.rte_lw(otau, olay, oinc, odec, np.repeat(emis.T, ng, 0), osfc)
    ok = np.isfinite(fu)
    assert np.max(np.abs(fl.flux_up.cpu().numpy()[::-1] - fu)[ok]) < FLUX_ATOL
    assert np.max(np.abs(fl.flux_dn.cpu().numpy()[::-1] - fd)[ok]) < FLUX_ATOL
    # other layer counts take the general route inside the same entry points (Planck kernel into library scratch, then the
    # register-resident solver; 137 layers also exercises its ring): same results
    for nl in (37, 137):
        c2 = synthetic.columns(5, 130, k.get_press_min(), nlay=nl)
        gc2 = helpers.product_gas_concs(pkg, c2, t)
        fl2 = pkg.FluxesBroadband(torch.zeros((nl + 1, 130), dtype=torch.float64, device=gpu), torch.zeros((nl + 1, 130), dtype=torch.float64, device=gpu))
        e2 = c2["sfc_emis"][:, None]
        assert k.lw_fluxes(t(c2["plev"]), t(c2["tlay"]), t(c2["tsfc"]), t(c2["tlev"]), gc2, True, t(e2), fl2, n_gauss_angles=2) == ""
        o = oracle_mod.gas_optics_int(m, c2["plev"], c2["tlay"], c2["tsfc"], helpers.oracle_gas_items(c2), c2["tlev"])
        fu, fd = oracle_mod.rte_lw(o[0], o[1], o[2], o[3], np.repeat(e2.T, ng, 0), o[4], nmus=2)
        assert np.max(np.abs(fl2.flux_up.cpu().numpy() - fu)) < FLUX_ATOL and np.max(np.abs(fl2.flux_dn.cpu().numpy() - fd)) < FLUX_ATOL
        hfl2 = pkg.FluxesBroadband(np.zeros((nl + 1, 130)), np.zeros((nl + 1, 130)))
        assert k.lw_fluxes(c2["plev"], c2["tlay"], c2["tsfc"], c2["tlev"], helpers.product_gas_concs(pkg, c2), True, np.ascontiguousarray(e2), hfl2,
                           n_gauss_angles=2) == ""
        assert np.array_equal(hfl2.flux_up, fl2.flux_up.cpu().numpy())


# ------------------------------------------------------------------------------------------------
# RTE-RRTMGP's kernel-level bind(C) interfaces (librte_kernels_hip.so, include/rte_kernels_hip.h)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("top_at_1,nmus", [(True, 1), (False, 3)])
def test_rte_kernel_level_interfaces(pkg, gpu, oracle_mod, top_at_1, nmus):
    """lw_solver_noscat_GaussQuad, sw_solver_2stream, sum_broadband, net_broadband_precalc called the way a Fortran
    bind(C) interface without VALUE calls them (every argument by reference, logical(wl) as C bool, host arrays, the
    incident fluxes parked in the top level of flux_dn / flux_dir): spectral fluxes against the oracle's."""
    import ctypes as C
    K = C.CDLL(pkg.RTE_KERNELS_LIB)
    rng = np.random.default_rng(21)
    ncol, nlay, ng = 150, 37, 5
    top = 0 if top_at_1 else nlay
    dp = C.POINTER(C.c_double)
    P = lambda a: a.ctypes.data_as(dp)
    I = lambda v: C.byref(C.c_int(v))
    B = C.byref(C.c_bool(top_at_1))
    # ---- longwave ----
    tau = rng.uniform(0, 2, (ng, nlay, ncol)) * rng.choice([1e-9, 1e-3, 1.0], size=(ng, nlay, ncol))
    lay, inc, dec = (rng.uniform(1, 9, (ng, nlay, ncol)) for _ in range(3))
    sfc = rng.uniform(1, 9, (ng, ncol)); emis = rng.uniform(0.7, 1.0, (ng, ncol)); incf = rng.uniform(0, 20, (ng, ncol))
    Ds = np.array([[1.66], [1.18350343, 2.81649655], [1.09719858, 1.69338507, 4.70941630]][nmus - 1])
    wts = np.array([[0.5], [0.3180413817, 0.1819586183], [0.2009319137, 0.2292411064, 0.0698269799]][nmus - 1])
    fu = np.zeros((ng, nlay + 1, ncol)); fd = np.zeros((ng, nlay + 1, ncol))
    fd[:, top, :] = incf                                            # RTE's apply_BC
    K.lw_solver_noscat_GaussQuad(I(ncol), I(nlay), I(ng), B, I(nmus), P(Ds), P(wts), P(tau), P(lay), P(inc), P(dec), P(emis),
                                 P(sfc), P(fu), P(fd))
    ou, od = oracle_mod.rte_lw_gpt(tau, lay, inc, dec, emis, sfc, top_at_1=top_at_1, nmus=nmus, inc_flux=incf)
    assert np.max(np.abs(fu - ou)) < FLUX_ATOL and np.max(np.abs(fd - od)) < FLUX_ATOL
    bb = np.zeros((nlay + 1, ncol)); bbd = np.zeros((nlay + 1, ncol)); net = np.zeros((nlay + 1, ncol))
    K.sum_broadband(I(ncol), I(nlay + 1), I(ng), P(fu), P(bb))
    K.sum_broadband(I(ncol), I(nlay + 1), I(ng), P(fd), P(bbd))
    obu, obd = oracle_mod.rte_lw(tau, lay, inc, dec, emis, sfc, top_at_1=top_at_1, nmus=nmus, inc_flux=incf)
    assert np.max(np.abs(bb - obu)) < FLUX_ATOL and np.max(np.abs(bbd - obd)) < FLUX_ATOL
    K.net_broadband_precalc(I(ncol), I(nlay + 1), P(bbd), P(bb), P(net))
    assert np.array_equal(net, bbd - bb)
    # ---- shortwave ----
    tau = rng.uniform(0.001, 2.0, (ng, nlay, ncol)); ssa = rng.uniform(0, 0.999, (ng, nlay, ncol)); g = rng.uniform(0, 0.8, (ng, nlay, ncol))
    mu0 = rng.uniform(0.1, 1.0, ncol); toa = rng.uniform(10, 100, (ng, ncol)); dif = rng.uniform(0, 5, (ng, ncol))
    ad = rng.uniform(0.05, 0.4, (ng, ncol)); af = rng.uniform(0.05, 0.4, (ng, ncol))
    fu = np.zeros((ng, nlay + 1, ncol)); fd = np.zeros((ng, nlay + 1, ncol)); fr = np.zeros((ng, nlay + 1, ncol))
    fr[:, top, :] = toa * mu0[None, :]
    fd[:, top, :] = dif
    K.sw_solver_2stream(I(ncol), I(nlay), I(ng), B, P(tau), P(ssa), P(g), P(mu0), P(ad), P(af), P(fu), P(fd), P(fr))
    ou, od, odir = oracle_mod.rte_sw_gpt(tau, ssa, g, mu0, toa, ad, af, top_at_1=top_at_1, inc_flux_dif=dif)
    assert np.max(np.abs(fu - ou)) < FLUX_ATOL and np.max(np.abs(fd - od)) < FLUX_ATOL and np.max(np.abs(fr - odir)) < FLUX_ATOL
    # device-pointer flavour of the same C entry points: the same bits as the host flavour
    import torch
    t = T(gpu)
    dfu, dfd, dfr = (torch.zeros((ng, nlay + 1, ncol), dtype=torch.float64, device=gpu) for _ in range(3))
    vp = lambda x: C.c_void_p(x.data_ptr())
    keep = [t(x) for x in (tau, ssa, g, mu0, toa * mu0[None, :], dif, ad, af)]
    rc = pkg.lib().ecckd_sw_solver_2stream_gpt(0, ncol, nlay, ng, int(top_at_1), *[vp(x) for x in keep], vp(dfu), vp(dfd), vp(dfr),
                                               pkg.DEVICE, None)
    assert rc == 0, pkg.last_error()
    torch.cuda.synchronize()
    assert np.array_equal(dfu.cpu().numpy(), fu) and np.array_equal(dfr.cpu().numpy(), fr)


# ------------------------------------------------------------------------------------------------
# multi-rank rehearsal of bench.py on one GPU
# ------------------------------------------------------------------------------------------------
def test_bench_two_ranks_rehearsal(pkg, gpu, tmp_path):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one process per rank), here
    with 2 ranks that both sit on cuda:0 over gloo (--rehearse-on-one-gpu: RCCL cannot put two ranks on one device):
    the product library is loaded and driven by every rank, rank r generates columns [r*ncol, (r+1)*ncol), the timed
    region is bracketed by barriers, the elapsed time is the MAX over ranks, every rank's own time is gathered, and
    rank 0 prints ONE JSON line whose value is the whole-job throughput."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--ncol", "60000",
           "--cpu-seconds", "0", "--no-side", "--rehearse-on-one-gpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["ncol_per_gpu"] == 60000 and d["config"]["ncol_total"] == 120000
    assert len(d["per_rank_ms_per_step"]["ranks"]) == 2
    assert d["per_rank_ms_per_step"]["max"] <= d["ms_per_step"] * 1.001 + 1e-3
    assert abs(d["value"] - 2 * 60000 * 60 * 32 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    assert d["check_max_abs_flux_diff_vs_oracle_Wm2"] < FLUX_ATOL and "REHEARSAL" in d["config"]["parallelism"]
    assert d["roofline"]["kernel"] in ("gas_lw_fused", "rte_lw") and d["cpu_baseline"] is None


# ------------------------------------------------------------------------------------------------
# merged slot: gases passed as one number for the call share one table (gas_merge_scalars)
# ------------------------------------------------------------------------------------------------
WELL_MIXED = dict(co2=420e-6, ch4=1.9e-6, n2o=3.3e-7, cfc11=2.3e-10, cfc12=5.2e-10, o2=0.209)


@pytest.mark.parametrize("path", [LW_FSCK, LW_RRTMGP])
def test_merged_scalar_gases_lw(pkg, gpu, oracle_mod, path, monkeypatch):
    """RFMIP's gas description: well-mixed gases are scalars (mo_rfmip_io.F90 *_GM), h2o and o3 profiles.  The
    merged table must give the per-gas result: against the oracle (1e-12) and against the unmerged kernel; a
    column must not depend on the path its wave takes (slab vs tables from global memory)."""
    k = pkg.GasOpticsEcckd()
    assert k.load(path, device=0) == ""
    m = oracle_mod.CkdModel(path)
    ncol = 1500
    cols = synthetic.columns(77, ncol, k.get_press_min())
    plan = k.plan(ncol, 60, synthetic.GAS_ORDER, scalar_gases=list(WELL_MIXED) + ["no2"])
    assert plan["merged"] == 6 and plan["slots"] == 2
    err, tau, lay, inc, dec, sfc = helpers.run_lw_gas_optics(pkg, k, cols, gpu, overrides=WELL_MIXED)
    assert err == ""
    otau, olay, oinc, odec, osfc, oerr = oracle_mod.gas_optics_int(
        m, cols["plev"], cols["tlay"], cols["tsfc"], helpers.oracle_gas_items(cols, overrides=WELL_MIXED), cols["tlev"])
    assert oerr == ""
    assert helpers.max_rel(tau, otau) < TAU_RTOL
    assert np.array_equal(lay, olay) and np.array_equal(inc, oinc) and np.array_equal(dec, odec) and np.array_equal(sfc, osfc)
    pkg.set_solver_option("gas_merge_scalars", 0)
    try:
        err, tau0, *_ = helpers.run_lw_gas_optics(pkg, k, cols, gpu, overrides=WELL_MIXED)
    finally:
        pkg.set_solver_option("gas_merge_scalars", 1)
    assert err == "" and helpers.max_rel(tau, tau0) < 1e-13 and not np.array_equal(tau, tau0)
    # a Planck window of 16 rows sends many waves down the tables-from-global-memory path: same bits
    monkeypatch.setenv("ECCKD_PLANCK_WINDOW", "16")
    err, tau_w, *_ = helpers.run_lw_gas_optics(pkg, k, cols, gpu, overrides=WELL_MIXED)
    monkeypatch.delenv("ECCKD_PLANCK_WINDOW")
    assert err == "" and np.array_equal(tau_w, tau)


def test_merged_scalar_gases_keep_the_per_gas_clamp(pkg, gpu, oracle_mod, lw):
    """A relative_linear gas below its reference concentration has negative optical depths, which the reference
    clamps per gas (:234-238): it must stay out of the merged table.  Bottom-up pressure (negative layer
    thickness) flips every sign: then the gases above their reference are the ones clamped."""
    k, m = lw
    ncol = 700
    cols = synthetic.columns(5, ncol, k.get_press_min())
    ov = dict(WELL_MIXED, ch4=0.7e-6, n2o=1.0e-7)        # pre-industrial-like: below the reference mole fractions
    assert k.plan(ncol, 60, synthetic.GAS_ORDER, scalar_gases=list(ov))["merged"] == 6   # (a plan call knows no values)
    for flip in (False, True):
        c = dict(cols)
        if flip:
            c["plev"] = np.ascontiguousarray(cols["plev"][::-1])
        err, tau, *_ = helpers.run_lw_gas_optics(pkg, k, c, gpu, overrides=ov)
        assert err == ""
        otau, *_rest, oerr = oracle_mod.gas_optics_int(m, c["plev"], c["tlay"], c["tsfc"],
                                                      helpers.oracle_gas_items(c, overrides=ov), c["tlev"])
        assert oerr == ""
        assert np.max(np.abs(tau - otau) / np.maximum(np.abs(otau), 1e-30)) < TAU_RTOL
        assert np.all(tau >= 0)


def test_merged_scalar_gases_sw_and_f32(pkg, gpu, oracle_mod, lw):
    import torch
    k = pkg.GasOpticsEcckd()
    assert k.load(SW_WIDE, device=0) == ""
    m = oracle_mod.CkdModel(SW_WIDE)
    ncol, nlay, ng = 900, 60, 27
    cols = synthetic.columns(3, ncol, k.get_press_min(), shortwave=True)
    t = T(gpu)
    names = ["co2", "ch4", "n2o", "o2", "h2o", "o3"]
    gc = helpers.product_gas_concs(pkg, cols, t, names, overrides=WELL_MIXED)
    op = pkg.OpticalProps2str(); op.alloc_2str(ncol, nlay, k, like=t(np.zeros(1)))
    toa = torch.empty((ng, ncol), dtype=torch.float64, device=gpu)
    assert k.gas_optics(None, t(cols["plev"]), t(cols["tlay"]), gc, op, toa) == ""
    otau, ossa, og, otoa, oerr = oracle_mod.gas_optics_ext(m, cols["plev"], cols["tlay"],
                                                           helpers.oracle_gas_items(cols, names, overrides=WELL_MIXED))
    assert oerr == ""
    assert helpers.max_rel(op.tau.cpu().numpy(), otau) < TAU_RTOL
    assert helpers.max_rel(op.ssa.cpu().numpy(), ossa) < TAU_RTOL
    # single precision, longwave, scalar gases (fp32 takes the per-gas path) against the fp64 oracle
    k32, m32 = lw
    cols = synthetic.columns(11, 1000, k32.get_press_min())
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(gpu)
    gc = helpers.product_gas_concs(pkg, cols, f, overrides=WELL_MIXED)
    op = pkg.OpticalProps1scl(); op.alloc_1scl(1000, 60, k32, like=f(np.zeros(1)))
    src = pkg.SourceFuncLW(); src.alloc(1000, 60, k32, like=f(np.zeros(1)))
    assert k32.gas_optics(None, f(cols["plev"]), f(cols["tlay"]), f(cols["tsfc"]), gc, op, src, tlev=f(cols["tlev"])) == ""
    torch.cuda.synchronize()
    r = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32), dtype=np.float64)
    c32 = {n: (r(v) if isinstance(v, np.ndarray) else v) for n, v in cols.items()}
    ov32 = {n: float(np.float32(v)) for n, v in WELL_MIXED.items()}
    otau, *_rest, oerr = oracle_mod.gas_optics_int(m32, c32["plev"], c32["tlay"], c32["tsfc"],
                                                  helpers.oracle_gas_items(c32, overrides=ov32), c32["tlev"])
    tau32 = op.tau.cpu().numpy().astype(np.float64)
    big = otau > 1e-6 * otau.max()
    assert np.max(np.abs(tau32 - otau)[big] / otau[big]) < 2e-5       # as test_single_precision_lw_path


def test_planck_sources_entry_point(pkg, gpu, oracle_mod, lw, arithmetic):
    """ecckd_planck_sources: the four source arrays alone.  Same bits as the sources gas_optics writes (one
    interpolation code for the fused kernel and the stand-alone one), oracle to 1e-15; without tlev only lay_source and
    sfc_source are written; ragged column counts."""
    import torch
    k, m = lw
    for ncol in (1, 63, 1000, 1537):
        cols = synthetic.columns(40, ncol, k.get_press_min())
        cols["tlay"][5, : min(3, ncol)] = 100.0           # below the Planck table: (T/t0)*B(:,1)
        cols["tlev"][9, : min(2, ncol)] = 365.0           # above it: linear extrapolation
        err, tau, lay, inc, dec, sfc = helpers.run_lw_gas_optics(pkg, k, cols, gpu)
        assert err == ""
        t = T(gpu)
        src = pkg.SourceFuncLW(); src.alloc(ncol, 60, k, like=t(np.zeros(1)))
        for a in (src.lay_source, src.lev_source_inc, src.lev_source_dec, src.sfc_source):
            a.fill_(-1.0)
        assert k.planck_sources(t(cols["tlay"]), t(cols["tsfc"]), src, tlev=t(cols["tlev"])) == ""
        torch.cuda.synchronize()
        for got, want in ((src.lay_source, lay), (src.lev_source_inc, inc), (src.lev_source_dec, dec), (src.sfc_source, sfc)):
            assert np.array_equal(got.cpu().numpy(), want)
        olay, oinc, odec, osfc = oracle_mod.gas_optics_int(m, cols["plev"], cols["tlay"], cols["tsfc"], [], cols["tlev"])[1:5]
        for got, want in ((src.lay_source, olay), (src.lev_source_inc, oinc), (src.lev_source_dec, odec), (src.sfc_source, osfc)):
            assert helpers.max_rel(got.cpu().numpy(), want) < 1e-15
        src.lev_source_inc.fill_(-1.0); src.lev_source_dec.fill_(-1.0); src.lay_source.fill_(-1.0)
        assert k.planck_sources(t(cols["tlay"]), t(cols["tsfc"]), src) == ""
        torch.cuda.synchronize()
        assert np.array_equal(src.lay_source.cpu().numpy(), lay) and bool((src.lev_source_inc == -1.0).all())


def test_rte_sw_zero_asymmetry_path_is_bit_identical(pkg, gpu):
    """rte_sw takes a specialised two-stream form for the layers of a wave whose asymmetry parameters are all zero (what
    ecCKD's gas optics writes).  A column's fluxes must not depend on which form its wave took: the same g = 0 columns,
    once alone (specialised form) and once next to a column with g != 0 in the same 16-column tile (general form)."""
    import torch
    rng = np.random.default_rng(11)
    ng, nlay, ncol = 27, 60, 64
    tau = rng.uniform(0.001, 2.0, (ng, nlay, ncol)); ssa = rng.uniform(0.0, 0.999, (ng, nlay, ncol))
    mu0 = rng.uniform(0.05, 1.0, ncol); toa = rng.uniform(10, 100, (ng, ncol)); alb = rng.uniform(0.05, 0.4, (ncol, 1))
    t = T(gpu)
    out = []
    for poison in (False, True):
        g = np.zeros((ng, nlay, ncol))
        if poison:
            g[:, :, 15::16] = 0.3           # the last column of every tile
        op = pkg.OpticalProps2str(); op.tau, op.ssa, op.g = t(tau), t(ssa), t(g)
        op.band2gpt = np.array([[1, ng]], dtype=np.int32)
        fl = pkg.FluxesBroadband(*(torch.empty((nlay + 1, ncol), dtype=torch.float64, device=gpu) for _ in range(3)))
        assert pkg.rte_sw(op, True, t(mu0), t(toa), t(alb), t(alb), fl) == ""
        out.append([f.cpu().numpy() for f in (fl.flux_up, fl.flux_dn, fl.flux_dn_dir)])
    keep = np.ones(ncol, bool); keep[15::16] = False
    for a, b in zip(*out):
        assert np.array_equal(a[:, keep], b[:, keep])
    assert not np.array_equal(out[0][0][:, ~keep], out[1][0][:, ~keep])     # (the changed columns do change: flux_up)


def test_merged_scalar_gases_in_a_two_pass_model(pkg, gpu, oracle_mod, lw):
    """Fourteen gases -> two kernel passes (<= 10 gases each), the second accumulating into tau; most gases are scalars
    and share the merged slot of their pass; one table has negative entries (its pass takes the per-g-point clamp
    instantiation and that gas keeps its own slot); orography makes some segments walk several slab positions."""
    import test_gpu_parity as tp
    k, m = lw
    rng = np.random.default_rng(9)
    tabs = []
    for n, t in zip(m.gas[:4], m.tables[:4]):
        tabs.append(dict(name=n, code=t["code"], composite_only=0, mole_fraction=t["mole_fraction"],
                         reference_mole_fraction=t["reference_mole_fraction"],
                         coefficient=t["coefficient"] if t["code"] == 2 else t["coefficient"][0]))
    lin = [t for t in m.tables if t["code"] in (1, 3)]
    extra = []
    for i in range(9):
        src = lin[i % len(lin)]
        extra.append("x%d" % i)
        tabs.append(dict(name="x%d" % i, code=1, composite_only=0, mole_fraction=None, reference_mole_fraction=0.0,
                         coefficient=src["coefficient"][0] * (0.5 + 0.1 * i)))
    neg = m.tables[2]["coefficient"][0] * rng.choice([1.0, -1.0], size=m.tables[2]["coefficient"][0].shape)
    tabs.append(dict(name="weird", code=1, composite_only=0, mole_fraction=None, reference_mole_fraction=0.0, coefficient=neg))
    k2 = pkg.GasOpticsEcckd()
    assert k2.init_from_tables(m.log_pressure, m.temperature, tabs, planck=(m.temperature_planck, m.planck_function)) == ""
    m2 = oracle_mod.CkdModel(LW_FSCK)
    m2.gas = [t["name"] for t in tabs]
    m2.tables = [dict(code=t["code"], composite_only=False,
                      mole_fraction=None if t["mole_fraction"] is None else np.ascontiguousarray(t["mole_fraction"], dtype=np.float64),
                      reference_mole_fraction=t["reference_mole_fraction"],
                      coefficient=np.ascontiguousarray(t["coefficient"] if t["coefficient"].ndim == 4 else t["coefficient"][None]),
                      nv=t["coefficient"].shape[0] if t["coefficient"].ndim == 4 else 1) for t in tabs]
    m2.num_gases = len(tabs)
    names = [t["name"] for t in tabs]
    assert len(names) == 14
    cols = tp.orography_ramp(k.get_press_min(), 1500, c0=21)
    over = {"weird": 1e-4}
    over.update({n: 1e-6 * (i + 1) for i, n in enumerate(extra)})
    over[m.gas[2]] = 4e-4                      # one of the file's own linear gases as a scalar too
    p = k2.plan(1500, 60, names, scalar_gases=list(over))
    assert p["passes"] == 2 and p["merged"] >= 6
    tp.check_lw(pkg, k2, m2, oracle_mod, cols, gpu, names=names, overrides=over)
    pkg.set_solver_option("gas_merge_scalars", 0)
    try:
        tp.check_lw(pkg, k2, m2, oracle_mod, cols, gpu, names=names, overrides=over)
    finally:
        pkg.set_solver_option("gas_merge_scalars", 1)


@pytest.mark.parametrize("seed", list(range(30)))
def test_random_gas_descriptions(pkg, gpu, oracle_mod, lw, arithmetic, seed):
    """Seeded differential test: a random column count, a random subset and order of the gas list (with unknown names
    mixed in), every gas in a random one of the four shapes ty_gas_concs knows -- scalar, per-layer profile, per-column
    value, full array -- some of them below their reference concentration, smooth or rough orography: tau against the
    oracle at 1e-12, sources bit for bit, in both arithmetic modes.  (Covers the merged slot next to per-gas slots,
    clamped gases, several slab positions, ragged waves.)"""
    k, m = lw
    rng = np.random.default_rng(1000 + seed)
    ncol = int(rng.choice([1, 17, 64, 65, 333, 512, 700, 1301, 2100]))
    cols = synthetic.columns(int(rng.integers(0, 10**6)), ncol, k.get_press_min())
    cols = {kk: (v.copy() if isinstance(v, np.ndarray) else v) for kk, v in cols.items()}
    if rng.random() < 0.5:                     # orography: smooth ramp or random surface pressure
        ps = np.linspace(52000.0, 103000.0, ncol) if rng.random() < 0.5 else rng.uniform(52000.0, 103000.0, ncol)
        eta = (np.arange(61, dtype=np.float64) / 60) ** 2
        ptop = cols["plev"][0, 0]
        cols["plev"] = np.ascontiguousarray(ptop + (ps[None, :] - ptop) * eta[:, None])
    pool = ["co2", "ch4", "n2o", "o2", "n2", "cfc11", "cfc12", "h2o", "o3", "no2", "xyz"]
    names = [n for n in rng.permutation(pool) if rng.random() < 0.8]
    base = dict(co2=4e-4, ch4=1.8e-6, n2o=3.3e-7, o2=0.209, n2=0.78, cfc11=2.3e-10, cfc12=5.2e-10, no2=1e-9, xyz=1e-3)
    over = {}
    for n in names:
        if n in ("h2o", "o3"):
            full = cols[n]
        else:
            lo = 0.3 if rng.random() < 0.3 else 1.0        # sometimes below the reference mole fraction
            full = base[n] * lo * rng.uniform(0.8, 1.6, (60, ncol))
        shape = rng.integers(0, 4)
        if shape == 0:
            over[n] = float(full.mean())
        elif shape == 1:
            over[n] = np.ascontiguousarray(full.mean(axis=1))          # (nlay,)
        elif shape == 2 and ncol != 60:
            over[n] = np.ascontiguousarray(full.mean(axis=0))          # (ncol,)
        else:
            over[n] = np.ascontiguousarray(full)
    err, tau, lay, inc, dec, sfc = helpers.run_lw_gas_optics(pkg, k, cols, gpu, names=names, overrides=over)
    assert err == ""
    otau, olay, oinc, odec, osfc, oerr = oracle_mod.gas_optics_int(
        m, cols["plev"], cols["tlay"], cols["tsfc"], helpers.oracle_gas_items(cols, names, over), cols["tlev"])
    assert oerr == ""
    assert np.array_equal(lay, olay) and np.array_equal(inc, oinc) and np.array_equal(dec, odec) and np.array_equal(sfc, osfc)
    assert np.max(np.abs(tau - otau) / np.maximum(np.abs(otau), 1e-300)) < TAU_RTOL
    assert np.array_equal(tau == 0, otau == 0)


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_sw_gas_descriptions_and_fluxes(pkg, gpu, oracle_mod, seed):
    """The shortwave pair on random gas descriptions (as test_random_gas_descriptions) and random surface albedos:
    tau / ssa / g / toa_src against the oracle, then rte_sw on the product's own optical properties against the
    oracle's solver on the oracle's."""
    import torch
    k = pkg.GasOpticsEcckd()
    assert k.load(SW_WIDE, device=0) == ""
    m = oracle_mod.CkdModel(SW_WIDE)
    rng = np.random.default_rng(5000 + seed)
    ncol = int(rng.choice([1, 16, 33, 257, 600, 1025]))
    nlay, ng = 60, 27
    cols = synthetic.columns(int(rng.integers(0, 10**6)), ncol, k.get_press_min(), shortwave=True)
    pool = ["co2", "ch4", "n2o", "o2", "n2", "h2o", "o3", "no2"]
    names = [n for n in rng.permutation(pool) if rng.random() < 0.85]
    base = dict(co2=4e-4, ch4=1.8e-6, n2o=3.3e-7, o2=0.209, n2=0.78, no2=1e-9)
    over = {}
    for n in names:
        full = cols[n] if n in ("h2o", "o3") else base[n] * (0.3 if rng.random() < 0.3 else 1.0) * rng.uniform(0.8, 1.6, (nlay, ncol))
        shape = rng.integers(0, 4)
        over[n] = (float(full.mean()) if shape == 0 else np.ascontiguousarray(full.mean(axis=1)) if shape == 1
                   else np.ascontiguousarray(full.mean(axis=0)) if (shape == 2 and ncol != nlay) else np.ascontiguousarray(full))
    t = T(gpu)
    gc = helpers.product_gas_concs(pkg, cols, t, names, over)
    op = pkg.OpticalProps2str(); op.alloc_2str(ncol, nlay, k, like=t(np.zeros(1)))
    toa = torch.empty((ng, ncol), dtype=torch.float64, device=gpu)
    assert k.gas_optics(None, t(cols["plev"]), t(cols["tlay"]), gc, op, toa) == ""
    otau, ossa, og, otoa, oerr = oracle_mod.gas_optics_ext(m, cols["plev"], cols["tlay"], helpers.oracle_gas_items(cols, names, over))
    assert oerr == ""
    assert helpers.max_rel(op.tau.cpu().numpy(), otau) < TAU_RTOL and helpers.max_rel(op.ssa.cpu().numpy(), ossa) < TAU_RTOL
    assert np.all(op.g.cpu().numpy() == 0) and np.array_equal(toa.cpu().numpy(), otoa)
    nband = k.get_nband()
    alb_dir = rng.uniform(0.02, 0.6, (ncol, nband)); alb_dif = rng.uniform(0.02, 0.6, (ncol, nband))
    fl = pkg.FluxesBroadband(*(torch.empty((nlay + 1, ncol), dtype=torch.float64, device=gpu) for _ in range(3)))
    assert pkg.rte_sw(op, True, t(cols["mu0"]), toa, t(alb_dir), t(alb_dif), fl) == ""
    g2b = m.gpt2band - 1
    fu, fd, fdir = oracle_mod.rte_sw(otau, ossa, og, cols["mu0"], otoa, np.ascontiguousarray(alb_dir[:, g2b].T),
                                     np.ascontiguousarray(alb_dif[:, g2b].T))
    # (each side solves on its OWN optical properties, 1e-15 apart: cells near the resonance k*mu0 = 1 of the two-stream
    # direct terms amplify that, up to ~1e-9 W m-2 here; the solver-only tests feed both sides the same arrays)
    for got, want in ((fl.flux_up, fu), (fl.flux_dn, fd), (fl.flux_dn_dir, fdir)):
        assert np.max(np.abs(got.cpu().numpy() - want)) < 10 * FLUX_ATOL


# tail split of the register-resident LW solver ("lw_tail_split"): tail tiles one g-point pair per wave + ordered sum
@pytest.mark.parametrize("ncol,nlay,ng,nmus,top_at_1,f32,shared", [
    (1000, 60, 32, 1, True, False, False),      # fewer tiles than SIMDs: every tile is a tail tile
    (33000, 60, 32, 1, True, False, True),      # one full round of 1 024 tiles + 8 tail tiles (the last one partly empty)
    (33000, 60, 32, 1, False, True, False),     # single precision
    (2500, 40, 27, 3, False, False, False),     # padded variant, odd g-point count, three angles
    (700, 91, 16, 2, True, False, True),
    (300, 137, 8, 1, True, False, False),       # beyond 96 layers: overflow variant, no split (option has no effect)
])
def test_rte_lw_tail_split_is_bit_identical(pkg, gpu, ncol, nlay, ng, nmus, top_at_1, f32, shared):
    """Fluxes with the tail split (default) against the whole-tile waves of round 1: the same bits, with and without
    incident flux.  The split sums the g-pair contributions in the order the whole-tile wave adds them."""
    import torch
    rng = np.random.default_rng(ncol + nlay)
    dt = torch.float32 if f32 else torch.float64
    tau = rng.uniform(0, 2, (ng, nlay, ncol)) * rng.choice([1e-9, 1e-3, 1.0], size=(ng, nlay, ncol))
    lay = rng.uniform(1, 9, (ng, nlay, ncol))
    lev = rng.uniform(1, 9, (ng, nlay + 1, ncol))
    inc, dec = np.ascontiguousarray(lev[:, 1:]), np.ascontiguousarray(lev[:, :-1])
    sfc = rng.uniform(1, 9, (ng, ncol))
    emis = rng.uniform(0.7, 1.0, (ncol, 2))
    incf = rng.uniform(0, 3, (ng, ncol))
    half = ng // 2
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu).to(dt)
    op = pkg.OpticalProps1scl(); op.tau = t(tau); op.band2gpt = np.array([[1, half], [half + 1, ng]], dtype=np.int32)
    src = pkg.SourceFuncLW()
    src.lay_source, src.lev_source_inc, src.lev_source_dec, src.sfc_source = t(lay), t(inc), t(dec), t(sfc)
    emis_d, incf_d = t(emis), t(incf)
    out = {}
    incs = (False,) if (f32 or shared) else (False, True)   # inc_flux: fp64, generic solver
    try:
        for split in (1, 0):
            pkg.set_solver_option("lw_tail_split", split)
            for with_inc in incs:
                fl = pkg.FluxesBroadband(torch.full((nlay + 1, ncol), -1., dtype=dt, device=gpu),
                                         torch.full((nlay + 1, ncol), -1., dtype=dt, device=gpu))
                kw = dict(inc_flux=incf_d) if with_inc else {}
                assert pkg.rte_lw(op, top_at_1, src, emis_d, fl, n_gauss_angles=nmus, shared_levels=shared, **kw) == ""
                out[split, with_inc] = (fl.flux_up.cpu().numpy(), fl.flux_dn.cpu().numpy())
    finally:
        pkg.set_solver_option("lw_tail_split", 1)
    for with_inc in incs:
        assert np.array_equal(out[1, with_inc][0], out[0, with_inc][0])
        assert np.array_equal(out[1, with_inc][1], out[0, with_inc][1])
        assert np.all(out[1, with_inc][0] > 0)
    if len(incs) == 2:
        assert not np.array_equal(out[1, True][1], out[1, False][1])


@pytest.mark.parametrize("ncol,nlay,ng,top_at_1,clamp", [
    (500, 60, 27, True, 0),       # fewer tiles than resident waves: every tile is a tail tile
    (50000, 60, 27, True, 0),     # one full round of 3 072 tiles + 53 tail tiles: no split beyond one round (option without effect)
    (9000, 60, 27, False, 0),     # 563 tiles: the largest calls that still split (partial sums below 64 MiB)
    (1003, 37, 14, False, 1),     # other layer count, bottom-up arrays, partly empty last tile and last g-point group
])
def test_rte_sw_tail_split_is_bit_identical(pkg, gpu, ncol, nlay, ng, top_at_1, clamp):
    """rte_sw, two-pass kernel (sw_solver = 1), with the tail split ("sw_tail_split", default) against whole-tile waves: the
    same bits in flux_up, flux_dn and flux_dn_dir, in both arithmetic modes.  (The layer-systolic default sums the
    g-points of SMALL calls in chunks, which is not bit-neutral: tests/test_gpu_round3.py.)"""
    import torch
    pkg.set_solver_option("sw_solver", 1)
    rng = np.random.default_rng(ncol)
    tau = rng.uniform(0.001, 2.0, (ng, nlay, ncol)); ssa = rng.uniform(0.0, 0.999, (ng, nlay, ncol))
    g = rng.uniform(0.0, 0.8, (ng, nlay, ncol)) * (rng.uniform(size=(1, 1, ncol)) < 0.5)
    mu0 = rng.uniform(0.05, 1.0, ncol); toa = rng.uniform(10, 100, (ng, ncol))
    albd, albf = rng.uniform(0.05, 0.4, (ncol, 2)), rng.uniform(0.05, 0.4, (ncol, 2))
    half = ng // 2
    t = T(gpu)
    op = pkg.OpticalProps2str(); op.tau, op.ssa, op.g = t(tau), t(ssa), t(g)
    op.band2gpt = np.array([[1, half], [half + 1, ng]], dtype=np.int32)
    args = (t(mu0), t(toa), t(albd), t(albf))
    pkg.set_solver_option("sw_dir_clamp", clamp)
    try:
        for arith in (0, 1):   # fast, reference expression order
            pkg.set_arithmetic(arith)
            out = []
            for split in (1, 0):
                pkg.set_solver_option("sw_tail_split", split)
                fl = pkg.FluxesBroadband(*(torch.full((nlay + 1, ncol), -1., dtype=torch.float64, device=gpu) for _ in range(3)))
                assert pkg.rte_sw(op, top_at_1, *args, fl) == ""
                out.append([f.cpu().numpy() for f in (fl.flux_up, fl.flux_dn, fl.flux_dn_dir)])
            for a, b in zip(*out):
                assert np.array_equal(a, b)
            assert np.all(out[0][1] >= 0) and np.all(out[0][0] >= 0)
    finally:
        pkg.set_arithmetic(0)
        pkg.set_solver_option("sw_tail_split", 1)
        pkg.set_solver_option("sw_dir_clamp", 0)
        pkg.set_solver_option("sw_solver", 0)
