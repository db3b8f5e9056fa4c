"""GPU parity tests added in round 3 (all through the C ABI, against the CPU oracle or against properties):

  * the layer-systolic shortwave solver (kernels_rte_sw_sys.hip, the default up to 60 layers) against the two-pass kernel
    and the oracle: layer counts, orientations, ragged tiles, per-band albedos, direct flux, solver switches, both
    arithmetic modes; the tail split (bit-identical); position independence; graph capture without warm-up;
  * host threads sharing the solver scratch pool.

Tolerances as in test_gpu_parity.py: fluxes 1e-9 W m-2 against the oracle (north_star: 1e-6).  The solvers restate
RTE-RRTMGP, which is not in the reference tree: parity unpinned."""
import threading

import numpy as np
import pytest

from test_gpu_round2 import T, FLUX_ATOL

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def default_options(pkg):
    pkg.reset_solver_options()
    pkg.set_solver_option("sw_solver", 0)
    pkg.set_solver_option("sw_tail_split", 1)
    yield
    pkg.reset_solver_options()
    pkg.set_solver_option("sw_solver", 0)
    pkg.set_solver_option("sw_tail_split", 1)
    pkg.set_arithmetic(pkg.FAST)


def sw_inputs(rng, ncol, nlay, ng, nband=2, g_zero=False):
    tau = rng.uniform(0.001, 2.0, (ng, nlay, ncol))
    ssa = rng.uniform(0.0, 0.999, (ng, nlay, ncol))
    g = np.zeros((ng, nlay, ncol)) if g_zero else rng.uniform(0.0, 0.8, (ng, nlay, ncol)) * (rng.uniform(size=(1, 1, ncol)) < 0.5)
    mu0 = rng.uniform(0.05, 1.0, ncol)
    toa = rng.uniform(10, 100, (ng, ncol))
    albd, albf = rng.uniform(0.05, 0.4, (ncol, nband)), rng.uniform(0.05, 0.4, (ncol, nband))
    edges = np.linspace(0, ng, nband + 1).astype(int)
    b2g = np.array([[edges[b] + 1, edges[b + 1]] for b in range(nband)], dtype=np.int32)
    return tau, ssa, g, mu0, toa, albd, albf, b2g


def run_sw(pkg, gpu, inp, top_at_1, with_dir=True):
    import torch
    tau, ssa, g, mu0, toa, albd, albf, b2g = inp
    ng, nlay, ncol = tau.shape
    t = T(gpu)
    op = pkg.OpticalProps2str(); op.tau, op.ssa, op.g = t(tau), t(ssa), t(g)
    op.band2gpt = b2g
    n = 3 if with_dir else 2
    fl = pkg.FluxesBroadband(*(torch.full((nlay + 1, ncol), -1., dtype=torch.float64, device=gpu) for _ in range(n)))
    assert pkg.rte_sw(op, top_at_1, t(mu0), t(toa), t(albd), t(albf), fl) == ""
    torch.cuda.synchronize()
    out = [fl.flux_up.cpu().numpy(), fl.flux_dn.cpu().numpy()]
    if with_dir:
        out.append(fl.flux_dn_dir.cpu().numpy())
    return out


def oracle_sw(oracle_mod, inp, top_at_1, options=None):
    tau, ssa, g, mu0, toa, albd, albf, b2g = inp
    ng = tau.shape[0]
    band = np.zeros(ng, dtype=int)
    for b, (lo, hi) in enumerate(b2g):
        band[lo - 1:hi] = b
    return oracle_mod.rte_sw(tau, ssa, g, mu0, toa, albd[:, band].T.copy(), albf[:, band].T.copy(), top_at_1=top_at_1, options=options)


@pytest.mark.parametrize("ncol,nlay,ng,top_at_1", [
    (1, 60, 27, True), (63, 60, 27, False), (65, 60, 5, True), (1000, 60, 27, True), (20000, 60, 27, False),
    (300, 1, 3, True), (300, 4, 7, False), (300, 5, 2, True), (700, 47, 14, True), (700, 59, 9, False), (130, 6, 1, True),
])
def test_systolic_rte_sw_vs_oracle_and_two_pass(pkg, gpu, oracle_mod, ncol, nlay, ng, top_at_1):
    """The layer-systolic solver against the oracle (1e-9 W m-2) and against the two-pass kernel (same arithmetic per
    (column, g-point); the g-point sums are ordered differently: a few ulp), with the tail split (tail tiles one
    g-point per block, summed in order: the same bits) switched on and off, in both arithmetic modes."""
    rng = np.random.default_rng(1000 * ncol + nlay)
    inp = sw_inputs(rng, ncol, nlay, ng, nband=min(2, ng))
    ref = oracle_sw(oracle_mod, inp, top_at_1)
    scale = max(1.0, float(np.max(ref[1])) / 1000.0)
    for arith in (pkg.FAST, pkg.REFERENCE_ORDER):
        pkg.set_arithmetic(arith)
        res = {}
        for solver, split in ((0, 1), (0, 0), (1, 0)):
            pkg.set_solver_option("sw_solver", solver)
            pkg.set_solver_option("sw_tail_split", split)
            res[solver, split] = run_sw(pkg, gpu, inp, top_at_1)
        for key, out in res.items():
            for a, b in zip(out, ref):
                assert np.max(np.abs(a - b)) < FLUX_ATOL * scale, (key, arith)
        for a, b, c in zip(res[0, 1], res[0, 0], res[1, 0]):
            assert np.array_equal(a, b) and np.allclose(b, c, rtol=1e-13, atol=1e-12)


def test_systolic_rte_sw_columns_do_not_depend_on_their_position(pkg, gpu):
    """A column's fluxes are the same bits wherever it sits in a call (shuffle), whatever the size of the call and whether
    or not it falls into a tail tile; calls repeat bit for bit."""
    rng = np.random.default_rng(5)
    ncol, nlay, ng = 3000, 60, 27
    inp = sw_inputs(rng, ncol, nlay, ng, g_zero=True)
    perm = rng.permutation(ncol)
    shuf = tuple(x[..., perm] if i < 5 else (x[perm] if i < 7 else x) for i, x in enumerate(inp))
    for split in (1, 0):
        pkg.set_solver_option("sw_tail_split", split)
        a = run_sw(pkg, gpu, inp, True)
        b = run_sw(pkg, gpu, inp, True)
        c = run_sw(pkg, gpu, shuf, True)
        for x, y, z in zip(a, b, c):
            assert np.array_equal(x, y) and np.array_equal(x[:, perm], z)
    big = sw_inputs(np.random.default_rng(6), 40000, nlay, ng, g_zero=True)
    whole = run_sw(pkg, gpu, big, True)                          # more tiles than CUs: sequential sum
    part = tuple(x[..., 17000:18000] if i < 5 else (x[17000:18000] if i < 7 else x) for i, x in enumerate(big))
    piece = run_sw(pkg, gpu, part, True)
    for x, y in zip(whole, piece):
        assert np.array_equal(x[:, 17000:18000], y)


def test_systolic_rte_sw_switches_and_no_direct_flux(pkg, gpu, oracle_mod):
    """Version switches (direct-beam clamps, k floor) on the layer-systolic solver against the oracle with the same
    switches; a call without flux_dir gives the same flux_up / flux_dn."""
    rng = np.random.default_rng(21)
    inp = sw_inputs(rng, 500, 60, 9)
    inp[1][:, ::7, :] = 1.0 - 1e-9     # nearly conservative layers: the k floor and the clamps matter
    for clamp, kfl in ((0, 1e-12), (1, 1e-12), (1, 1e-4)):
        pkg.set_solver_option("sw_dir_clamp", clamp)
        pkg.set_solver_option("sw_k_floor", kfl)
        opt = oracle_mod.solver_options()
        opt.sw_dir_clamp, opt.sw_k_floor = clamp, kfl
        ref = oracle_sw(oracle_mod, inp, True, options=opt)
        scale = max(1.0, float(np.max(ref[1])) / 1000.0)
        out = run_sw(pkg, gpu, inp, True)
        for a, b in zip(out, ref):
            assert np.max(np.abs(a - b)) < FLUX_ATOL * scale * 10, (clamp, kfl)
        two = run_sw(pkg, gpu, inp, True, with_dir=False)
        assert np.array_equal(two[0], out[0]) and np.array_equal(two[1], out[1])


def test_systolic_rte_sw_captured_without_warm_up(pkg, gpu):
    """The layer-systolic solver keeps nothing in global scratch: a call is captured in a HIP graph on a stream that
    has never seen it (whole tiles inside the capture -- the partial sums of the optional tail split are not allocated
    there -- and the same bits as the eager call with the split)."""
    import torch
    rng = np.random.default_rng(3)
    inp = sw_inputs(rng, 20000, 60, 9)       # 313 tiles: one round of 256 and 57 tail tiles
    tau, ssa, g, mu0, toa, albd, albf, b2g = inp
    t = T(gpu)
    op = pkg.OpticalProps2str(); op.tau, op.ssa, op.g = t(tau), t(ssa), t(g)
    op.band2gpt = b2g
    args = (t(mu0), t(toa), t(albd), t(albf))
    fl = pkg.FluxesBroadband(*(torch.zeros((61, 20000), dtype=torch.float64, device=gpu) for _ in range(3)))
    assert pkg.rte_sw(op, True, *args, fl) == ""
    torch.cuda.synchronize()
    ref = fl.flux_up.clone()
    fresh = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    fl.flux_up.zero_()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=fresh):
        assert pkg.rte_sw(op, True, *args, fl) == ""
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(fl.flux_up, ref)


def test_host_threads_share_the_solver_scratch(pkg, gpu, oracle_mod):
    """two host threads call rte_sw (two-pass kernel: always needs the scratch ring) through ECCKD_HOST with
    growing column counts, so that each keeps outgrowing the block the other may be about to launch on; every call
    must still return the oracle's fluxes."""
    pkg.set_solver_option("sw_solver", 1)
    sizes = [40, 700, 90, 2500, 300, 6000, 1500, 9000]
    cases = {}
    for n in sizes:
        inp = sw_inputs(np.random.default_rng(n), n, 60, 6, nband=1)
        cases[n] = (inp, oracle_sw(oracle_mod, inp, True))
    errors = []

    def worker(order):
        import numpy as np
        try:
            for n in order:
                inp, ref = cases[n]
                tau, ssa, g, mu0, toa, albd, albf, b2g = inp
                op = pkg.OpticalProps2str(); op.tau, op.ssa, op.g = tau, ssa, g
                op.band2gpt = b2g
                fl = pkg.FluxesBroadband(np.empty((61, n)), np.empty((61, n)), np.empty((61, n)))
                msg = pkg.rte_sw(op, True, mu0, toa, albd, albf, fl, device=0)
                if msg:
                    errors.append(msg)
                elif max(np.max(np.abs(fl.flux_up - ref[0])), np.max(np.abs(fl.flux_dn - ref[1]))) > FLUX_ATOL:
                    errors.append("wrong fluxes for %d columns" % n)
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(sizes * 3,)), threading.Thread(target=worker, args=(sizes[::-1] * 3,))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    pkg.release_scratch(0)


# ------------------------------------------------------------------------------------------------
# fused shortwave path (ecckd_sw_fluxes), single precision outside the longwave fused path
# ------------------------------------------------------------------------------------------------
SW_NAMES = ["co2", "ch4", "n2o", "o2", "h2o", "o3"]


@pytest.fixture(scope="module")
def sw(pkg, gpu, oracle_mod):
    from conftest import SW_WIDE
    k = pkg.GasOpticsEcckd()
    assert k.load(SW_WIDE, device=0) == ""
    return k, oracle_mod.CkdModel(SW_WIDE)


def sw_columns(k, c0, ncol, rng):
    from rte_ecckd_amd import synthetic
    cols = synthetic.columns(c0, ncol, k.get_press_min(), shortwave=True)
    nband = k.get_nband()
    cols["alb_dir"] = rng.uniform(0.02, 0.6, (ncol, nband))
    cols["alb_dif"] = rng.uniform(0.02, 0.6, (ncol, nband))
    cols["scale"] = rng.uniform(0.97, 1.03, ncol)
    return cols


def sw_api_path(pkg, k, cols, to, dtype, top_at_1=True, scale=False):
    """gas_optics + (driver-side rescaling of toa_flux) + rte_sw through the API objects; returns numpy fluxes."""
    import helpers
    ncol = cols["plev"].shape[1]
    nlay, ng = cols["tlay"].shape[0], k.get_ngpt()
    gc = helpers.product_gas_concs(pkg, cols, to, SW_NAMES)
    like = to(np.zeros(1, dtype=dtype))
    op = pkg.OpticalProps2str(); op.alloc_2str(ncol, nlay, k, like=like)
    toa = to(np.empty((ng, ncol), dtype=dtype))
    assert k.gas_optics(None, to(cols["plev"]), to(cols["tlay"]), gc, op, toa) == ""
    if scale:
        toa = toa * to(cols["scale"])[None, :]           # ecckd_rfmip_sw.F90:126-133
    fl = pkg.FluxesBroadband(*(to(np.empty((nlay + 1, ncol), dtype=dtype)) for _ in range(3)))
    assert pkg.rte_sw(op, top_at_1, to(cols["mu0"]), toa, to(cols["alb_dir"]), to(cols["alb_dif"]), fl) == ""
    back = (lambda a: a.cpu().numpy()) if hasattr(fl.flux_up, "cpu") else (lambda a: a)
    return [back(fl.flux_up), back(fl.flux_dn), back(fl.flux_dn_dir)], op, toa


def sw_fused_path(pkg, k, cols, to, dtype, top_at_1=True, scale=False, with_dir=True):
    import helpers
    ncol = cols["plev"].shape[1]
    nlay = cols["tlay"].shape[0]
    gc = helpers.product_gas_concs(pkg, cols, to, SW_NAMES)
    fl = pkg.FluxesBroadband(*(to(np.full((nlay + 1, ncol), -1, dtype=dtype)) for _ in range(3 if with_dir else 2)))
    assert k.sw_fluxes(to(cols["plev"]), to(cols["tlay"]), gc, top_at_1, to(cols["mu0"]), to(cols["alb_dir"]), to(cols["alb_dif"]),
                       fl, toa_scale=to(cols["scale"]) if scale else None) == ""
    back = (lambda a: a.cpu().numpy()) if hasattr(fl.flux_up, "cpu") else (lambda a: a)
    return [back(fl.flux_up), back(fl.flux_dn)] + ([back(fl.flux_dn_dir)] if with_dir else [])


@pytest.mark.parametrize("ncol,top_at_1,scale", [(333, True, False), (1500, True, True), (20000, True, False), (700, False, True)])
def test_fused_sw_path(pkg, gpu, oracle_mod, sw, ncol, top_at_1, scale):
    """ecckd_sw_fluxes -- gas optics writes the total optical depth only, the solver derives ssa = tau_rayleigh/tau,
    g = 0 and the incoming beam as gas_optics_ext does (src/gas_optics_ecckd.f90:455-472) -- against gas_optics +
    rte_sw through the API (bit-identical fluxes: the same arithmetic per cell) and against the oracle pair; device
    and host arrays, with and without the direct flux, the drivers' rescaling of the incoming beam, and (the model's
    layers reversed) the bottom-up orientation."""
    import torch
    import helpers
    k, m = sw
    rng = np.random.default_rng(ncol)
    cols = sw_columns(k, 7 * ncol, ncol, rng)
    if not top_at_1:   # bottom-up arrays: reverse the vertical axis of every profile.  (gas_optics takes the layer mass from
        # plev(l+1) - plev(l), src/gas_optics_ecckd.f90:143,313 -- negative here, optical depths clamped or negative as in the
        # reference: this case only checks that the fused path follows the two calls bit for bit in this orientation too)
        for n in ("plev", "tlay", "tlev", "h2o", "o3"):
            cols[n] = np.ascontiguousarray(cols[n][::-1])
    t = T(gpu)
    api, op, toa = sw_api_path(pkg, k, cols, t, np.float64, top_at_1, scale)
    fused = sw_fused_path(pkg, k, cols, t, np.float64, top_at_1, scale)
    for a, b in zip(api, fused):
        assert np.array_equal(a, b, equal_nan=True)
    two = sw_fused_path(pkg, k, cols, t, np.float64, top_at_1, scale, with_dir=False)
    assert np.array_equal(two[0], fused[0], equal_nan=True) and np.array_equal(two[1], fused[1], equal_nan=True)
    if ncol <= 1500:
        host = sw_fused_path(pkg, k, cols, np.ascontiguousarray, np.float64, top_at_1, scale)
        for a, b in zip(host, fused):
            assert np.array_equal(a, b, equal_nan=True)
    if ncol <= 1500 and top_at_1:
        otau, ossa, og, otoa, oerr = oracle_mod.gas_optics_ext(m, cols["plev"], cols["tlay"], helpers.oracle_gas_items(cols, SW_NAMES))
        assert oerr == ""
        if scale:
            otoa = otoa * cols["scale"][None, :]
        g2b = m.gpt2band - 1
        ref = oracle_mod.rte_sw(otau, ossa, og, cols["mu0"], otoa, np.ascontiguousarray(cols["alb_dir"][:, g2b].T),
                                np.ascontiguousarray(cols["alb_dif"][:, g2b].T), top_at_1=top_at_1)
        for a, b in zip(fused, ref):   # (each side on its own optical properties: see test_random_sw_gas_descriptions_and_fluxes)
            assert np.max(np.abs(a - b)) < 10 * FLUX_ATOL
    pkg.set_arithmetic(pkg.REFERENCE_ORDER)
    gc = helpers.product_gas_concs(pkg, cols, t, SW_NAMES)
    fl = pkg.FluxesBroadband(*(torch.empty((61, ncol), dtype=torch.float64, device=gpu) for _ in range(2)))
    msg = k.sw_fluxes(t(cols["plev"]), t(cols["tlay"]), gc, top_at_1, t(cols["mu0"]), t(cols["alb_dir"]), t(cols["alb_dif"]), fl)
    assert "fast arithmetic" in msg


def test_single_precision_sw_path(pkg, gpu, oracle_mod, sw):
    """float32 arrays take ecckd_gas_optics_sw_f32 / ecckd_rte_sw_f32 / ecckd_sw_fluxes_f32 (a host built with wp = real32,
    src/gas_optics_ecckd.f90:6).  Against the fp64 oracle on the float32-rounded inputs, single-precision bars: tau 5e-5
    relative (where tau is not tiny), ssa 5e-5 absolute, fluxes 0.05 W m-2 in 99 % of the columns and 0.5 W m-2 (of up
    to 1 300) for the worst -- cells near the resonance k*mu0 = 1 of the two-stream direct terms lose digits to
    1 - (k*mu0)**2 in any single-precision evaluation; the fused path gives the fluxes of the two calls bit for bit."""
    import torch
    import helpers
    k, m = sw
    ncol, nlay, ng = 900, 60, 27
    rng = np.random.default_rng(2)
    cols = sw_columns(k, 11, ncol, rng)
    t32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(gpu)
    api, op, toa = sw_api_path(pkg, k, cols, t32, np.float32)
    assert op.tau.dtype == torch.float32 and api[0].dtype == np.float32
    r = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32), dtype=np.float64)
    c32 = {n: (r(v) if isinstance(v, np.ndarray) else v) for n, v in cols.items()}
    otau, ossa, og, otoa, oerr = oracle_mod.gas_optics_ext(m, c32["plev"], c32["tlay"], helpers.oracle_gas_items(c32, SW_NAMES))
    g2b = m.gpt2band - 1
    ref = oracle_mod.rte_sw(otau, ossa, og, c32["mu0"], otoa, np.ascontiguousarray(c32["alb_dir"][:, g2b].T),
                            np.ascontiguousarray(c32["alb_dif"][:, g2b].T))
    gt = op.tau.cpu().numpy().astype(np.float64)
    big = otau > 1e-6 * otau.max()
    assert np.max(np.abs(gt - otau)[big] / otau[big]) < 5e-5
    assert np.max(np.abs(op.ssa.cpu().numpy() - ossa)) < 5e-5 and bool((op.g == 0).all())
    assert np.max(np.abs(toa.cpu().numpy() - otoa)) < 1e-4
    for a, b in zip(api, ref):
        d = np.abs(a - b)
        assert np.max(d) < 0.5 and np.percentile(d.max(axis=0), 99) < 0.05     # (per column: all, and 99 % of them)
    fused = sw_fused_path(pkg, k, cols, t32, np.float32)
    for a, b in zip(api, fused):
        assert np.array_equal(a, b)


def test_single_precision_incident_flux(pkg, gpu, oracle_mod):
    """ecckd_rte_lw_inc_flux_f32 against the fp64 oracle on the same (float32-rounded) arrays: 2e-3 W m-2."""
    import torch
    rng = np.random.default_rng(12)
    ng, nlay, ncol = 6, 60, 500
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    tau, lay, inc, dec = f(rng.uniform(0.001, 1.5, (ng, nlay, ncol))), f(rng.uniform(1, 9, (ng, nlay, ncol))), \
        f(rng.uniform(1, 9, (ng, nlay, ncol))), f(rng.uniform(1, 9, (ng, nlay, ncol)))
    sfc, incf, emis = f(rng.uniform(1, 9, (ng, ncol))), f(rng.uniform(0, 5, (ng, ncol))), f(rng.uniform(0.9, 1.0, (ncol, 1)))
    t = lambda a: torch.from_numpy(a).to(gpu)
    op = pkg.OpticalProps1scl(); op.tau = t(tau); op.band2gpt = np.array([[1, ng]], dtype=np.int32)
    src = pkg.SourceFuncLW(); src.lay_source, src.lev_source_inc, src.lev_source_dec, src.sfc_source = t(lay), t(inc), t(dec), t(sfc)
    for nmus, top in ((1, True), (3, False)):
        fl = pkg.FluxesBroadband(torch.empty((nlay + 1, ncol), dtype=torch.float32, device=gpu),
                                 torch.empty((nlay + 1, ncol), dtype=torch.float32, device=gpu))
        assert pkg.rte_lw(op, top, src, t(emis), fl, n_gauss_angles=nmus, inc_flux=t(incf)) == ""
        d = lambda a: a.astype(np.float64)
        fu, fd = oracle_mod.rte_lw(d(tau), d(lay), d(inc), d(dec), np.repeat(d(emis).T, ng, 0), d(sfc), top_at_1=top, nmus=nmus,
                                   inc_flux=d(incf))
        assert np.max(np.abs(fl.flux_up.cpu().numpy() - fu)) < 2e-3 * nmus and np.max(np.abs(fl.flux_dn.cpu().numpy() - fd)) < 2e-3 * nmus


def test_full_size_sw_properties(pkg, gpu, oracle_mod, sw):
    """BASELINE configs[2] at full size (1e5 synthetic columns x 60 layers x 27 g-points, gas_optics + rte_sw): the oracle
    cannot run that in seconds, so size-independent properties -- oracle spot checks at both ends and in the tail tiles,
    calls repeat bit for bit, a column's fluxes do not depend on its position (shuffle) nor on the size of the call,
    energy conservation bounds, the mu0 edge (sun at the horizon) and the drivers' night columns (mu0 = 1, fluxes zeroed
    by the caller: ecckd_rfmip_sw.F90:143-145,156-161) stay finite."""
    import helpers
    k, m = sw
    ncol = 100000
    rng = np.random.default_rng(1)
    cols = sw_columns(k, 0, ncol, rng)
    cols["mu0"][:50] = 1e-3                 # sun at the horizon
    cols["mu0"][50:100] = 1.0               # what the drivers put in night columns
    t = T(gpu)
    out, op, toa = sw_api_path(pkg, k, cols, t, np.float64)
    again, _, _ = sw_api_path(pkg, k, cols, t, np.float64)
    for a, b in zip(out, again):
        assert np.array_equal(a, b)
    up, dn, dr = out
    assert np.all(np.isfinite(up)) and np.all(np.isfinite(dn)) and np.all(up >= 0) and np.all(dr >= 0) and np.all(dn >= dr)
    tsi = m.solar_irradiance.sum() if hasattr(m, "solar_irradiance") else k.get_total_solar_irradiance()
    assert np.all(dn[0] <= tsi * cols["mu0"] * (1 + 1e-12)) and np.all(up[0] <= dn[0] * (1 + 1e-12))   # nothing gains energy
    assert np.allclose(dn[0], tsi * cols["mu0"], rtol=1e-12)
    g2b = m.gpt2band - 1
    for lo, hi in ((0, 128), (49990, 50040), (ncol - 200, ncol)):      # first tile, middle, the tail tiles of the last round
        sl = slice(lo, hi)
        sub = {n: (np.ascontiguousarray(v[..., sl]) if isinstance(v, np.ndarray) and v.shape[-1] == ncol else v)
               for n, v in cols.items() if n not in ("alb_dir", "alb_dif")}
        otau, ossa, og, otoa, _ = oracle_mod.gas_optics_ext(m, sub["plev"], sub["tlay"], helpers.oracle_gas_items(sub, SW_NAMES))
        ref = oracle_mod.rte_sw(otau, ossa, og, sub["mu0"], otoa, np.ascontiguousarray(cols["alb_dir"][sl][:, g2b].T),
                                np.ascontiguousarray(cols["alb_dif"][sl][:, g2b].T))
        for a, b in zip(out, ref):
            assert np.max(np.abs(a[:, sl] - b)) < 10 * FLUX_ATOL
    # position and call size: 3000 columns taken from all over the call, shuffled, as a call of their own
    pick = rng.permutation(ncol)[:3000]
    sub = {}
    for n, v in cols.items():
        if isinstance(v, np.ndarray) and v.ndim == 2 and v.shape[1] == ncol:
            sub[n] = np.ascontiguousarray(v[:, pick])
        elif isinstance(v, np.ndarray) and v.shape[0] == ncol:
            sub[n] = np.ascontiguousarray(v[pick])
        else:
            sub[n] = v
    small, _, _ = sw_api_path(pkg, k, sub, t, np.float64)
    for a, b in zip(out, small):
        assert np.array_equal(a[:, pick], b)


def test_bench_launches_its_own_ranks(pkg, gpu, tmp_path):
    """`python bench.py --gpus 2` with NO launcher in front: the parent starts the ranks itself (torch.distributed.run as
    a child, before anything touches the GPU) and passes rank 0's JSON line through.  Here both ranks sit on cuda:0
    over gloo (--rehearse-on-one-gpu); on a node with N GPUs the same command runs one rank per GPU over RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k2: v for k2, v in os.environ.items() if k2 not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--ncol", "60000",
           "--cpu-seconds", "0", "--no-side", "--rehearse-on-one-gpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["ncol_total"] == 120000
    assert len(d["per_rank_ms_per_step"]["ranks"]) == 2 and d["check_max_abs_flux_diff_vs_oracle_Wm2"] < FLUX_ATOL


# ------------------------------------------------------------------------------------------------
# fp64 gas optics over the float32 image of the tables in LDS ("gas_slab_f32")
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("orography", ["default", "shuffled 50-103 kPa"])
def test_float32_table_image_gives_the_same_bits(pkg, gpu, oracle_mod, orography):
    """Every table of the ecCKD files is float32 on disk (widened exactly on read, mo_simple_netcdf.F90:44-142), so the
    fp64 longwave gas optics may stage them in LDS as float32 and widen them when it uses them: "gas_slab_f32" = 0 (fp64
    slab), 1 (float32 image: 8 instead of 3 pressure rows) and 2 (default: a probe picks per call) give the same bits in
    tau and in the Planck sources, on the default columns and on surface pressures shuffled over 50-103 kPa (where the
    probe picks the float32 image); a model whose tables are not float32 numbers never takes it."""
    import torch
    import helpers
    from conftest import LW_FSCK
    from rte_ecckd_amd import synthetic
    k = pkg.GasOpticsEcckd()
    assert k.load(LW_FSCK, device=0) == ""
    ncol, nlay = 6000, 60
    cols = synthetic.columns(123, ncol, k.get_press_min())
    if orography != "default":
        rng = np.random.default_rng(4)
        eta = (np.arange(nlay + 1, dtype=np.float64) / nlay) ** 2
        ptop = cols["plev"][0, 0]
        ps = 50000 + 53000 * rng.random(ncol)
        cols["plev"] = ptop + (ps[None, :] - ptop) * eta[:, None]
    out = {}
    try:
        for opt in (0, 1, 2):
            pkg.set_solver_option("gas_slab_f32", opt)
            err, tau, lay, inc, dec, sfc = helpers.run_lw_gas_optics(pkg, k, cols, gpu)
            assert err == ""
            out[opt] = (tau, lay, inc, dec, sfc)
    finally:
        pkg.set_solver_option("gas_slab_f32", 2)
    for opt in (1, 2):
        for a, b in zip(out[0], out[opt]):
            assert np.array_equal(a, b)
    m = oracle_mod.CkdModel(LW_FSCK)
    sub = {n: (np.ascontiguousarray(v[..., :256]) if isinstance(v, np.ndarray) and v.shape[-1] == ncol else v) for n, v in cols.items()}
    otau, olay, oinc, odec, osfc, _ = oracle_mod.gas_optics_int(m, sub["plev"], sub["tlay"], sub["tsfc"], helpers.oracle_gas_items(sub), sub["tlev"])
    assert helpers.max_rel(out[1][0][..., :256], otau) < 1e-12 and np.array_equal(out[1][1][..., :256], olay)


def test_fused_lw_path_with_a_64_g_point_model(pkg, gpu, oracle_mod):
    """the Planck-recomputing solver of ecckd_lw_fluxes keeps the model's Planck table in LDS; the 231 x 64
    table of a 64-g-point model does not fit next to its accumulators.  Such a model takes the general route (sources
    through library scratch) at 60 layers too -- same fluxes as gas_optics + rte_lw -- instead of failing at launch.
    (Tables that are not float32 numbers: "gas_slab_f32" never applies to this model.)"""
    import torch
    import helpers
    from conftest import LW_FSCK
    from rte_ecckd_amd import synthetic
    m = oracle_mod.CkdModel(LW_FSCK)
    rng = np.random.default_rng(64)
    ng = 64
    tabs = []
    for name, tb in zip(m.gas[:3], m.tables[:3]):
        c = tb["coefficient"]
        big = np.concatenate([c, c * rng.uniform(0.5, 1.5, c.shape)], axis=-1)            # (nv, nt, np, 64)
        tabs.append(dict(name=name, code=tb["code"], composite_only=0, mole_fraction=tb.get("mole_fraction"),
                         reference_mole_fraction=tb["reference_mole_fraction"], coefficient=big))
    planck = np.concatenate([m.planck_function, m.planck_function * 0.37], axis=-1)       # (ntp, 64)
    k = pkg.GasOpticsEcckd()
    assert k.init_from_tables(m.log_pressure, m.temperature, tabs, planck=(m.temperature_planck, planck), device=0) == ""
    assert k.get_ngpt() == ng
    ncol, nlay = 300, 60
    cols = synthetic.columns(9, ncol, float(np.exp(m.log_pressure[0])))
    t = T(gpu)
    names = list(m.gas[:3])
    gc = helpers.product_gas_concs(pkg, cols, t, names)
    plev, tlay, tlev, tsfc = t(cols["plev"]), t(cols["tlay"]), t(cols["tlev"]), t(cols["tsfc"])
    emis = t(cols["sfc_emis"][:, None])
    op = pkg.OpticalProps1scl(); op.alloc_1scl(ncol, nlay, k, like=plev)
    src = pkg.SourceFuncLW(); src.alloc(ncol, nlay, k, like=plev)
    assert k.gas_optics(None, plev, tlay, tsfc, gc, op, src, tlev=tlev) == ""
    fl = pkg.FluxesBroadband(*(torch.empty((nlay + 1, ncol), dtype=torch.float64, device=gpu) for _ in range(2)))
    assert pkg.rte_lw(op, True, src, emis, fl) == ""
    f2 = pkg.FluxesBroadband(*(torch.empty((nlay + 1, ncol), dtype=torch.float64, device=gpu) for _ in range(2)))
    assert k.lw_fluxes(plev, tlay, tsfc, tlev, gc, True, emis, f2) == ""
    assert float((fl.flux_up - f2.flux_up).abs().max()) < FLUX_ATOL and float((fl.flux_dn - f2.flux_dn).abs().max()) < FLUX_ATOL
    assert float(fl.flux_up.min()) > 0


def test_tail_scratch_size_queries_and_a_caller_owned_block(pkg, gpu):
    """ecckd_rte_sw_tail_scratch_bytes / ecckd_rte_lw_tail_scratch_bytes say what the optional tail splits of a
    call would take; a caller-owned block of exactly that size serves the call (captured in a graph without a warm-up),
    one that is a byte short makes the split step aside -- the same fluxes either way."""
    import torch
    rng = np.random.default_rng(17)
    ncol, nlay, ng = 1000, 60, 9
    need = pkg.rte_sw_tail_scratch_bytes(ncol, nlay, ng)
    assert need == 8 * 3 * (nlay + 1) * 64 * 16 * ng                       # 16 tiles, one unit per (tile, g-point)
    assert pkg.rte_sw_tail_scratch_bytes(256 * 64, nlay, ng) == 0           # whole rounds only: nothing to split
    assert pkg.rte_lw_tail_scratch_bytes(100000, 60, 32) > 0 and pkg.rte_lw_tail_scratch_bytes(1024 * 32, 60, 32) == 0
    pkg.set_solver_option("sw_tail_split", 0)
    assert pkg.rte_sw_tail_scratch_bytes(ncol, nlay, ng) == 0
    pkg.set_solver_option("sw_tail_split", 1)
    inp = sw_inputs(rng, ncol, nlay, ng)
    ref = run_sw(pkg, gpu, inp, True)
    tau, ssa, g, mu0, toa, albd, albf, b2g = inp
    t = T(gpu)
    op = pkg.OpticalProps2str(); op.tau, op.ssa, op.g = t(tau), t(ssa), t(g)
    op.band2gpt = b2g
    args = (t(mu0), t(toa), t(albd), t(albf))
    for size in (need, need - 1):
        stream = torch.cuda.Stream()
        buf = torch.empty(size, dtype=torch.uint8, device=gpu)
        pkg.set_stream_scratch(buf, stream=stream)
        fl = pkg.FluxesBroadband(*(torch.zeros((nlay + 1, ncol), dtype=torch.float64, device=gpu) for _ in range(3)))
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=stream):
            assert pkg.rte_sw(op, True, *args, fl) == ""
        graph.replay()
        torch.cuda.synchronize()
        for a, b in zip(ref, (fl.flux_up, fl.flux_dn, fl.flux_dn_dir)):
            assert np.array_equal(a, b.cpu().numpy())
        pkg.set_stream_scratch(None, stream=stream)
        del graph


def test_bench_rccl_branch_with_one_rank(pkg, gpu, tmp_path):
    """The RCCL branch of bench.py (init_process_group("nccl", device_id=...), barrier(device_ids=...), all_reduce(MAX),
    all_gather on device tensors) executed for real: one rank under torch.distributed.run on the one GPU of the box.  (Two
    ranks cannot share a device over RCCL -- the two-rank tests rehearse the rank logic over gloo.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k2: v for k2, v in os.environ.items() if k2 not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--ncol", "60000",
           "--cpu-seconds", "0", "--no-side"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and len(d["per_rank_ms_per_step"]["ranks"]) == 1 and "REHEARSAL" not in d["config"]["parallelism"]
    assert d["check_max_abs_flux_diff_vs_oracle_Wm2"] < FLUX_ATOL


# ------------------------------------------------------------------------------------------------
# late round 3: the written-out exp / sqrt / division of the solvers at the edges of their ranges
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_shortwave_solver_extreme_and_nan_columns(pkg, gpu, oracle_mod):
    """The fast arithmetic mode evaluates sqrt and exp without the device library's range handling (sw_two_stream.hpp) and
    hands the adding recurrence on in its projective form: optically black layers (tau 1e3 ... 1e30), nearly transparent
    ones (1e-12), nearly conservative scattering and a grazing sun stay within the flux tolerance of the oracle; a NaN or inf
    optical depth poisons its own column only, and the other columns keep their bits."""
    rng = np.random.default_rng(77)
    ncol, nlay, ng = 200, 60, 9
    inp = list(sw_inputs(rng, ncol, nlay, ng, nband=2, g_zero=True))
    tau, ssa, mu0 = inp[0], inp[1], inp[3]
    tau[:, 10:13, 0] = 1.0e3
    tau[:, 20, 1] = 1.0e30
    tau[:, :, 2] = 1.0e-12
    ssa[:, :, 3] = 1.0 - 1.0e-13
    mu0[4] = 1.0e-3
    tau[:, :, 5] = 50.0; ssa[:, :, 5] = 0.999999
    ref = oracle_sw(oracle_mod, inp, True)
    scale = max(1.0, float(np.max(ref[1])) / 1000.0)
    for arith in (pkg.FAST, pkg.REFERENCE_ORDER):
        pkg.set_arithmetic(arith)
        out = run_sw(pkg, gpu, inp, True)
        for a, b in zip(out, ref):
            assert np.all(np.isfinite(a)) and np.max(np.abs(a - b)) < FLUX_ATOL * scale * 10, arith
    pkg.set_arithmetic(pkg.FAST)
    clean = run_sw(pkg, gpu, inp, True)
    bad = [x.copy() for x in inp]
    bad[0][3, 17, 7] = np.nan
    bad[0][5, 40, 8] = np.inf
    out = run_sw(pkg, gpu, bad, True)
    keep = np.ones(ncol, bool); keep[[7, 8]] = False
    for a, b in zip(out, clean):
        assert np.array_equal(a[:, keep], b[:, keep])
    assert np.all(np.isnan(out[0][:, 7])) and np.any(np.isnan(out[1][:, 7]))    # NaN optical depth: NaN column
    oref = oracle_sw(oracle_mod, bad, True)
    for a, b in zip(out, oref):                                                  # inf optical depth: what the oracle gives
        assert np.array_equal(np.isnan(a[:, 8]), np.isnan(b[:, 8]))
        ok = ~np.isnan(b[:, 8])
        assert np.max(np.abs(a[ok, 8] - b[ok, 8]), initial=0.0) < FLUX_ATOL * scale * 10


@pytest.mark.gpu
def test_longwave_solver_extreme_and_nan_columns(pkg, gpu, oracle_mod):
    """rte_lw divides and exponentiates with the sequences of lw_layer.hpp: optical depths from 1e-300 (series branch) over
    the threshold of the series to 1e30 stay within the flux tolerance of the oracle; NaN / inf optical depths poison their
    own column as the oracle's do."""
    from test_gpu_round2 import lw_objects
    rng = np.random.default_rng(78)
    ncol, nlay, ng = 96, 60, 8
    tau = rng.uniform(1e-3, 3.0, (ng, nlay, ncol))
    lay = rng.uniform(1.0, 10.0, (ng, nlay, ncol))
    lev = rng.uniform(1.0, 10.0, (ng, nlay + 1, ncol))
    inc, dec = np.ascontiguousarray(lev[:, 1:, :]), np.ascontiguousarray(lev[:, :-1, :])
    sfc = rng.uniform(1.0, 10.0, (ng, ncol))
    emis = rng.uniform(0.9, 1.0, (1, ncol))
    tau[:, :, 0] = 1e-300
    tau[:, :, 1] = 1.4e-8 / 1.66          # just below / above the series threshold sqrt(eps) after the secant
    tau[:, :, 2] = 1.6e-8 / 1.66
    tau[:, 30, 3] = 1e30
    tau[:, 10:20, 4] = 800.0
    fu, fd = oracle_mod.rte_lw(tau, lay, inc, dec, np.repeat(emis, ng, 0), sfc, top_at_1=True, nmus=1)
    t = T(gpu)
    op, src, fl = lw_objects(pkg, gpu, tau, lay, inc, dec, sfc)
    assert pkg.rte_lw(op, True, src, t(emis.T.copy()), fl, n_gauss_angles=1) == ""
    gu, gd = fl.flux_up.cpu().numpy().copy(), fl.flux_dn.cpu().numpy().copy()
    assert np.all(np.isfinite(gu)) and np.max(np.abs(gu - fu)) < FLUX_ATOL and np.max(np.abs(gd - fd)) < FLUX_ATOL
    tau2 = tau.copy()
    tau2[2, 5, 9] = np.nan
    tau2[4, 50, 11] = np.inf
    fu2, fd2 = oracle_mod.rte_lw(tau2, lay, inc, dec, np.repeat(emis, ng, 0), sfc, top_at_1=True, nmus=1)
    op, src, fl = lw_objects(pkg, gpu, tau2, lay, inc, dec, sfc)
    assert pkg.rte_lw(op, True, src, t(emis.T.copy()), fl, n_gauss_angles=1) == ""
    hu, hd = fl.flux_up.cpu().numpy(), fl.flux_dn.cpu().numpy()
    keep = np.ones(ncol, bool); keep[[9, 11]] = False
    assert np.array_equal(hu[:, keep], gu[:, keep]) and np.array_equal(hd[:, keep], gd[:, keep])
    for a, b in ((hu, fu2), (hd, fd2)):
        assert np.array_equal(np.isnan(a), np.isnan(b))
        ok = ~np.isnan(b)
        assert np.max(np.abs(a[ok] - b[ok])) < FLUX_ATOL


@pytest.mark.gpu
def test_single_precision_fluxes_byband(pkg, gpu, oracle_mod):
    """float32 arrays with a FluxesByband take ecckd_rte_lw_byband_f32 / ecckd_rte_sw_byband_f32: per-band fluxes against the
    fp64 oracle run on each band's g-points of the float32-rounded inputs (single-precision bars: LW 2e-3 W m-2 per band,
    SW 0.05 W m-2 in 99 % of the columns, 0.5 in the worst), band sums against the broadband float32 call, host arrays
    against device arrays bit for bit."""
    import torch
    rng = np.random.default_rng(79)
    ng, nlay, ncol = 11, 60, 150
    b2g = np.array([[1, 2], [3, 3], [4, 8], [9, 11]], dtype=np.int32)
    nband = b2g.shape[0]
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    d = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    t = lambda a: torch.from_numpy(f(a)).to(gpu)
    z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=gpu)
    # ---- longwave ----
    tau = f(rng.uniform(0, 2, (ng, nlay, ncol)))
    lay, inc, dec = (f(rng.uniform(1, 9, (ng, nlay, ncol))) for _ in range(3))
    sfc = f(rng.uniform(1, 9, (ng, ncol)))
    emis = f(rng.uniform(0.7, 1.0, (ncol, nband)))
    op = pkg.OpticalProps1scl(); op.tau = t(tau); op.band2gpt = b2g
    src = pkg.SourceFuncLW()
    src.lay_source, src.lev_source_inc, src.lev_source_dec, src.sfc_source = t(lay), t(inc), t(dec), t(sfc)
    bb = pkg.FluxesBroadband(z(nlay + 1, ncol), z(nlay + 1, ncol))
    assert pkg.rte_lw(op, True, src, t(emis), bb, n_gauss_angles=2) == ""
    fb = pkg.FluxesByband(z(nband, nlay + 1, ncol), z(nband, nlay + 1, ncol), flux_up=z(nlay + 1, ncol), flux_dn=z(nlay + 1, ncol))
    assert pkg.rte_lw(op, True, src, t(emis), fb, n_gauss_angles=2) == ""
    assert fb.bnd_flux_up.dtype == torch.float32
    for b, (lo, hi) in enumerate(b2g):
        sl = slice(lo - 1, hi)
        fu, fd = oracle_mod.rte_lw(d(tau[sl]), d(lay[sl]), d(inc[sl]), d(dec[sl]), np.repeat(d(emis)[None, :, b], hi - lo + 1, 0),
                                   d(sfc[sl]), nmus=2)
        assert np.max(np.abs(fb.bnd_flux_up[b].cpu().numpy() - fu)) < 2e-3 * (hi - lo + 1)
        assert np.max(np.abs(fb.bnd_flux_dn[b].cpu().numpy() - fd)) < 2e-3 * (hi - lo + 1)
    assert torch.allclose(fb.flux_up, fb.bnd_flux_up.sum(0), rtol=1e-6, atol=1e-3)
    assert torch.allclose(fb.flux_up, bb.flux_up, rtol=1e-6, atol=2e-3) and torch.allclose(fb.flux_dn, bb.flux_dn, rtol=1e-6, atol=2e-3)
    hb = pkg.FluxesByband(np.zeros((nband, nlay + 1, ncol), np.float32), np.zeros((nband, nlay + 1, ncol), np.float32),
                          flux_up=np.zeros((nlay + 1, ncol), np.float32))
    oph = pkg.OpticalProps1scl(); oph.tau = tau; oph.band2gpt = b2g
    sh = pkg.SourceFuncLW(); sh.lay_source, sh.lev_source_inc, sh.lev_source_dec, sh.sfc_source = lay, inc, dec, sfc
    assert pkg.rte_lw(oph, True, sh, emis, hb, n_gauss_angles=2) == ""
    assert np.array_equal(hb.bnd_flux_up, fb.bnd_flux_up.cpu().numpy())
    assert np.allclose(hb.flux_up, fb.flux_up.cpu().numpy(), rtol=1e-6, atol=1e-3)
    # ---- shortwave ----
    ssa = f(rng.uniform(0, 1, (ng, nlay, ncol))); gg = f(rng.uniform(-0.3, 0.8, (ng, nlay, ncol)))
    mu0 = f(rng.uniform(0.1, 1.0, ncol)); toa = f(rng.uniform(1, 50, (ng, ncol)))
    adir = f(rng.uniform(0.05, 0.4, (ncol, nband))); adif = f(rng.uniform(0.05, 0.4, (ncol, nband)))
    op2 = pkg.OpticalProps2str(); op2.tau, op2.ssa, op2.g, op2.band2gpt = t(tau), t(ssa), t(gg), b2g
    sb = pkg.FluxesBroadband(z(nlay + 1, ncol), z(nlay + 1, ncol), z(nlay + 1, ncol))
    assert pkg.rte_sw(op2, True, t(mu0), t(toa), t(adir), t(adif), sb) == ""
    fs = pkg.FluxesByband(z(nband, nlay + 1, ncol), z(nband, nlay + 1, ncol), z(nband, nlay + 1, ncol),
                          flux_up=z(nlay + 1, ncol), flux_dn=z(nlay + 1, ncol), flux_dn_dir=z(nlay + 1, ncol))
    assert pkg.rte_sw(op2, True, t(mu0), t(toa), t(adir), t(adif), fs) == ""
    for b, (lo, hi) in enumerate(b2g):
        sl = slice(lo - 1, hi)
        n = hi - lo + 1
        ref = oracle_mod.rte_sw(d(tau[sl]), d(ssa[sl]), d(gg[sl]), d(mu0), d(toa[sl]), np.repeat(d(adir)[None, :, b], n, 0),
                                np.repeat(d(adif)[None, :, b], n, 0))
        for got, want in zip((fs.bnd_flux_up[b], fs.bnd_flux_dn[b], fs.bnd_flux_dn_dir[b]), ref):
            dd = np.abs(got.cpu().numpy() - want)
            assert np.max(dd) < 0.5 and np.percentile(dd.max(axis=0), 99) < 0.05
    for got, want in ((fs.flux_up, sb.flux_up), (fs.flux_dn, sb.flux_dn), (fs.flux_dn_dir, sb.flux_dn_dir)):
        assert torch.allclose(got, want, rtol=1e-5, atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("top_at_1,f32", [(True, False), (False, False), (True, True)])
def test_rte_lw_lane_offsets_32_and_64_bit(pkg, gpu, monkeypatch, top_at_1, f32):
    """The 60-layer solver addresses its inputs with 32-bit lane offsets on wave-uniform plane pointers whenever two planes
    span less than 4 GiB, and with 64-bit offsets beyond (calls of more than 4.4e6 columns, which no test can hold): the
    environment switch forces the 64-bit form, and both give the same bits (ragged column count, both orientations, fp32)."""
    import torch
    from test_gpu_round2 import lw_objects
    rng = np.random.default_rng(80)
    ncol, nlay, ng = 1000 + 13, 60, 7
    dt = np.float32 if f32 else np.float64
    tau = rng.uniform(1e-3, 3.0, (ng, nlay, ncol)).astype(dt)
    lay = rng.uniform(1.0, 10.0, (ng, nlay, ncol)).astype(dt)
    lev = rng.uniform(1.0, 10.0, (ng, nlay + 1, ncol)).astype(dt)
    inc, dec = np.ascontiguousarray(lev[:, 1:, :]), np.ascontiguousarray(lev[:, :-1, :])
    sfc = rng.uniform(1.0, 10.0, (ng, ncol)).astype(dt)
    emis = rng.uniform(0.9, 1.0, (ncol, 1)).astype(dt)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    res = []
    for force64 in (False, True):
        if force64:
            monkeypatch.setenv("ECCKD_LW_NO_OFF32", "1")
        op = pkg.OpticalProps1scl(); op.tau = tt(tau); op.band2gpt = np.array([[1, ng]], dtype=np.int32)
        src = pkg.SourceFuncLW()
        src.lay_source, src.lev_source_inc, src.lev_source_dec, src.sfc_source = tt(lay), tt(inc), tt(dec), tt(sfc)
        tdt = torch.float32 if f32 else torch.float64
        fl = pkg.FluxesBroadband(torch.zeros((nlay + 1, ncol), dtype=tdt, device=gpu), torch.zeros((nlay + 1, ncol), dtype=tdt, device=gpu))
        assert pkg.rte_lw(op, top_at_1, src, tt(emis), fl, n_gauss_angles=2) == ""
        res.append((fl.flux_up.cpu().numpy().copy(), fl.flux_dn.cpu().numpy().copy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.all(np.isfinite(res[0][0])) and float(res[0][0].max()) > 1.0
