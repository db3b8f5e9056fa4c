"""CPU tests of the oracle itself: it is pinned against the known-answer values that SURVEY.md
§8(a) records from the unmodified reference module, checked against the committed golden
fixture, and the solver half (parity unpinned: RTE-RRTMGP is absent) against analytic cases."""
import json
import os

import numpy as np
import pytest

from conftest import LW_FSCK, LW_RRTMGP, SW_WIDE

HERE = os.path.dirname(os.path.abspath(__file__))


def kat_inputs():
    j = np.arange(1, 62)
    x = (j - 1) / 60.0
    plev = 1 + (101325 - 1) * x ** 2 * (1 + 0.01 * np.sin(1.0))
    tlev = 200 + 90 * x + 3 * np.cos(0.37)
    tlay = 0.5 * (tlev[1:] + tlev[:-1])
    tsfc = np.array([tlev[60] + 1])
    jl = np.arange(1, 61)
    h2o = 1e-6 + 0.02 * (jl / 60.0) ** 4
    o3 = 5e-6 * np.exp(-((jl - 12) / 8.0) ** 2) + 2e-8
    s = lambda v: np.array([v])
    gases = [("h2o", h2o, 0, 1), ("o3", o3, 0, 1), ("co2", s(400e-6), 0, 0), ("ch4", s(1.8e-6), 0, 0),
             ("n2o", s(3.3e-7), 0, 0), ("cfc11", s(2.3e-10), 0, 0), ("cfc12", s(5.2e-10), 0, 0),
             ("o2", s(0.209), 0, 0)]
    return plev[:, None], tlev[:, None], tlay[:, None], tsfc, gases


def test_known_answers_from_reference_run(oracle_mod):
    """Bit-for-bit: these six numbers came out of the reference's own Fortran."""
    kat = json.load(open(os.path.join(HERE, "golden", "kat_survey.json")))
    m = oracle_mod.CkdModel(LW_FSCK)
    plev, tlev, tlay, tsfc, gases = kat_inputs()
    tau, lay, inc, dec, sfc, err = oracle_mod.gas_optics_int(m, plev, tlay, tsfc, gases, tlev)
    assert err == ""
    assert tau[0, 0, 0] == kat["tau(1,1,1)"]
    assert tau[0, 59, 0] == kat["tau(1,60,1)"]
    assert tau[16, 29, 0] == kat["tau(1,30,17)"]
    assert lay[4, 29, 0] == kat["lay_source(1,30,5)"]
    assert sfc[4, 0] == kat["sfc_source(1,5)"]
    assert inc[4, 59, 0] == kat["lev_source_inc(1,60,5)"]
    assert abs(np.pi * sfc.sum() - kat["pi_times_sum_sfc_source"]) < 1e-3


def test_f32_literal_constants():
    """gravity, molar mass, 0.001 and pi are default-real literals."""
    kat = json.load(open(os.path.join(HERE, "golden", "kat_survey.json")))
    gw = 1.0 / (float(np.float32(9.80665)) * float(np.float32(0.001)) * float(np.float32(28.970)))
    assert gw == kat["global_weight"]
    assert float(np.float32(3.14159265359)) == 3.14159274101257324


def test_load_and_init_bookkeeping(oracle_mod):
    """mo_load_coefficients.F90:104-144: file-order gases, then composite constituents that are
    not already tables (o2, n2) as composite-only copies."""
    m = oracle_mod.CkdModel(LW_FSCK)
    assert m.gas == ["h2o", "o3", "co2", "ch4", "n2o", "cfc11", "cfc12", "o2", "n2"]
    assert [t["code"] for t in m.tables] == [2, 1, 1, 3, 3, 1, 1, 0, 0]
    assert [t["composite_only"] for t in m.tables] == [False] * 7 + [True, True]
    assert (m.ng, m.np_, m.nt, m.ntp) == (32, 53, 6, 231)
    assert m.tables[3]["reference_mole_fraction"] == float(np.float32(1.921e-6))
    s = oracle_mod.CkdModel(SW_WIDE)
    assert s.gas == ["h2o", "o3", "co2", "ch4", "n2o", "o2", "n2"] and s.ng == 27 and s.shortwave
    assert abs(s.total_solar_irradiance - 1361.0) < 1e-3
    assert s.band2gpt.shape == (5, 2) and s.band2gpt[0, 0] == 1 and s.band2gpt[-1, 1] == 27
    r = oracle_mod.CkdModel(LW_RRTMGP)
    assert r.ng == 36 and r.band2gpt.shape == (16, 2)
    assert r.gas == ["h2o", "o3", "co2", "ch4", "n2o", "cfc11", "cfc12", "o2", "n2"]


def test_tokenize_quirk(oracle_mod):
    assert oracle_mod.tokenize("composite h2o o3") == ["composite", "h2o", "o3"]
    assert oracle_mod.tokenize("  a1  b2   ") == ["a1", "b2"]
    assert oracle_mod.tokenize("o2 n2 x") == ["o2", "n2"]   # single-character last token is lost


def test_composite_once_and_unknown_gas(oracle_mod):
    m = oracle_mod.CkdModel(LW_FSCK)
    plev, tlev, tlay, tsfc, gases = kat_inputs()
    base = oracle_mod.gas_optics_int(m, plev, tlay, tsfc, gases, tlev)[0]
    s = lambda v: np.array([v])
    with_n2 = gases + [("n2", s(0.78), 0, 0), ("no2", s(1e-9), 0, 0)]
    assert np.array_equal(oracle_mod.gas_optics_int(m, plev, tlay, tsfc, with_n2, tlev)[0], base)
    no_comp = [g for g in gases if g[0] != "o2"]
    assert np.all(oracle_mod.gas_optics_int(m, plev, tlay, tsfc, no_comp, tlev)[0] <= base)
    assert np.any(oracle_mod.gas_optics_int(m, plev, tlay, tsfc, no_comp, tlev)[0] < base)


def test_error_messages(oracle_mod):
    m = oracle_mod.CkdModel(LW_FSCK)
    plev, tlev, tlay, tsfc, gases = kat_inputs()
    assert oracle_mod.gas_optics_int(m, plev, tlay, tsfc, gases, None)[5] == "tlev is required for ecckd"
    s = oracle_mod.CkdModel(SW_WIDE)
    assert oracle_mod.gas_optics_ext(s, plev, tlay, gases, two_stream=False)[4] == \
        "shortwave must use ty_optical_props_2str"


def test_golden_fixture(oracle_mod):
    """tests/golden/lw_fsck_synth16.npz was written by tests/golden/make_golden.py from this
    oracle (itself pinned above); it guards the oracle against drift."""
    from rte_ecckd_amd import synthetic
    z = np.load(os.path.join(HERE, "golden", "lw_fsck_synth16.npz"))
    m = oracle_mod.CkdModel(LW_FSCK)
    cols = synthetic.columns(0, 16, float(np.exp(m.log_pressure[0])))
    tau, lay, inc, dec, sfc, _ = oracle_mod.gas_optics_int(m, cols["plev"], cols["tlay"], cols["tsfc"],
                                                           synthetic.gas_items(cols), cols["tlev"])
    fu, fd = oracle_mod.rte_lw(tau, lay, inc, dec, np.repeat(cols["sfc_emis"][None], m.ng, 0), sfc)
    for name, a in (("tau", tau), ("lay_source", lay), ("lev_source_inc", inc), ("sfc_source", sfc),
                    ("flux_up", fu), ("flux_dn", fd)):
        assert np.array_equal(z[name], a), name


# ---------------------------- solver: analytic known answers ----------------------------
def test_rte_lw_transparent_atmosphere(oracle_mod):
    ng, nlay, ncol = 4, 7, 3
    tau = np.zeros((ng, nlay, ncol))
    B = np.random.default_rng(1).uniform(1, 5, (ng, nlay, ncol))
    sfc = np.random.default_rng(2).uniform(1, 5, (ng, ncol))
    emis = np.full((ng, ncol), 0.9)
    fu, fd = oracle_mod.rte_lw(tau, B, B, B, emis, sfc)
    assert np.all(fd == 0)
    expect = (2 * np.pi * 0.5 * (0.9 * sfc)).sum(0)
    assert np.allclose(fu, expect[None, :], rtol=1e-14)


def test_rte_lw_isothermal(oracle_mod):
    ng, nlay, ncol = 5, 9, 2
    rng = np.random.default_rng(3)
    tau = rng.uniform(0.01, 2.0, (ng, nlay, ncol))
    Bg = rng.uniform(1, 3, (ng, 1, ncol))
    B = np.repeat(Bg, nlay, 1)
    fu, fd = oracle_mod.rte_lw(tau, B, B, B, np.ones((ng, ncol)), Bg[:, 0, :].copy())
    assert np.allclose(fu, (np.pi * Bg[:, 0, :]).sum(0)[None, :], rtol=1e-13)
    trans = np.exp(-1.66 * np.cumsum(tau, 1))
    expect = (np.pi * Bg * (1 - trans)).sum(0)
    assert np.allclose(fd[1:], expect, rtol=1e-12)
    for nmus in (2, 3, 4):   # quadrature weights sum to 1/2: isothermal black surface unchanged
        fu2, _ = oracle_mod.rte_lw(tau, B, B, B, np.ones((ng, ncol)), Bg[:, 0, :].copy(), nmus=nmus)
        assert np.allclose(fu2, fu, rtol=1e-9)


def test_rte_lw_orientation(oracle_mod):
    rng = np.random.default_rng(4)
    ng, nlay, ncol = 3, 6, 4
    tau = rng.uniform(0.01, 1.0, (ng, nlay, ncol))
    lay = rng.uniform(1, 3, (ng, nlay, ncol))
    inc = rng.uniform(1, 3, (ng, nlay, ncol))
    dec = rng.uniform(1, 3, (ng, nlay, ncol))
    sfc = rng.uniform(1, 3, (ng, ncol))
    emis = rng.uniform(0.8, 1.0, (ng, ncol))
    fu, fd = oracle_mod.rte_lw(tau, lay, inc, dec, emis, sfc, top_at_1=True)
    f = lambda a: np.ascontiguousarray(a[:, ::-1, :])
    fu2, fd2 = oracle_mod.rte_lw(f(tau), f(lay), f(dec), f(inc), emis, sfc, top_at_1=False)
    assert np.array_equal(fu2[::-1], fu) and np.array_equal(fd2[::-1], fd)


def test_rte_sw_beer_lambert_and_conservation(oracle_mod):
    rng = np.random.default_rng(5)
    ng, nlay, ncol = 3, 8, 4
    tau = rng.uniform(0.01, 0.5, (ng, nlay, ncol))
    zeros = np.zeros_like(tau)
    mu0 = rng.uniform(0.2, 1.0, ncol)
    toa = rng.uniform(10, 100, (ng, ncol))
    alb = np.full((ng, ncol), 0.3)
    fu, fd, fdir = oracle_mod.rte_sw(tau, zeros, zeros, mu0, toa, alb, alb)
    expect = (toa[:, None, :] * mu0 * np.exp(-np.cumsum(tau, 1) / mu0)).sum(0)
    assert np.allclose(fdir[1:], expect, rtol=1e-13)
    assert np.allclose(fd, fdir, rtol=1e-13)            # no scattering: no diffuse down
    # conservative scattering over a black surface: everything that enters leaves or is absorbed by the surface
    ones = np.ones_like(tau)
    fu, fd, fdir = oracle_mod.rte_sw(tau, ones, zeros, mu0, toa, np.zeros((ng, ncol)), np.zeros((ng, ncol)))
    assert np.allclose(fd[0] - fu[0], fd[-1] - fu[-1], rtol=1e-5)


# ---------------- node-identity known answers from the reference-held LUT files ----------------
@pytest.mark.parametrize("path", [LW_FSCK, LW_RRTMGP])
def test_planck_sources_on_table_nodes(oracle_mod, path):
    """T == temperature_planck(k)  =>  sources == planck_function(:,k)/pi_f32, bit for bit."""
    import helpers
    m = oracle_mod.CkdModel(path)
    cols, want = helpers.planck_node_case(m)
    tau, lay, inc, dec, sfc, err = oracle_mod.gas_optics_int(m, cols["plev"], cols["tlay"], cols["tsfc"], [], cols["tlev"])
    assert err == ""
    assert np.array_equal(sfc, want)
    for a in (lay, inc, dec):
        assert np.array_equal(a, np.repeat(want[:, None, :], 3, 1))


@pytest.mark.parametrize("path", [LW_FSCK, SW_WIDE])
def test_tau_on_table_nodes(oracle_mod, path):
    """p, T (and the h2o mole fraction) on table nodes => tau == weight * coefficient(:,ip,it), bit for bit."""
    import helpers
    m = oracle_mod.CkdModel(path)
    for name, cols, item, want in helpers.tau_node_cases(m):
        if m.shortwave:
            tau = oracle_mod.gas_optics_ext(m, cols["plev"], cols["tlay"], [item], two_stream=False)[0]
            ray = helpers.GLOBAL_WEIGHT * (cols["plev"][1] - cols["plev"][0])[None, None, :] * m.rayleigh[:, None, None]
            assert np.array_equal(tau, want + ray), name      # :456 tau = gas + Rayleigh
        else:
            tau = oracle_mod.gas_optics_int(m, cols["plev"], cols["tlay"], cols["tsfc"], [item], cols["tlev"])[0]
            assert np.array_equal(tau, want), name
        assert np.all(want >= 0) and np.any(want > 0)


# ---------------- solver switches and the incident-flux boundary condition ----------------
def test_rte_lw_incident_flux(oracle_mod):
    """Transparent column: F_dn = sum_g inc_flux at every level; an absorbing one
    attenuates it by exp(-D tau)."""
    rng = np.random.default_rng(11)
    ng, nlay, ncol = 4, 6, 5
    z = np.zeros((ng, nlay, ncol))
    inc = rng.uniform(1, 9, (ng, ncol))
    emis = np.ones((ng, ncol))
    fu, fd = oracle_mod.rte_lw(z, z, z, z, emis, np.zeros((ng, ncol)), inc_flux=inc)
    assert np.allclose(fd, inc.sum(0)[None, :], rtol=1e-15) and np.all(fu == 0)
    fu0, fd0 = oracle_mod.rte_lw(z, z, z, z, emis, np.zeros((ng, ncol)))
    assert np.all(fd0 == 0)
    tau = rng.uniform(0.1, 1.0, (ng, nlay, ncol))
    _, fd = oracle_mod.rte_lw(tau, z, z, z, emis, np.zeros((ng, ncol)), inc_flux=inc)
    want = (inc[:, None, :] * np.exp(-1.66 * np.cumsum(tau, 1))).sum(0)
    assert np.allclose(fd[1:], want, rtol=1e-13)
    # bottom-up orientation gives the mirrored answer
    f = lambda a: np.ascontiguousarray(a[:, ::-1, :])
    _, fd2 = oracle_mod.rte_lw(f(tau), z, z, z, emis, np.zeros((ng, ncol)), inc_flux=inc, top_at_1=False)
    assert np.array_equal(fd2[::-1], fd)
    # n angles: the literal B.1 form feeds every angle the full flux; the isotropic switch conserves it
    _, fd3 = oracle_mod.rte_lw(z, z, z, z, emis, np.zeros((ng, ncol)), inc_flux=inc, nmus=3)
    assert np.allclose(fd3, 3 * inc.sum(0)[None, :], rtol=1e-14)
    iso = oracle_mod.solver_options(lw_inc_flux_isotropic=1)
    _, fd4 = oracle_mod.rte_lw(z, z, z, z, emis, np.zeros((ng, ncol)), inc_flux=inc, nmus=3, options=iso)
    assert np.allclose(fd4, inc.sum(0)[None, :], rtol=1e-9)


def test_solver_switches(oracle_mod):
    rng = np.random.default_rng(12)
    ng, nlay, ncol = 3, 5, 40
    # LW: the series branch only matters below the threshold; a 3-term series moves tiny-tau layers by O(tau^4)
    tau = rng.uniform(0, 1e-3, (ng, nlay, ncol))
    B = rng.uniform(1, 9, (ng, nlay, ncol)); B2 = rng.uniform(1, 9, (ng, nlay, ncol))
    sfc = rng.uniform(1, 9, (ng, ncol)); emis = np.full((ng, ncol), 0.95)
    base = oracle_mod.rte_lw(tau, B, B2, B2, emis, sfc)
    hi = oracle_mod.rte_lw(tau, B, B2, B2, emis, sfc, options=oracle_mod.solver_options(lw_tau_thresh=1e-2, lw_series_terms=3))
    assert not np.array_equal(hi[1], base[1]) and np.allclose(hi[1], base[1], rtol=0, atol=1e-8)
    # SW: mu0 close to 1/k makes Rdir/Tdir ill-conditioned; the clamps keep 0 <= Rdir <= 1 - Tnoscat
    tau = rng.uniform(0.01, 3.0, (ng, nlay, ncol)); ssa = rng.uniform(0.0, 0.999999, (ng, nlay, ncol)); g = rng.uniform(0, 0.9, (ng, nlay, ncol))
    mu0 = rng.uniform(0.05, 1.0, ncol); toa = rng.uniform(10, 100, (ng, ncol)); alb = np.full((ng, ncol), 0.2)
    a = oracle_mod.rte_sw(tau, ssa, g, mu0, toa, alb, alb)
    b = oracle_mod.rte_sw(tau, ssa, g, mu0, toa, alb, alb, options=oracle_mod.solver_options(sw_dir_clamp=1))
    assert np.allclose(a[0], b[0], rtol=0, atol=5.0) and np.array_equal(a[2], b[2])   # the direct beam is untouched
    c = oracle_mod.rte_sw(tau, ssa, g, mu0, toa, alb, alb, options=oracle_mod.solver_options(sw_k_floor=1e-3))
    assert np.allclose(a[0], c[0], rtol=1e-2)
