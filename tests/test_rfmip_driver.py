"""The RFMIP RAD-IRF drivers (BASELINE config 1, "plumbing"): executables `ecckd_rfmip_lw` /
`ecckd_rfmip_sw` with the reference's names and command line (example/rfmip-rad-irf/Makefile:12-40,
utils.f90:26-37), run on an RFMIP-SCHEMA file synthesised here -- the real RFMIP input and output
template files are an FTP download (download-data-files.sh:4-17) and are not available offline.
Schema = mo_rfmip_io.F90:38-46,78-99,130-138,167-171,217-240."""
import os
import subprocess

import numpy as np
import pytest
from scipy.io import netcdf_file

from conftest import LW_FSCK, SW_WIDE
from rte_ecckd_amd import synthetic

NSITE, NEXP, NLAY = 100, 3, 60          # "100 column test cases" (README.md:25); RFMIP proper has 18 experiments
GM = {"carbon_dioxide": ("1.e-6", [400.0, 800.0, 280.0]), "methane": ("1.e-9", [1800.0, 1800.0, 720.0]),
      "nitrous_oxide": ("1.e-9", [330.0, 330.0, 270.0]), "oxygen": ("1.", [0.209, 0.209, 0.209]),
      "cfc11": ("1.e-12", [230.0, 230.0, 0.0]), "cfc11eq": ("1.e-12", [800.0, 800.0, 30.0]),
      "cfc12": ("1.e-12", [520.0, 520.0, 0.0])}
KDIST = ["co2", "ch4", "n2o", "o2", "cfc11", "cfc12"]


def make_rfmip_files(d, press_min):
    """Returns the per-(site,expt) inputs as arrays with the flattened column index site-fastest."""
    cols = [synthetic.columns(1000 * e, NSITE, press_min, shortwave=True) for e in range(NEXP)]
    plev = cols[0]["plev"].copy()                    # pressures do not depend on the experiment
    plev[0, :] = 1e-3                                # RFMIP's top level; the drivers clamp it
    play = 0.5 * (plev[1:] + plev[:-1])
    f = netcdf_file(os.path.join(d, "rfmip.nc"), "w")
    for n, k in (("site", NSITE), ("layer", NLAY), ("level", NLAY + 1), ("expt", NEXP)):
        f.createDimension(n, k)

    def var(name, dims, data, units=None):
        v = f.createVariable(name, "d", dims)
        v[:] = data
        if units is not None:
            v.units = units
    var("pres_layer", ("site", "layer"), play.T)
    var("pres_level", ("site", "level"), plev.T)
    var("temp_layer", ("expt", "site", "layer"), np.stack([c["tlay"].T for c in cols]))
    var("temp_level", ("expt", "site", "level"), np.stack([c["tlev"].T for c in cols]))
    var("surface_temperature", ("expt", "site"), np.stack([c["tsfc"] for c in cols]))
    var("surface_emissivity", ("site",), cols[0]["sfc_emis"])
    var("surface_albedo", ("site",), cols[0]["albedo"])
    tsi = 1361.0 + 0.5 * np.arange(NSITE)
    sza = np.linspace(0.0, 110.0, NSITE)             # the last sites are night columns (> 90 degrees)
    var("total_solar_irradiance", ("site",), tsi)
    var("solar_zenith_angle", ("site",), sza)
    var("water_vapor", ("expt", "site", "layer"), np.stack([c["h2o"].T for c in cols]) * 1e6, "1.e-6")
    var("ozone", ("expt", "site", "layer"), np.stack([c["o3"].T for c in cols]) * 1e9, "1.e-9")
    for name, (units, vals) in GM.items():
        var(name + "_GM", ("expt",), np.array(vals), units)
    f.close()
    for fn, vn in (("rlu", "rlu"), ("rld", "rld"), ("rsu", "rsu"), ("rsd", "rsd")):
        for p, ff in (("1", "1"), ("2", "1"), ("1", "2"), ("2", "2")):
            if fn.startswith("rs") and p != "1":
                continue
            o = netcdf_file(os.path.join(d, "%s_Efx_RTE-ecckd_rad-irf_r1i1p%sf%s_gn.nc" % (fn, p, ff)), "w")
            for n, k in (("expt", NEXP), ("site", NSITE), ("level", NLAY + 1)):
                o.createDimension(n, k)
            v = o.createVariable(vn, "d", ("expt", "site", "level"))
            v[:] = -999.0
            o.close()
    return cols, plev, tsi, sza


def flat_inputs(cols, plev, press_min, forcing_index):
    """Oracle-side view: columns flattened site-fastest over the experiments, top level clamped."""
    ncol = NSITE * NEXP
    pl = np.tile(plev, (1, NEXP))
    pl[0, :] = press_min + np.spacing(press_min)     # ecckd_rfmip_lw.F90:90-94
    cat = lambda k: np.concatenate([c[k] for c in cols], axis=-1)
    rf = ["carbon_dioxide", "methane", "nitrous_oxide", "oxygen", "cfc11" if forcing_index == 1 else "cfc11eq", "cfc12"]
    items = []
    for kd, name in zip(KDIST, rf):
        units, vals = GM[name]
        percol = np.repeat(np.array(vals) * float(units), NSITE)
        items.append((kd, percol, 1, 0))
    items.append(("h2o", np.ascontiguousarray(cat("h2o") * 1e6 * 1e-6), 1, ncol))
    items.append(("o3", np.ascontiguousarray(cat("o3") * 1e9 * 1e-9), 1, ncol))
    items.append(("no2", np.zeros(1), 0, 0))
    return pl, cat("tlay"), cat("tlev"), cat("tsfc"), items


def read_flux(path, var):
    f = netcdf_file(path, "r", mmap=False)
    a = np.array(f.variables[var].data, dtype=np.float64)      # (expt, site, level)
    f.close()
    return a.reshape(NEXP * NSITE, NLAY + 1).T                 # (level, flattened column)


@pytest.mark.gpu
@pytest.mark.parametrize("flags,f_idx,p_idx", [([], 1, 1), (["-b", "1", "-n", "7"], 1, 1), (["-f", "2", "-p", "2", "-b", "60"], 2, 2)])
def test_rfmip_lw(pkg, gpu, oracle_mod, tmp_path, flags, f_idx, p_idx):
    if not os.path.exists(pkg.RFMIP_LW) and pkg.build_fortran() is None:
        pytest.skip("no Fortran toolchain / binaries")
    m = oracle_mod.CkdModel(LW_FSCK)
    pmin = float(np.exp(m.log_pressure[0]))
    cols, plev, _, _ = make_rfmip_files(str(tmp_path), pmin)
    r = subprocess.run([pkg.RFMIP_LW, "rfmip.nc", LW_FSCK] + flags, cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    pl, tlay, tlev, tsfc, items = flat_inputs(cols, plev, pmin, f_idx)
    tau, lay, inc, dec, sfc, _ = oracle_mod.gas_optics_int(m, pl, tlay, tsfc, items, tlev)
    emis = np.tile(cols[0]["sfc_emis"], NEXP)
    fu, fd = oracle_mod.rte_lw(tau, lay, inc, dec, np.repeat(emis[None], 32, 0), sfc, nmus=3 if p_idx == 2 else 1)
    gu = read_flux(str(tmp_path / ("rlu_Efx_RTE-ecckd_rad-irf_r1i1p%df%d_gn.nc" % (p_idx, f_idx))), "rlu")
    gd = read_flux(str(tmp_path / ("rld_Efx_RTE-ecckd_rad-irf_r1i1p%df%d_gn.nc" % (p_idx, f_idx))), "rld")
    n = 7 if "-n" in flags else NSITE * NEXP                    # "-b 1 -n 7": only the first 7 blocks of 1
    assert np.max(np.abs(gu[:, :n] - fu[:, :n])) < 1e-9 and np.max(np.abs(gd[:, :n] - fd[:, :n])) < 1e-9
    if n < NSITE * NEXP:
        assert np.all(gu[:, n:] == 0)                          # unprocessed blocks stay zero, like the reference's 1700
    # -d: device-resident optical properties and sources (ECCKD_MIXED), the same fluxes bit for bit
    r = subprocess.run([pkg.RFMIP_LW, "rfmip.nc", LW_FSCK] + flags + ["-d"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(read_flux(str(tmp_path / ("rlu_Efx_RTE-ecckd_rad-irf_r1i1p%df%d_gn.nc" % (p_idx, f_idx))), "rlu"), gu)
    assert np.array_equal(read_flux(str(tmp_path / ("rld_Efx_RTE-ecckd_rad-irf_r1i1p%df%d_gn.nc" % (p_idx, f_idx))), "rld"), gd)


@pytest.mark.gpu
def test_rfmip_sw(pkg, gpu, oracle_mod, tmp_path):
    if not os.path.exists(pkg.RFMIP_SW) and pkg.build_fortran() is None:
        pytest.skip("no Fortran toolchain / binaries")
    m = oracle_mod.CkdModel(SW_WIDE)
    pmin = float(np.exp(m.log_pressure[0]))
    cols, plev, tsi, sza = make_rfmip_files(str(tmp_path), pmin)
    r = subprocess.run([pkg.RFMIP_SW, "rfmip.nc", SW_WIDE, "-b", "150"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    pl, tlay, tlev, tsfc, items = flat_inputs(cols, plev, pmin, 1)
    tau, ssa, g, toa, _ = oracle_mod.gas_optics_ext(m, pl, tlay, items)
    tsi_c, sza_c = np.tile(tsi, NEXP), np.tile(sza, NEXP)
    toa = toa * tsi_c[None, :] / toa.sum(0)[None, :]           # ecckd_rfmip_sw.F90:126-133
    use = sza_c < 90.0 - 2.0 * np.spacing(90.0)
    mu0 = np.where(use, np.cos(sza_c * (np.arccos(-1.0) / 180.0)), 1.0)
    alb = np.repeat(np.tile(cols[0]["albedo"], NEXP)[None], 27, 0)
    fu, fd, _ = oracle_mod.rte_sw(tau, ssa, g, mu0, np.ascontiguousarray(toa), alb, alb)
    fu[:, ~use] = 0.0
    fd[:, ~use] = 0.0
    gu = read_flux(str(tmp_path / "rsu_Efx_RTE-ecckd_rad-irf_r1i1p1f1_gn.nc"), "rsu")
    gd = read_flux(str(tmp_path / "rsd_Efx_RTE-ecckd_rad-irf_r1i1p1f1_gn.nc"), "rsd")
    assert np.max(np.abs(gu - fu)) < 1e-9 and np.max(np.abs(gd - fd)) < 1e-9
    assert np.all(gu[:, ~use] == 0) and (~use).sum() > 0
    # -d: device-resident optical properties, the same fluxes bit for bit
    r = subprocess.run([pkg.RFMIP_SW, "rfmip.nc", SW_WIDE, "-b", "150", "-d"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(read_flux(str(tmp_path / "rsu_Efx_RTE-ecckd_rad-irf_r1i1p1f1_gn.nc"), "rsu"), gu)
    assert np.array_equal(read_flux(str(tmp_path / "rsd_Efx_RTE-ecckd_rad-irf_r1i1p1f1_gn.nc"), "rsd"), gd)


def test_rfmip_cli_and_io_errors(pkg, tmp_path):
    """CPU: usage/help/flag validation and missing-file behaviour (stop 1 with the reference's texts)."""
    if pkg.build_fortran() is None:
        pytest.skip("no amdflang in this image")
    r = subprocess.run([pkg.RFMIP_LW], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage:" in r.stderr and "rfmip_file ecckd_file" in r.stderr
    r = subprocess.run([pkg.RFMIP_LW, "--help"], capture_output=True, text=True)
    assert r.returncode == 1                      # as the reference: fewer than 2 arguments -> usage, stop 1
    r = subprocess.run([pkg.RFMIP_LW, "a.nc", "b.nc", "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "-f [1,2] - Forcing index." in r.stderr and "-p [1,2] - Physics index." in r.stderr
    r = subprocess.run([pkg.RFMIP_LW, "a.nc", "b.nc", "-f", "3"], capture_output=True, text=True)
    assert r.returncode == 1 and "forcing index must be either 1 or 2." in r.stderr
    r = subprocess.run([pkg.RFMIP_SW, str(tmp_path / "nope.nc"), "b.nc"], capture_output=True, text=True)
    assert r.returncode == 1 and "read_size: can't find file" in r.stderr
