"""Deterministic synthetic atmospheres for parity tests and bench.py.

No library RNG: ``u(c,f,j)`` = top 53 bits of one splitmix64 step of
``20221128 ^ (f << 56) ^ (c << 8) ^ j`` times 2**-53, so the same columns can be regenerated
bit for bit in C, Fortran or Python for any column range (column-range sharding across GPUs
needs no communication and no shared files).

Arrays follow the package convention: C order, reversed Fortran shape (``plev`` is
``(nlay+1, ncol)``).  Gas order is the RFMIP one (mo_rfmip_io.F90:204-259 / utils.f90:41-70):
co2, ch4, n2o, o2, cfc11, cfc12, h2o, o3, no2 -- no2 is unknown to the ecCKD tables and is
skipped by gas_optics (src/gas_optics_ecckd.f90:358-364).
"""
import numpy as np

SEED = 20221128
NLAY = 60
F_PS, F_TS, F_TLEV, F_TSFC, F_EMIS, F_H2O, F_O3, F_CO2, F_CH4, F_N2O, F_CFC11, F_CFC12, F_MU0, F_ALB = range(1, 15)
GAS_ORDER = ["co2", "ch4", "n2o", "o2", "cfc11", "cfc12", "h2o", "o3", "no2"]


def uniform(c, f, j):
    """u(c,f,j) in [0,1); c, j broadcastable integer arrays, f a field id."""
    with np.errstate(over="ignore"):
        x = (np.uint64(SEED) ^ (np.uint64(f) << np.uint64(56)) ^ (np.asarray(c, dtype=np.uint64) << np.uint64(8))
             ^ np.asarray(j, dtype=np.uint64))
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def columns(c0, ncol, press_min, nlay=NLAY, shortwave=False):
    """Columns ``c0 .. c0+ncol-1``.  ``press_min`` = ``ecckd%get_press_min()`` (the driver clamps
    the top level to it, ecckd_rfmip_lw.F90:90-94).  Returns a dict of float64 arrays."""
    c = np.arange(c0, c0 + ncol, dtype=np.uint64)[None, :]
    jl = np.arange(1, nlay + 2, dtype=np.uint64)[:, None]       # levels 1..nlay+1
    jm = np.arange(1, nlay + 1, dtype=np.uint64)[:, None]       # layers 1..nlay
    eta = ((np.arange(nlay + 1, dtype=np.float64)) / nlay) ** 2  # eta_j, j = 1..nlay+1
    eta = eta[:, None]
    ptop = press_min * (1 + 2.3e-16)
    ps = 95000.0 + 8000.0 * uniform(c, F_PS, 0)
    plev = ptop + (ps - ptop) * eta
    ts = 250.0 + 60.0 * uniform(c, F_TS, 0)
    tlev = ts - 70.0 * (1.0 - eta) ** 0.8 + 4.0 * (uniform(c, F_TLEV, jl) - 0.5)
    tlay = 0.5 * (tlev[1:] + tlev[:-1])
    tsfc = tlev[nlay] + 2.0 * (uniform(c, F_TSFC, 0)[0] - 0.5)
    sfc_emis = 0.95 + 0.05 * uniform(c, F_EMIS, 0)[0]
    eta_mid = 0.5 * (eta[1:] + eta[:-1])
    p_mid = 0.5 * (plev[1:] + plev[:-1])
    h2o = np.maximum(1e-7, 0.03 * uniform(c, F_H2O, jm) * eta_mid ** 3)
    o3 = 2e-8 + 8e-6 * np.exp(-((np.log(p_mid) - np.log(1000.0)) / 1.2) ** 2) * (0.5 + uniform(c, F_O3, jm))
    lin = lambda f, lo, hi: lo + (hi - lo) * uniform(c, f, 0)[0]
    out = dict(
        plev=np.ascontiguousarray(plev), tlev=np.ascontiguousarray(tlev), tlay=np.ascontiguousarray(tlay),
        tsfc=np.ascontiguousarray(tsfc), sfc_emis=np.ascontiguousarray(sfc_emis),
        h2o=np.ascontiguousarray(h2o), o3=np.ascontiguousarray(o3),
        co2=lin(F_CO2, 180e-6, 2240e-6), ch4=lin(F_CH4, 350e-9, 3500e-9), n2o=lin(F_N2O, 190e-9, 540e-9),
        cfc11=lin(F_CFC11, 0.0, 2000e-12), cfc12=lin(F_CFC12, 0.0, 550e-12), o2=0.209, no2=0.0)
    if shortwave:
        out["mu0"] = 0.05 + 0.95 * uniform(c, F_MU0, 0)[0]
        out["albedo"] = 0.05 + 0.3 * uniform(c, F_ALB, 0)[0]
        out["tsi"] = 1361.0
    return out


def gas_items(cols):
    """[(name, array, col_stride, lay_stride)] in GAS_ORDER for oracle-style consumers: full
    arrays are (nlay,ncol) -> strides (1,ncol); per-column arrays (1,0); scalars (0,0)."""
    ncol = cols["plev"].shape[1]
    items = []
    for n in GAS_ORDER:
        v = cols[n]
        if np.isscalar(v):
            items.append((n, np.array([v], dtype=np.float64), 0, 0))
        elif v.ndim == 1:
            items.append((n, v, 1, 0))
        else:
            items.append((n, v, 1, ncol))
    return items
