"""Shared test plumbing: product-side objects built from the synthetic generator."""
import numpy as np

from rte_ecckd_amd import synthetic


def product_gas_concs(pkg, cols, to=lambda a: a, names=None, overrides=None):
    """GasConcs in RFMIP order from synthetic columns; `to` moves arrays (e.g. to the GPU)."""
    names = list(synthetic.GAS_ORDER if names is None else names)
    gc = pkg.GasConcs(names)
    for n in names:
        v = (overrides or {}).get(n, cols.get(n, 0.0))
        if np.isscalar(v):
            assert gc.set_vmr(n, float(v)) == ""
        elif v.ndim == 1 and v.shape[0] == cols["plev"].shape[1]:
            assert gc.set_vmr_column(n, to(v)) == ""
        else:
            assert gc.set_vmr(n, to(v)) == ""
    return gc


def oracle_gas_items(cols, names=None, overrides=None):
    names = list(synthetic.GAS_ORDER if names is None else names)
    ncol = cols["plev"].shape[1]
    nlay = cols["tlay"].shape[0]
    items = []
    for n in names:
        v = (overrides or {}).get(n, cols.get(n, 0.0))
        if np.isscalar(v):
            items.append((n, np.array([v], dtype=np.float64), 0, 0))
        elif v.ndim == 1 and v.shape[0] == ncol:
            items.append((n, np.ascontiguousarray(v), 1, 0))
        elif v.ndim == 1 and v.shape[0] == nlay:
            items.append((n, np.ascontiguousarray(v), 0, 1))
        else:
            items.append((n, np.ascontiguousarray(v), 1, ncol))
    return items


def max_rel(a, b, floor=1e-300):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def run_lw_gas_optics(pkg, k, cols, device=None, names=None, overrides=None, tlev=True):
    """Product LW gas optics on numpy (host memspace) or torch (device memspace) arrays.
    Returns (err, tau, lay, inc, dec, sfc) as numpy arrays."""
    ncol = cols["plev"].shape[1]
    nlay = cols["tlay"].shape[0]
    if device is None:
        to = lambda a: np.ascontiguousarray(a)
        like = np.empty(0)
        back = lambda a: a
    else:
        import torch
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        like = to(np.zeros(1))
        back = lambda a: a.cpu().numpy()
    gc = product_gas_concs(pkg, cols, to, names, overrides)
    op = pkg.OpticalProps1scl()
    op.alloc_1scl(ncol, nlay, k, like=like)
    src = pkg.SourceFuncLW()
    src.alloc(ncol, nlay, k, like=like)
    err = k.gas_optics(None, to(cols["plev"]), to(cols["tlay"]), to(cols["tsfc"]), gc, op, src,
                       tlev=to(cols["tlev"]) if tlev else None)
    if device is not None:
        import torch
        torch.cuda.synchronize()
    return err, back(op.tau), back(src.lay_source), back(src.lev_source_inc), back(src.lev_source_dec), back(src.sfc_source)


# ------------------------------------------------------------------------------------------------
# Node-identity known answers from data the reference DOES hold: the three LUT files.
# These do not pin the code path (parity stays "unpinned"); they tie the oracle and the HIP path to
# reference-held numbers instead of to each other.
# ------------------------------------------------------------------------------------------------
PI_F32 = float(np.float32(3.14159265359))          # src/gas_optics_ecckd.f90:53
GLOBAL_WEIGHT = 1.0 / (float(np.float32(9.80665)) * float(np.float32(0.001)) * float(np.float32(28.970)))   # :107


def planck_node_case(m):
    """Temperatures that sit exactly on temperature_planck(k): the interpolation weights are (1, 0), so every
    source must equal planck_function(:,k)/pi (pi the f32-rounded literal), bit for bit (:275-288).
    Returns (cols-like dict with ncol = ntp, expected (ng, ntp))."""
    ntp, ng = m.ntp, m.ng
    T = m.temperature_planck.copy()
    nlay = 3
    p = np.exp(m.log_pressure[20]) * np.array([1.0, 1.1, 1.2, 1.3])
    cols = dict(plev=np.repeat(p[:, None], ntp, 1), tlay=np.repeat(T[None], nlay, 0),
                tlev=np.repeat(T[None], nlay + 1, 0), tsfc=T.copy())
    return cols, np.ascontiguousarray(m.planck_function.T) / PI_F32


def tau_node_cases(m, names=("co2", "ch4", "h2o")):
    """One gas at a time, layer pressure and temperature on table nodes where the index arithmetic is exact
    (pressure nodes 1 and 2: (lp - lp0)/dlp is 0 or 1 by construction; every temperature node but the last: the
    grid is exactly 20 K apart; h2o at its first mole-fraction node): tau must equal weight * coefficient(:,ip,it[,iv])
    with weight = global_weight*(p1-p0)*vmr (linear, look-up table) or *(vmr - ref) (relative-linear)
    (src/gas_optics_ecckd.f90:143-149,167-221), bit for bit.  Yields (gas, cols, gas item, expected (ng, 1, ncol))."""
    out = []
    for name in names:
        t = m.tables[m.gas.index(name)]
        cases = [(ip, it) for ip in (0, 1) for it in range(m.nt - 1)]   # (the last node is clamped to nt - 1.0001: :137)
        ncol = len(cases)
        plev = np.empty((2, ncol)); tlay = np.empty((1, ncol)); exp = np.empty((m.ng, 1, ncol))
        vmr = {"co2": 4e-4, "ch4": 2.5e-6}.get(name)     # (ch4 above its reference mole fraction: below it the per-gas clamp zeroes tau, :234-238)
        if t["code"] == 2:
            vmr = float(t["mole_fraction"][0])
        for c, (ip, it) in enumerate(cases):
            P = float(np.exp(m.log_pressure[ip]))
            if np.log(P) != m.log_pressure[ip]:            # exp/log round trip not exact: take the file's pressure
                from scipy.io import netcdf_file
                P = float(netcdf_file(m.path, mmap=False).variables["pressure"].data[ip])
            d = 2.0 ** np.floor(np.log2(P / 4))
            plev[0, c], plev[1, c] = P - d, P + d
            assert 0.5 * (plev[0, c] + plev[1, c]) == P
            tlay[0, c] = m.temperature[it, ip]
            sw = GLOBAL_WEIGHT * (plev[1, c] - plev[0, c])
            w = sw * (vmr - t["reference_mole_fraction"]) if t["code"] == 3 else sw * vmr
            exp[:, 0, c] = w * t["coefficient"][0, it, ip, :]
        cols = dict(plev=plev, tlay=tlay, tlev=np.repeat(tlay, 2, 0), tsfc=tlay[0].copy())
        out.append((name, cols, (name, np.array([vmr]), 0, 0), exp))
    return out
