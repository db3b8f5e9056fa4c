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
