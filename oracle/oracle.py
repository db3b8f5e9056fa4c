"""ctypes binding of the CPU oracle (oracle/ecckd_oracle.c) plus a Python restatement of
``load_and_init`` (example/rfmip-rad-irf/mo_load_coefficients.F90:19-203).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py -- never by the product package.

Array convention: every numpy array is C-ordered with the *reverse* of the Fortran shape, so
the memory is exactly the reference's column-major layout (column index fastest):
``tau`` is ``(ngpt, nlay, ncol)``, ``plev`` is ``(nlay+1, ncol)``, fluxes ``(nlay+1, ncol)``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libecckd_oracle.so")
MAX_GASES = 16
NONE_, LINEAR, LOOK_UP_TABLE, RELATIVE_LINEAR = 0, 1, 2, 3

_dp = C.POINTER(C.c_double)


class _Gas(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("coefficient", _dp), ("nv", C.c_int),
                ("composite_only", C.c_int), ("concentration_dependence_code", C.c_int),
                ("mole_fraction", _dp), ("reference_mole_fraction", C.c_double)]


class _Model(C.Structure):
    _fields_ = [("ng", C.c_int), ("np", C.c_int), ("nt", C.c_int), ("ntp", C.c_int),
                ("num_gases", C.c_int), ("log_pressure", _dp), ("temperature", _dp),
                ("planck_function", _dp), ("temperature_planck", _dp),
                ("solar_irradiance", _dp), ("rayleigh_molar_scattering_coeff", _dp),
                ("gas", _Gas * MAX_GASES)]


class _GasConcs(C.Structure):
    _fields_ = [("ngas", C.c_int), ("names", C.POINTER(C.c_char_p)), ("vmr", C.POINTER(_dp)),
                ("col_stride", C.POINTER(C.c_long)), ("lay_stride", C.POINTER(C.c_long))]


def build():
    """Compile the oracle with gcc (oracle/Makefile)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.oracle_lw_pipeline.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# --------------------------------------------------------------------------------------
# load_and_init restated (mo_load_coefficients.F90:19-203) on top of scipy's CDF reader.
# --------------------------------------------------------------------------------------
def tokenize(buffer):
    """mo_load_coefficients.F90:244-293, including its quirk: a final token that is a single
    character is dropped (found_token is only set, never closed, when i == n)."""
    buffer = buffer.rstrip(" ")
    n = len(buffer)
    tokens, found, start = [], False, 0
    for i in range(n):
        ch = buffer[i]
        if found:
            if i == n - 1 and ch != " ":
                tokens.append(buffer[start:i + 1])
                break
            elif ch == " ":
                tokens.append(buffer[start:i])
                found = False
        elif ch != " ":
            found, start = True, i
    return tokens


class CkdModel:
    """What ``load_and_init`` leaves in ``ty_gas_optics_ecckd`` (src/gas_optics_ecckd.f90:23-48)."""

    def __init__(self, path):
        from scipy.io import netcdf_file
        f = netcdf_file(path, mmap=False)
        v = f.variables
        self.path = path
        self.log_pressure = np.log(_f64(v["pressure"].data))            # :46-49
        self.temperature = _f64(v["temperature"].data)                   # :51-53 (nt,np) C-order
        band_number = np.asarray(v["band_number"].data, dtype=np.int64) + 1  # :60-63
        self.band_lims_wvn = np.stack([_f64(v["wavenumber1_band"].data),
                                       _f64(v["wavenumber2_band"].data)], axis=1)  # (nband,2)
        nband = self.band_lims_wvn.shape[0]
        ng = band_number.shape[0]
        b2g = np.zeros((nband, 2), dtype=np.int64)                       # :64-73
        b2g[0, 0] = 1
        b2g[nband - 1, 1] = ng
        band = 1
        for i in range(2, ng + 1):
            if band_number[i - 1] > band:
                b2g[band - 1, 1] = i - 1
                band += 1
                b2g[band - 1, 0] = i
        self.band2gpt = b2g
        self.gpt2band = np.zeros(ng, dtype=np.int64)
        for b in range(nband):
            self.gpt2band[b2g[b, 0] - 1:b2g[b, 1]] = b + 1
        self.ng = v["gpoint_fraction"].shape[0]                          # size(gpoint_fraction,2)
        self.np_ = self.log_pressure.shape[0]
        self.nt = self.temperature.shape[0]
        self.shortwave = "solar_irradiance" in v                         # :84
        self.planck_function = self.temperature_planck = None
        self.solar_irradiance = self.rayleigh = None
        self.ntp = 0
        if self.shortwave:
            self.solar_irradiance = _f64(v["solar_irradiance"].data)
            self.total_solar_irradiance = float(np.sum(self.solar_irradiance))
            self.rayleigh = _f64(v["rayleigh_molar_scattering_coeff"].data)
        else:
            self.temperature_planck = _f64(v["temperature_planck"].data)
            self.planck_function = _f64(v["planck_function"].data)       # (ntp,ng) C-order
            self.ntp = self.temperature_planck.shape[0]
        gas = tokenize(f._attributes["constituent_id"].decode())         # :104-107
        self.gas, self.tables = [], []
        composite = []
        if "composite" in gas:                                           # :108-117
            composite = tokenize(f._attributes["composite_constituent_id"].decode())
        for name in gas:                                                 # :118-126
            if name != "composite":
                self.gas.append(name)
                self.tables.append(self._read_gas(v, name, False))
        for name in composite:                                           # :127-143
            if name not in gas:
                self.gas.append(name)
                self.tables.append(self._read_gas(v, "composite", True))
        self.num_gases = len(self.gas)
        f.close()

    @staticmethod
    def _read_gas(v, name, composite_only):
        """read_gas_input_data, mo_load_coefficients.F90:149-203."""
        t = dict(composite_only=composite_only, mole_fraction=None, reference_mole_fraction=0.0)
        mf = name + "_mole_fraction"
        if mf in v and len(v[mf].shape) == 1:                            # :160-175
            t["code"] = LOOK_UP_TABLE
            t["mole_fraction"] = _f64(v[mf].data)
            t["coefficient"] = _f64(v[name + "_molar_absorption_coeff"].data)  # (nv,nt,np,ng)
        else:
            n = int(v[name + "_conc_dependence_code"].data)              # :178-192
            if n not in (0, 1, 3):
                raise ValueError("load_and_init_ecckd: bad concentration code for " + name)
            t["code"] = n
            if n == 3:
                t["reference_mole_fraction"] = float(
                    np.float64(v[name + "_reference_mole_fraction"].data))
            c = v[name + "_molar_absorption_coeff"]
            if len(c.shape) != 3:
                raise ValueError("load_and_init_ecckd: absorption coefficient not 3d for " + name)
            t["coefficient"] = _f64(c.data)[None]                        # (1,nt,np,ng)
        t["coefficient"] = np.ascontiguousarray(t["coefficient"])
        t["nv"] = t["coefficient"].shape[0]
        return t

    # C view -------------------------------------------------------------------------
    def cstruct(self):
        m = _Model()
        m.ng, m.np, m.nt, m.ntp, m.num_gases = self.ng, self.np_, self.nt, self.ntp, self.num_gases
        m.log_pressure = _p(self.log_pressure)
        m.temperature = _p(self.temperature)
        m.planck_function = _p(self.planck_function)
        m.temperature_planck = _p(self.temperature_planck)
        m.solar_irradiance = _p(self.solar_irradiance)
        m.rayleigh_molar_scattering_coeff = _p(self.rayleigh)
        for i, (name, t) in enumerate(zip(self.gas, self.tables)):
            g = m.gas[i]
            g.name = name.encode()
            g.coefficient = _p(t["coefficient"])
            g.nv = t["nv"]
            g.composite_only = int(t["composite_only"])
            g.concentration_dependence_code = t["code"]
            g.mole_fraction = _p(t["mole_fraction"])
            g.reference_mole_fraction = t["reference_mole_fraction"]
        return m


class _GC:
    """Keeps the ctypes arrays of an oracle_gas_concs_t alive."""

    def __init__(self, gases):
        # gases: iterable of (name, float64 ndarray, col_stride, lay_stride)
        gases = list(gases)
        n = len(gases)
        self.arrays = [_f64(np.atleast_1d(a)) for _, a, _, _ in gases]
        self.names = (C.c_char_p * n)(*[g[0].encode() for g in gases])
        self.vmr = (_dp * n)(*[_p(a) for a in self.arrays])
        self.cs = (C.c_long * n)(*[int(g[2]) for g in gases])
        self.ls = (C.c_long * n)(*[int(g[3]) for g in gases])
        self.c = _GasConcs(n, self.names, self.vmr, self.cs, self.ls)


def gas_optics_int(model, plev, tlay, tsfc, gases, tlev):
    """Returns (tau, lay_source, lev_source_inc, lev_source_dec, sfc_source, errmsg)."""
    plev, tlay, tsfc = _f64(plev), _f64(tlay), _f64(tsfc)
    nlay, ncol = tlay.shape
    ng = model.ng
    m = model.cstruct()
    gc = _GC(gases)
    tau = np.empty((ng, nlay, ncol))
    lay = np.empty_like(tau)
    inc = np.empty_like(tau)
    dec = np.empty_like(tau)
    sfc = np.empty((ng, ncol))
    err = C.create_string_buffer(128)
    tl = None if tlev is None else _f64(tlev)
    lib().oracle_gas_optics_int(C.byref(m), ncol, nlay, _p(plev), _p(tlay), _p(tsfc), C.byref(gc.c),
                                _p(tl), _p(tau), _p(lay), _p(inc), _p(dec), _p(sfc), err)
    return tau, lay, inc, dec, sfc, err.value.decode()


def gas_optics_ext(model, plev, tlay, gases, two_stream=True):
    """Returns (tau, ssa, g, toa_src, errmsg)."""
    plev, tlay = _f64(plev), _f64(tlay)
    nlay, ncol = tlay.shape
    ng = model.ng
    m = model.cstruct()
    gc = _GC(gases)
    tau = np.empty((ng, nlay, ncol))
    ssa = np.empty_like(tau) if two_stream else None
    g = np.empty_like(tau) if two_stream else None
    toa = np.empty((ng, ncol))
    err = C.create_string_buffer(128)
    lib().oracle_gas_optics_ext(C.byref(m), ncol, nlay, _p(plev), _p(tlay), C.byref(gc.c), _p(tau),
                                _p(ssa), _p(g), _p(toa), err)
    return tau, ssa, g, toa, err.value.decode()


class _SolverOptions(C.Structure):
    _fields_ = [("lw_tau_thresh", C.c_double), ("lw_series_terms", C.c_int), ("lw_inc_flux_isotropic", C.c_int),
                ("sw_k_floor", C.c_double), ("sw_dir_clamp", C.c_int)]


def solver_options(**kw):
    """oracle_solver_options_t with the v1.5-era defaults, overridden by keyword (lw_tau_thresh,
    lw_series_terms, lw_inc_flux_isotropic, sw_k_floor, sw_dir_clamp)."""
    o = _SolverOptions()
    lib().oracle_default_solver_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, v)
    return o


def rte_lw(tau, lay_source, lev_source_inc, lev_source_dec, sfc_emis_gpt, sfc_source,
           top_at_1=True, nmus=1, inc_flux=None, options=None):
    """inc_flux: (ng, ncol) incident diffuse flux at the top of the domain, or None."""
    ng, nlay, ncol = tau.shape
    fu = np.empty((nlay + 1, ncol))
    fd = np.empty_like(fu)
    opt = options if options is not None else solver_options()
    inc = None if inc_flux is None else _f64(inc_flux)
    lib().oracle_rte_lw_opt(ncol, nlay, ng, int(top_at_1), nmus, _p(_f64(tau)), _p(_f64(lay_source)),
                            _p(_f64(lev_source_inc)), _p(_f64(lev_source_dec)), _p(_f64(sfc_emis_gpt)),
                            _p(_f64(sfc_source)), _p(inc), C.byref(opt), _p(fu), _p(fd))
    return fu, fd


def rte_sw(tau, ssa, g, mu0, toa, alb_dir_gpt, alb_dif_gpt, top_at_1=True, options=None):
    ngp, nlay, ncol = tau.shape
    fu = np.empty((nlay + 1, ncol))
    fd = np.empty_like(fu)
    fdir = np.empty_like(fu)
    opt = options if options is not None else solver_options()
    lib().oracle_rte_sw_opt(ncol, nlay, ngp, int(top_at_1), _p(_f64(tau)), _p(_f64(ssa)), _p(_f64(g)),
                            _p(_f64(mu0)), _p(_f64(toa)), _p(_f64(alb_dir_gpt)), _p(_f64(alb_dif_gpt)),
                            C.byref(opt), _p(fu), _p(fd), _p(fdir))
    return fu, fd, fdir


def rte_lw_gpt(tau, lay_source, lev_source_inc, lev_source_dec, sfc_emis_gpt, sfc_source, top_at_1=True, nmus=1,
               inc_flux=None, options=None):
    """Spectral fluxes (ng, nlay+1, ncol) as lw_solver_noscat_GaussQuad returns them, before sum_broadband."""
    ng, nlay, ncol = tau.shape
    gu = np.empty((ng, nlay + 1, ncol))
    gd = np.empty_like(gu)
    opt = options if options is not None else solver_options()
    inc = None if inc_flux is None else _f64(inc_flux)
    lib().oracle_rte_lw_gpt(ncol, nlay, ng, int(top_at_1), nmus, _p(_f64(tau)), _p(_f64(lay_source)),
                            _p(_f64(lev_source_inc)), _p(_f64(lev_source_dec)), _p(_f64(sfc_emis_gpt)),
                            _p(_f64(sfc_source)), _p(inc), C.byref(opt), None, None, _p(gu), _p(gd))
    return gu, gd


def rte_sw_gpt(tau, ssa, g, mu0, toa, alb_dir_gpt, alb_dif_gpt, top_at_1=True, inc_flux_dif=None, options=None):
    """Spectral fluxes (ng, nlay+1, ncol) up, down (total), direct as sw_solver_2stream returns them."""
    ngp, nlay, ncol = tau.shape
    gu = np.empty((ngp, nlay + 1, ncol))
    gd = np.empty_like(gu)
    gr = np.empty_like(gu)
    opt = options if options is not None else solver_options()
    dif = None if inc_flux_dif is None else _f64(inc_flux_dif)
    lib().oracle_rte_sw_gpt(ncol, nlay, ngp, int(top_at_1), _p(_f64(tau)), _p(_f64(ssa)), _p(_f64(g)),
                            _p(_f64(mu0)), _p(_f64(toa)), _p(dif), _p(_f64(alb_dir_gpt)), _p(_f64(alb_dif_gpt)),
                            C.byref(opt), None, None, None, _p(gu), _p(gd), _p(gr))
    return gu, gd, gr


def lw_pipeline(model, plev, tlay, tlev, tsfc, gases, sfc_emis, block=1, nthreads=1, nmus=1):
    """gas_optics_int + rte_lw block by block (ecckd_rfmip_lw.F90:107-136)."""
    plev, tlay, tlev, tsfc, sfc_emis = map(_f64, (plev, tlay, tlev, tsfc, sfc_emis))
    nlay, ncol = tlay.shape
    m = model.cstruct()
    gc = _GC(gases)
    fu = np.empty((nlay + 1, ncol))
    fd = np.empty_like(fu)
    rc = lib().oracle_lw_pipeline(C.byref(m), ncol, nlay, int(block), int(nthreads), _p(plev),
                                  _p(tlay), _p(tlev), _p(tsfc), C.byref(gc.c), _p(sfc_emis),
                                  int(nmus), _p(fu), _p(fd))
    if rc:
        raise RuntimeError("oracle_lw_pipeline failed")
    return fu, fd
