"""rte-ecckd hot path for MI355X -- Python host side.

This package is a thin mirror of the reference's Fortran interface on top of the C ABI in
``include/ecckd_hip.h`` (``librte_ecckd_hip.so``, hand-written HIP for gfx950):

* :class:`GasOpticsEcckd`  <-> ``type(ty_gas_optics_ecckd)`` (src/gas_optics_ecckd.f90:23-48)
  with ``load()`` (= ``load_and_init``, example/rfmip-rad-irf/mo_load_coefficients.F90:19) and
  the generic ``gas_optics()`` (LW: ``gas_optics_int`` :381, SW: ``gas_optics_ext`` :431);
* :class:`GasConcs`, :class:`OpticalProps1scl`, :class:`OpticalProps2str`,
  :class:`SourceFuncLW`, :class:`FluxesBroadband` <-> the RTE-RRTMGP types the reference
  passes around (only the members it touches);
* :func:`rte_lw`, :func:`rte_sw` <-> RTE-RRTMGP's solvers as called at
  ecckd_rfmip_lw.F90:130-135 and ecckd_rfmip_sw.F90:148-154.

Error behaviour follows the reference: the entry points return a message string, empty on
success.  Arrays are numpy (host: staged through the GPU by the library) or torch CUDA tensors
(device resident, asynchronous on the current stream); either way C-ordered with the REVERSE
of the Fortran shape, i.e. the reference's column-major memory: ``tau`` is
``(ngpt, nlay, ncol)``, ``plev`` is ``(nlay+1, ncol)``, fluxes are ``(nlay+1, ncol)``.

There is no CPU fallback anywhere in this package: without the HIP library and a GPU every
compute call raises / returns an error.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("ECCKD_LIB", os.path.join(_HERE, "librte_ecckd_hip.so"))   # override: kernel experiments
HOST, DEVICE = 0, 1
NAME_LEN = 32

_SOURCES = ["kernels_gas_fused.hip", "kernels_tau.hip", "kernels_planck.hip", "kernels_rte_lw.hip", "kernels_rte_lw_split.hip",
            "kernels_rte_sw.hip", "kernels_rte_sw_sys.hip", "kernels_rte_gpt.hip",
            "capi.cpp", "nc_capi.cpp", "model.cpp", "cdf1.cpp"]
_HEADERS = ["kernels.hpp", "wave_pair.hpp", "sw_two_stream.hpp", "sw_two_stream_body.inc", "lw_layer.hpp", "model.hpp", "cdf1.hpp", os.path.join("..", "..", "include", "ecckd_hip.h"),
            os.path.join("..", "..", "include", "ecckd_nc.h"), os.path.join("..", "..", "include", "rte_kernels_hip.h")]
# second library: RTE-RRTMGP's kernel-level bind(C) names over the C ABI of the first (include/rte_kernels_hip.h)
RTE_KERNELS_LIB = os.path.join(_HERE, "librte_kernels_hip.so")
_RTE_KERNELS_SRC = "rte_kernels_capi.cpp"
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def build(force=False, verbose=False):
    """Compile the HIP library for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, s) for s in _SOURCES]
    deps = srcs + [os.path.join(_CSRC, h) for h in _HEADERS] + [os.path.join(_CSRC, _RTE_KERNELS_SRC)]
    if not force and os.path.exists(LIB_PATH) and os.path.exists(RTE_KERNELS_LIB):
        t = min(os.path.getmtime(LIB_PATH), os.path.getmtime(RTE_KERNELS_LIB))
        if all(os.path.getmtime(d) <= t for d in deps):
            return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # one object per source (in parallel, recompiled only when the source or a header is newer), then the link
    objdir = os.path.join(_HERE, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.join(_CSRC, h)) for h in _HEADERS)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s).rsplit(".", 1)[0] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_t):
            cmd = [hipcc] + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-rpath,/opt/rocm/lib", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    cmd = [hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", RTE_KERNELS_LIB, os.path.join(_CSRC, _RTE_KERNELS_SRC),
           "-L" + _HERE, "-lrte_ecckd_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


FORTRAN_DIR = os.path.join(_HERE, "fortran")
FORTRAN_BUILD = os.path.join(FORTRAN_DIR, "build")
FORTRAN_DRIVER = os.path.join(FORTRAN_BUILD, "ecckd_driver")
RFMIP_LW = os.path.join(FORTRAN_BUILD, "ecckd_rfmip_lw")
RFMIP_SW = os.path.join(FORTRAN_BUILD, "ecckd_rfmip_sw")
_FORTRAN_MODULES = ["mo_rte_min.F90", "mo_ecckd_device.F90", "gas_optics_ecckd.F90", "mo_rte_solvers.F90", "rfmip_support.F90"]
_FORTRAN_PROGRAMS = [("ecckd_driver", "ecckd_driver.F90", []), ("ecckd_rfmip_lw", "ecckd_rfmip.F90", []),
                     ("ecckd_rfmip_sw", "ecckd_rfmip.F90", ["-DSHORTWAVE"])]


def build_fortran(force=False, verbose=False):
    """Compile the Fortran drop-in module, the solver shims, the RFMIP support modules and the host
    programs (ecckd_driver, ecckd_rfmip_lw, ecckd_rfmip_sw) with amdflang and link them against
    librte_ecckd_hip.so.  Returns the path of ecckd_driver, or None if no Fortran compiler is
    installed (the C ABI and the Python mirror do not need one)."""
    fc = os.environ.get("FC", "/opt/rocm/bin/amdflang")
    if not os.path.exists(fc):
        return None
    srcs = [os.path.join(FORTRAN_DIR, s) for s in _FORTRAN_MODULES + sorted({p[1] for p in _FORTRAN_PROGRAMS})]
    exes = [os.path.join(FORTRAN_BUILD, p[0]) for p in _FORTRAN_PROGRAMS]
    if not force and all(os.path.exists(e) for e in exes):
        t = min(os.path.getmtime(e) for e in exes)
        if all(os.path.getmtime(d) <= t for d in srcs + [LIB_PATH]):
            return FORTRAN_DRIVER
    os.makedirs(FORTRAN_BUILD, exist_ok=True)

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    objs = []
    for s in _FORTRAN_MODULES:
        o = os.path.join(FORTRAN_BUILD, s[:-4] + ".o")
        run([fc, "-O2", "-module-dir", FORTRAN_BUILD, "-I" + FORTRAN_BUILD, "-c", os.path.join(FORTRAN_DIR, s), "-o", o])
        objs.append(o)
    for exe, src, flags in _FORTRAN_PROGRAMS:
        o = os.path.join(FORTRAN_BUILD, exe + ".o")
        run([fc, "-O2", "-cpp"] + flags + ["-module-dir", FORTRAN_BUILD, "-I" + FORTRAN_BUILD, "-c",
                                           os.path.join(FORTRAN_DIR, src), "-o", o])
        run([fc, "-o", os.path.join(FORTRAN_BUILD, exe), o] + objs +
            ["-L" + _HERE, "-lrte_ecckd_hip", "-Wl,-rpath,$ORIGIN/../..", "-Wl,-rpath,/opt/rocm/lib"])
    return FORTRAN_DRIVER


_lib = None
_dp = C.POINTER(C.c_double)


def lib():
    """Load librte_ecckd_hip.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("librte_ecckd_hip.so is missing: run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` (there is no CPU fallback)")
    try:  # share torch's HIP runtime when torch is around (it must be loaded first)
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    L = C.CDLL(LIB_PATH)
    L.ecckd_last_error.restype = C.c_char_p
    L.ecckd_build_info.restype = C.c_char_p
    for f in ("press_min", "press_max", "temp_min", "temp_max", "total_solar_irradiance"):
        getattr(L, "ecckd_model_get_" + f).restype = C.c_double
        getattr(L, "ecckd_model_get_" + f).argtypes = [C.c_void_p]
    for f in ("ngpt", "nband", "ngas", "device"):
        getattr(L, "ecckd_model_get_" + f).argtypes = [C.c_void_p]
    L.ecckd_model_source_is_internal.argtypes = [C.c_void_p]
    L.ecckd_model_source_is_external.argtypes = [C.c_void_p]
    L.ecckd_model_destroy.argtypes = [C.c_void_p]
    L.ecckd_model_destroy.restype = None
    L.ecckd_model_add_gas.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_double, C.c_void_p]
    L.ecckd_planck_sources.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_void_p]
    L.ecckd_gas_optics_plan.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_void_p]
    L.ecckd_gas_optics_plan_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_void_p, C.c_int,
                                           C.c_void_p]
    L.ecckd_set_solver_option.argtypes = [C.c_char_p, C.c_double]
    L.ecckd_get_solver_option.argtypes = [C.c_char_p, C.POINTER(C.c_double)]
    for f in ("ecckd_rte_lw_scratch_bytes", "ecckd_rte_sw_scratch_bytes"):
        getattr(L, f).restype = C.c_size_t
        getattr(L, f).argtypes = [C.c_int, C.c_int, C.c_int]
    L.ecckd_rte_sw_tail_scratch_bytes.restype = C.c_size_t
    L.ecckd_rte_sw_tail_scratch_bytes.argtypes = [C.c_int] * 4
    L.ecckd_rte_lw_tail_scratch_bytes.restype = C.c_size_t
    L.ecckd_rte_lw_tail_scratch_bytes.argtypes = [C.c_int] * 6
    L.ecckd_set_stream_scratch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    L.ecckd_release_scratch.argtypes = [C.c_int]
    _lib = L
    return L


def last_error():
    return lib().ecckd_last_error().decode()


FAST, REFERENCE_ORDER = 0, 1


def set_arithmetic(mode):
    """0 = fast (fused kernel, re-associated FMAs; default), 1 = reference expression order
    (bit-faithful kernels); see ecckd_set_arithmetic in include/ecckd_hip.h."""
    if lib().ecckd_set_arithmetic(int(mode)):
        raise ValueError(last_error())


def get_arithmetic():
    return lib().ecckd_get_arithmetic()


SOLVER_OPTIONS = ("lw_tau_thresh", "lw_series_terms", "lw_inc_flux_isotropic", "sw_k_floor", "sw_dir_clamp", "lw_solver",
                  "lw_split_seg", "gas_merge_scalars", "lw_tail_split", "sw_tail_split", "sw_solver", "gas_slab_f32")


def set_solver_option(name, value):
    """Version switches of rte_lw / rte_sw (ecckd_set_solver_option in include/ecckd_hip.h); the defaults
    are the RTE-RRTMGP v1.5-era forms."""
    if lib().ecckd_set_solver_option(name.encode(), float(value)):
        raise ValueError(last_error())


def get_solver_option(name):
    v = C.c_double()
    if lib().ecckd_get_solver_option(name.encode(), C.byref(v)):
        raise ValueError(last_error())
    return v.value


def solver_options():
    """The active switches as a dict (bench.py prints it)."""
    return {n: get_solver_option(n) for n in SOLVER_OPTIONS}


def reset_solver_options():
    """The version switches back to their (v1.5-era) defaults; the implementation choices (lw_solver, lw_split_seg)
    are left alone."""
    for n, v in (("lw_tau_thresh", 0.0), ("lw_series_terms", 2), ("lw_inc_flux_isotropic", 0), ("sw_k_floor", 1e-12),
                 ("sw_dir_clamp", 0)):
        set_solver_option(n, v)


def rte_sw_scratch_bytes(ncol, nlay, ngpt):
    return int(lib().ecckd_rte_sw_scratch_bytes(int(ncol), int(nlay), int(ngpt)))


def rte_lw_scratch_bytes(ncol, nlay, ngpt):
    return int(lib().ecckd_rte_lw_scratch_bytes(int(ncol), int(nlay), int(ngpt)))


def rte_sw_tail_scratch_bytes(ncol, nlay, ngpt, device=0):
    return int(lib().ecckd_rte_sw_tail_scratch_bytes(int(device), int(ncol), int(nlay), int(ngpt)))


def rte_lw_tail_scratch_bytes(ncol, nlay, ngpt, n_gauss_angles=1, single_precision=False, device=0):
    return int(lib().ecckd_rte_lw_tail_scratch_bytes(int(device), int(ncol), int(nlay), int(ngpt), int(n_gauss_angles),
                                                     int(bool(single_precision))))


def set_stream_scratch(buffer, device=None, stream=None):
    """Hand a caller-owned device buffer (a torch CUDA uint8/any tensor, or None to take it back) to the solver
    calls on `stream` (default: torch's current stream) -- ecckd_set_stream_scratch."""
    import torch
    st = torch.cuda.current_stream() if stream is None else stream
    dev = (buffer.device.index if buffer is not None else torch.cuda.current_device()) if device is None else device
    ptr = C.c_void_p(buffer.data_ptr()) if buffer is not None else None
    n = buffer.numel() * buffer.element_size() if buffer is not None else 0
    if lib().ecckd_set_stream_scratch(int(dev or 0), C.c_void_p(st.cuda_stream), ptr, n):
        raise ValueError(last_error())


def release_scratch(device=0):
    if lib().ecckd_release_scratch(int(device)):
        raise RuntimeError(last_error())


# ------------------------------------------------------------------------------------------
# array plumbing: numpy (host) or torch.cuda tensors (device)
# ------------------------------------------------------------------------------------------
def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _space_of(arrays):
    dev = [a for a in arrays if a is not None and _is_torch(a) and a.is_cuda]
    if not dev:
        return HOST
    for a in arrays:
        if a is not None and not (_is_torch(a) and a.is_cuda):
            raise TypeError("mixing host and device arrays in one call")
    return DEVICE


def _is_f32(a):
    """True for float32 arrays/tensors (the single-precision flavour of the entry points)."""
    if a is None:
        return False
    if _is_torch(a):
        import torch
        return a.dtype == torch.float32
    return isinstance(a, np.ndarray) and a.dtype == np.float32


def _ptr(a, shape=None, what="array", f32=False):
    """Data pointer of a C-contiguous float64 (or, with f32, float32) array (numpy or torch)."""
    if a is None:
        return None
    want = "float32" if f32 else "float64"
    if _is_torch(a):
        import torch
        if a.dtype != (torch.float32 if f32 else torch.float64) or not a.is_contiguous():
            raise TypeError(what + ": need a contiguous " + want + " tensor")
        if shape is not None and tuple(a.shape) != tuple(shape):
            raise ValueError("%s: shape %s, expected %s" % (what, tuple(a.shape), tuple(shape)))
        return C.c_void_p(a.data_ptr())
    if not isinstance(a, np.ndarray) or a.dtype != (np.float32 if f32 else np.float64) or not a.flags.c_contiguous:
        raise TypeError(what + ": need a C-contiguous " + want + " ndarray")
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("%s: shape %s, expected %s" % (what, a.shape, tuple(shape)))
    return C.c_void_p(a.ctypes.data)


def _stream(space):
    if space == DEVICE:
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return None


def _empty_like_space(shape, like):
    """Uninitialised array in the memory space (and precision, if float32) of `like`."""
    if _is_torch(like):
        import torch
        return torch.empty(shape, dtype=torch.float32 if _is_f32(like) else torch.float64, device=like.device)
    return np.empty(shape, dtype=np.float32 if _is_f32(like) else np.float64)


# ------------------------------------------------------------------------------------------
# RTE-RRTMGP data types, as far as the reference touches them
# ------------------------------------------------------------------------------------------
class GasConcs:
    """``ty_gas_concs``: init / set_vmr / get_vmr / get_gas_names / get_num_gases
    (used at src/gas_optics_ecckd.f90:340-351 and mo_rfmip_io.F90:202-259)."""

    def __init__(self, gas_names=()):
        self.init(gas_names)

    def init(self, gas_names):
        names = [n.strip().lower() for n in gas_names]
        if len(set(names)) != len(names):
            return "ty_gas_concs%init: duplicate gas names aren't allowed"
        if any(len(n) == 0 for n in names):
            return "ty_gas_concs%init: must provide non-empty gas names"
        self._names = names
        self._conc = {}
        return ""

    def set_vmr(self, gas, w):
        """w: scalar, profile ``(nlay,)``, or full ``(nlay, ncol)`` [Fortran (ncol,nlay)]."""
        gas = gas.strip().lower()
        if gas not in self._names:
            return "ty_gas_concs%set_vmr: trying to set " + gas + " but name not provided at initialization"
        if np.isscalar(w):
            if w < 0 or w > 1:
                return "ty_gas_concs%set_vmr: concentrations should be >= 0, <= 1"
            self._conc[gas] = float(w)
        else:
            if w.ndim not in (1, 2):
                return "ty_gas_concs%set_vmr: need a scalar, (nlay) or (ncol,nlay) array"
            self._conc[gas] = w
        return ""

    def set_vmr_column(self, gas, w):
        """Extension of the C ABI (per-column value, broadcast over layers)."""
        gas = gas.strip().lower()
        if gas not in self._names:
            return "ty_gas_concs%set_vmr: trying to set " + gas + " but name not provided at initialization"
        self._conc[gas] = ("column", w)
        return ""

    def get_num_gases(self):
        return len(self._names)

    def get_gas_names(self):
        return list(self._names)

    def entries(self, ncol, nlay, known=None):
        """[(name, array-or-None, col_stride, lay_stride, scalar)] in gas order; raises KeyError
        with the ty_gas_concs%get_vmr message for a gas that was never set.  `known`: gas names of the
        k-distribution -- like the reference (src/gas_optics_ecckd.f90:348-364), get_vmr is only
        consulted for gases the model holds a table for; the others cross the boundary as a name with
        a null pointer (nothing is staged for them, and an unset one is not an error)."""
        out = []
        for n in self._names:
            if known is not None and n not in known:
                out.append((n, None, 0, 0, 0.0))
                continue
            if n not in self._conc:
                raise KeyError("ty_gas_concs%get_vmr; gas " + n + " not found")
            w = self._conc[n]
            if isinstance(w, float):
                out.append((n, None, 0, 0, w))
            elif isinstance(w, tuple):
                if tuple(w[1].shape) != (ncol,):
                    raise KeyError("ty_gas_concs%get_vmr; gas " + n + " array is inconsistent with ncol")
                out.append((n, w[1], 1, 0, 0.0))
            elif w.ndim == 1:
                if w.shape[0] != nlay:
                    raise KeyError("ty_gas_concs%get_vmr; gas " + n + " array is inconsistent with nlay")
                out.append((n, w, 0, 1, 0.0))
            else:
                if tuple(w.shape) != (nlay, ncol):
                    raise KeyError("ty_gas_concs%get_vmr; gas " + n + " array is inconsistent with ncol/nlay")
                out.append((n, w, 1, ncol, 0.0))
        return out


class OpticalProps1scl:
    """``ty_optical_props_1scl``: tau(ncol,nlay,ngpt) (+ band structure of the parent)."""

    def __init__(self):
        self.tau = None
        self.band2gpt = None

    def alloc_1scl(self, ncol, nlay, spectral_desc, like=None):
        self.band2gpt = spectral_desc.get_band2gpt()
        ng = spectral_desc.get_ngpt()
        self.tau = _empty_like_space((ng, nlay, ncol), like if like is not None else np.empty(0))
        return ""

    def get_ngpt(self):
        return self.tau.shape[0]


class OpticalProps2str(OpticalProps1scl):
    """``ty_optical_props_2str``: tau, ssa, g."""

    def __init__(self):
        super().__init__()
        self.ssa = None
        self.g = None

    def alloc_2str(self, ncol, nlay, spectral_desc, like=None):
        self.alloc_1scl(ncol, nlay, spectral_desc, like)
        self.ssa = _empty_like_space(tuple(self.tau.shape), self.tau)
        self.g = _empty_like_space(tuple(self.tau.shape), self.tau)
        return ""


class SourceFuncLW:
    """``ty_source_func_lw``: lay_source, lev_source_inc, lev_source_dec, sfc_source."""

    def __init__(self):
        self.lay_source = self.lev_source_inc = self.lev_source_dec = self.sfc_source = None
        self.levels_shared = False   # True after ecckd's gas_optics: one value per level (:419-424)

    def alloc(self, ncol, nlay, spectral_desc, like=None):
        ng = spectral_desc.get_ngpt()
        like = like if like is not None else np.empty(0)
        self.lay_source = _empty_like_space((ng, nlay, ncol), like)
        self.lev_source_inc = _empty_like_space((ng, nlay, ncol), like)
        self.lev_source_dec = _empty_like_space((ng, nlay, ncol), like)
        self.sfc_source = _empty_like_space((ng, ncol), like)
        return ""


class FluxesBroadband:
    """``ty_fluxes_broadband``: flux_up, flux_dn (ncol,nlay+1) [+ flux_dn_dir]."""

    def __init__(self, flux_up=None, flux_dn=None, flux_dn_dir=None):
        self.flux_up, self.flux_dn, self.flux_dn_dir = flux_up, flux_dn, flux_dn_dir


class FluxesByband(FluxesBroadband):
    """``ty_fluxes_byband``: bnd_flux_up, bnd_flux_dn (nband,nlay+1,ncol in numpy order) [+ bnd_flux_dn_dir],
    and optionally the broadband members of the parent."""

    def __init__(self, bnd_flux_up, bnd_flux_dn, bnd_flux_dn_dir=None, flux_up=None, flux_dn=None, flux_dn_dir=None):
        super().__init__(flux_up, flux_dn, flux_dn_dir)
        self.bnd_flux_up, self.bnd_flux_dn, self.bnd_flux_dn_dir = bnd_flux_up, bnd_flux_dn, bnd_flux_dn_dir


# ------------------------------------------------------------------------------------------
# ty_gas_optics_ecckd
# ------------------------------------------------------------------------------------------
class GasOpticsEcckd:
    """``type(ty_gas_optics_ecckd)`` backed by a device-resident model handle."""

    def __init__(self):
        self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.finalize()
        except Exception:
            pass

    def finalize(self):
        if self._h:
            lib().ecckd_model_destroy(self._h)
            self._h = C.c_void_p(None)

    # -- construction ---------------------------------------------------------------
    def load(self, filename, available_gases=None, device=0):
        """``load_and_init(ecckd, filename, available_gases)``; ``available_gases`` is accepted
        and ignored exactly as in the reference (mo_load_coefficients.F90:19,23)."""
        self.finalize()
        h = C.c_void_p(None)
        if lib().ecckd_model_load(os.fsencode(filename), int(device), C.byref(h)):
            return last_error()
        self._h = h
        return ""

    def init_from_tables(self, log_pressure, temperature, gases, planck=None, solar=None, bands=None,
                         device=0):
        """Fill the public members directly (src/gas_optics_ecckd.f90:24-36).  ``temperature`` is
        ``(nt,np)``, tables ``(nv,nt,np,ng)``; ``planck=(temperature_planck, planck_function(ntp,ng))``;
        ``solar=(solar_irradiance, rayleigh)``; ``bands=(band_lims_wvn(nband,2), band2gpt(nband,2))``;
        ``gases`` = list of dicts(name, code, composite_only, mole_fraction, reference_mole_fraction,
        coefficient)."""
        self.finalize()
        L = lib()
        f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        lp, T = f8(log_pressure), f8(temperature)
        ng = gases[0]["coefficient"].shape[-1]
        h = C.c_void_p(None)
        if L.ecckd_model_begin(ng, lp.shape[0], T.shape[0], _ptr(lp), _ptr(T), C.byref(h)):
            return last_error()
        try:
            if planck is not None:
                tp, pf = f8(planck[0]), f8(planck[1])
                if L.ecckd_model_set_planck(h, tp.shape[0], _ptr(tp), _ptr(pf)):
                    raise RuntimeError(last_error())
            if solar is not None:
                si, ray = f8(solar[0]), f8(solar[1])
                if L.ecckd_model_set_solar(h, _ptr(si), _ptr(ray)):
                    raise RuntimeError(last_error())
            if bands is not None:
                lims = f8(bands[0])
                b2g = np.ascontiguousarray(bands[1], dtype=np.int32)
                if L.ecckd_model_set_bands(h, b2g.shape[0], _ptr(lims), C.c_void_p(b2g.ctypes.data)):
                    raise RuntimeError(last_error())
            for g in gases:
                coef = f8(g["coefficient"])
                mf = g.get("mole_fraction")
                mf = None if mf is None else f8(mf)
                if L.ecckd_model_add_gas(h, g["name"].encode(), int(g["code"]), int(g.get("composite_only", 0)),
                                         coef.shape[0] if coef.ndim == 4 else 1, _ptr(mf),
                                         float(g.get("reference_mole_fraction", 0.0)), _ptr(coef)):
                    raise RuntimeError(last_error())
            if L.ecckd_model_finalize(h, int(device)):
                raise RuntimeError(last_error())
        except RuntimeError as e:
            L.ecckd_model_destroy(h)
            return str(e)
        self._h = h
        return ""

    # -- type-bound getters (src/gas_optics_ecckd.f90:477-553 + parent) -----------------
    def _need(self):
        if not self._h:
            raise RuntimeError("ty_gas_optics_ecckd: not loaded")
        return self._h

    def get_ngpt(self):
        return lib().ecckd_model_get_ngpt(self._need())

    def get_nband(self):
        return lib().ecckd_model_get_nband(self._need())

    def get_ngas(self):
        return lib().ecckd_model_get_ngas(self._need())

    def get_gases(self):
        out = []
        buf = C.create_string_buffer(NAME_LEN)
        for i in range(self.get_ngas()):
            lib().ecckd_model_get_gas_name(self._need(), i, buf)
            out.append(buf.value.decode())
        return out

    def source_is_internal(self):
        return bool(lib().ecckd_model_source_is_internal(self._need()))

    def source_is_external(self):
        return bool(lib().ecckd_model_source_is_external(self._need()))

    def plan(self, ncol, nlay, gas_names, single_precision=False, scalar_gases=()):
        """``ecckd_gas_optics_plan_ex``: how gas_optics would run for this model, gas list and size (works
        without a GPU on a ``device=-1`` model).  ``scalar_gases`` names the gases that would be passed as
        one number (they can share the merged slot).  Returns a dict, or raises RuntimeError with the
        library's message."""
        names = b"".join(n.strip().lower().encode().ljust(NAME_LEN, b" ") for n in gas_names)
        sc = {n.strip().lower() for n in scalar_gases}
        flags = (C.c_int * max(1, len(gas_names)))(*[int(n.strip().lower() in sc) for n in gas_names])
        out = (C.c_int * 10)()
        if lib().ecckd_gas_optics_plan_ex(self._need(), int(ncol), int(nlay), int(bool(single_precision)),
                                          len(gas_names), names, flags, 10, out):
            raise RuntimeError(last_error())
        keys = ("passes", "fused", "planck_fused", "slab_rows", "planck_rows", "col_chunks", "lds_bytes", "g_chunk",
                "slots", "merged")
        return dict(zip(keys, list(out)))

    def get_press_min(self):
        return lib().ecckd_model_get_press_min(self._need())

    def get_press_max(self):
        return lib().ecckd_model_get_press_max(self._need())

    def get_temp_min(self):
        return lib().ecckd_model_get_temp_min(self._need())

    def get_temp_max(self):
        return lib().ecckd_model_get_temp_max(self._need())

    def get_total_solar_irradiance(self):
        return lib().ecckd_model_get_total_solar_irradiance(self._need())

    def get_band2gpt(self):
        b = np.zeros((self.get_nband(), 2), dtype=np.int32)
        lib().ecckd_model_get_band2gpt(self._need(), C.c_void_p(b.ctypes.data))
        return b

    def get_band_lims_wavenumber(self):
        b = np.zeros((self.get_nband(), 2), dtype=np.float64)
        lib().ecckd_model_get_band_lims_wvn(self._need(), C.c_void_p(b.ctypes.data))
        return b

    def get_device(self):
        return lib().ecckd_model_get_device(self._need())

    # -- gas_optics ---------------------------------------------------------------------
    def _gas_args(self, gas_desc, ncol, nlay, space, f32=False):
        ent = gas_desc.entries(ncol, nlay, known=set(self.get_gases()))
        n = len(ent)
        names = b"".join(e[0].encode().ljust(NAME_LEN, b" ") for e in ent)
        keep = []
        ptrs = (C.c_void_p * max(n, 1))()
        for i, e in enumerate(ent):
            if e[1] is None:
                ptrs[i] = None
            else:
                if (_is_torch(e[1]) and e[1].is_cuda) != (space == DEVICE):
                    raise TypeError("gas " + e[0] + ": vmr array is not in the same memory space as the inputs")
                p = _ptr(e[1], what="vmr of " + e[0], f32=f32)
                keep.append(e[1])
                ptrs[i] = p.value
        cs = (C.c_longlong * max(n, 1))(*[e[2] for e in ent])
        ls = (C.c_longlong * max(n, 1))(*[e[3] for e in ent])
        sc = (C.c_double * max(n, 1))(*[e[4] for e in ent])
        return n, names, ptrs, cs, ls, sc, keep

    def gas_optics(self, play, plev, tlay, *args, **kw):
        """Generic ``gas_optics``: ``(play, plev, tlay, tsfc, gas_desc, optical_props, sources,
        col_dry=None, tlev=None)`` -> gas_optics_int; ``(play, plev, tlay, gas_desc, optical_props,
        toa_src, col_dry=None)`` -> gas_optics_ext.  Returns the error message ('' = success)."""
        if len(args) >= 1 and isinstance(args[0], GasConcs):
            return self.gas_optics_ext(play, plev, tlay, *args, **kw)
        return self.gas_optics_int(play, plev, tlay, *args, **kw)

    def gas_optics_int(self, play, plev, tlay, tsfc, gas_desc, optical_props, sources, col_dry=None,
                       tlev=None):
        """float64 arrays -> ecckd_gas_optics_lw; float32 arrays -> ecckd_gas_optics_lw_f32."""
        nlay, ncol = tlay.shape
        ng = self.get_ngpt()
        f32 = _is_f32(plev)
        try:
            space = _space_of([plev, tlay, tsfc, tlev, optical_props.tau, sources.lay_source])
            n, names, ptrs, cs, ls, sc, keep = self._gas_args(gas_desc, ncol, nlay, space, f32)
        except KeyError as e:
            return str(e.args[0])
        fn = lib().ecckd_gas_optics_lw_f32 if f32 else lib().ecckd_gas_optics_lw
        P = lambda a, shape, what: _ptr(a, shape, what, f32)
        rc = fn(self._need(), ncol, nlay, P(plev, (nlay + 1, ncol), "plev"), P(tlay, (nlay, ncol), "tlay"),
                P(tsfc, (ncol,), "tsfc"), P(tlev, (nlay + 1, ncol), "tlev"), n, names, ptrs, cs, ls, sc,
                P(optical_props.tau, (ng, nlay, ncol), "tau"), P(sources.lay_source, (ng, nlay, ncol), "lay_source"),
                P(sources.lev_source_inc, (ng, nlay, ncol), "lev_source_inc"),
                P(sources.lev_source_dec, (ng, nlay, ncol), "lev_source_dec"),
                P(sources.sfc_source, (ng, ncol), "sfc_source"), space, _stream(space))
        sources.levels_shared = rc == 0 and tlev is not None
        return last_error() if rc else ""

    def planck_sources(self, tlay, tsfc, sources, tlev=None):
        """``ecckd_planck_sources``: the four Planck source arrays alone (device tensors, fp64)."""
        nlay, ncol = tlay.shape
        ng = self.get_ngpt()
        space = _space_of([tlay, tsfc, sources.lay_source])
        rc = lib().ecckd_planck_sources(
            self._need(), ncol, nlay, _ptr(tlay, (nlay, ncol), "tlay"),
            None if tlev is None else _ptr(tlev, (nlay + 1, ncol), "tlev"), _ptr(tsfc, (ncol,), "tsfc"),
            _ptr(sources.lay_source, (ng, nlay, ncol), "lay_source"),
            None if tlev is None else _ptr(sources.lev_source_inc, (ng, nlay, ncol), "lev_source_inc"),
            None if tlev is None else _ptr(sources.lev_source_dec, (ng, nlay, ncol), "lev_source_dec"),
            _ptr(sources.sfc_source, (ng, ncol), "sfc_source"), space, _stream(space))
        return last_error() if rc else ""

    def gas_optics_tau(self, plev, tlay, gas_desc, optical_props):
        """``ecckd_gas_optics_lw_tau``: gas_optical_depth alone (tau only), device tensors."""
        nlay, ncol = tlay.shape
        ng = self.get_ngpt()
        try:
            space = _space_of([plev, tlay, optical_props.tau])
            n, names, ptrs, cs, ls, sc, keep = self._gas_args(gas_desc, ncol, nlay, space)
        except KeyError as e:
            return str(e.args[0])
        rc = lib().ecckd_gas_optics_lw_tau(self._need(), ncol, nlay, _ptr(plev, (nlay + 1, ncol), "plev"),
                                           _ptr(tlay, (nlay, ncol), "tlay"), n, names, ptrs, cs, ls, sc,
                                           _ptr(optical_props.tau, (ng, nlay, ncol), "tau"), space, _stream(space))
        return last_error() if rc else ""

    def rte_lw_fused(self, optical_props, top_at_1, tlay, tlev, tsfc, sfc_emis, fluxes, n_gauss_angles=1, inc_flux=None):
        """``ecckd_rte_lw_fused``: rte_lw that recomputes the Planck sources from the temperatures (device tensors)."""
        ng, nlay, ncol = optical_props.tau.shape
        space = _space_of([optical_props.tau, tlay, tlev, tsfc, sfc_emis, inc_flux, fluxes.flux_up, fluxes.flux_dn])
        rc = lib().ecckd_rte_lw_fused(self._need(), ncol, nlay, int(bool(top_at_1)), int(n_gauss_angles),
                                      _ptr(optical_props.tau), _ptr(tlay, (nlay, ncol), "tlay"),
                                      _ptr(tlev, (nlay + 1, ncol), "tlev"), _ptr(tsfc, (ncol,), "tsfc"),
                                      _ptr(sfc_emis, (ncol, self.get_nband()), "sfc_emis"),
                                      _ptr(inc_flux, (ng, ncol), "inc_flux"), _ptr(fluxes.flux_up, (nlay + 1, ncol), "flux_up"),
                                      _ptr(fluxes.flux_dn, (nlay + 1, ncol), "flux_dn"), space, _stream(space))
        return last_error() if rc else ""

    def lw_fluxes(self, plev, tlay, tsfc, tlev, gas_desc, top_at_1, sfc_emis, fluxes, n_gauss_angles=1, inc_flux=None):
        """``ecckd_lw_fluxes``: gas optics + rte_lw in one call for hosts that only need broadband fluxes (tau in
        library-owned scratch, sources recomputed in the solver); numpy or device tensors."""
        nlay, ncol = tlay.shape
        ng = self.get_ngpt()
        try:
            space = _space_of([plev, tlay, tsfc, tlev, sfc_emis, inc_flux, fluxes.flux_up, fluxes.flux_dn])
            n, names, ptrs, cs, ls, sc, keep = self._gas_args(gas_desc, ncol, nlay, space)
        except KeyError as e:
            return str(e.args[0])
        rc = lib().ecckd_lw_fluxes(self._need(), ncol, nlay, _ptr(plev, (nlay + 1, ncol), "plev"),
                                   _ptr(tlay, (nlay, ncol), "tlay"), _ptr(tsfc, (ncol,), "tsfc"),
                                   _ptr(tlev, (nlay + 1, ncol), "tlev"), n, names, ptrs, cs, ls, sc, int(bool(top_at_1)),
                                   int(n_gauss_angles), _ptr(sfc_emis, (ncol, self.get_nband()), "sfc_emis"),
                                   _ptr(inc_flux, (ng, ncol), "inc_flux"), _ptr(fluxes.flux_up, (nlay + 1, ncol), "flux_up"),
                                   _ptr(fluxes.flux_dn, (nlay + 1, ncol), "flux_dn"), space, _stream(space))
        return last_error() if rc else ""

    def gas_optics_ext(self, play, plev, tlay, gas_desc, optical_props, toa_src, col_dry=None):
        nlay, ncol = tlay.shape
        ng = self.get_ngpt()
        """float64 arrays -> ecckd_gas_optics_sw; float32 arrays -> ecckd_gas_optics_sw_f32."""
        two = isinstance(optical_props, OpticalProps2str)
        f32 = _is_f32(plev)
        try:
            space = _space_of([plev, tlay, optical_props.tau, toa_src])
            n, names, ptrs, cs, ls, sc, keep = self._gas_args(gas_desc, ncol, nlay, space, f32)
        except KeyError as e:
            return str(e.args[0])
        P = lambda a, shape, what: _ptr(a, shape, what, f32)
        rc = (lib().ecckd_gas_optics_sw_f32 if f32 else lib().ecckd_gas_optics_sw)(
            self._need(), ncol, nlay, P(plev, (nlay + 1, ncol), "plev"), P(tlay, (nlay, ncol), "tlay"),
            n, names, ptrs, cs, ls, sc, P(optical_props.tau, (ng, nlay, ncol), "tau"),
            P(optical_props.ssa, (ng, nlay, ncol), "ssa") if two else None,
            P(optical_props.g, (ng, nlay, ncol), "g") if two else None,
            P(toa_src, (ng, ncol), "toa_src"), space, _stream(space))
        return last_error() if rc else ""

    def sw_fluxes(self, plev, tlay, gas_desc, top_at_1, mu0, sfc_alb_dir, sfc_alb_dif, fluxes, toa_scale=None):
        """``ecckd_sw_fluxes`` (float32 arrays: ``_f32``): gas optics + rte_sw in one call for hosts that only need
        broadband fluxes -- the total optical depth alone goes through (library-owned) memory, the solver derives
        ssa, g = 0 and the incoming beam from plev and the model's tables as gas_optics_ext does.  ``toa_scale``
        ``(ncol,)``: the drivers' rescaling of the incoming beam (total solar irradiance), or None."""
        nlay, ncol = tlay.shape
        nband = self.get_nband()
        f32 = _is_f32(plev)
        try:
            space = _space_of([plev, tlay, mu0, toa_scale, sfc_alb_dir, sfc_alb_dif, fluxes.flux_up, fluxes.flux_dn])
            n, names, ptrs, cs, ls, sc, keep = self._gas_args(gas_desc, ncol, nlay, space, f32)
        except KeyError as e:
            return str(e.args[0])
        P = lambda a, shape, what: _ptr(a, shape, what, f32)
        rc = (lib().ecckd_sw_fluxes_f32 if f32 else lib().ecckd_sw_fluxes)(
            self._need(), ncol, nlay, P(plev, (nlay + 1, ncol), "plev"), P(tlay, (nlay, ncol), "tlay"), n, names, ptrs, cs,
            ls, sc, int(bool(top_at_1)), P(mu0, (ncol,), "mu0"), P(toa_scale, (ncol,), "toa_scale"),
            P(sfc_alb_dir, (ncol, nband), "sfc_alb_dir"), P(sfc_alb_dif, (ncol, nband), "sfc_alb_dif"),
            P(fluxes.flux_up, (nlay + 1, ncol), "flux_up"), P(fluxes.flux_dn, (nlay + 1, ncol), "flux_dn"),
            P(fluxes.flux_dn_dir, (nlay + 1, ncol), "flux_dn_dir"), space, _stream(space))
        return last_error() if rc else ""


# ------------------------------------------------------------------------------------------
# RTE solvers
# ------------------------------------------------------------------------------------------
def _device_of(a):
    if _is_torch(a) and a.is_cuda:
        return a.device.index if a.device.index is not None else 0
    return 0


def rte_lw(optical_props, top_at_1, sources, sfc_emis, fluxes, n_gauss_angles=1, device=None,
           shared_levels=False, inc_flux=None):
    """``rte_lw(optical_props, top_at_1, sources, sfc_emis(nband,ncol), fluxes, n_gauss_angles=)``
    (ecckd_rfmip_lw.F90:130-135).  ``sfc_emis`` is ``(ncol, nband)`` in numpy order.  float32 arrays
    take the single-precision entry point.  ``shared_levels=True`` asserts that the level sources hold
    one value per level (``sources.levels_shared``, set by ecckd's gas_optics) and takes
    ``ecckd_rte_lw_shared_levels`` (fp64 only).  ``inc_flux`` ``(ngpt, ncol)``: incident diffuse flux at the
    top of the domain (rte_lw's optional argument; ``ecckd_rte_lw_inc_flux`` / ``_f32``, generic solver)."""
    ng, nlay, ncol = optical_props.tau.shape
    b2g = np.ascontiguousarray(optical_props.band2gpt, dtype=np.int32)
    nband = b2g.shape[0]
    f32 = _is_f32(optical_props.tau)
    space = _space_of([optical_props.tau, sources.lay_source, sfc_emis, fluxes.flux_up, fluxes.flux_dn])
    dev = _device_of(optical_props.tau) if device is None else device
    if isinstance(fluxes, FluxesByband):
        if shared_levels:
            return "rte_lw: per-band fluxes are implemented for the generic solver"
        space = _space_of([optical_props.tau, sources.lay_source, sfc_emis, fluxes.bnd_flux_up, fluxes.bnd_flux_dn])
        Pb = lambda a, shape=None, what="array": _ptr(a, shape, what, f32)
        opt = lambda a, what: Pb(a, (nlay + 1, ncol), what) if a is not None else None
        rc = (lib().ecckd_rte_lw_byband_f32 if f32 else lib().ecckd_rte_lw_byband)(
            int(dev), ncol, nlay, ng, int(bool(top_at_1)), int(n_gauss_angles), Pb(optical_props.tau),
            Pb(sources.lay_source, (ng, nlay, ncol), "lay_source"),
            Pb(sources.lev_source_inc, (ng, nlay, ncol), "lev_source_inc"),
            Pb(sources.lev_source_dec, (ng, nlay, ncol), "lev_source_dec"),
            Pb(sources.sfc_source, (ng, ncol), "sfc_source"), nband, C.c_void_p(b2g.ctypes.data),
            Pb(sfc_emis, (ncol, nband), "sfc_emis"), Pb(fluxes.bnd_flux_up, (nband, nlay + 1, ncol), "bnd_flux_up"),
            Pb(fluxes.bnd_flux_dn, (nband, nlay + 1, ncol), "bnd_flux_dn"), opt(fluxes.flux_up, "flux_up"),
            opt(fluxes.flux_dn, "flux_dn"), space, _stream(space))
        return last_error() if rc else ""
    if shared_levels and f32:
        return "rte_lw: shared_levels is implemented for float64 arrays"
    if inc_flux is not None:
        if shared_levels:
            return "rte_lw: inc_flux is implemented for the generic solver"
        space = _space_of([optical_props.tau, sources.lay_source, sfc_emis, inc_flux, fluxes.flux_up, fluxes.flux_dn])
        Pi = lambda a, shape=None, what="array": _ptr(a, shape, what, f32)
        rc = (lib().ecckd_rte_lw_inc_flux_f32 if f32 else lib().ecckd_rte_lw_inc_flux)(
            int(dev), ncol, nlay, ng, int(bool(top_at_1)), int(n_gauss_angles), Pi(optical_props.tau),
            Pi(sources.lay_source, (ng, nlay, ncol), "lay_source"),
            Pi(sources.lev_source_inc, (ng, nlay, ncol), "lev_source_inc"),
            Pi(sources.lev_source_dec, (ng, nlay, ncol), "lev_source_dec"),
            Pi(sources.sfc_source, (ng, ncol), "sfc_source"), nband, C.c_void_p(b2g.ctypes.data),
            Pi(sfc_emis, (ncol, nband), "sfc_emis"), Pi(inc_flux, (ng, ncol), "inc_flux"),
            Pi(fluxes.flux_up, (nlay + 1, ncol), "flux_up"), Pi(fluxes.flux_dn, (nlay + 1, ncol), "flux_dn"),
            space, _stream(space))
        return last_error() if rc else ""
    fn = lib().ecckd_rte_lw_f32 if f32 else (lib().ecckd_rte_lw_shared_levels if shared_levels else lib().ecckd_rte_lw)
    P = lambda a, shape=None, what="array": _ptr(a, shape, what, f32)
    rc = fn(int(dev), ncol, nlay, ng, int(bool(top_at_1)), int(n_gauss_angles), P(optical_props.tau),
            P(sources.lay_source, (ng, nlay, ncol), "lay_source"),
            P(sources.lev_source_inc, (ng, nlay, ncol), "lev_source_inc"),
            P(sources.lev_source_dec, (ng, nlay, ncol), "lev_source_dec"),
            P(sources.sfc_source, (ng, ncol), "sfc_source"), nband, C.c_void_p(b2g.ctypes.data),
            P(sfc_emis, (ncol, nband), "sfc_emis"), P(fluxes.flux_up, (nlay + 1, ncol), "flux_up"),
            P(fluxes.flux_dn, (nlay + 1, ncol), "flux_dn"), space, _stream(space))
    return last_error() if rc else ""


def rte_sw(optical_props, top_at_1, mu0, toa_flux, sfc_alb_dir, sfc_alb_dif, fluxes, device=None):
    """``rte_sw(optical_props, top_at_1, mu0, toa_flux, sfc_alb_dir, sfc_alb_dif, fluxes)``
    (ecckd_rfmip_sw.F90:148-154).  Albedos are ``(ncol, nband)`` in numpy order."""
    if not isinstance(optical_props, OpticalProps2str):
        return "rte_sw: two-stream optical properties required"
    ng, nlay, ncol = optical_props.tau.shape
    b2g = np.ascontiguousarray(optical_props.band2gpt, dtype=np.int32)
    nband = b2g.shape[0]
    space = _space_of([optical_props.tau, mu0, toa_flux, sfc_alb_dir, sfc_alb_dif, fluxes.flux_up])
    dev = _device_of(optical_props.tau) if device is None else device
    if isinstance(fluxes, FluxesByband):
        space = _space_of([optical_props.tau, mu0, toa_flux, sfc_alb_dir, sfc_alb_dif, fluxes.bnd_flux_up])
        fb = _is_f32(optical_props.tau)   # float32 arrays take ecckd_rte_sw_byband_f32
        Pb = lambda a, shape=None, what="array": _ptr(a, shape, what, fb)
        opt = lambda a, shape, what: Pb(a, shape, what) if a is not None else None
        rc = (lib().ecckd_rte_sw_byband_f32 if fb else lib().ecckd_rte_sw_byband)(
            int(dev), ncol, nlay, ng, int(bool(top_at_1)), Pb(optical_props.tau),
            Pb(optical_props.ssa, (ng, nlay, ncol), "ssa"), Pb(optical_props.g, (ng, nlay, ncol), "g"),
            Pb(mu0, (ncol,), "mu0"), Pb(toa_flux, (ng, ncol), "toa_flux"), nband,
            C.c_void_p(b2g.ctypes.data), Pb(sfc_alb_dir, (ncol, nband), "sfc_alb_dir"),
            Pb(sfc_alb_dif, (ncol, nband), "sfc_alb_dif"),
            Pb(fluxes.bnd_flux_up, (nband, nlay + 1, ncol), "bnd_flux_up"),
            Pb(fluxes.bnd_flux_dn, (nband, nlay + 1, ncol), "bnd_flux_dn"),
            opt(fluxes.bnd_flux_dn_dir, (nband, nlay + 1, ncol), "bnd_flux_dn_dir"),
            opt(fluxes.flux_up, (nlay + 1, ncol), "flux_up"), opt(fluxes.flux_dn, (nlay + 1, ncol), "flux_dn"),
            opt(fluxes.flux_dn_dir, (nlay + 1, ncol), "flux_dn_dir"), space, _stream(space))
        return last_error() if rc else ""
    f32 = _is_f32(optical_props.tau)   # float32 arrays take ecckd_rte_sw_f32
    P = lambda a, shape=None, what="array": _ptr(a, shape, what, f32)
    rc = (lib().ecckd_rte_sw_f32 if f32 else lib().ecckd_rte_sw)(
        int(dev), ncol, nlay, ng, int(bool(top_at_1)), P(optical_props.tau),
        P(optical_props.ssa, (ng, nlay, ncol), "ssa"), P(optical_props.g, (ng, nlay, ncol), "g"),
        P(mu0, (ncol,), "mu0"), P(toa_flux, (ng, ncol), "toa_flux"), nband,
        C.c_void_p(b2g.ctypes.data), P(sfc_alb_dir, (ncol, nband), "sfc_alb_dir"),
        P(sfc_alb_dif, (ncol, nband), "sfc_alb_dif"), P(fluxes.flux_up, (nlay + 1, ncol), "flux_up"),
        P(fluxes.flux_dn, (nlay + 1, ncol), "flux_dn"),
        P(fluxes.flux_dn_dir, (nlay + 1, ncol), "flux_dn_dir"), space, _stream(space))
    return last_error() if rc else ""
