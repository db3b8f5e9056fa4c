// sw_two_stream.hpp -- sw_two_stream of RTE-RRTMGP for one cell (Zdunkowski PIFM / Meador-Weaver two-stream
// coefficients, diffuse and direct reflectance / transmittance, direct-beam transmittance), shared by the shortwave
// solvers (kernels_rte_sw.hip, kernels_rte_sw_sys.hip) so that both produce the same bits per cell.
// [RTE-ext: restated from the public v1.5-era mo_rte_solver_kernels.F90; the library is not in the reference tree --
// SURVEY.md Appendix B.2; call site example/rfmip-rad-irf/ecckd_rfmip_sw.F90:148-154.]
#pragma once
#include <hip/hip_runtime.h>

namespace ecckd {
namespace {

template <typename real> struct TwoStreamT { real Rdif, Tdif, Rdir, Tdir, Tnoscat; };
typedef TwoStreamT<double> TwoStream;

// 1/x: the IEEE division sequence in the reference-order arithmetic mode; in the fast mode the hardware reciprocal
// and Newton steps (~1 ulp: v_rcp_f64 + two steps, 6 instructions against ~15; v_rcp_f32 + one step).
template <bool FAST>
__device__ __forceinline__ double rcp(double x) {
  if (!FAST) return 1. / x;
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.), r, r);
  r = fma(fma(-x, r, 1.), r, r);
  return r;
}
template <bool FAST>
__device__ __forceinline__ float rcp(float x) {
  if (!FAST) return 1.f / x;
  float r = __builtin_amdgcn_rcpf(x);
  r = fmaf(fmaf(-x, r, 1.f), r, r);
  return r;
}
template <typename real> __device__ __forceinline__ real sw_eps();
template <> __device__ __forceinline__ double sw_eps<double>() { return 2.220446049250313e-16; }   // epsilon(1._wp)
template <> __device__ __forceinline__ float sw_eps<float>() { return 1.1920928955078125e-07f; }

// G0: the asymmetry parameter of the whole wave is zero -- what ecCKD's gas optics writes (g = 0,
// src/gas_optics_ecckd.f90:460); the callers vote on the values they have loaded anyway.  With gq a literal 0 the
// compiler folds (1 - g), 3*mu0*g and the duplicated alpha / k*gamma terms: every folded operation is exact (x*1,
// x+0), so the same bits as the general form for finite mu0 (mu0 = inf or NaN: 3*mu0*0 is NaN in the general form,
// 0 here -- such a column is NaN through exp(-tau/mu0) / toa*mu0 either way).
template <typename real, bool FAST, bool CLAMP, bool G0>
__device__ __forceinline__ TwoStreamT<real> two_stream(real tau, real w0, real gq_in, real mu0, real mu0_inv, real k_floor) {
#include "sw_two_stream_body.inc"
}
// (The same body under `#pragma clang fp contract(fast)` -- multiply-add pairs fused, ~20 % fewer fp64 instructions -- was
// measured in round 3: no gain on the layer-systolic solver, whose sweeps set the pace, and the direct-beam terms lose
// two digits to the fused cancellations: fluxes 1.2e-9 W m-2 from the oracle instead of 1e-11.  Not kept.)


}  // namespace
}  // namespace ecckd
