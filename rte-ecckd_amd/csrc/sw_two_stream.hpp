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
// (relative error <= 2^-23 in fp64) and ONE third-order step, r = r0 + r0 (e + e^2), e = 1 - x r0: error e^3 ~ 2^-69, the
// correctly rounded reciprocal up to the last bit -- 4 instructions against ~15 (until late in round 3: two Newton steps, 5).
template <bool FAST>
__device__ __forceinline__ double rcp(double x) {
  if (!FAST) return 1. / x;
  const double r0 = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, r0, 1.);
  return fma(fma(e, e, e), r0, r0);
}
template <bool FAST>
__device__ __forceinline__ float rcp(float x) {
  if (!FAST) return 1.f / x;
  float r = __builtin_amdgcn_rcpf(x);
  r = fmaf(fmaf(-x, r, 1.f), r, r);
  return r;
}
// sqrt(x) for the eigenvalue k of the two-stream equations.  Fast arithmetic mode, fp64: v_rsq_f64 and Goldschmidt steps with a
// final residual correction (<= 1 ulp; tools/check_sw_sqrt.hip) -- 10 instructions against the 22 of the device library's
// sqrt, whose scaling of subnormal / huge arguments and special-case selects this argument never needs: x = max((g1-g2)(g1+g2),
// k_floor) lies in [k_floor, 4], and the host raises a subnormal k_floor to the smallest normal number in this mode.
// NaN stays NaN.
#ifndef ECCKD_SW_LEAN_SQRT
#define ECCKD_SW_LEAN_SQRT 1
#endif
template <bool FAST>
__device__ __forceinline__ double sw_sqrt(double x) {
  if (!FAST || !ECCKD_SW_LEAN_SQRT) return sqrt(x);
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  return fma(fma(-g, g, x), h, g);
}
template <bool FAST>
__device__ __forceinline__ float sw_sqrt(float x) { return sqrtf(x); }
// exp(x) for the two transmittances of a cell, exp(-k tau) and exp(-tau / mu0).  Fast arithmetic mode, fp64: the usual
// reduction x = n ln2 + r, |r| <= ln2 / 2, and a degree-11 polynomial (1 + r + r^2 g(r), g interpolated at Chebyshev
// nodes: <= 0.84 ulp on 4e4 random arguments against 200-bit arithmetic).  Instead of the device library's two compares
// and three selects for results beyond the double range, ONE compare-and-select keeps the argument above -1100
// (v_ldexp_f64 turns n < -1074 into +0 by itself; without the bound the reduction of |x| > 1e15 leaves a remainder whose
// powers overflow): 20 instructions against 22, any optical depth up to inf gives what exp() gives, a NaN stays a NaN
// (the compare is false for it).  Arguments above +709 (a negative optical depth) give inf through v_ldexp_f64 up to
// 1e15 and are not meaningful beyond.
#ifndef ECCKD_SW_LEAN_EXP
#define ECCKD_SW_LEAN_EXP 1
#endif
template <bool FAST>
__device__ __forceinline__ double sw_exp(double x) {
  if (!FAST || !ECCKD_SW_LEAN_EXP) return exp(x);
  x = x < -1100. ? -1100. : x;
  const double n = __builtin_rint(x * 0x1.71547652b82fep+0);   // log2(e)
  double r = fma(n, -0x1.62e42fee00000p-1, x);                // ln2, upper 32 bits: n * hi is exact
  r = fma(n, -0x1.a39ef35793c76p-33, r);                        // ln2 - hi
  double p = 0x1.af38d53857513p-26;
  p = fma(p, r, 0x1.2891a8c1d838dp-22);
  p = fma(p, r, 0x1.71de0d9c145d0p-19);
  p = fma(p, r, 0x1.a019b8ef67c6cp-16);
  p = fma(p, r, 0x1.a01a01a7c8d47p-13);
  p = fma(p, r, 0x1.6c16c17893833p-10);
  p = fma(p, r, 0x1.11111111109adp-7);
  p = fma(p, r, 0x1.5555555553d4fp-5);
  p = fma(p, r, 0x1.5555555555556p-3);
  p = fma(p, r, 0x1.0000000000001p-1);
  p = fma(p, r, 1.);
  p = fma(p, r, 1.);
  return __builtin_amdgcn_ldexp(p, (int)n);
}
template <bool FAST>
__device__ __forceinline__ float sw_exp(float x) { return expf(x); }
template <typename real> __device__ __forceinline__ real sw_eps();
template <> __device__ __forceinline__ double sw_eps<double>() { return 2.220446049250313e-16; }   // epsilon(1._wp)
template <> __device__ __forceinline__ float sw_eps<float>() { return 1.1920928955078125e-07f; }

// G0: the asymmetry parameter of the whole wave is zero -- what ecCKD's gas optics writes (g = 0,
// src/gas_optics_ecckd.f90:460); the callers vote on the values they have loaded anyway.  Then gamma3 = gamma4 = 1/2,
// alpha1 = alpha2 and k gamma3 = k gamma4: the body writes those out (the compiler may not fold x * 0), every dropped
// operation is exact (x * 1, x + 0, (2 - 0) / 4), so the same bits as the general form for finite mu0 (mu0 = inf or NaN:
// 3*mu0*0 is NaN in the general form, 0 here -- such a column is NaN through exp(-tau/mu0) / toa*mu0 either way).
template <typename real, bool FAST, bool CLAMP, bool G0>
__device__ __forceinline__ TwoStreamT<real> two_stream(real tau, real w0, real gq_in, real mu0, real mu0_inv, real k_floor) {
#include "sw_two_stream_body.inc"
}
// (The same body under `#pragma clang fp contract(fast)` -- multiply-add pairs fused, ~20 % fewer fp64 instructions -- was
// measured in round 3: no gain on the layer-systolic solver, whose sweeps set the pace, and the direct-beam terms lose
// two digits to the fused cancellations: fluxes 1.2e-9 W m-2 from the oracle instead of 1e-11.  Not kept.)


}  // namespace
}  // namespace ecckd
