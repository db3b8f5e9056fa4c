// lw_layer.hpp -- exp() and `/` of lw_source_noscat / the transmittance of a layer, shared by the longwave solvers
// (kernels_rte_lw.hip, kernels_rte_lw_split.hip) so that they produce the same bits per cell.
// [RTE-ext: the expressions they serve are restated from the public v1.5-era mo_rte_solver_kernels.F90 -- SURVEY.md
// Appendix B.1; call site example/rfmip-rad-irf/ecckd_rfmip_lw.F90:130-135.]
//
// The register-resident solver runs one wave per SIMD and issues one fp64 instruction every ~10 clocks: at 1e6 columns
// its ~75 vector instructions per cell ARE its run time (8.5e6 wave-instructions per CU at 0.40 per clock = 10.2 ms
// against 10.4 measured), so every instruction of the per-cell body counts.
#pragma once
#include <hip/hip_runtime.h>

namespace ecckd {
namespace {

// exp(x), fp64: the reduction x = n ln2 + r, |r| <= ln2 / 2, and a degree-11 polynomial (1 + r + r^2 g(r), g interpolated at
// Chebyshev nodes; <= 0.84 ulp on 4e4 random arguments against 200-bit arithmetic -- the accuracy class of the device
// library's exp).  Instead of that routine's two compares and three selects for results beyond the double range the
// argument is clamped to [-1100, 1100] (v_ldexp_f64 turns n < -1074 into +0 and n > 1023 into inf): 19 instructions
// against 22.  A NaN argument gives exp(-1100) = 0 (v_max_f64 returns its other operand): the callers' NaN reaches the
// fluxes through the optical depth itself (omt / tl, tl * (...)), see lw_source_noscat below each call.
__device__ __forceinline__ double lw_exp(double x) {
#ifdef ECCKD_LW_OLD_MATH   // (A/B builds: the device library's exp and the compiler's `/`)
  return exp(x);
#endif
  x = __builtin_fmin(__builtin_fmax(x, -1100.), 1100.);
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
__device__ __forceinline__ float lw_exp(float x) { return expf(x); }

// x / d, fp64: the instruction sequence the compiler emits for `/` (reciprocal, two Newton steps, quotient, one residual
// correction) WITHOUT its v_div_scale / v_div_fixup frame -- 9 instructions against 11.  That frame rescales operands whose
// quotient or reciprocal leaves the normal range; for the one division of lw_source_noscat, (1 - t) / tl with
// tl in (tau_thresh, 1e290) and 1 - t in [tl / 2, 1], it never acts, and the quotient is the same bits as `/`
// (tools/check_lw_div.hip).  Outside: tl <= tau_thresh selects the series (whatever this returns, NaN and inf included,
// is dropped by the select); the divisor is bounded by 1e290, so an optical depth beyond -- up to inf -- gives ~1e-290
// where `/` gives less: both vanish against the flux.  A NaN divisor becomes 1e290 here (v_min_f64 returns its other
// operand) and reaches the fluxes through the series branch, which the select takes for it.
__device__ __forceinline__ double lw_div(double x, double d) {
#ifdef ECCKD_LW_OLD_MATH
  return x / d;
#endif
  d = __builtin_fmin(d, 1e290);
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.), r, r);
  r = fma(fma(-d, r, 1.), r, r);
  const double q = x * r;
  return fma(fma(-d, q, x), r, q);
}
__device__ __forceinline__ float lw_div(float x, float d) { return x / d; }

}  // namespace
}  // namespace ecckd
