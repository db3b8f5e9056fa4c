// wave_pair.hpp -- device helpers shared by the gas-optics kernels (kernels_gas_fused.hip, kernels_planck.hip):
// the column <-> lane mapping and the paired 16-byte stores, the Planck interpolation point, division by
// wave-uniform constants.  One definition, so that both kernels produce the same bits.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace ecckd {
namespace {

// Row strides are == 2 (mod 4) doubles: every row starts 16-byte aligned (ds_read_b128 of two
// consecutive g-points) and consecutive rows are shifted by 4 banks, so the 16 lanes of a b128
// lane group that sit in different rows do not collide.
__host__ __device__ inline int row_stride(int n) { return n + ((6 - (n & 3)) & 3); }

// x / d for a wave-uniform divisor d with r = 1/d precomputed (correctly rounded): Markstein's
// correction step returns the correctly rounded quotient, i.e. exactly what `x / d` returns, in
// 3 instructions instead of the ~30 of the IEEE division sequence.  (Precondition checked on the
// host: d finite, non-zero, significand not all ones; otherwise exact is 0 and `/` is used.)
template <typename real> struct UDivT { real d, r; int exact; };
template <typename real> __device__ __forceinline__ UDivT<real> make_udiv_t(const UDiv &u) {
  // d and r come from the host already rounded to the working precision (make_udiv): they stay in SGPRs.  (Computing
  // 1/d here cost a division sequence per thread and two VGPRs per divisor for the whole kernel.)
  UDivT<real> o;
  o.d = (real)u.d;
  o.r = (real)u.r;
  o.exact = u.exact;
  return o;
}
template <typename real> __device__ __forceinline__ real udiv(real x, const UDivT<real> &u) {
  if (!u.exact) return x / u.d;
  const real q = x * u.r;
  const real rem = fma(-q, u.d, x);
  return fma(rem, u.r, q);
}

// Planck interpolation point (:275-285): rows `row`, `row + 1` of the table (0-based) with weights
// w0, w1.  Below the table the reference uses (T/t0)*B(:,1); that is row 0 with weights (T/t0, 0):
// w0*b0 + 0*b1 == w0*b0 exactly, so no branch is needed.  `off` is filled in by the caller (LDS
// offset of the row inside the staged window).
template <typename real> struct PlPoint { int row, off; real w0, w1; };
template <typename real> __device__ __forceinline__ PlPoint<real> planck_point(real Tk, real t0, const UDivT<real> &dt, int ntp) {
  PlPoint<real> p;
  real temperature_index = udiv(Tk - t0, dt);
  if (temperature_index >= 0) {
    temperature_index = real(1) + temperature_index;
    const int it0 = temperature_index >= (real)(ntp - 1) ? ntp - 1 : (int)temperature_index;
    p.w1 = temperature_index - it0;
    p.w0 = real(1) - p.w1;
    p.row = it0 - 1;
  } else {
    p.w0 = Tk / t0;
    p.w1 = real(0);
    p.row = 0;
  }
  p.off = 0;
  return p;
}

template <typename real> __device__ __forceinline__ real div_pi(real x, real pi, real rpi) {
  const real q = x * rpi;              // correctly rounded x/pi (Markstein), see kernels_planck.hip
  const real r = fma(-q, pi, x);
  return fma(r, rpi, q);
}

// Column <-> lane mapping inside a wave: lane l < 32 holds column 2l of the wave's 64, lane l + 32 column 2l + 1.  The
// two lanes of a column pair are 32 apart, so that ONE v_permlane32_swap_b32 per dword (gfx950) hands each lane the pair
// it stores: after pair_exchange(v0, v1) -- v0 / v1 the lane's values of planes g / g+1 -- lane l < 32 holds
// (v0 of column 2l, v0 of column 2l+1) and lane l + 32 holds (v1 of column 2l, v1 of column 2l+1).
// (Until round 2 the pairs were adjacent lanes: two DPP moves and six v_cndmask per store, 16 VALU instructions per
// cell in a kernel that is bound by VALU issue.)
__device__ __forceinline__ int wave_column(int lane) { return 2 * (lane & 31) + (lane >> 5); }
__device__ __forceinline__ void pair_exchange(double &v0, double &v1) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v0), (unsigned)__double2loint(v1), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v0), (unsigned)__double2hiint(v1), false, false);
  v0 = __hiloint2double((int)hi[0], (int)lo[0]);
  v1 = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void pair_exchange(float &v0, float &v1) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
  v0 = __uint_as_float(r[0]);
  v1 = __uint_as_float(r[1]);
}

// Stores the values of two consecutive g-points (planes g, g+1 of a column-fastest array) as ONE
// 16-byte store per lane instead of two 8-byte ones (see pair_exchange): the lower half-wave writes the column
// pairs of plane g, the upper half-wave those of plane g+1.  Per CU the store path moves ~7 B/clk with dwordx2 and
// about twice that with dwordx4 (the kernel was store-issue bound).  Called by all lanes of the wave.
//   masked (wave-uniform) == false: every lane stores, no exec mask.
//   masked == true: the wave holds lanes that are not handled in this pass (columns beyond ncol,
//   or columns that belong to another slab position): the lanes with `active` store their own
//   column of both planes with two 8-byte stores, the others store nothing.
//   base: wave-uniform RUNNING pointer to (column 0, plane g) of the array, advanced by two planes
//   (plane2 elements) after the store -- the g-pairs of an array are stored in ascending order, so
//   one pointer per array walks the whole tile and nothing per (array, g-pair) is loop invariant;
//   voff: per-lane BYTE offset sizeof(real) * ((c - upper) + (upper ? plane : 0)), 32 bits, shared by
//   the four output arrays; coff: sizeof(real) * c.
// The stores are buffer stores: descriptor (SGPRs, rebuilt from `base` with three scalar instructions) + the per-lane
// 32-bit offset, no vector address arithmetic.  (global_store with the 64-bit address built per store cost one
// v_lshl_add_u64 each: instruction selection does not fold a zero-extension hoisted out of the loop.)
//   nplanes (wave-uniform): 2 = both planes exist; 1 = plane g is the last g-point of the array (g-point counts that are
//   not a multiple of the chunk): only the half-wave that holds plane g stores.  `upper`: this lane is in that other half.
template <typename real>
__device__ __forceinline__ void store_pair(real *&base, long plane2, unsigned voff, unsigned coff, real v0, real v1,
                                           bool masked, bool active, int nplanes = 2, bool upper = false) {
  typedef unsigned uint2_t __attribute__((ext_vector_type(2)));
  typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
  typedef real real2_t __attribute__((ext_vector_type(2)));
  // Pin the pointer in SGPRs right here: left alone, the optimiser precomputes one pointer per
  // (array, g-pair) outside the loops (32 SGPRs, spilled to VGPR lanes and read back per store).
  asm volatile("" : "+s"(base));
  // raw buffer over the whole address range above `base` (offsets are checked against 2^32 - 1 on the host side:
  // launch_gas_fused); word 3 = DATA_FORMAT_32, the value untyped buffer accesses use on gfx9
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, -1, 0x00020000);
#ifndef ECCKD_PLAIN_STORES    // nontemporal: the outputs are written once and read by the next kernel
  constexpr int aux = 2;
#else
  constexpr int aux = 0;
#endif
  if (!masked) {
    pair_exchange(v0, v1);
    real2_t out;
    out[0] = v0;
    out[1] = v1;
#ifndef ECCKD_DEBUG_NOSTORE   // (compile-time switch for timing experiments: arithmetic without stores)
    if (nplanes >= 2 || !upper) {
      if constexpr (sizeof(real) == 8) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, out), rsrc, (int)voff, 0, aux);
      else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uint2_t, out), rsrc, (int)voff, 0, aux);
    }
#else
    asm volatile("" :: "v"(out));
#endif
  } else if (active) {
    const int plane_bytes = (int)((plane2 / 2) * (long)sizeof(real));
    if constexpr (sizeof(real) == 8) {
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uint2_t, v0), rsrc, (int)coff, 0, 0);
      if (nplanes >= 2) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uint2_t, v1), rsrc, (int)coff, plane_bytes, 0);
    } else {
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rsrc, (int)coff, 0, 0);
      if (nplanes >= 2) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), rsrc, (int)coff, plane_bytes, 0);
    }
  }
  base += plane2;
}

}  // namespace
}  // namespace ecckd
