// kernels_planck.hip -- Planck sources: lay_source, lev_source_inc, lev_source_dec, sfc_source.
//
// Replaces calculate_planck_function and its three call sites in gas_optics_int
// (src/gas_optics_ecckd.f90:245-289, :407-424).  The reference interpolates tlev into a
// (ncol,nlay+1,ng) buffer and copies it twice (:419-424); here each level value is computed
// once and stored to lev_source_dec(:,l,:) and lev_source_inc(:,l-1,:) directly.
//
// Mapping (gfx950): lane -> column, block = 512 columns x a chunk of levels; the whole Planck
// table (ntp rows, padded to an odd row length) sits in LDS.  Pure store-bandwidth kernel:
// 24 B/cell written, 4 ds_read_b64 + ~12 fp64 ops per cell.
//
// `x/pi` of :288 is evaluated as q=x*(1/pi), r=fma(-q,pi,x), q+=r*(1/pi) which returns the
// correctly rounded quotient (Markstein) -- bit-identical to the division, at 3 flops.
#include "kernels.hpp"

namespace ecckd {
namespace {

constexpr int kPlBlock = 512;

__device__ __forceinline__ double div_pi(double x, double pi, double rpi) {
  const double q = x * rpi;
  const double r = fma(-q, pi, x);
  return fma(r, rpi, q);
}

struct PlanckPoint {
  int row;          // 0-based lower row
  double w0, w1;    // interpolation weights, or (ratio, unused) when below the table
  bool below;
};

// :275-285 for one temperature
__device__ __forceinline__ PlanckPoint planck_point(double T, double t0, double dt, int ntp) {
  PlanckPoint p;
  double temperature_index = (T - t0) / dt;
  if (temperature_index >= 0) {
    temperature_index = 1. + temperature_index;
    // min(int(idx), ntp-1); the comparison is done in fp so that a huge idx cannot overflow int
    const int it0 = temperature_index >= (double)(ntp - 1) ? ntp - 1 : (int)temperature_index;
    p.w1 = temperature_index - it0;
    p.w0 = 1. - p.w1;
    p.row = it0 - 1;
    p.below = false;
  } else {
    p.w0 = T / t0;
    p.w1 = 0.;
    p.row = 0;
    p.below = true;
  }
  return p;
}

template <int GC>
__device__ __forceinline__ void planck_rows(const double *lds, int SR, const PlanckPoint &p, int gb,
                                            int ng, double pi, double rpi, double (&out)[GC]) {
  const int o = p.row * SR + gb;
  if (!p.below) {
#pragma unroll
    for (int g = 0; g < GC; ++g)
      if (gb + g < ng) out[g] = div_pi(p.w0 * lds[o + g] + p.w1 * lds[o + SR + g], pi, rpi);
  } else {
#pragma unroll
    for (int g = 0; g < GC; ++g)
      if (gb + g < ng) out[g] = div_pi(p.w0 * lds[gb + g], pi, rpi);
  }
}

template <int GC>
__global__ void __launch_bounds__(kPlBlock) planck_kernel(const PlanckArgs a) {
  extern __shared__ double lds[];
  const int ng = a.ng, ntp = a.ntp, ncol = a.ncol, nlay = a.nlay;
  const int SR = ng | 1;
  for (int q = threadIdx.x; q < ntp * ng; q += kPlBlock) {
    const int r = q / ng, g = q - r * ng;
    lds[r * SR + g] = a.planck[q];
  }
  __syncthreads();

  const double pi = (double)3.14159265359f;   // src/gas_optics_ecckd.f90:53 (f32 literal)
  const double rpi = 1. / pi;
  const long c = (long)blockIdx.x * kPlBlock + threadIdx.x;
  if (c >= ncol) return;
  const int nlev = nlay + 1;
  const int l0 = (int)((long)nlev * blockIdx.y / gridDim.y);
  const int l1 = (int)((long)nlev * (blockIdx.y + 1) / gridDim.y);
  double v[GC];

  for (int l = l0; l < l1; ++l) {
    if (a.tlev) {   // :419-424
      const PlanckPoint p = planck_point(a.tlev[c + (long)ncol * l], a.t0, a.dt, ntp);
      for (int gb = 0; gb < ng; gb += GC) {
        planck_rows<GC>(lds, SR, p, gb, ng, pi, rpi, v);
#pragma unroll
        for (int g = 0; g < GC; ++g) {
          if (gb + g < ng) {
            if (l < nlay) a.lev_source_dec[c + (long)ncol * (l + (long)nlay * (gb + g))] = v[g];
            if (l > 0) a.lev_source_inc[c + (long)ncol * ((l - 1) + (long)nlay * (gb + g))] = v[g];
          }
        }
      }
    }
    if (l < nlay) {   // :407
      const PlanckPoint p = planck_point(a.tlay[c + (long)ncol * l], a.t0, a.dt, ntp);
      for (int gb = 0; gb < ng; gb += GC) {
        planck_rows<GC>(lds, SR, p, gb, ng, pi, rpi, v);
#pragma unroll
        for (int g = 0; g < GC; ++g)
          if (gb + g < ng) a.lay_source[c + (long)ncol * (l + (long)nlay * (gb + g))] = v[g];
      }
    }
  }
  if (blockIdx.y == 0) {   // :408-413
    const PlanckPoint p = planck_point(a.tsfc[c], a.t0, a.dt, ntp);
    for (int gb = 0; gb < ng; gb += GC) {
      planck_rows<GC>(lds, SR, p, gb, ng, pi, rpi, v);
#pragma unroll
      for (int g = 0; g < GC; ++g)
        if (gb + g < ng) a.sfc_source[c + (long)ncol * (gb + g)] = v[g];
    }
  }
}

}  // namespace

hipError_t launch_planck(PlanckArgs &a, hipStream_t s) {
  if (a.ncol <= 0) return hipSuccess;
  const size_t lds = sizeof(double) * (size_t)a.ntp * (a.ng | 1);
  if (lds > (size_t)kLdsBudget) return hipErrorInvalidValue;
  const int nb = (a.ncol + kPlBlock - 1) / kPlBlock;
  // enough blocks to fill 256 CUs several times over; levels are split when columns are few
  int chunks = 1;
  while (chunks < 8 && (long)nb * chunks < 2048 && chunks * 4 <= a.nlay + 1) chunks *= 2;
  a.lev_chunks = chunks;
  auto k = planck_kernel<16>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(nb, chunks), dim3(kPlBlock), lds, s, a);
  return hipGetLastError();
}

}  // namespace ecckd
