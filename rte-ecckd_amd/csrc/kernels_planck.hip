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
#include "wave_pair.hpp"

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

// ---------------------------------------------------------------------------------------------------------
// The Planck sources as a kernel of their own in the fast arithmetic mode (ecckd_planck_sources; gas_optics calls whose
// sources cannot ride in the fused kernel): the table in LDS, the same column <-> lane mapping and paired 16-byte buffer
// stores (wave_pair.hpp) and the same interpolation code -- so the same bits -- as the fused kernel; 40 registers.
// 7.2-7.4 ms for the 46 GB of 1e6 columns x 60 x 32 (6.3 TB/s).
// Block = 512 columns x a range of layers; per (column, layer) two interpolation points (layer, level j+1; the top level
// as well in the first layer) serve 32 g-points each: lay_source(:,j,:), lev_source_inc(:,j,:) == lev_source_dec(:,j+1,:).
// (Round 2 tried it on a second stream BESIDE the tau-only gas-optics kernel, in place of the fused kernel: the two
// overlap in time but the pair takes 13.1-13.5 ms against the fused kernel's 12.4 -- the GPU sits at its power limit
// in all three cases)
constexpr int kPpBlock = 512;

template <typename real>
__global__ void __launch_bounds__(kPpBlock) planck_pair_kernel(const PlanckArgs a, const UDiv ud_dt_h) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  real *lds = reinterpret_cast<real *>(lds_raw);
  typedef real real2_t __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) const volatile char lds_cvc;
  typedef __attribute__((address_space(3))) const volatile real2_t lds_cv2;
  typedef __attribute__((address_space(3))) const volatile real lds_cv1;
  constexpr int ES = (int)sizeof(real);
  auto P = [](const double *p) { return reinterpret_cast<const real *>(p); };
  auto Q = [](double *p) { return reinterpret_cast<real *>(p); };
  const int ng = a.ng, ntp = a.ntp, ncol = a.ncol, nlay = a.nlay;
  const int SP = row_stride(ng + (ng & 1));
  const int tid = threadIdx.x, lane = tid & 63;
  for (int q = tid; q < ntp * SP; q += kPpBlock) {
    const int r = q / SP, g = q - r * SP;
    lds[q] = g < ng ? P(a.planck)[(long)r * ng + g] : real(0);
  }
  __syncthreads();
  lds_cvc *lb = (lds_cvc *)lds;
  auto ld2b = [&](int bytes, int elem) -> real2_t { return *(lds_cv2 *)(lb + bytes + elem * ES); };
  auto ld1b = [&](int bytes, int elem) -> real { return *(lds_cv1 *)(lb + bytes + elem * ES); };

  const UDivT<real> ud_dt = make_udiv_t<real>(ud_dt_h);
  const real t0 = (real)a.t0;
  const real pi = (real)3.14159265359f, rpi = real(1) / pi;   // src/gas_optics_ecckd.f90:53
  const long c = (long)blockIdx.x * kPpBlock + (tid & ~63) + wave_column(lane);
  const bool valid = c < ncol, upper = lane >= 32;
  const long cc = valid ? c : (long)ncol - 1;
  const bool masked = !__all(valid);
  const unsigned plane = (unsigned)ncol * (unsigned)nlay;
  const long plane2 = 2L * plane;
  const unsigned coff = (unsigned)ES * (unsigned)cc;
  const unsigned voff = (unsigned)ES * ((unsigned)(c - (upper ? 1 : 0)) + (upper ? plane : 0u));
  const int l0 = (int)((long)nlay * blockIdx.y / gridDim.y), l1 = (int)((long)nlay * (blockIdx.y + 1) / gridDim.y);
  const int npairs = ng / 2;
  typedef __attribute__((address_space(1))) const real greal;

  auto interp = [&](const PlPoint<real> &q, int bytes, int g, real (&v)[2]) {
    const real2_t b0 = ld2b(bytes, g), b1 = ld2b(bytes + SP * ES, g);
    v[0] = div_pi(q.w0 * b0[0] + q.w1 * b1[0], pi, rpi);   // :275-288, the reference's order
    v[1] = div_pi(q.w0 * b0[1] + q.w1 * b1[1], pi, rpi);
  };
  auto single = [&](const PlPoint<real> &q, int bytes, int g) -> real {
    return div_pi(q.w0 * ld1b(bytes, g) + q.w1 * ld1b(bytes + SP * ES, g), pi, rpi);
  };

  // temperatures of a layer are requested one layer ahead of their use
  real n_lay = ((greal *)(P(a.tlay) + (long)ncol * l0))[cc];
  real n_lev = a.tlev ? ((greal *)(P(a.tlev) + (long)ncol * (l0 + 1)))[cc] : real(0);
  for (int j = l0; j < l1; ++j) {
    const real Tl = n_lay, T1 = n_lev;
    if (j + 1 < l1) {
      n_lay = ((greal *)(P(a.tlay) + (long)ncol * (j + 1)))[cc];
      if (a.tlev) n_lev = ((greal *)(P(a.tlev) + (long)ncol * (j + 2)))[cc];
    }
    const PlPoint<real> qlay = planck_point<real>(Tl, t0, ud_dt, ntp);
    int alay = qlay.row * SP * ES;
    real *w_lay = Q(a.lay_source) + (long)ncol * j;
    if (!a.tlev) {   // sources of the layers only (gas_optics without tlev still fills lay_source: :407)
      asm volatile("" : "+v"(alay));
      for (int p = 0; p < npairs; ++p, alay += 2 * ES) {
        real vl[2];
        interp(qlay, alay, 0, vl);
        store_pair<real>(w_lay, plane2, voff, coff, vl[0], vl[1], masked, valid);
      }
      if (ng & 1) {
        if (valid) w_lay[c] = single(qlay, alay, 0);
      }
      continue;
    }
    const PlPoint<real> q1 = planck_point<real>(T1, t0, ud_dt, ntp);
    int a1 = q1.row * SP * ES;
    real *w_inc = Q(a.lev_source_inc) + (long)ncol * j;
    real *w_dec = Q(a.lev_source_dec) + (long)ncol * (j + 1);
    const bool has_next = j + 1 < nlay;
    asm volatile("" : "+v"(alay), "+v"(a1));
#pragma unroll 4
    for (int p = 0; p < npairs; ++p, alay += 2 * ES, a1 += 2 * ES) {
      real vl[2], v1[2];
      interp(qlay, alay, 0, vl);
      interp(q1, a1, 0, v1);
      store_pair<real>(w_lay, plane2, voff, coff, vl[0], vl[1], masked, valid);
      store_pair<real>(w_inc, plane2, voff, coff, v1[0], v1[1], masked, valid);            // :423-424
      if (has_next) store_pair<real>(w_dec, plane2, voff, coff, v1[0], v1[1], masked, valid);
    }
    if (ng & 1) {   // odd g-point count: the last one alone
      const real vl = single(qlay, alay, 0), v1 = single(q1, a1, 0);
      if (valid) {
        w_lay[c] = vl;
        w_inc[c] = v1;
        if (has_next) w_dec[c] = v1;
      }
    }
    if (j == 0) {   // the top level: lev_source_dec(:,1,:)
      const PlPoint<real> q0 = planck_point<real>(((greal *)P(a.tlev))[cc], t0, ud_dt, ntp);
      int a0 = q0.row * SP * ES;
      real *w_dec0 = Q(a.lev_source_dec);
      for (int p = 0; p < npairs; ++p, a0 += 2 * ES) {
        real v0[2];
        interp(q0, a0, 0, v0);
        store_pair<real>(w_dec0, plane2, voff, coff, v0[0], v0[1], masked, valid);
      }
      if ((ng & 1) && valid) w_dec0[c] = single(q0, a0, 0);
    }
  }
  if (blockIdx.y == 0 && valid) {   // :408-413
    const PlPoint<real> qs = planck_point<real>(P(a.tsfc)[c], t0, ud_dt, ntp);
    const int as = qs.row * SP * ES;
    for (int g = 0; g < ng; ++g) Q(a.sfc_source)[c + (long)ncol * g] = single(qs, as, g);
  }
}

}  // namespace

size_t planck_pair_lds_bytes(int ng, int ntp, int f32) {
  return (f32 ? sizeof(float) : sizeof(double)) * (size_t)ntp * row_stride(ng + (ng & 1));
}

// The fast Planck kernel; f32 = 1: every data pointer addresses float arrays.
hipError_t launch_planck_pair(const PlanckArgs &a, int f32, hipStream_t s) {
  if (a.ncol <= 0) return hipSuccess;
  const size_t lds = planck_pair_lds_bytes(a.ng, a.ntp, f32);
  if (lds > (size_t)kLdsBudget) return hipErrorInvalidValue;
  if (((size_t)a.ncol * (size_t)a.nlay + (size_t)a.ncol) * (f32 ? sizeof(float) : sizeof(double)) >= (size_t)0xFFFFFFF0u)
    return hipErrorInvalidValue;   // store_pair(): 32-bit byte offsets inside a plane pair
  const int nb = (a.ncol + kPpBlock - 1) / kPpBlock;
  int chunks = 1;
  while (chunks < 6 && (long)nb * chunks < 4096 && (chunks + 1) * 4 <= a.nlay) ++chunks;
  const UDiv ud = make_udiv(a.dt, f32);
  hipError_t e;
  auto go = [&](auto k) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(nb, chunks), dim3(kPpBlock), lds, s, a, ud);
    return hipGetLastError();
  };
  return f32 ? go(planck_pair_kernel<float>) : go(planck_pair_kernel<double>);
}

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
