// kernels_rte_lw_split.hip -- longwave no-scattering solver, layer-split form: the NW waves of a block share one
// tile of (CW columns x 64/CW g-points) and each walks a SEGMENT of SEG = nlay/NW layers.
//
// Why: the register-resident solver of kernels_rte_lw.hip keeps trans(l) and source_up(l) of all 60 layers in
// registers (2 x 60 doubles: one wave per SIMD), and one wave alone issues an fp64 instruction only every ~10
// clocks (tools/ubench.hip: 0.40 wave-instructions/clk/CU at 4 waves per CU against 0.74-0.81 at 12-16): that
// kernel is bound by its own issue cadence, not by HBM.  The layer recurrences are AFFINE,
//     I_dn(l+1) = t(l) I_dn(l) + s_dn(l),      I_up(l) = t(l) I_up(l+1) + s_up(l),
// so a segment of layers composes into (T, D, U) with  I_out = T I_in + D  and  U_out = T U_in + U.  Each wave
// does the expensive part (exp, the source functions) for its own 15 layers only -- 3 x 15 doubles in registers,
// three waves per SIMD -- then the NW composites are exchanged through LDS (one block barrier per g-point group),
// every wave folds them into the intensities that enter its segment from above and from below, and finishes
// its 15 levels of both sweeps out of registers.  Same arithmetic per cell as lw_solver_noscat; the intensities
// entering a segment are composed in a different association than a top-to-bottom walk would (relative 1e-16).
//
// PLANCK variant ("fused longwave"): the three source arrays are not read from HBM
// but recomputed from tlay / tlev and the Planck table (src/gas_optics_ecckd.f90:245-289, :407-424) inside
// the solver, so gas optics only has to write tau: 16 instead of 64 B/cell between the two kernels.
#include <cstdlib>

#include "kernels.hpp"
#include "lw_layer.hpp"

namespace ecckd {
namespace {

#ifndef ECCKD_SPLIT_SEG
#define ECCKD_SPLIT_SEG 10   // layers per wave; 60 / SEG waves per block
#endif
#ifndef ECCKD_SPLIT_PF
#define ECCKD_SPLIT_PF 2
#endif
#ifndef ECCKD_SPLIT_WAVES_PER_SIMD
#define ECCKD_SPLIT_WAVES_PER_SIMD 3
#endif
#ifndef ECCKD_SPLIT_PF_PLANCK
#define ECCKD_SPLIT_PF_PLANCK 2
#endif
// layers in flight per lane
constexpr int split_pf(bool planck) { return planck ? ECCKD_SPLIT_PF_PLANCK : ECCKD_SPLIT_PF; }

template <int CW>
__device__ __forceinline__ double gsum(double v) {
#pragma unroll
  for (int o = CW; o < 64; o <<= 1) v = v + __shfl_xor(v, o);
  return v;
}

// acc += v by the owner lane (the others add +0.0): one fire-and-forget ds_add_f64 (see kernels_rte_lw.hip)
__device__ __forceinline__ void acc_add(double *p, double v, bool owner) {
  __hip_atomic_fetch_add(p, owner ? v : 0., __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Planck source of one temperature for one g-point (calculate_planck_function, :275-288), table rows from
// global memory (59 KB: L1/L2 resident).  tp0 = temperature_planck(1), rdt = 1/(temperature_planck(2)-(1)).
struct PlanckTab { const double *tab; double t0, dt, rdt; int ntp, ng; };
// `tab` / `stride`: the table the rows are read from -- the model's (ng,ntp) table in global memory (stride ng), or the
// block's copy in LDS whose rows are padded to an odd number of doubles: the 32 columns of a wave sit in different
// (neighbouring) rows, and with a stride of 32 doubles they would all hit the same two banks.
__device__ __forceinline__ double planck_at(const PlanckTab &P, const double *tab, int stride, double T, int g, double pi, double rpi) {
  double ti = (T - P.t0) * P.rdt;
  {   // exact quotient (Markstein) so that the row and the weights are the reference's
    const double rem = fma(-ti, P.dt, T - P.t0);
    ti = fma(rem, P.rdt, ti);
  }
  double v;
  if (ti >= 0.) {
    ti = 1. + ti;
    const int it0 = ti >= (double)(P.ntp - 1) ? P.ntp - 1 : (int)ti;
    const double w1 = ti - it0, w0 = 1. - w1;
    const double *r = tab + (it0 - 1) * stride + g;
    v = w0 * r[0] + w1 * r[stride];
  } else {
    v = (T / P.t0) * tab[g];
  }
  const double q = v * rpi;   // correctly rounded v / pi
  return fma(fma(-q, pi, v), rpi, q);
}

// NG tile groups of NW waves per block.  The Planck-recomputing form runs two groups per block (8 waves) that share one
// copy of the Planck table in LDS (61-68 KB): read from global memory instead, the two dependent table loads per source
// leave the kernel latency-bound at its two waves per SIMD (measured 21 ms per 1e6 columns against 11 from LDS).
constexpr int split_groups(bool planck) { return planck ? 2 : 1; }
__host__ __device__ constexpr int planck_stride(int ng) { return ng | 1; }   // odd number of doubles per row

template <int SEG, int NW, int CW, bool SHARED, bool SER3, bool PLANCK, int WPS>
__global__ void __launch_bounds__(64 * NW * split_groups(PLANCK), WPS) rte_lw_split_kernel(const RteLwArgs a, const PlanckTab pt,
                                                                                          const double *tlay, const double *tlev,
                                                                                          const double *tsfc) {
  constexpr int GW = 64 / CW;
  constexpr int NL = SEG * NW;
  constexpr int NG = split_groups(PLANCK);
  constexpr int kSplitPF = split_pf(PLANCK);
  constexpr int kGroupDoubles = 2 * (NL + 1) * CW + 2 * NW * 3 * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int grp = (tid >> 6) / NW, w = (tid >> 6) % NW, gtid = tid - grp * 64 * NW;
  double *acc_dn = reinterpret_cast<double *>(lds_raw) + grp * kGroupDoubles;   // [NL+1][CW]
  double *acc_up = acc_dn + (NL + 1) * CW;                                         // [NL+1][CW]
  double *xch = acc_up + (NL + 1) * CW;                                            // [2][NW][3][64]
  [[maybe_unused]] const int pstride = PLANCK ? planck_stride(a.ng) : 0;
  [[maybe_unused]] double *ptab = reinterpret_cast<double *>(lds_raw) + NG * kGroupDoubles;   // PLANCK: [ntp][pstride]
  if (PLANCK) {
    for (int i = tid; i < pt.ntp * pt.ng; i += 64 * NW * NG) {
      const int r = i / pt.ng, g = i - r * pt.ng;
      ptab[r * pstride + g] = pt.tab[i];
    }
  }
  const int cl = lane % CW, gs = lane / CW;
  const bool owner = gs == 0;
  const int ncol = a.ncol, ng = a.ng;
  const double pi = acos(-1.);
  const double pi_f32 = (double)3.14159265359f, rpi_f32 = 1. / pi_f32;   // src/gas_optics_ecckd.f90:53 (PLANCK)
  const double tau_thresh = a.tau_thresh;
  const long lay0 = a.top_at_1 ? 0 : NL - 1, lev0 = a.top_at_1 ? 0 : NL;
  const long lstep = a.top_at_1 ? 1 : -1;
  const double *Bdn = a.top_at_1 ? a.lev_source_inc : a.lev_source_dec;
  const double *Bup = a.top_at_1 ? a.lev_source_dec : a.lev_source_inc;
  const int ngroups = (ng + GW - 1) / GW;
  const int niter = ngroups * a.nmus;
  const long ntiles = ((long)ncol + CW - 1) / CW;
  const int s0 = w * SEG;   // first layer of this wave, in walking order from the top

  for (int i = gtid; i < 2 * (NL + 1) * CW; i += 64 * NW) acc_dn[i] = 0.;
  __syncthreads();

  // every group of the block walks the same number of tiles (block barriers inside): a group whose tile lies beyond the
  // end computes on clamped columns and stores nothing
  for (long tile0 = (long)blockIdx.x * NG; tile0 < ntiles; tile0 += (long)gridDim.x * NG) {
    const long tile = tile0 + grp;
    const long col = tile * CW + cl;
    const bool valid = col < ncol;
    const long cc = valid ? col : (long)ncol - 1;

    double ptau[kSplitPF];
    [[maybe_unused]] double play_[PLANCK ? 1 : kSplitPF], pbdn[PLANCK ? 1 : kSplitPF], pbup[(PLANCK || SHARED) ? 1 : kSplitPF];
    [[maybe_unused]] double ptl[PLANCK ? kSplitPF : 1], ptv[PLANCK ? kSplitPF : 1];   // PLANCK: tlay(l), tlev(far edge of l)
    long qn = 0;
    const long qstep = (long)ncol * lstep;
    [[maybe_unused]] long q2 = 0;      // PLANCK: offset into tlay / tlev rows (no g dimension)
    [[maybe_unused]] double near_first = 0.;   // SHARED / PLANCK: near-edge source (temperature) of the segment's first layer

    auto pair_start = [&](int it) {
      const int g = (it / a.nmus) * GW + gs;
      const int gg = g < ng ? g : ng - 1;
      qn = cc + (long)ncol * NL * gg + (long)ncol * (lay0 + lstep * s0);
      asm volatile("" : "+v"(qn));
      if (PLANCK) {
        q2 = cc + (long)ncol * (lay0 + lstep * s0);
        // level at the near (upper, in walking order) edge of the first layer of the segment
        near_first = tlev[cc + (long)ncol * (lev0 + lstep * s0)];
      } else if (SHARED) {
        near_first = __builtin_nontemporal_load(Bup + qn);
      }
    };
    auto issue = [&](int slot) {
      ptau[slot] = __builtin_nontemporal_load(a.tau + qn);
      if (PLANCK) {
        ptl[slot] = tlay[q2];
        ptv[slot] = tlev[q2 + (a.top_at_1 ? (long)ncol : 0)];   // far edge of the layer: level index l+1 (top_at_1) or l
        q2 += qstep;
      } else {
        play_[slot] = __builtin_nontemporal_load(a.lay_source + qn);
        pbdn[slot] = __builtin_nontemporal_load(Bdn + qn);
        if (!SHARED) pbup[slot] = __builtin_nontemporal_load(Bup + qn);
      }
      qn += qstep;
      asm volatile("" : "+v"(qn));
    };

    pair_start(0);
#pragma unroll
    for (int s = 0; s < kSplitPF; ++s) issue(s);

    for (int it = 0; it < niter; ++it) {
      const int gi = it / a.nmus, k = it - gi * a.nmus;
      const int g = gi * GW + gs;
      const bool gact = g < ng;
      const int gg = gact ? g : ng - 1;
      const double D = a.Ds[k];
      const double wfac = gact ? 2. * pi * a.wts[k] : 0.;

      // ---------------- phase 1: this wave's SEG layers ----------------
      double T[SEG], SDN[SEG], SU[SEG];
      double Tq = 1., Dq = 0.;
      [[maybe_unused]] double carry = PLANCK ? planck_at(pt, ptab, pstride, near_first, gg, pi_f32, rpi_f32) : near_first;
#pragma unroll
      for (int s = 0; s < SEG; ++s) {
        const double tau = ptau[s % kSplitPF];
        double lay, bdn, bup;
        if (PLANCK) {
          lay = planck_at(pt, ptab, pstride, ptl[s % kSplitPF], gg, pi_f32, rpi_f32);
          bdn = planck_at(pt, ptab, pstride, ptv[s % kSplitPF], gg, pi_f32, rpi_f32);
          bup = carry;
        } else {
          lay = play_[s % kSplitPF];
          bdn = pbdn[s % kSplitPF];
          bup = SHARED ? carry : pbup[s % kSplitPF];
        }
        if (s + kSplitPF < SEG) {
          asm volatile("" : "+v"(qn), "+v"(Dq));   // keep the loads of layer s+PF below layer s-1
          issue(s % kSplitPF);
        }
        const double tl = tau * D;
        const double t = lw_exp(-tl);
        const double omt = 1. - t;
        const double fact_big = lw_div(omt, tl) - t;
        const double fact_small = SER3 ? tl * (0.5 + tl * (-1. / 3. + tl * (1. / 8.))) : tl * (0.5 - 1. / 3. * tl);
        const double fact = tl > tau_thresh ? fact_big : fact_small;
        const double sdn = omt * bdn + 2. * fact * (lay - bdn);
        double su = omt * bup + 2. * fact * (lay - bup);
        asm volatile("" : "+v"(su));
        T[s] = t; SDN[s] = sdn; SU[s] = su;
        Dq = t * Dq + sdn;
        Tq = Tq * t;
        if (SHARED || PLANCK) carry = bdn;
      }
      double Uq = 0.;
#pragma unroll
      for (int s = SEG - 1; s >= 0; --s) Uq = T[s] * Uq + SU[s];
      // the next pair's first layers start streaming now
      if (it + 1 < niter) {
        pair_start(it + 1);
#pragma unroll
        for (int s = 0; s < kSplitPF; ++s) issue(s);
      }
      double *x = xch + (it & 1) * (NW * 3 * 64);
      x[(w * 3 + 0) * 64 + lane] = Tq;
      x[(w * 3 + 1) * 64 + lane] = Dq;
      x[(w * 3 + 2) * 64 + lane] = Uq;
      __syncthreads();

      // ---------------- phase 2: boundary intensities of this segment, then both sweeps ----------------
      double I = 0.;   // radn_dn(top): no incident diffuse flux, or inc_flux as an intensity (Appendix B.1)
      if (a.inc_flux) {
        const double f = a.inc_flux[cc + (long)ncol * gg];
        I = a.inc_isotropic ? f / pi : f / (2. * pi * a.wts[k]);
      }
      double Iin = I;
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        if (q == w) Iin = I;
        I = x[(q * 3 + 0) * 64 + lane] * I + x[(q * 3 + 1) * 64 + lane];
      }
      const double eps = a.sfc_emis[a.gpt2band[gg] + (long)a.nband * cc];
      const double sfc_src = PLANCK ? planck_at(pt, ptab, pstride, tsfc[cc], gg, pi_f32, rpi_f32) : a.sfc_source[cc + (long)ncol * gg];
      double U = I * (1. - eps) + eps * sfc_src;   // surface
      double Uin = U;
#pragma unroll
      for (int q = NW - 1; q >= 0; --q) {
        if (q == w) Uin = U;
        U = x[(q * 3 + 0) * 64 + lane] * U + x[(q * 3 + 2) * 64 + lane];
      }
      // down sweep of the segment: levels s0 .. s0+SEG-1 (the level above each layer); the last wave adds the surface
      I = Iin;
#pragma unroll
      for (int s = 0; s < SEG; ++s) {
        acc_add(&acc_dn[(s0 + s) * CW + cl], gsum<CW>(wfac * I), owner);
        I = T[s] * I + SDN[s];
      }
      if (w == NW - 1) acc_add(&acc_dn[NL * CW + cl], gsum<CW>(wfac * I), owner);
      // up sweep: levels s0+SEG .. s0+1 (the level below each layer); the first wave adds the top
      U = Uin;
#pragma unroll
      for (int s = SEG - 1; s >= 0; --s) {
        acc_add(&acc_up[(s0 + s + 1) * CW + cl], gsum<CW>(wfac * U), owner);
        U = T[s] * U + SU[s];
      }
      if (w == 0) acc_add(&acc_up[cl], gsum<CW>(wfac * U), owner);
    }

    // broadband fluxes of the tile: level s-th from the top -> lev0 + lstep*s
    __syncthreads();
    for (int i = gtid; i < (NL + 1) * CW; i += 64 * NW) {
      const int s = i / CW, c = i - s * CW;
      const long cg = tile * CW + c;
      if (cg < ncol) {
        const long q = cg + (long)ncol * (lev0 + lstep * s);
        a.flux_dn[q] = acc_dn[i];
        a.flux_up[q] = acc_up[i];
      }
      acc_dn[i] = 0.;
      acc_up[i] = 0.;
    }
    __syncthreads();
  }
}

template <int SEG, int NW, int CW, bool SHARED, bool SER3, bool PLANCK, int WPS = ECCKD_SPLIT_WAVES_PER_SIMD>
hipError_t launch_split(const RteLwArgs &a, const PlanckTab &pt, const double *tlay, const double *tlev, const double *tsfc,
                        hipStream_t s) {
  auto k = rte_lw_split_kernel<SEG, NW, CW, SHARED, SER3, PLANCK, WPS>;
  constexpr int NG = split_groups(PLANCK);
  const size_t lds = sizeof(double) * (NG * (2 * (size_t)(SEG * NW + 1) * CW + 2 * NW * 3 * 64) +
                                       (PLANCK ? (size_t)pt.ntp * planck_stride(a.ng) : 0));
  if (lds > (size_t)kLdsBudget) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  long blocks = (((long)a.ncol + CW - 1) / CW + NG - 1) / NG;
  const long cap = PLANCK ? 256L * 4 : 256L * WPS * 4;   // four rounds of resident blocks; the rest by grid stride
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * NW * NG), lds, s, a, pt, tlay, tlev, tsfc);
  return hipGetLastError();
}

}  // namespace

// LDS of the Planck-recomputing form (launch_seg<..., PLANCK = true>: 15 layers per wave, 4 waves, 32 columns): does the
// model's Planck table fit next to the accumulators?  (The shipped 32 / 36-g models do; a 64-g table does not.)
bool rte_lw_planck_fits(int ng, int ntp) {
  constexpr int SEG = 15, NW = 4, CW = 32, NG = split_groups(true);
  const size_t lds = sizeof(double) * (NG * (2 * (size_t)(SEG * NW + 1) * CW + 2 * NW * 3 * 64) + (size_t)ntp * planck_stride(ng));
  return lds <= (size_t)kLdsBudget;
}

bool rte_lw_split_applies(const RteLwArgs &a) {
  return !a.f32 && a.nlay == 60 && a.ncol > 0;
}

template <bool SHARED, bool SER3, bool PLANCK>
static hipError_t launch_seg(const RteLwArgs &a, const PlanckTab &pt, const double *tlay, const double *tlev, const double *tsfc,
                             hipStream_t s) {
  // The Planck-recomputing form needs 244 VGPRs: 15 layers per wave, two waves per SIMD (three spill 25-54 registers)
  if constexpr (PLANCK) {
    return launch_split<15, 4, 32, SHARED, SER3, PLANCK, 2>(a, pt, tlay, tlev, tsfc, s);
  } else {
  switch (a.split_seg) {   // layers per wave: 10 (6 waves per block, 157 VGPRs, 12 waves per CU), 12 or 15
    case 15: return launch_split<15, 4, 32, SHARED, SER3, PLANCK>(a, pt, tlay, tlev, tsfc, s);
    case 12: return launch_split<12, 5, 32, SHARED, SER3, PLANCK>(a, pt, tlay, tlev, tsfc, s);
    default: return launch_split<10, 6, 32, SHARED, SER3, PLANCK>(a, pt, tlay, tlev, tsfc, s);
  }
  }
}

hipError_t launch_rte_lw_split(const RteLwArgs &a, hipStream_t s) {
  const PlanckTab none{nullptr, 0., 1., 1., 2, a.ng};
  if (a.shared_levels)
    return a.series3 ? launch_seg<true, true, false>(a, none, nullptr, nullptr, nullptr, s)
                     : launch_seg<true, false, false>(a, none, nullptr, nullptr, nullptr, s);
  return a.series3 ? launch_seg<false, true, false>(a, none, nullptr, nullptr, nullptr, s)
                   : launch_seg<false, false, false>(a, none, nullptr, nullptr, nullptr, s);
}

hipError_t launch_rte_lw_planck(const RteLwArgs &a, const double *planck, int ntp, double t0, double dt, const double *tlay,
                                const double *tlev, const double *tsfc, hipStream_t s) {
  if (a.f32 || a.nlay != 60 || a.ncol <= 0) return hipErrorInvalidValue;
  const PlanckTab pt{planck, t0, dt, 1. / dt, ntp, a.ng};
  return a.series3 ? launch_seg<false, true, true>(a, pt, tlay, tlev, tsfc, s)
                   : launch_seg<false, false, true>(a, pt, tlay, tlev, tsfc, s);
}

}  // namespace ecckd
