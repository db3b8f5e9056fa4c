// kernels_rte_sw.hip -- shortwave two-stream + adding flux solver with the broadband g-point
// reduction fused in, and the toa_src broadcast of gas_optics_ext.
//
// Replaces RTE-RRTMGP's rte_sw as the reference calls it (example/rfmip-rad-irf/
// ecckd_rfmip_sw.F90:148-154): sw_two_stream (Zdunkowski PIFM / Meador-Weaver), sw_source_2str,
// adding, flux_dn = diffuse + direct, then ty_fluxes_broadband%reduce.
//
// Mapping (gfx950): as the LW solver -- one wave = CW columns x GW=64/CW g-points, lanes that
// share a column are summed with a wave shuffle butterfly into wave-private LDS accumulators.
// Two passes per (column, g-point): bottom->top computes the two-stream coefficients and runs the
// adding recurrences with the source normalised by the direct beam (which is only known on the way
// down); top->bottom propagates the direct beam and the fluxes.  What the second pass needs goes
// through a per-wave global scratch ring ([array][level][lane], 512 B coalesced rows); see the
// RECOMPUTE note at the kernel for what is stored and what is computed twice.
// (Registers cannot hold it next to a useful occupancy: the coefficient arithmetic is ~250 fp64
// instructions per cell and needs several waves per SIMD to issue at rate.)
#include <type_traits>

#include "kernels.hpp"
#include "sw_two_stream.hpp"

namespace ecckd {
namespace {

// f(integral_constant<int, I>) for I = I0 .. N-1 as straight-line code (compile-time slot indices)
template <int I, int N, class F>
__device__ __forceinline__ void static_for_sw(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_sw<I + 1, N>(f);
  }
}

template <int CW>
__device__ __forceinline__ double gsum(double v) {
#pragma unroll
  for (int o = CW; o < 64; o <<= 1) v = v + __shfl_xor(v, o);
  return v;
}

#ifndef ECCKD_SW_WAVES
#define ECCKD_SW_WAVES 4096   // 12 waves per CU are resident (143 VGPRs); a grid of 4096 measured 15 % faster than 2048
#endif
// Registers: the kernel needs 143 VGPRs, i.e. three waves per SIMD (12 per CU).  Forcing it under 128 for four waves
// per SIMD was measured in round 2 (same box, tools/ab.py): with the compiler's 2-9 spills 3.80-3.88 ms, with a prefetch
// depth of 2 (no spill) 3.81-3.85 ms, against 3.78 ms as is -- the kernel is bound by fp64 issue, which three waves per
// SIMD already saturate (tools/ubench.hip: 0.74 of 0.86 wave-instructions/clk/CU), so the allocation is left alone.
#ifndef ECCKD_SW_WAVES_PER_SIMD
#define ECCKD_SW_WAVES_PER_SIMD 3
#endif
#ifndef ECCKD_SW_CW
#define ECCKD_SW_CW 16
#endif
#ifndef ECCKD_SW_RECOMPUTE
#define ECCKD_SW_RECOMPUTE 1
#endif
constexpr bool kSwRecompute = ECCKD_SW_RECOMPUTE != 0;
#ifndef ECCKD_SW_PF
#define ECCKD_SW_PF 3
#endif
constexpr int kPF = ECCKD_SW_PF;   // layers of optical properties in flight per lane
constexpr int kSwWaves = ECCKD_SW_WAVES;

// acc += v by the owner lane only (the other lanes add +0.0): one fire-and-forget ds_add_f64,
// order-independent result (see kernels_rte_lw.hip).
__device__ __forceinline__ void acc_add(double *p, double v, bool owner) {
  __hip_atomic_fetch_add(p, owner ? v : 0., __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// RECOMPUTE = true: pass 2 reads tau/ssa/g again and recomputes the two-stream coefficients; only the
// two per-level quantities go through the scratch ring (24 + 16 + 24 + 16 = 80 B/cell of traffic,
// twice the arithmetic).  RECOMPUTE = false: pass 1 stores the four per-layer products pass 2
// needs as well (24 + 48 + 48 = 120 B/cell).  The kernel is bound by that traffic, not by the
// arithmetic (0.25 VALU wave-instr/clk/CU of 0.81 available at this occupancy): measured 4.52 ms
// stored vs 3.85 ms recomputed per 1e5 columns x 27 g-points (recomputed: 52 % of the fp64 VALU rate).
template <int CW, bool RECOMPUTE, bool FAST, bool CLAMP>
__global__ void __launch_bounds__(64, ECCKD_SW_WAVES_PER_SIMD) rte_sw_kernel(const RteSwArgs a) {
  constexpr int GW = 64 / CW;
  extern __shared__ double acc[];   // [3][nlay+1][CW]: up, dn, dir
  const int lane = threadIdx.x;
  const int cl = lane % CW, gs = lane / CW;
  const bool owner = gs == 0;
  const int ncol = a.ncol, nlay = a.nlay, ng = a.ng, nlev = nlay + 1;
  double *acc_up = acc, *acc_dn = acc + nlev * CW, *acc_dir = acc + 2 * nlev * CW;
  const long lay0 = a.top_at_1 ? 0 : nlay - 1, lev0 = a.top_at_1 ? 0 : nlay;
  const long lstep = a.top_at_1 ? 1 : -1;
  // scratch ring of this wave: level arrays (albedo, normalised source) first, then -- unless they
  // are recomputed -- the four layer arrays; each [index][64 lanes]
  constexpr int NLAYARR = RECOMPUTE ? 0 : 4;
  double *sc = a.scratch + (long)blockIdx.x * ((long)NLAYARR * nlay + 2L * nlev) * 64 + lane;
  double *sAlb = sc, *sSrc = sc + 64L * nlev;
  double *sA = sSrc + 64L * nlev, *sB = sA + 64L * nlay, *sC = sB + 64L * nlay, *sTn = sC + 64L * nlay;
  const int ngroups = (ng + GW - 1) / GW;
  const long ntiles = ((long)ncol + CW - 1) / CW;
  const double k_floor = a.k_floor;

  // Work units (see rte_sw_tail_plan): units [0, tail_first) are whole tiles; beyond that a unit is ONE g-point group
  // of a tail tile and leaves its sums in `partials` for rte_sw_tail_reduce.
  const long tail_first = a.tail_first < 0 ? ntiles : a.tail_first;
  const long nunits = tail_first + (ntiles - tail_first) * ngroups;
  for (long unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
    long tile = unit;
    int g0 = 0, g1 = ngroups;
    if (unit >= tail_first) {
      const long u = unit - tail_first;
      tile = tail_first + u / ngroups;
      g0 = (int)(u - (tile - tail_first) * ngroups);
      g1 = g0 + 1;
    }
    const long col = tile * CW + cl;
    const bool valid = col < ncol;
    const long cc = valid ? col : (long)ncol - 1;
    for (int i = lane; i < 3 * nlev * CW; i += 64) acc[i] = 0.;
    const double mu0 = a.mu0[cc];
    const double mu0_inv = 1. / mu0;

    for (int gi = g0; gi < g1; ++gi) {
      const int g = gi * GW + gs;
      const bool gact = g < ng;
      const int gg = gact ? g : ng - 1;
      const double keep = gact ? 1. : 0.;
      const long base = cc + (long)ncol * nlay * gg;
      const int band = a.gpt2band[gg];

      // ---- pass 1, bottom -> top: two-stream coefficients (sw_two_stream) and the adding
      // recurrences (albedo, source of upward radiation).  The direct beam is not known yet on the
      // way up, so the source is carried normalised by the direct flux at its own level:
      //   src(l) = nsrc(l) * F_dir(l),  F_dir(l+1) = Tnoscat(l) * F_dir(l)
      double albedo = a.alb_dif[band + (long)a.nband * cc];
      double nsrc = a.alb_dir[band + (long)a.nband * cc];   // src_sfc = F_dir(sfc) * sfc_alb_dir
      sAlb[64L * nlay] = albedo;
      sSrc[64L * nlay] = nsrc;
      // (the optical properties of layer s - kPF are requested before the arithmetic of layer s: the
      // loop is a serial recurrence and the compiler does not pipeline it)
      double ptau[kPF], pssa[kPF], pg[kPF];
#pragma unroll
      for (int d = 0; d < kPF; ++d) {
        const int sl = nlay - 1 - d > 0 ? nlay - 1 - d : 0;
        const long q = base + (long)ncol * (lay0 + lstep * sl);
        ptau[d] = a.tau[q]; pssa[d] = a.ssa[q]; pg[d] = a.g[q];
      }
      // The layer loop is unrolled by the prefetch depth so that every layer has a FIXED slot of the prefetch registers:
      // shifting the slots along (ptau[d] = ptau[d + 1]) reads registers whose loads are still in flight and made the
      // compiler wait with vmcnt(0) in every iteration -- one layer of loads in flight instead of kPF (round 2).
      auto layer1 = [&](int s, auto slot_c) __attribute__((always_inline)) {
        constexpr int d = decltype(slot_c)::value;
        const double ctau = ptau[d], cssa = pssa[d], cg = pg[d];
        {
          const long q = base + (long)ncol * (lay0 + lstep * (s - kPF > 0 ? s - kPF : 0));
          ptau[d] = a.tau[q]; pssa[d] = a.ssa[q]; pg[d] = a.g[q];
        }
        const TwoStream ts = __all(cg == 0.) ? two_stream<double, FAST, CLAMP, true>(ctau, cssa, cg, mu0, mu0_inv, k_floor)
                                             : two_stream<double, FAST, CLAMP, false>(ctau, cssa, cg, mu0, mu0_inv, k_floor);
        const double denom = rcp<FAST>(1. - ts.Rdif * albedo);                             // adding, Eq 10
        if (!RECOMPUTE) {
          sA[64L * s] = ts.Tdif * denom;
          sB[64L * s] = ts.Rdif * denom;
          sC[64L * s] = ts.Tdir * denom;
          sTn[64L * s] = ts.Tnoscat;
        }
        // Eq 11 divided by F_dir(l): src_up = Rdir*F_dir(l), src_dn = Tdir*F_dir(l), src(l+1) = nsrc*Tnoscat*F_dir(l)
        nsrc = ts.Rdir + ts.Tdif * denom * (nsrc * ts.Tnoscat + albedo * ts.Tdir);
        albedo = ts.Rdif + ts.Tdif * ts.Tdif * albedo * denom;                        // Eq 9
        sAlb[64L * s] = albedo;
        sSrc[64L * s] = nsrc;
      };
      {
        int s = nlay - 1;
        for (; s >= kPF - 1; s -= kPF)                 // whole groups: slot d serves layer s - d
          static_for_sw<0, kPF>([&](auto dc) __attribute__((always_inline)) { layer1(s - decltype(dc)::value, dc); });
        static_for_sw<0, kPF>([&](auto dc) __attribute__((always_inline)) {   // the last, partial group
          if (s - decltype(dc)::value >= 0) layer1(s - decltype(dc)::value, dc);
        });
      }

      // ---- pass 2, top -> bottom: direct beam and fluxes (Eq 12, 13) ----
      double fdir = a.toa[cc + (long)ncol * gg] * mu0;
      double fdn = 0.;
      {
        const double fup = fdn * albedo + nsrc * fdir;
        const double vu = gsum<CW>(keep * fup), vd = gsum<CW>(keep * (fdn + fdir)), vr = gsum<CW>(keep * fdir);
        acc_add(&acc_up[cl], vu, owner);
        acc_add(&acc_dn[cl], vd, owner);
        acc_add(&acc_dir[cl], vr, owner);
      }
      double palb[kPF], pnsrc[kPF];
#pragma unroll
      for (int d = 0; d < kPF; ++d) {
        const int sl = d < nlay ? d : nlay - 1;
        palb[d] = sAlb[64L * (sl + 1)]; pnsrc[d] = sSrc[64L * (sl + 1)];
        if (RECOMPUTE) {
          const long q = base + (long)ncol * (lay0 + lstep * sl);
          ptau[d] = a.tau[q]; pssa[d] = a.ssa[q]; pg[d] = a.g[q];
        }
      }
      auto layer2 = [&](int s, auto slot_c) __attribute__((always_inline)) {   // (fixed prefetch slots: see pass 1)
        constexpr int d = decltype(slot_c)::value;
        const double alb_next = palb[d], nsrc_next = pnsrc[d];
        const double ctau = ptau[d], cssa = pssa[d], cg = pg[d];
        {
          const int sn = s + kPF < nlay ? s + kPF : nlay - 1;
          palb[d] = sAlb[64L * (sn + 1)]; pnsrc[d] = sSrc[64L * (sn + 1)];
          if (RECOMPUTE) {
            const long q = base + (long)ncol * (lay0 + lstep * sn);
            ptau[d] = a.tau[q]; pssa[d] = a.ssa[q]; pg[d] = a.g[q];
          }
        }
        double A, B, C, Tn;
        if (RECOMPUTE) {
          const TwoStream ts = __all(cg == 0.) ? two_stream<double, FAST, CLAMP, true>(ctau, cssa, cg, mu0, mu0_inv, k_floor)
                                               : two_stream<double, FAST, CLAMP, false>(ctau, cssa, cg, mu0, mu0_inv, k_floor);
          const double denom = rcp<FAST>(1. - ts.Rdif * alb_next);     // the same expression as in pass 1: same bits
          A = ts.Tdif * denom; B = ts.Rdif * denom; C = ts.Tdir * denom; Tn = ts.Tnoscat;
        } else {
          A = sA[64L * s]; B = sB[64L * s]; C = sC[64L * s]; Tn = sTn[64L * s];
        }
        const double fdir_next = Tn * fdir;
        const double src_next = nsrc_next * fdir_next;
        fdn = A * fdn + B * src_next + C * fdir;
        const double fup = fdn * alb_next + src_next;
        fdir = fdir_next;
        const double vu = gsum<CW>(keep * fup), vd = gsum<CW>(keep * (fdn + fdir)), vr = gsum<CW>(keep * fdir);
        acc_add(&acc_up[(s + 1) * CW + cl], vu, owner);
        acc_add(&acc_dn[(s + 1) * CW + cl], vd, owner);
        acc_add(&acc_dir[(s + 1) * CW + cl], vr, owner);
      };
      {
        int s = 0;
        for (; s + kPF <= nlay; s += kPF)              // whole groups: slot d serves layer s + d
          static_for_sw<0, kPF>([&](auto dc) __attribute__((always_inline)) { layer2(s + decltype(dc)::value, dc); });
        static_for_sw<0, kPF>([&](auto dc) __attribute__((always_inline)) {   // the last, partial group
          if (s + decltype(dc)::value < nlay) layer2(s + decltype(dc)::value, dc);
        });
      }
    }

    if (unit >= tail_first) {   // one g-point group of a tail tile: [unit][up, dn, dir][nlev][CW]
      double *pp = a.partials + (unit - tail_first) * 3 * nlev * CW;
      for (int s = gs; s < nlev; s += GW) {
        pp[s * CW + cl] = acc_up[s * CW + cl];
        pp[(nlev + s) * CW + cl] = acc_dn[s * CW + cl];
        pp[(2 * nlev + s) * CW + cl] = acc_dir[s * CW + cl];
      }
    } else if (valid) {
      for (int s = gs; s < nlev; s += GW) {
        const long q = col + (long)ncol * (lev0 + lstep * s);
        a.flux_up[q] = acc_up[s * CW + cl];
        a.flux_dn[q] = acc_dn[s * CW + cl];
        if (a.flux_dir) a.flux_dir[q] = acc_dir[s * CW + cl];
      }
    }
  }
}

// Sums the per-group partial fluxes of the tail tiles in group order: the order in which a whole-tile wave adds the
// same values to accumulators that start at +0, hence the same bits (see rte_lw_tail_reduce, kernels_rte_lw.hip).
template <typename real>
__global__ void __launch_bounds__(256) rte_sw_tail_reduce(const double *partials, int ngroups, int nlev, int cw, long tail_first,
                                                          long ntail, int ncol, long lev0, long lstep, real *flux_up,
                                                          real *flux_dn, real *flux_dir) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= ntail * nlev * cw) return;
  const int cl = (int)(idx % cw), s = (int)((idx / cw) % nlev);
  const long t = idx / ((long)cw * nlev);
  const long col = (tail_first + t) * cw + cl;
  if (col >= ncol) return;
  const double *p = partials + (t * ngroups * 3 * nlev + s) * cw + cl;
  double up = 0., dn = 0., dir = 0.;
  for (int gi = 0; gi < ngroups; ++gi, p += 3L * nlev * cw) {
    up += p[0];
    dn += p[(long)nlev * cw];
    dir += p[2L * nlev * cw];
  }
  const long q = col + (long)ncol * (lev0 + lstep * s);
  flux_up[q] = (real)up;
  flux_dn[q] = (real)dn;
  if (flux_dir) flux_dir[q] = (real)dir;
}

template <typename real>
__global__ void toa_src_kernel(const real *solar, int ncol, int ng, real *toa) {
  const long n = (long)ncol * ng;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x)
    toa[q] = solar[q / ncol];   // src/gas_optics_ecckd.f90:468-472
}

template <typename real>
__global__ void sum_planes_kernel(const real *planes, int nplanes, size_t n, real *out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    real acc = 0;
    for (int b = 0; b < nplanes; ++b) acc += planes[(size_t)b * n + i];
    out[i] = acc;
  }
}

}  // namespace

size_t rte_sw_scratch_bytes(int ncol, int nlay, int ng) {
  (void)ng;
  long tiles = ((long)ncol + ECCKD_SW_CW - 1) / ECCKD_SW_CW;
  if (tiles > kSwWaves) tiles = kSwWaves;
  return sizeof(double) * (size_t)((kSwRecompute ? 0L : 4L) * nlay + 2L * (nlay + 1)) * 64 * (size_t)tiles;
}

namespace {
size_t sw_ring_bytes(int nlay, long blocks) {
  return sizeof(double) * (size_t)((kSwRecompute ? 0L : 4L) * nlay + 2L * (nlay + 1)) * 64 * (size_t)blocks;
}
}  // namespace

// Tail split of rte_sw, as rte_lw_tail_plan (kernels_rte_lw.hip): the grid is persistent (kSwWaves blocks, three per
// SIMD); the tiles of a call that does not fill one round are handed out one g-point group per wave.  Returns the bytes the split
// needs in all -- the scratch ring of the (possibly larger) grid, then the partial sums at offset *partials_at -- or 0.
size_t rte_sw_tail_plan(const RteSwArgs &a, long *tail_first, size_t *partials_at) {
  constexpr double kTailGain = 0.03;
  constexpr size_t kTailMaxBytes = (size_t)64 << 20;
  *tail_first = -1;
  *partials_at = 0;
  if (a.ncol <= 0) return 0;
  constexpr int CW = ECCKD_SW_CW, GW = 64 / CW;
  const long ngroups = (a.ng + GW - 1) / GW;
  const long tiles = ((long)a.ncol + CW - 1) / CW, full = tiles / kSwWaves * kSwWaves, ntail = tiles - full;
  if (ntail == 0 || ngroups < 2) return 0;
  // Only calls of less than one round: with three waves per SIMD a part-empty last round costs next to nothing (the
  // waves left on a SIMD run faster: 100 000 columns 3.36 against 3.39 ms with and without the split, 50 000 columns
  // 1.85 against 1.81), while a small call gains the factor the groups run side by side (1 000 columns: 0.43 -> 0.08 ms).
  if (full > 0) return 0;
  const double before = (double)(full / kSwWaves + 1);
  const double after = (double)(full / kSwWaves) + (double)((ntail * ngroups + kSwWaves - 1) / kSwWaves) / (double)ngroups;
  if (before - after < kTailGain * before) return 0;
  const size_t part = sizeof(double) * 3 * (size_t)(a.nlay + 1) * CW * (size_t)(ntail * ngroups);
  if (part > kTailMaxBytes) return 0;
  long blocks = full + ntail * ngroups;
  if (blocks > kSwWaves) blocks = kSwWaves;
  *tail_first = full;
  *partials_at = (sw_ring_bytes(a.nlay, blocks) + 255) & ~(size_t)255;
  return *partials_at + part;
}

hipError_t launch_rte_sw(const RteSwArgs &a, hipStream_t s) {
  if (a.ncol <= 0) return hipSuccess;
  if (a.f32 || a.derive) return hipErrorInvalidValue;   // single precision / fused form: layer-systolic solver only
  constexpr int CW = ECCKD_SW_CW;
  auto k = a.dir_clamp ? (a.exact_division ? rte_sw_kernel<CW, kSwRecompute, false, true> : rte_sw_kernel<CW, kSwRecompute, true, true>)
                       : (a.exact_division ? rte_sw_kernel<CW, kSwRecompute, false, false> : rte_sw_kernel<CW, kSwRecompute, true, false>);
  const size_t lds = sizeof(double) * 3 * (size_t)(a.nlay + 1) * CW;
  if (lds > (size_t)kLdsBudget) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const long tiles = ((long)a.ncol + CW - 1) / CW;
  const long ntail = a.tail_first >= 0 ? tiles - a.tail_first : 0;
  const int ngroups = (a.ng + 64 / CW - 1) / (64 / CW);
  long blocks = tiles - ntail + ntail * ngroups;
  if (blocks > kSwWaves) blocks = kSwWaves;
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64), lds, s, a);
  e = hipGetLastError();
  if (e != hipSuccess || ntail == 0) return e;
  return launch_rte_sw_tail_reduce(a, ngroups, CW, a.tail_first, ntail, s);
}

hipError_t launch_rte_sw_tail_reduce(const RteSwArgs &a, int nchunks, int cw, long tail_first, long ntail, hipStream_t s) {
  const int nlev = a.nlay + 1;
  const long n = ntail * nlev * cw;
  if (n <= 0) return hipSuccess;
  const long lev0 = a.top_at_1 ? 0 : a.nlay, lstep = a.top_at_1 ? 1 : -1;
  if (a.f32)
    hipLaunchKernelGGL(rte_sw_tail_reduce<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.partials, nchunks, nlev, cw,
                       tail_first, ntail, a.ncol, lev0, lstep, reinterpret_cast<float *>(a.flux_up),
                       reinterpret_cast<float *>(a.flux_dn), reinterpret_cast<float *>(a.flux_dir));
  else
    hipLaunchKernelGGL(rte_sw_tail_reduce<double>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.partials, nchunks, nlev, cw,
                       tail_first, ntail, a.ncol, lev0, lstep, a.flux_up, a.flux_dn, a.flux_dir);
  return hipGetLastError();
}

hipError_t launch_toa_src(const double *solar, int ncol, int ng, double *toa_src, int f32, hipStream_t s) {
  const long n = (long)ncol * ng;
  if (n <= 0) return hipSuccess;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (f32)
    hipLaunchKernelGGL(toa_src_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float *>(solar), ncol, ng,
                       reinterpret_cast<float *>(toa_src));
  else
    hipLaunchKernelGGL(toa_src_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, s, solar, ncol, ng, toa_src);
  return hipGetLastError();
}

hipError_t launch_sum_planes(const double *planes, int nplanes, size_t n, double *out, int f32, hipStream_t s) {
  if (n == 0) return hipSuccess;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (f32)
    hipLaunchKernelGGL(sum_planes_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s,
                       reinterpret_cast<const float *>(planes), nplanes, n, reinterpret_cast<float *>(out));
  else
    hipLaunchKernelGGL(sum_planes_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, s, planes, nplanes, n, out);
  return hipGetLastError();
}

}  // namespace ecckd
