// kernels_rte_lw.hip -- longwave no-scattering flux solver with the broadband g-point
// reduction fused in.
//
// Replaces RTE-RRTMGP's rte_lw as the reference calls it (example/rfmip-rad-irf/
// ecckd_rfmip_lw.F90:130-135): lw_solver_noscat_GaussQuad (transmittance, lw_source_noscat,
// lw_transport_noscat) followed by ty_fluxes_broadband%reduce (sum_broadband).
//
// Mapping (gfx950): one wave = CW columns x GW=64/CW g-points (CW = 32: measured 5 % faster than
// 16).  A load instruction therefore touches GW segments of CW*8 B (whole 128 B lines) of the
// column-fastest inputs.
// Each lane owns one (column, g-point) pair at a time and walks the layer recurrence:
//   down sweep: reads tau/lay_source/lev_source_{inc,dec} ONCE (4x8 B per cell, software
//               prefetched PF layers ahead), keeps trans(l) and source_up(l) in registers
//               (fully unrolled over NL layers), accumulates 2*pi*w*I_dn per level;
//   up sweep:   runs out of registers, accumulates 2*pi*w*I_up per level.
// The sum over g-points is a wave shuffle butterfly over the GW lanes that share a column
// followed by an add into a wave-private LDS accumulator (no inter-wave traffic, no atomics,
// deterministic).  After the last g-point group the accumulators are the broadband fluxes.
// Registers (512 per lane at one wave per SIMD) are the only place on the chip large enough
// for the 2*NL doubles per pair that the up sweep needs; the HBM latency is hidden by the
// explicit prefetch ring instead of by occupancy.
//
// NL > 0, EXACT: nlay == NL exactly, no per-layer predicates in the unrolled code (the 60-layer
//   benchmark and RFMIP shape).
// NL > 0, !EXACT: any nlay <= NL.  Layers s >= nlay of the unrolled code are made transparent
//   (tau = 0: trans = 1, sources 0) and accumulate into a dummy LDS row; the predicate s < nlay is
//   re-derived from an opaque copy of nlay at every use -- as a plain comparison the optimiser
//   hoists NL loop-invariant booleans and spills the SGPR file.  Costs NL/nlay of the arithmetic.
// NL > 0, OVER: nlay > NL.  The bottom NL layers run from registers as above; trans/source_up of
//   the top nlay - NL layers go through a per-wave global scratch ring (16 B/cell written and read
//   back for those layers only).
// SHARED: the caller asserts that the two level-source arrays describe ONE value per level,
//   lev_source_inc(:,l,:) == lev_source_dec(:,l+1,:) -- what ecckd's gas optics produces
//   (src/gas_optics_ecckd.f90:419-424: both are slices of one buffer).  The source at the far edge
//   of a layer is then the near-edge source of the next one: it is carried in a register and only
//   the first layer reads the second array (24 instead of 32 B/cell).  Same arithmetic, same bits.
#include <cstdlib>

#include "kernels.hpp"
#include "lw_layer.hpp"

namespace ecckd {
namespace {

#ifndef ECCKD_LW_PF
#define ECCKD_LW_PF 8
#endif
#ifndef ECCKD_LW_CW
#define ECCKD_LW_CW 32
#endif
#ifndef ECCKD_LW_CW_F32
#define ECCKD_LW_CW_F32 32
#endif
// prefetch depth in layers: 8 (measured 2 % faster than 4, equal to 12 and 16); 4 for the 96-layer variants,
// whose 2 x 96 resident values leave no room for a deeper ring
constexpr int prefetch_depth(int NL) { return NL >= 96 ? 4 : ECCKD_LW_PF; }
#ifndef ECCKD_LW_SPAN
#define ECCKD_LW_SPAN 2
#endif
constexpr int kSchedSpan = ECCKD_LW_SPAN;   // layers the instruction scheduler may interleave

template <typename real, int CW>
__device__ __forceinline__ real gsum(real v) {
  // sum over the lanes that share a column: lane = cl + CW*gs
#pragma unroll
  for (int o = CW; o < 64; o <<= 1) v = v + __shfl_xor(v, o);
  return v;
}

// acc += v by the owner lane (gs == 0) only, as one fire-and-forget ds_add_f64 (an exec-masked
// read/wait/add/write would expose the full LDS latency twice per layer at one wave per SIMD).
// The accumulators are double in both precisions: ds_add_f32 retires one lane every 3 clocks on
// gfx950 (193 clocks per wave instruction, tools/ubench_atomic.hip) against 6-13 clocks for
// ds_add_f64, and the broadband sum is the better for it.
// Every lane issues the atomic and non-owners add +0.0, which leaves the sum bit-identical whatever
// order the LDS unit serialises them in.  Exec-masking the owners (ECCKD_LW_MASKED) makes the
// atomic itself cheaper (10.0 against 12.9 clocks at 4 waves per CU) but the exec juggling costs
// more than that in the kernel: measured 2-5 % slower.
template <typename real>
__device__ __forceinline__ void acc_add(double *p, real v, bool owner) {
#ifdef ECCKD_LW_MASKED
  if (owner) __hip_atomic_fetch_add(p, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#else
  __hip_atomic_fetch_add(p, owner ? (double)v : 0., __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#endif
}

// OFF32 (the exact-layer-count variants): the inputs of a g-point group are addressed as a wave-uniform pointer to the
// group's first plane plus ONE 32-bit byte offset per lane, advanced by a layer per request -- the loads take the pointer in
// SGPRs and no vector instruction builds an address (five 64-bit vector operations per layer otherwise, in a kernel that is
// paid per instruction).  Needs GW planes to span less than 4 GiB; launch_real() falls back to 64-bit offsets beyond.
template <typename real, int NL, int CW, bool EXACT, bool OVER, bool SHARED, bool SER3, bool OFF32 = false>
__global__ void __launch_bounds__(64) rte_lw_kernel(const RteLwArgs a) {
  static_assert(NL > 0 && !(EXACT && OVER), "unrolled layer count; overflow only in the padded form");
  static_assert(!OFF32 || EXACT, "32-bit offsets: exact layer count only");
  constexpr int kPF = prefetch_depth(NL);
  static_assert(kPF <= NL, "prefetch ring deeper than the unrolled layer count");
  constexpr int GW = 64 / CW;
  extern __shared__ __attribute__((aligned(16))) unsigned char acc_raw[];
  double *acc = reinterpret_cast<double *>(acc_raw);   // [2][nlay+1][CW], double in both precisions
  // RteLwArgs carries `double` pointers; in the single-precision instantiation they address float data
  auto P = [](const double *p) { return reinterpret_cast<const real *>(p); };
  auto Q = [](double *p) { return reinterpret_cast<real *>(p); };
  const int lane = threadIdx.x;
  const int cl = lane % CW, gs = lane / CW;
  const bool owner = gs == 0;
  const int ncol = a.ncol, nlay = a.nlay, ng = a.ng;
  const int nlev = nlay + 1;
  // (padded variants: one extra row per array, index nlev, swallows the absent layers' adds)
  constexpr bool PAD = !EXACT;
  const int nover = OVER ? (nlay > NL ? nlay - NL : 0) : 0;   // layers above the register-resident ones
  double *acc_dn = acc, *acc_up = acc + (nlev + (PAD ? 1 : 0)) * CW;
  // s < nlay with nlay read through an opaque asm (see the header comment)
  // index of the s_-th register-resident layer among all layers (opaque for the same reason)
  auto abs_layer = [&](int s_) {
    if (!OVER) return s_;
    int no = nover;
    asm volatile("" : "+s"(no));
    return no + s_;
  };
  auto present = [&](int s_) {
    int nl = nlay;
    asm volatile("" : "+s"(nl));
    return s_ < nl;
  };
  const real pi = (real)acos(-1.);
  const real tau_thresh = (real)a.tau_thresh;   // sqrt(epsilon(1._wp)) unless ecckd_set_solver_option moved it
  // layer / level walked s-th from the top lives at index l0 + s*lstep
  const long lay0 = a.top_at_1 ? 0 : nlay - 1, lev0 = a.top_at_1 ? 0 : nlay;
  const long lstep = a.top_at_1 ? 1 : -1;
  const real *Bdn = P(a.top_at_1 ? a.lev_source_inc : a.lev_source_dec);
  const real *Bup = P(a.top_at_1 ? a.lev_source_dec : a.lev_source_inc);
  const int ngroups = (ng + GW - 1) / GW;
  const int niter = ngroups * a.nmus;
  const long ntiles = ((long)ncol + CW - 1) / CW;

  // Work units: blocks [0, tail_first) take one whole tile (all g-pairs); beyond that a block takes ONE g-pair
  // iteration of a tail tile and leaves its sums in `partials` for rte_lw_tail_reduce (see launch_ser).
  const long tail_first = (OVER || a.tail_first < 0) ? ntiles : a.tail_first;
  const long nunits = tail_first + (ntiles - tail_first) * niter;
  for (long unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
    long tile = unit;
    int it0 = 0, it1 = niter;
    if (!OVER && unit >= tail_first) {
      const long u = unit - tail_first;
      tile = tail_first + u / niter;
      it0 = (int)(u - (tile - tail_first) * niter);
      it1 = it0 + 1;
    }
    const long col = tile * CW + cl;
    const bool valid = col < ncol;
    const long cc = valid ? col : (long)ncol - 1;
    for (int i = lane; i < 2 * (nlev + (PAD ? 1 : 0)) * CW; i += 64) acc[i] = 0.;

    real T[NL], SU[NL];
    [[maybe_unused]] real *sT = nullptr, *sSU = nullptr;
    if constexpr (OVER) {
      sT = Q(a.scratch) + ((long)blockIdx.x * 2 * nover) * 64 + lane;
      sSU = sT + (long)nover * 64;
    }
    real ptau[kPF], play[kPF], pbdn[kPF];
    [[maybe_unused]] real pbup[SHARED ? 1 : kPF];
    [[maybe_unused]] real bup_first = real(0);   // SHARED: near-edge source of the first register-resident layer

    // Element offset of the next layer to prefetch.  It is advanced step by step and made
    // opaque to the optimiser after every advance: otherwise the fully unrolled layer loop is
    // rewritten as base + s*step with 60 loop-invariant scalar offsets that spill the SGPR file.
    long qn = 0;
    const long qstep = (long)ncol * lstep;
    typedef __attribute__((address_space(1))) const char gcchar_t;
    typedef __attribute__((address_space(1))) const real greal_t;
    [[maybe_unused]] unsigned vo = 0;                                           // OFF32: byte offset of this lane's next request
    [[maybe_unused]] const unsigned vstep = (unsigned)((long)sizeof(real) * qstep);   // (wraps for bottom-up storage)
    [[maybe_unused]] const real *tau_b = nullptr, *lay_b = nullptr, *bdn_b = nullptr, *bup_b = nullptr;   // first plane of the group
    auto at32 = [&](const real *plane) -> real {
      return __builtin_nontemporal_load((greal_t *)((gcchar_t *)plane + vo));
    };
    auto pair_start = [&](int it) {
      const int gb = (it / a.nmus) * GW;
      const int g = gb + gs;
      const int gg = g < ng ? g : ng - 1;
      if constexpr (OFF32) {
        const long pl = (long)ncol * nlay * gb;
        tau_b = P(a.tau) + pl; lay_b = P(a.lay_source) + pl; bdn_b = Bdn + pl; bup_b = Bup + pl;
        vo = (unsigned)sizeof(real) * (unsigned)(cc + (long)ncol * nlay * (gg - gb) + (long)ncol * lay0);
        asm volatile("" : "+v"(vo));
        if (SHARED) bup_first = at32(bup_b);
        return;
      }
      qn = cc + (long)ncol * nlay * gg + (long)ncol * (lay0 + lstep * nover);
      asm volatile("" : "+v"(qn));
      if (SHARED && !OVER) bup_first = __builtin_nontemporal_load(Bup + qn);
    };
    // `sl` = layer being requested (compile-time in the unrolled code); in the padded variants the
    // offset stops advancing at the last real layer, so absent layers re-read it (finite data).
    auto issue = [&](int slot, int sl) {
      if constexpr (OFF32) {
        ptau[slot] = at32(tau_b);
        play[slot] = at32(lay_b);
        pbdn[slot] = at32(bdn_b);
        if (!SHARED) pbup[slot] = at32(bup_b);
        vo += vstep;
        asm volatile("" : "+v"(vo));
        return;
      }
#ifndef ECCKD_LW_PLAIN_LOADS   // nontemporal: read-once streams
      ptau[slot] = __builtin_nontemporal_load(P(a.tau) + qn);
      play[slot] = __builtin_nontemporal_load(P(a.lay_source) + qn);
      pbdn[slot] = __builtin_nontemporal_load(Bdn + qn);
      if (!SHARED) pbup[slot] = __builtin_nontemporal_load(Bup + qn);
#else
      ptau[slot] = P(a.tau)[qn];
      play[slot] = P(a.lay_source)[qn];
      pbdn[slot] = Bdn[qn];
      if (!SHARED) pbup[slot] = Bup[qn];
#endif
      if (!PAD || present(abs_layer(sl + 1))) qn += qstep;
      asm volatile("" : "+v"(qn));
    };
    // Same, but pinned into the recurrence: the empty asm also "modifies" the running
    // intensity, so the loads for layer s+kPF cannot be hoisted above layer s-1 (without this
    // the scheduler issues dozens of layers of loads up front and spills the register file).
    auto issue_after = [&](int slot, int sl, real &pin) {
      if constexpr (OFF32) asm volatile("" : "+v"(vo), "+v"(pin));
      else asm volatile("" : "+v"(qn), "+v"(pin));
      issue(slot, sl);
    };

    pair_start(it0);
#pragma unroll
    for (int s = 0; s < kPF; ++s) issue(s, s);

    for (int it = it0; it < it1; ++it) {
      const int gi = it / a.nmus, k = it - gi * a.nmus;
      const int g = gi * GW + gs;
      const bool gact = g < ng;
      const int gg = gact ? g : ng - 1;
      const long base = cc + (long)ncol * nlay * gg;
      const real D = (real)a.Ds[k];
      const real wfac = gact ? real(2) * pi * (real)a.wts[k] : real(0);
      const real eps = P(a.sfc_emis)[a.gpt2band[gg] + (long)a.nband * cc];
      const real sfc_src = P(a.sfc_source)[cc + (long)ncol * gg];

      // ---------------- down sweep ----------------
      // radn_dn(top): no incident diffuse flux, or inc_flux turned into an intensity
      real I = real(0);
      if (a.inc_flux) {
        const real f = P(a.inc_flux)[cc + (long)ncol * gg];
        I = a.inc_isotropic ? f / pi : f / (real(2) * pi * (real)a.wts[k]);
      }
      auto layer = [&](int s, real tau, real lay, real bdn, real bup, real &t_out,
                       real &su_out) {
        const bool act = !PAD || present(s);
        if (PAD) tau = act ? tau : real(0);   // absent layer: trans = 1, both sources 0
        const real tl = tau * D;
        const real t = lw_exp(-tl);
        const real omt = real(1) - t;
        // both branches of lw_source_noscat's merge() are evaluated and selected (no branch)
        const real fact_big = lw_div(omt, tl) - t;
        const real fact_small = SER3 ? tl * (real(0.5) + tl * (-real(1) / real(3) + tl * (real(1) / real(8))))
                                     : tl * (real(0.5) - real(1) / real(3) * tl);
        const real fact = tl > tau_thresh ? fact_big : fact_small;
        const real sdn = omt * bdn + real(2) * fact * (lay - bdn);
        real su = omt * bup + real(2) * fact * (lay - bup);
        // Materialise source_up here: left alone, the optimiser sinks this expression into the
        // up sweep and keeps its five inputs alive per layer instead of the one result.
        asm volatile("" : "+v"(su));
        su_out = su;
        t_out = t;
        const real v = gsum<real, CW>(wfac * I);
        acc_add(&acc_dn[(act ? s : nlev) * CW + cl], v, owner);
        I = t * I + sdn;
      };
      [[maybe_unused]] real carry = bup_first;   // SHARED: far-edge source of the layer above
      if constexpr (OVER) {   // the layers above the register-resident ones: plain loads, scratch ring
        for (int s = 0; s < nover; ++s) {
          const long q = base + (long)ncol * (lay0 + lstep * s);
          const real bdn = Bdn[q];
          const real bup = (SHARED && s > 0) ? carry : Bup[q];
          real t, su;
          layer(s, P(a.tau)[q], P(a.lay_source)[q], bdn, bup, t, su);
          carry = bdn;
          sT[(long)s * 64] = t;
          sSU[(long)s * 64] = su;
        }
      }
#pragma unroll
      for (int s = 0; s < NL; ++s) {
        const real tau = ptau[s % kPF], lay = play[s % kPF], bdn = pbdn[s % kPF];
        const real bup = SHARED ? carry : pbup[s % kPF];
        if (s + kPF < NL) issue_after(s % kPF, s + kPF, I);
        layer(abs_layer(s), tau, lay, bdn, bup, T[s], SU[s]);
        if (SHARED) carry = bdn;
        if (s % kSchedSpan == kSchedSpan - 1) __builtin_amdgcn_sched_barrier(0);
      }
      // the next pair's first register-resident layers start streaming while the up sweep runs
      if (it + 1 < it1) {
        pair_start(it + 1);
#pragma unroll
        for (int s = 0; s < kPF; ++s) issue(s, s);
      }
      {
        const real v = gsum<real, CW>(wfac * I);
        acc_add(&acc_dn[nlay * CW + cl], v, owner);
      }
      // ---------------- surface + up sweep ----------------
      real U = I * (real(1) - eps) + eps * sfc_src;
      auto up = [&](int s, real t, real su) {
        const real v = gsum<real, CW>(wfac * U);
        acc_add(&acc_up[((!PAD || present(s)) ? s + 1 : nlev) * CW + cl], v, owner);
        U = t * U + su;
      };
#pragma unroll
      for (int s = NL - 1; s >= 0; --s) up(abs_layer(s), T[s], SU[s]);
      if constexpr (OVER) {
        for (int s = nover - 1; s >= 0; --s) up(s, sT[(long)s * 64], sSU[(long)s * 64]);
      }
      {
        const real v = gsum<real, CW>(wfac * U);
        acc_add(&acc_up[cl], v, owner);
      }
    }

    if (!OVER && unit >= tail_first) {   // one g-pair iteration of a tail tile: [unit][dn, up][nlev][CW]
      double *pp = a.partials + (unit - tail_first) * 2 * nlev * CW;
      for (int s = gs; s < nlev; s += GW) {
        pp[s * CW + cl] = acc_dn[s * CW + cl];
        pp[(nlev + s) * CW + cl] = acc_up[s * CW + cl];
      }
    } else
    // broadband fluxes: level s-th from the top -> lev0 + lstep*s
    if (valid) {
      for (int s = gs; s < nlev; s += GW) {
        const long q = col + (long)ncol * (lev0 + lstep * s);
        Q(a.flux_dn)[q] = (real)acc_dn[s * CW + cl];
        Q(a.flux_up)[q] = (real)acc_up[s * CW + cl];
      }
    }
  }
}

// Sums the per-iteration partial fluxes of the tail tiles in iteration order.  A whole-tile wave adds the value of
// iteration 0, 1, ... to an accumulator that starts at +0; a tail unit holds 0 + v_it == v_it, so adding the units in
// the same order reproduces the whole-tile sum bit for bit: whether a column lands in a tail tile does not show.
template <typename real>
__global__ void __launch_bounds__(256) rte_lw_tail_reduce(const double *partials, int niter, int nlev, int cw, long tail_first,
                                                          long ntail, int ncol, long lev0, long lstep, real *flux_dn, real *flux_up) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= ntail * nlev * cw) return;
  const int cl = (int)(idx % cw), s = (int)((idx / cw) % nlev);
  const long t = idx / ((long)cw * nlev);
  const long col = (tail_first + t) * cw + cl;
  if (col >= ncol) return;
  const double *p = partials + (t * niter * 2 * nlev + s) * cw + cl;
  double dn = 0., up = 0.;
  for (int it = 0; it < niter; ++it, p += 2L * nlev * cw) {
    dn += p[0];
    up += p[(long)nlev * cw];
  }
  const long q = col + (long)ncol * (lev0 + lstep * s);
  flux_dn[q] = (real)dn;
  flux_up[q] = (real)up;
}

constexpr int kOverWaves = 2048;   // grid of the overflow variant (its scratch ring is per wave)
constexpr int kMaxRegisterLayers = 96;   // largest unrolled variant: 2 * 96 values + ~90 working registers of 512
constexpr int kOverCW = 16;   // 16 * (nlay + 2) * CW bytes of LDS accumulators per wave: 16 columns keep 4 waves per CU

template <typename real, int NL, int CW, bool EXACT, bool OVER, bool SHARED, bool SER3, bool OFF32 = false>
hipError_t launch_ser(const RteLwArgs &a, hipStream_t s) {
  auto k = rte_lw_kernel<real, NL, CW, EXACT, OVER, SHARED, SER3, OFF32>;
  const size_t lds = sizeof(double) * 2 * (size_t)(a.nlay + 1 + (EXACT ? 0 : 1)) * CW;
  if (lds > (size_t)kLdsBudget) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  long tiles = ((long)a.ncol + CW - 1) / CW;
  if (OVER && tiles > kOverWaves) tiles = kOverWaves;
  const long ntail = (!OVER && a.tail_first >= 0) ? tiles - a.tail_first : 0;
  const int niter = ((a.ng + 64 / CW - 1) / (64 / CW)) * a.nmus;
  const long blocks = tiles - ntail + ntail * niter;
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64), lds, s, a);
  e = hipGetLastError();
  if (e != hipSuccess || ntail == 0) return e;
  const int nlev = a.nlay + 1;
  const long n = ntail * nlev * CW;
  const long lev0 = a.top_at_1 ? 0 : a.nlay, lstep = a.top_at_1 ? 1 : -1;
  hipLaunchKernelGGL(rte_lw_tail_reduce<real>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.partials, niter, nlev, CW,
                     a.tail_first, ntail, a.ncol, lev0, lstep, reinterpret_cast<real *>(a.flux_dn), reinterpret_cast<real *>(a.flux_up));
  return hipGetLastError();
}

template <typename real, int NL, int CW, bool EXACT, bool OVER, bool SHARED>
hipError_t launch_one(const RteLwArgs &a, hipStream_t s) {
  if constexpr (EXACT) {   // 32-bit lane offsets when the planes of a g-point group span less than 4 GiB (see OFF32)
#ifndef ECCKD_LW_NO_OFF32
    // (ECCKD_LW_NO_OFF32 in the environment: tests run the 64-bit form, which only calls beyond 4.4e6 columns take otherwise)
    if ((double)a.ncol * a.nlay * (64 / CW) * sizeof(real) < 4294967296. && !getenv("ECCKD_LW_NO_OFF32"))
      return a.series3 ? launch_ser<real, NL, CW, EXACT, OVER, SHARED, true, true>(a, s)
                       : launch_ser<real, NL, CW, EXACT, OVER, SHARED, false, true>(a, s);
#endif
  }
  return a.series3 ? launch_ser<real, NL, CW, EXACT, OVER, SHARED, true>(a, s)
                   : launch_ser<real, NL, CW, EXACT, OVER, SHARED, false>(a, s);
}

template <typename real, bool SHARED>
hipError_t launch_real(const RteLwArgs &a, hipStream_t s) {
  constexpr int CW = sizeof(real) == 8 ? ECCKD_LW_CW : ECCKD_LW_CW_F32;
  if (a.nlay == 60) return launch_one<real, 60, CW, true, false, SHARED>(a, s);
  if (a.nlay <= 32) return launch_one<real, 32, CW, false, false, SHARED>(a, s);
  if (a.nlay <= 48) return launch_one<real, 48, CW, false, false, SHARED>(a, s);
  if (a.nlay <= 64) return launch_one<real, 64, CW, false, false, SHARED>(a, s);
  if (a.nlay <= 80) return launch_one<real, 80, CW, false, false, SHARED>(a, s);
  if (a.nlay <= kMaxRegisterLayers) return launch_one<real, 96, CW, false, false, SHARED>(a, s);
  return launch_one<real, 96, kOverCW, false, true, SHARED>(a, s);
}

}  // namespace

// Tail split of the register-resident solver.  A wave owns a SIMD (512 registers), so `slots` tiles run at a time and
// the tiles beyond the last full round keep a fraction of the SIMDs busy for a whole tile time (1e5 columns: 3 125
// tiles on 1 024 SIMDs, the 4th round holds 53).  When that costs more than kTailGain of the call, the tail tiles are
// handed out one g-pair iteration per wave instead (53 x 16 units, one sixteenth of a round) and summed by
// rte_lw_tail_reduce, with the same bits.  Returns the bytes of `partials` the split needs (0: no split) and the first
// tail tile; the caller sets a.tail_first / a.partials when it can provide them and leaves tail_first = -1 otherwise.
size_t rte_lw_tail_plan(const RteLwArgs &a, int slots, long *tail_first) {
  constexpr double kTailGain = 0.03;
  constexpr size_t kTailMaxBytes = (size_t)64 << 20;
  *tail_first = -1;
  if (a.ncol <= 0 || a.nlay > kMaxRegisterLayers || slots <= 0) return 0;
  if (a.use_split && rte_lw_split_applies(a)) return 0;
  const int cw = a.f32 ? ECCKD_LW_CW_F32 : ECCKD_LW_CW, gw = 64 / cw;
  const long niter = (long)((a.ng + gw - 1) / gw) * a.nmus;
  const long tiles = ((long)a.ncol + cw - 1) / cw, full = tiles / slots * slots, ntail = tiles - full;
  if (ntail == 0 || niter < 2) return 0;
  const double before = (double)(full / slots + 1);
  const double after = (double)(full / slots) + (double)((ntail * niter + slots - 1) / slots) / (double)niter;
  if (before - after < kTailGain * before) return 0;
  const size_t bytes = sizeof(double) * 2 * (size_t)(a.nlay + 1) * cw * (size_t)(ntail * niter);
  if (bytes > kTailMaxBytes) return 0;
  *tail_first = full;
  return bytes;
}

size_t rte_lw_scratch_bytes(int ncol, int nlay, int ng) {
  (void)ng;
  if (nlay <= kMaxRegisterLayers) return 0;
  long tiles = ((long)ncol + kOverCW - 1) / kOverCW;
  if (tiles > kOverWaves) tiles = kOverWaves;
  return sizeof(double) * 2 * (size_t)(nlay - kMaxRegisterLayers) * 64 * (size_t)tiles;
}

hipError_t launch_rte_lw(const RteLwArgs &a, hipStream_t s) {
  if (a.ncol <= 0) return hipSuccess;
  if (a.use_split && rte_lw_split_applies(a)) return launch_rte_lw_split(a, s);
  if (a.shared_levels && !a.f32) return launch_real<double, true>(a, s);   // (no single-precision entry point sets it)
  return a.f32 ? launch_real<float, false>(a, s) : launch_real<double, false>(a, s);
}

}  // namespace ecckd
