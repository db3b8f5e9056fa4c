// kernels_rte_sw_sys.hip -- shortwave two-stream + adding solver, "layer-systolic" form: the two-stream
// coefficients of a cell are computed ONCE and stay in registers between the two sweeps.
//
// Replaces RTE-RRTMGP's rte_sw as the reference calls it (example/rfmip-rad-irf/ecckd_rfmip_sw.F90:148-154):
// sw_two_stream, sw_source_2str, adding, flux_dn = diffuse + direct, ty_fluxes_broadband%reduce
// [RTE-ext: SURVEY.md Appendix B.2].  Same arithmetic per (column, g-point) as kernels_rte_sw.hip -- same
// expressions in the same order, shared sw_two_stream.hpp -- so a (column, g-point) pair gets the same bits from
// either kernel; the g-point sum is taken in g-point order 1..ngpt here (what sum_broadband does).
//
// Why another form.  The adding method needs the coefficients of every layer twice, bottom -> top (albedo and
// source of the stack below) and top -> bottom (fluxes).  kernels_rte_sw.hip gives a lane one (column, g-point) and
// all its layers: nothing on the chip holds 60 layers x 5 coefficients per lane, so its second pass reads tau / ssa / g
// again and recomputes the two exp, the sqrt and the divisions (373 VALU instructions per cell, 80 B/cell of traffic).
// Here the LAYERS of a column are spread over the waves of a block instead:
//   block = kSysWaves waves, tile = 64 columns (lane = column), one g-point at a time;
//   wave w owns layers [w*LPW, (w+1)*LPW) counted from the top, for all 64 columns;
//   P  every wave computes the coefficients of its own 5 x 64 cells (independent: all lanes of all waves busy);
//   U  the adding recurrence climbs through the waves, bottom wave first: a wave takes (albedo, source) of the stack
//      below it from the wave below (LDS hand-off + flag), folds its layers in and hands on upwards;
//   D  the flux recurrence comes back down the same way; every wave adds the fluxes at the levels it owns to its
//      own rows of the block's LDS accumulators.
// 5 layers x 6 values per lane = 60 VGPRs carry everything from P to D: tau, ssa, g are read once (24 B/cell), no
// scratch ring, one two-stream evaluation per cell.  The sweeps are serial chains in which one wave after the other is
// active -- the price of keeping the reference's operation order instead of composing layer segments algebraically --
// and they overlap with the P work of the waves that are done: a wave computes the coefficients of the NEXT g-point
// as soon as its part of D is over (its registers are free again), so only the bottom wave's P sits on the
// critical path.  Waves spin on LDS flags (bounded; s_sleep) -- point-to-point, no block barrier in the loop.
#include <type_traits>

#include "kernels.hpp"
#include "sw_two_stream.hpp"

namespace ecckd {
namespace {

#ifndef ECCKD_SYS_WAVES
#define ECCKD_SYS_WAVES 12
#endif
#ifndef ECCKD_SYS_LPW
#define ECCKD_SYS_LPW 5
#endif
constexpr int kSysWaves = ECCKD_SYS_WAVES;   // waves per block: three per SIMD (<= 168 VGPRs), one block per CU
constexpr int kSysLPW = ECCKD_SYS_LPW;       // layers per wave
constexpr int kSysMaxLay = kSysWaves * kSysLPW;
#ifndef ECCKD_SYS_FMA_CHAIN
#define ECCKD_SYS_FMA_CHAIN 1
#endif
constexpr bool kSysFmaChain = ECCKD_SYS_FMA_CHAIN != 0;
// Fast arithmetic mode, what a wave does while it holds the token (see the sweeps below):
//   ECCKD_SYS_PROJ  U: the pair handed upwards comes from a division-free (projective) form of the adding recurrence -- two
//                   dependent operations per layer instead of six -- and the wave's own per-layer values, which only its D
//                   needs, are computed after the token has moved on;
//   ECCKD_SYS_DPRE  D: everything that does not depend on the incoming pair is multiplied out before the token arrives
//                   (direct-beam transmittances as prefix products): one dependent FMA per layer on the way to the hand-off.
//                   Worth 6 % while the sweeps set the pace; with the one-round hand-off, the polling priority and the
//                   leaner coefficients the kernel runs within 4 % of its instruction stream without the waits, and the three
//                   extra instructions per cell cost more than the shorter D sweep returns (1.57 -> 1.54 ms without it,
//                   profiles/r03_ab_sw20.txt, r03_ab_sw21.txt): off by default, kept as a build option.
#ifndef ECCKD_SYS_PROJ
#define ECCKD_SYS_PROJ 1
#endif
#ifndef ECCKD_SYS_DPRE
#define ECCKD_SYS_DPRE 0
#endif
constexpr int kSysSpinLimit = 1 << 22;   // polls of a flag before the block gives up (a lost hand-off never hangs the GPU)

#ifdef ECCKD_SYS_TIMING   // (variant build only: s_memtime stamps of one g-point step of block 0, read back by tools/sys_timing.py)
__device__ long long g_sys_times[kSysWaves][8];
#define SYS_STAMP(i) do { if (blockIdx.x == 0 && unit == blockIdx.x + gridDim.x && g == g_begin + 10 && lane == 0) g_sys_times[w][i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define SYS_STAMP(i) do {} while (0)
#endif

struct SysLds {
  int flag_u[kSysWaves];   // sequence number of the (albedo, source) pair waiting in hand-off slot w
  int flag_d[kSysWaves];   // ... of the (fdn, fdir) pair
  int abort_;              // set when a wait ran into kSysSpinLimit: every later wait returns at once, the fluxes become NaN
  int pad_;
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for_sys(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_sys<I + 1, N>(f);
  }
}

// Hand-off between neighbouring waves of a block: the pair of values of every lane in an LDS slot, then a flag word
// with the sequence number of the step (release); the consumer polls the flag -- one broadcast dword per poll -- and
// reads its pair after it (acquire).  Measured in round 3 (same box, 1e5 columns, kernel times): this form 1.75 ms;
// no flag, the slot's own "empty" NaN mark polled instead (one LDS round trip less, but every poll moves 1 KiB per wave
// and eleven waves poll) 1.84; that with long naps ended by the producer's s_wakeup 1.87; s_sleep 0 / 1 / 2 between
// the polls: no difference.
#ifndef ECCKD_SYS_SLEEP
#define ECCKD_SYS_SLEEP 2   // x 64 clocks
#endif
#ifndef ECCKD_SYS_LIGHT_FENCES
#define ECCKD_SYS_LIGHT_FENCES 0
#endif
// Waits until *flag == seq (set by another wave of this block with publish()).  Wave-uniform: every lane reads the same
// word and the comparison is scalar.
__device__ __forceinline__ void wait_flag(int *flag, int seq, int *abort_) {
#ifdef ECCKD_SYS_DEBUG_NOWAIT   // (timing experiments only: the work of the sweeps without their serial dependence)
  return;
#endif
  int spins = 0;
#if ECCKD_SYS_LIGHT_FENCES
  // (LDS executes the DS instructions of a wave in order: the producer's flag store cannot overtake its data stores and
  // this wave's data reads cannot overtake the flag read whose value it has waited for -- compiler fences are enough)
  while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != seq) {
#else
  while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) != seq) {
#endif
    __builtin_amdgcn_s_sleep(ECCKD_SYS_SLEEP);
    if ((++spins & 63) == 0) {
      if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) return;
      if (spins > kSysSpinLimit) {
        __hip_atomic_store(abort_, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
      }
    }
  }
#if ECCKD_SYS_LIGHT_FENCES
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
__device__ __forceinline__ void publish(int *flag, int seq) {
#if ECCKD_SYS_LIGHT_FENCES
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}

// The token of a sweep: the pair of values of every lane in the receiving wave's LDS slot (one 16-byte word per lane in
// fp64), then the slot's flag word with the sequence number of the step.  ECCKD_SYS_HANDOFF:
//   0  (until late in round 3) release store of the flag behind an s_waitcnt on the data; the receiver polls the flag
//      with a nap between polls and reads its pair afterwards: three LDS round trips between "the sender is done" and
//      "the receiver computes" -- measured ~500 clocks per hand-off, 24 hand-offs per g-point step, half the step;
//   1  the LDS executes the DS instructions of a wave in order, so (i) the sender issues data and flag back to back and
//      (ii) the receiver asks for flag AND data in one round (flag first): a flag that reads `seq` vouches for the data read
//      behind it.  So that eleven waiting waves do not crowd the LDS queue with 1 KiB polls (the sentinel experiment of
//      §5.4), a wave first naps on the flag of the wave the token comes FROM -- only the next wave in line polls its slot.
#ifndef ECCKD_SYS_HANDOFF
#define ECCKD_SYS_HANDOFF 1
#endif
#ifndef ECCKD_SYS_POLL_PRIO
#define ECCKD_SYS_POLL_PRIO 3
#endif
#ifndef ECCKD_SYS_SLEEP2
#define ECCKD_SYS_SLEEP2 0   // nap between the polls of the next wave in line (x 64 clocks; 0: none)
#endif
template <typename real> struct SysPair { typedef real type __attribute__((ext_vector_type(2))); };

// near: flag word of the wave the token comes from (nullptr: that wave starts the sweep), flag / slot: this wave's own.
template <typename real>
__device__ __forceinline__ typename SysPair<real>::type take_token(int *near, int *flag, const real *slot, int seq, int *abort_) {
  typedef typename SysPair<real>::type pair_t;
  typedef __attribute__((address_space(3))) const volatile pair_t lds_pair;
  typedef __attribute__((address_space(3))) const volatile int lds_int;
#if ECCKD_SYS_HANDOFF == 0
  (void)near;
  wait_flag(flag, seq, abort_);
  return *(lds_pair *)slot;
#else
#ifdef ECCKD_SYS_DEBUG_NOWAIT
  return *(lds_pair *)slot;
#endif
  if (near) wait_flag(near, seq, abort_);
#if ECCKD_SYS_POLL_PRIO && !defined(ECCKD_SYS_NOPRIO)
  // the next wave in line polls at the token holder's priority: its poll is not queued behind the coefficient arithmetic
  // of the other waves of its SIMD (-3 %; profiles/r03_ab_sw17.txt, r03_ab_sw18.txt)
  __builtin_amdgcn_s_setprio(ECCKD_SYS_POLL_PRIO);
#endif
  int spins = 0;
  pair_t v;
  for (;;) {
    const int f = *(lds_int *)flag;   // (volatile: the two reads stay in this order, one s_waitcnt behind both)
    v = *(lds_pair *)slot;
    if (__builtin_amdgcn_readfirstlane(f) == seq) break;
#if ECCKD_SYS_SLEEP2
    __builtin_amdgcn_s_sleep(ECCKD_SYS_SLEEP2);
#endif
    if ((++spins & 255) == 0) {
      if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) break;
      if (spins > kSysSpinLimit) {
        __hip_atomic_store(abort_, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        break;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return v;
#endif
}
template <typename real>
__device__ __forceinline__ void give_token(int *flag, real *slot, real a, real b, int seq) {
  typedef typename SysPair<real>::type pair_t;
  typedef __attribute__((address_space(3))) volatile pair_t lds_pair;
  typedef __attribute__((address_space(3))) volatile int lds_int;
  pair_t v;
  v[0] = a; v[1] = b;
  *(lds_pair *)slot = v;
#if ECCKD_SYS_HANDOFF == 0
  publish(flag, seq);
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  *(lds_int *)flag = seq;
#endif
}

// 1/x for the adding recurrence of the fast arithmetic mode: rcp<true> of sw_two_stream.hpp (hardware reciprocal and one
// third-order step: three dependent operations behind the v_rcp).
__device__ __forceinline__ double rcp_chain(double x) { return rcp<true>(x); }
__device__ __forceinline__ float rcp_chain(float x) { return rcp<true>(x); }

// real: storage and arithmetic type.  FAST / CLAMP: as rte_sw_kernel.  DERIVE: fused shortwave path (RteSwArgs::derive).
// FULL: every wave that owns layers owns LPW of them (nlay a multiple of LPW): no per-layer branches.
template <typename real, bool FAST, bool CLAMP, bool DERIVE, bool FULL>
__global__ void __launch_bounds__(64 * kSysWaves) rte_sw_sys_kernel(const RteSwArgs a) {
  constexpr int LPW = kSysLPW, NW = kSysWaves;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index: uniform, and the compiler has to know it
  const int ncol = a.ncol, nlay = a.nlay, ng = a.ng, nlev = nlay + 1;
  // LDS: flags | band of every g-point | hand-off slots [NW][4][64] real | accumulators [3][nlev][64] double (up, dn, dir)
  SysLds *ctl = reinterpret_cast<SysLds *>(lds_raw);
  unsigned char *bandmap = lds_raw + 128;   // gpt2band: an LDS byte instead of a load from the argument block per g-point
  real *hand = reinterpret_cast<real *>(lds_raw + 384);
  double *acc = reinterpret_cast<double *>(lds_raw + 384 + sizeof(real) * NW * 4 * 64);
  double *acc_up = acc, *acc_dn = acc + (long)nlev * 64, *acc_dir = acc + 2L * nlev * 64;
  auto P = [](const double *p) { return reinterpret_cast<const real *>(p); };
  auto Q = [](double *p) { return reinterpret_cast<real *>(p); };

  const int nwa = (nlay + LPW - 1) / LPW;          // waves that own layers
  const int s0 = w * LPW;                           // first layer of this wave, counted from the top
  const int nl = nlay - s0 < 0 ? 0 : (nlay - s0 < LPW ? nlay - s0 : LPW);
  if (threadIdx.x < 2 * NW + 2) reinterpret_cast<int *>(ctl)[threadIdx.x] = 0;
  if (threadIdx.x < 256) bandmap[threadIdx.x] = a.gpt2band[threadIdx.x];
  __syncthreads();
  if (w >= nwa) return;                             // (no barrier below this line)
  const bool top_wave = w == 0, bottom_wave = w == nwa - 1;
  const long lay0 = a.top_at_1 ? 0 : nlay - 1, lev0 = a.top_at_1 ? 0 : nlay, lstep = a.top_at_1 ? 1 : -1;
  // hand-off slots: [wave][U, D][lane] pairs
  real *slot_u = hand + (((long)w * 2 + 0) * 64 + lane) * 2, *slot_d = hand + (((long)w * 2 + 1) * 64 + lane) * 2;
  real *slot_u_above = slot_u - 4 * 64, *slot_d_below = slot_d + 4 * 64;
  const real k_floor = (real)a.k_floor;
  const real gw = (real)a.gw;
  double *my_up = acc_up + s0 * 64 + lane, *my_dn = acc_dn + s0 * 64 + lane, *my_dir = acc_dir + s0 * 64 + lane;

  const long ntiles = ((long)ncol + 63) / 64;
  const long tail_first = a.sys_tail_first < 0 ? ntiles : a.sys_tail_first;
  const int gchunk = a.sys_gchunk > 0 ? a.sys_gchunk : ng;
  const int nchunks = (ng + gchunk - 1) / gchunk;
  const long nunits = tail_first + (ntiles - tail_first) * nchunks;

  int seq = 0;   // hand-off sequence number: one per (unit, g-point), the same in every wave of the block
  for (long unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
    long tile = unit;
    int g_begin = 0, g_end = ng;
    if (unit >= tail_first) {
      const long u = unit - tail_first;
      tile = tail_first + u / nchunks;
      g_begin = (int)(u - (tile - tail_first) * nchunks) * gchunk;
      g_end = g_begin + gchunk < ng ? g_begin + gchunk : ng;
    }
    const long col = tile * 64 + lane;
    const bool valid = col < ncol;
    const long cc = valid ? col : (long)ncol - 1;
    // the rows of the accumulators this wave owns: the level below each of its layers; the top wave also level 0
#pragma unroll
    for (int l = 0; l <= LPW; ++l) {
      const int lev = l == LPW ? 0 : s0 + l + 1;
      if (l == LPW ? top_wave : l < nl) {
        acc_up[lev * 64 + lane] = 0.;
        acc_dn[lev * 64 + lane] = 0.;
        if (a.flux_dir) acc_dir[lev * 64 + lane] = 0.;
      }
    }
    const real mu0 = P(a.mu0)[cc];
    const real mu0_inv = real(1) / mu0;
    real moles[LPW];   // DERIVE: (plev(l+1) - plev(l)) * gw, src/gas_optics_ecckd.f90:313-314
    if (DERIVE) {
#pragma unroll
      for (int l = 0; l < LPW; ++l) {
        moles[l] = real(0);
        if (l < nl) {
          const long lm = lay0 + lstep * (s0 + l);   // layer index in memory
          moles[l] = (P(a.plev)[cc + (long)ncol * (lm + 1)] - P(a.plev)[cc + (long)ncol * lm]) * gw;   // :313-314
        }
      }
    }

    // optical properties of this wave's cells at one g-point, and the boundary values of the column that the bottom
    // wave (surface albedos) and the top wave (incoming beam) feed into the sweeps.  All of it is requested one g-point
    // ahead, in one round of global loads; nothing in the sweeps below touches global memory (a global load behind a
    // select of a hand-off slot would become a flat load behind s_waitcnt vmcnt(0) -- on the critical path).
    // (addresses: a wave-uniform row pointer, recomputed with scalar instructions where it is used, plus ONE 32-bit
    // per-lane byte offset -- per-layer vector addresses kept across the g-point loop spill, and every reload of a
    // spilled address waits for the loads already in flight)
    typedef __attribute__((address_space(1))) const char gcchar_t;
    typedef __attribute__((address_space(1))) const real greal_t;
    const unsigned co = (unsigned)cc * (unsigned)sizeof(real);
    auto at = [&](const real *row) -> real { return *reinterpret_cast<greal_t *>((gcchar_t *)row + co); };
    real ptau[LPW], pssa[LPW], pg[LPW], pb0 = real(0), pb1 = real(0), ptoa = real(0);
    // boundary values of the column at one g-point: surface albedos (bottom wave), incoming beam (top wave).  Each is
    // requested for g + 1 right after the sweep of g has read it -- NOT together with the optical properties, which the
    // parking waves request a g-point further ahead.
    auto load_albedos = [&](int g) {
      const int band = bandmap[g];
      pb0 = P(a.alb_dif)[band + (long)a.nband * cc];
      pb1 = P(a.alb_dir)[band + (long)a.nband * cc];   // src_sfc = F_dir(sfc) * sfc_alb_dir
    };
    const real tscale = (DERIVE && a.toa_scale) ? P(a.toa_scale)[cc] : real(1);
    auto load_toa = [&](int g) { ptoa = DERIVE ? P(a.solar)[g] * tscale : at(P(a.toa) + (long)ncol * g); };
    auto load_props = [&](int g) {
      // (no select on a value just requested -- it would wait for the load on the spot: a layer this wave does not
      // own keeps whatever its registers hold, nothing reads them)
#pragma unroll
      for (int l = 0; l < LPW; ++l) {
        if (l < nl) {
          const long row = (long)ncol * ((lay0 + lstep * (s0 + l)) + (long)nlay * g);
          ptau[l] = at(P(a.tau) + row);
          if (!DERIVE) { pssa[l] = at(P(a.ssa) + row); pg[l] = at(P(a.g) + row); }
        }
      }
    };
    if (bottom_wave) load_albedos(g_begin);
    if (top_wave) load_toa(g_begin);
    load_props(g_begin);

    // Coefficients of this wave's cells at the g-point whose optical properties are in ptau / pssa / pg: results
    // through put(l, Rdif, Tdif, Rdir, Tdir, Tnoscat).  Two call sites: waves that keep them in registers (after D), and
    // the lower waves of the column, which compute them a g-point ahead while the token is away and park them in LDS.
    auto coefficients = [&](int g, auto &&put) __attribute__((always_inline)) {
      real ray = real(0);
      if (DERIVE) ray = P(a.rayleigh)[g];
#pragma unroll
      for (int l = 0; l < LPW; ++l) {
        if (FULL || l < nl) {
          real cssa, cg;
          if (DERIVE) { cssa = (moles[l] * ray) / ptau[l]; cg = real(0); }   // :316, :459-460
          else { cssa = pssa[l]; cg = pg[l]; }
          const TwoStreamT<real> ts = (DERIVE || __all(cg == real(0)))
                                          ? two_stream<real, FAST, CLAMP, true>(ptau[l], cssa, cg, mu0, mu0_inv, k_floor)
                                          : two_stream<real, FAST, CLAMP, false>(ptau[l], cssa, cg, mu0, mu0_inv, k_floor);
          put(l, ts);
        }
        // one cell after the other: left alone, the scheduler interleaves the five independent evaluations and their
        // temporaries no longer fit the register file (hundreds of spills)
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    real *park = nullptr;   // [LPW][5][64] of this wave, if it parks
    const bool parking = w >= nwa - a.sys_npark;
    if (parking) park = reinterpret_cast<real *>(lds_raw + a.sys_park_at) + (long)(w - (nwa - a.sys_npark)) * LPW * 5 * 64 + lane;

    // (a parking wave runs one half-step ahead: the iteration before its first g-point only computes and parks)
    for (int g = g_begin - 1; g < g_end; ++g) {
      const bool step = g >= g_begin;
      // st: P leaves Rdif, Tdif, Rdir, Tdir, Tnoscat in slots 0..4; U turns them into what D needs:
      // B = Rdif*denom, A = Tdif*denom, albedo below the layer, C = Tdir*denom, Tnoscat, normalised source below
      real st[LPW][6];
      real albedo = real(0), nsrc = real(0);
      if (step) {
        ++seq;
        SYS_STAMP(0);
        // ---- P ----
        if (parking) {
#pragma unroll
          for (int l = 0; l < LPW; ++l)
            if (FULL || l < nl)
#pragma unroll
              for (int v = 0; v < 5; ++v) st[l][v] = park[(l * 5 + v) * 64];
        } else {
          coefficients(g, [&](int l, const TwoStreamT<real> &ts) __attribute__((always_inline)) {
            st[l][0] = ts.Rdif; st[l][1] = ts.Tdif; st[l][2] = ts.Rdir; st[l][3] = ts.Tdir; st[l][4] = ts.Tnoscat;
          });
        }
        SYS_STAMP(1);
      }
      if (step) {
        if (!parking && g + 1 < g_end) load_props(g + 1);   // in flight during the sweeps

        // ---- U: adding, bottom -> top.  The source is carried normalised by the direct beam at its own level (the
        // beam is only known on the way down): src(l) = nsrc(l) * F_dir(l), F_dir(l+1) = Tnoscat(l) * F_dir(l) ----
        SYS_STAMP(2);
        typename SysPair<real>::type tok;
        tok[0] = pb0; tok[1] = pb1;
        if (!bottom_wave) tok = take_token<real>(w + 1 < nwa - 1 ? &ctl->flag_u[w + 1] : nullptr, &ctl->flag_u[w], slot_u, seq, &ctl->abort_);
        SYS_STAMP(3);
#ifndef ECCKD_SYS_NOPRIO
        // the sweeps are the critical path of the block: the wave that holds the token issues ahead of the waves of
        // its SIMD that are still computing coefficients
        __builtin_amdgcn_s_setprio(3);
#endif
        albedo = tok[0];
        nsrc = tok[1];
        if (bottom_wave && g + 1 < g_end) load_albedos(g + 1);
        constexpr bool kProj = FAST && kSysFmaChain && ECCKD_SYS_PROJ != 0;
        constexpr bool kDpre = FAST && kSysFmaChain && ECCKD_SYS_DPRE != 0;
        if constexpr (kProj) {
          // The pair for the wave above, ahead of everything else.  With albedo = p/q and nsrc = s/q the recurrence
          //   albedo' = Rdif + Tdif^2 albedo / (1 - Rdif albedo),  nsrc' = Rdir + Tdif (nsrc Tn + albedo Tdir) / (1 - Rdif albedo)
          // is linear in (p, q, s):  q' = q - Rdif p,  p' = Rdif q' + Tdif^2 p,  s' = Rdir q' + Tdif Tn s + Tdif Tdir p,
          // two dependent operations per layer (p -> q' -> p'; s trails by a constant) and ONE reciprocal per wave instead
          // of one per layer.  q' / q = 1 - Rdif albedo lies in (0, 1]: five layers cannot underflow, and the pair is
          // handed on normalised (q = 1).  The values differ from the per-layer form below in the last bits only.
          if (!top_wave) {
            real p = albedo, q = real(1), sN = nsrc;
#pragma unroll
            for (int l = LPW - 1; l >= 0; --l) {
              if (FULL || l < nl) {
                const real Rdif = st[l][0], Tdif = st[l][1], Rdir = st[l][2], Tdir = st[l][3], Tn = st[l][4];
                const real qn = fma(-Rdif, p, q);
                sN = fma(Tdif * Tn, sN, fma(Rdir, qn, Tdif * (Tdir * p)));
                p = fma(Rdif, qn, (Tdif * Tdif) * p);
                q = qn;
              }
            }
            const real rq = rcp_chain(q);
            give_token<real>(&ctl->flag_u[w - 1], slot_u_above, p * rq, sN * rq, seq);
            SYS_STAMP(7);
#ifndef ECCKD_SYS_NOPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
          }
        }
        real dn[LPW];
#pragma unroll
        for (int l = LPW - 1; l >= 0; --l) {
          if (FULL || l < nl) {
            const real Rdif = st[l][0], Tdif = st[l][1], Rdir = st[l][2], Tdir = st[l][3], Tn = st[l][4];
            // adding, Eq 10 / 11 / 9.  Eq 11 is divided by F_dir(l): src_up = Rdir*F_dir(l), src_dn = Tdir*F_dir(l),
            // src(l+1) = nsrc*Tnoscat*F_dir(l).  Fast arithmetic mode (kSysFmaChain): the multiply-add pairs of the recurrence
            // are FMAs -- the wave that holds the token issues one fp64 instruction every ~10 clocks, and every instruction
            // less shortens the critical path of the block: -5.5 % on the kernel, fluxes unchanged to 1e-13 W m-2.  The
            // reference-order mode rounds every operation on its own, as kernels_rte_sw.hip does.
            real denom, A;
            if constexpr (FAST && kSysFmaChain) {
              denom = rcp_chain(fma(-Rdif, albedo, real(1)));
              A = Tdif * denom;
              dn[l] = denom;
              st[l][2] = albedo;
              st[l][5] = nsrc;
              nsrc = fma(A, fma(nsrc, Tn, albedo * Tdir), Rdir);
              albedo = fma(Tdif * Tdif * albedo, denom, Rdif);
            } else {
              denom = rcp<FAST>(real(1) - Rdif * albedo);
              A = Tdif * denom;
              dn[l] = denom;   // (B and C wait until the token has been handed on)
              st[l][2] = albedo;
              st[l][5] = nsrc;
              nsrc = Rdir + A * (nsrc * Tn + albedo * Tdir);
              albedo = Rdif + Tdif * Tdif * albedo * denom;
            }
            st[l][1] = A;
          }
        }
        if (!kProj && !top_wave) {
          give_token<real>(&ctl->flag_u[w - 1], slot_u_above, albedo, nsrc, seq);
#ifndef ECCKD_SYS_NOPRIO
          __builtin_amdgcn_s_setprio(0);   // (waiting for the token to come back down)
#endif
        }
#ifndef ECCKD_SYS_NOPRIO
        if (kProj && top_wave) __builtin_amdgcn_s_setprio(0);
#endif
        if constexpr (kDpre) {
          // D-form of a layer: Y, A, albedo below, P, -, normalised source below, with P(l) = product of the direct-beam
          // transmittances of this wave's layers above and including l, and Y = B nsrc_below P(l) + C P(l-1): the diffuse
          // flux below layer l is A fdn + Y fdir_in, the beam P(l) fdir_in -- fdir_in the beam that arrives with the token.
          real Pacc = real(1);
#pragma unroll
          for (int l = 0; l < LPW; ++l) {
            if (FULL || l < nl) {
              const real B = st[l][0] * dn[l], C = st[l][3] * dn[l];
              const real Pn = Pacc * st[l][4];
              st[l][0] = fma(B * st[l][5], Pn, C * Pacc);
              st[l][3] = Pn;
              Pacc = Pn;
            }
          }
        } else {
#pragma unroll
          for (int l = 0; l < LPW; ++l) {
            if (FULL || l < nl) { st[l][0] = st[l][0] * dn[l]; st[l][3] = st[l][3] * dn[l]; }
          }
        }
      }
      // ---- the lower waves: coefficients of the next g-point while the token is away ----
      if (parking && g + 1 < g_end) {
        coefficients(g + 1, [&](int l, const TwoStreamT<real> &ts) __attribute__((always_inline)) {
          park[(l * 5 + 0) * 64] = ts.Rdif; park[(l * 5 + 1) * 64] = ts.Tdif; park[(l * 5 + 2) * 64] = ts.Rdir;
          park[(l * 5 + 3) * 64] = ts.Tdir; park[(l * 5 + 4) * 64] = ts.Tnoscat;
        });
        if (g + 2 < g_end) load_props(g + 2);
      }
      if (step) {
        // ---- D: direct beam and fluxes, top -> bottom (Eq 12, 13) ----
        SYS_STAMP(4);
        typename SysPair<real>::type tokd;
        tokd[0] = real(0); tokd[1] = real(0);
        if (!top_wave) tokd = take_token<real>(w > 1 ? &ctl->flag_d[w - 1] : nullptr, &ctl->flag_d[w], slot_d, seq, &ctl->abort_);
        const real h_fdn = tokd[0], h_dir = tokd[1];
        SYS_STAMP(5);
#ifndef ECCKD_SYS_NOPRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        real fdir = top_wave ? ptoa * mu0 : h_dir;
        real fdn = top_wave ? real(0) : h_fdn;
        if (top_wave && g + 1 < g_end) load_toa(g + 1);
        const real fup0 = fdn * albedo + nsrc * fdir, fdn0 = fdn + fdir, fdir0 = fdir;   // level 0 (top wave)
        real fu[LPW], fd[LPW], fr[LPW];
        constexpr bool kDpreD = FAST && kSysFmaChain && ECCKD_SYS_DPRE != 0;
        if constexpr (kDpreD) {
          const real fdir_in = fdir;
          real Plast = real(1);
#pragma unroll
          for (int l = 0; l < LPW; ++l) {
            if (FULL || l < nl) {
              fdn = fma(st[l][1], fdn, st[l][0] * fdir_in);   // Eq 12: the one dependent operation per layer
              fd[l] = fdn;
              Plast = st[l][3];
            }
          }
          if (!bottom_wave) {
            give_token<real>(&ctl->flag_d[w + 1], slot_d_below, fdn, Plast * fdir_in, seq);
          }
#ifndef ECCKD_SYS_NOPRIO
          __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
          for (int l = 0; l < LPW; ++l) {
            if (FULL || l < nl) {
              const real fdir_next = st[l][3] * fdir_in;
              fu[l] = fma(fd[l], st[l][2], st[l][5] * fdir_next);   // Eq 13
              fd[l] = fd[l] + fdir_next;
              fr[l] = fdir_next;
            }
          }
        }
#pragma unroll
        for (int l = 0; l < LPW; ++l) {
          if (!kDpreD && (FULL || l < nl)) {
            const real B = st[l][0], A = st[l][1], alb_next = st[l][2], C = st[l][3], Tn = st[l][4], nsrc_next = st[l][5];
            const real fdir_next = Tn * fdir;
            const real src_next = nsrc_next * fdir_next;
            if constexpr (FAST && kSysFmaChain) {
              // Eq 12 with denom multiplied in, associated so that the recurrence of the diffuse flux is ONE dependent
              // operation per layer (the direct-beam terms run ahead of it): a dependent fp64 operation costs ~20 clocks
              fdn = fma(A, fdn, fma(B, src_next, C * fdir));
              fu[l] = fma(fdn, alb_next, src_next);                          // Eq 13
            } else {
              fdn = A * fdn + B * src_next + C * fdir;
              fu[l] = fdn * alb_next + src_next;
            }
            fdir = fdir_next;
            fd[l] = fdn + fdir;
            fr[l] = fdir;
          }
        }
        if (!kDpreD && !bottom_wave) {
          give_token<real>(&ctl->flag_d[w + 1], slot_d_below, fdn, fdir, seq);
        }
#ifndef ECCKD_SYS_NOPRIO
        if (!kDpreD) __builtin_amdgcn_s_setprio(0);
#endif
        // the token has moved on: now the sums (one fire-and-forget ds_add_f64 each; every lane owns its words)
        if (top_wave) {
          __hip_atomic_fetch_add(&acc_up[lane], (double)fup0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          __hip_atomic_fetch_add(&acc_dn[lane], (double)fdn0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          if (a.flux_dir) __hip_atomic_fetch_add(&acc_dir[lane], (double)fdir0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
#pragma unroll
        for (int l = 0; l < LPW; ++l) {
          if (FULL || l < nl) {   // (one base address per array, the layer is an immediate offset)
            __hip_atomic_fetch_add(&my_up[(l + 1) * 64], (double)fu[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(&my_dn[(l + 1) * 64], (double)fd[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (a.flux_dir) __hip_atomic_fetch_add(&my_dir[(l + 1) * 64], (double)fr[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          }
        }
        SYS_STAMP(6);
      }
    }

    // ---- the levels this wave owns: broadband fluxes of the tile, or the chunk's partial sums ----
    const bool failed = __hip_atomic_load(&ctl->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
    const double poison = failed ? __builtin_nan("") : 0.;
#pragma unroll
    for (int l = 0; l <= LPW; ++l) {
      const int lev = l == LPW ? 0 : s0 + l + 1;
      if (l == LPW ? top_wave : l < nl) {
        const double vu = acc_up[lev * 64 + lane] + poison, vd = acc_dn[lev * 64 + lane] + poison;
        const double vr = a.flux_dir ? acc_dir[lev * 64 + lane] + poison : 0.;
        if (unit >= tail_first) {   // [unit][up, dn, dir][nlev][64]
          double *pp = a.partials + (unit - tail_first) * 3 * nlev * 64;
          pp[lev * 64 + lane] = vu;
          pp[(nlev + lev) * 64 + lane] = vd;
          pp[(2 * nlev + lev) * 64 + lane] = vr;
        } else if (valid) {
          const long q = col + (long)ncol * (lev0 + lstep * lev);
          Q(a.flux_up)[q] = (real)vu;
          Q(a.flux_dn)[q] = (real)vd;
          if (a.flux_dir) Q(a.flux_dir)[q] = (real)vr;
        }
      }
    }
  }
}

// LDS of a block: control words and band map (384 B), hand-off slots, accumulators (up, dn and, if asked for, dir), and
// behind them the parking areas: as many of the lower waves as fit keep the coefficients of the next g-point there.
size_t sys_lds_base(int nlay, int f32, bool with_dir) {
  return 384 + (f32 ? sizeof(float) : sizeof(double)) * kSysWaves * 4 * 64 + sizeof(double) * (with_dir ? 3 : 2) * (size_t)(nlay + 1) * 64;
}
size_t sys_park_bytes(int f32) { return (f32 ? sizeof(float) : sizeof(double)) * kSysLPW * 5 * 64; }

template <typename real, bool DERIVE, bool FULL>
hipError_t launch_sys2(RteSwArgs a, long blocks, hipStream_t s) {
  auto k = a.dir_clamp ? (a.exact_division ? rte_sw_sys_kernel<real, false, true, DERIVE, FULL> : rte_sw_sys_kernel<real, true, true, DERIVE, FULL>)
                       : (a.exact_division ? rte_sw_sys_kernel<real, false, false, DERIVE, FULL> : rte_sw_sys_kernel<real, true, false, DERIVE, FULL>);
  const size_t base = (sys_lds_base(a.nlay, a.f32, a.flux_dir != nullptr) + 15) & ~(size_t)15;
  if (base > (size_t)kLdsBudget) return hipErrorInvalidValue;
  const int nwa = (a.nlay + kSysLPW - 1) / kSysLPW;
  long npark = (long)(((size_t)kLdsBudget - base) / sys_park_bytes(a.f32));
#ifdef ECCKD_SYS_NPARK
  if (npark > ECCKD_SYS_NPARK) npark = ECCKD_SYS_NPARK;
#endif
  if (npark > nwa) npark = nwa;
  a.sys_npark = (int)npark;
  a.sys_park_at = (unsigned)base;
  const size_t lds = base + (size_t)npark * sys_park_bytes(a.f32);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * kSysWaves), lds, s, a);
  return hipGetLastError();
}
template <typename real, bool DERIVE>
hipError_t launch_sys(const RteSwArgs &a, long blocks, hipStream_t s) {
  return a.nlay % kSysLPW == 0 ? launch_sys2<real, DERIVE, true>(a, blocks, s) : launch_sys2<real, DERIVE, false>(a, blocks, s);
}

}  // namespace

#ifdef ECCKD_SYS_TIMING
extern "C" int ecckd_debug_sys_times(long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sys_times), sizeof(long long) * kSysWaves * 8);
}
#endif

bool rte_sw_sys_applies(const RteSwArgs &a) { return a.nlay >= 1 && a.nlay <= kSysMaxLay && a.ncol > 0; }

// One block per CU is resident (LDS), so a call is a whole number of rounds of `cus` tiles, and a tile takes the time of
// its serial sweeps however few columns it holds.  The tiles beyond the last full round (all tiles of a small call) are
// handed out ONE g-point per block; each such unit leaves its 3 x (nlay+1) x 64 flux contributions in `partials` and
// rte_sw_tail_reduce adds the g-points of a tile in order -- exactly what a whole-tile block does with its LDS
// accumulators, which start at +0: the fluxes are the same bits whether or not a column falls into a tail tile (the idea of
// rte_lw_tail_plan).  (Chunks of several g-points would need fewer partial sums but re-associate the sum.)  Taken only
// when it shortens the call by 3 % and the partial sums stay below 128 MiB: 1e5 columns = 6 rounds + 27 tiles, 7 x 27
// g-point steps without the split, 6 x 27 + 3 with it.
size_t rte_sw_sys_plan(RteSwArgs &a, int cus) {
  constexpr size_t kTailMaxBytes = (size_t)128 << 20;
  a.sys_tail_first = -1;
  a.sys_gchunk = 0;
  if (a.ncol <= 0 || cus <= 0 || a.ng < 2) return 0;
  const long tiles = ((long)a.ncol + 63) / 64, full = tiles / cus * cus, ntail = tiles - full;
  if (ntail == 0) return 0;
  const double before = (double)(full / cus + 1) * a.ng;
  const double after = (double)(full / cus) * a.ng + (double)((ntail * a.ng + cus - 1) / cus);
  if (before - after < 0.03 * before) return 0;
  const size_t part = sizeof(double) * 3 * (size_t)(a.nlay + 1) * 64 * (size_t)(ntail * a.ng);
  if (part > kTailMaxBytes) return 0;
  a.sys_tail_first = full;
  a.sys_gchunk = 1;
  return part;
}

hipError_t launch_rte_sw_sys(const RteSwArgs &a, int cus, hipStream_t s) {
  if (a.ncol <= 0) return hipSuccess;
  if (!rte_sw_sys_applies(a) || cus <= 0) return hipErrorInvalidValue;
  const long tiles = ((long)a.ncol + 63) / 64;
  const long tail_first = a.sys_tail_first < 0 ? tiles : a.sys_tail_first;
  const int gchunk = a.sys_gchunk > 0 ? a.sys_gchunk : a.ng;
  const int nchunks = (a.ng + gchunk - 1) / gchunk;
  const long units = tail_first + (tiles - tail_first) * nchunks;
  long blocks = units < cus ? units : cus;
  hipError_t e;
  if (a.f32) e = a.derive ? launch_sys<float, true>(a, blocks, s) : launch_sys<float, false>(a, blocks, s);
  else e = a.derive ? launch_sys<double, true>(a, blocks, s) : launch_sys<double, false>(a, blocks, s);
  if (e != hipSuccess || tail_first >= tiles) return e;
  return launch_rte_sw_tail_reduce(a, nchunks, 64, tail_first, tiles - tail_first, s);
}

}  // namespace ecckd
