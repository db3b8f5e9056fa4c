// capi.cpp -- the C ABI of include/ecckd_hip.h: model construction, gas_optics (LW/SW) and the
// RTE solvers, in device-pointer and host-pointer flavours.  Host logic only; the arithmetic
// is in kernels_*.hip.  There is deliberately no CPU fallback: a missing/unusable GPU is an
// error returned to the caller.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ecckd_hip.h"
#include "kernels.hpp"
#include "model.hpp"

namespace {

thread_local std::string g_err;

int fail(const std::string &msg) {
  g_err = msg;
  return 1;
}
}  // namespace

namespace ecckd {
int set_last_error(const std::string &msg) { return fail(msg); }   // for nc_capi.cpp
}  // namespace ecckd

namespace {

#define HIPCHK(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// trim() of a blank- or NUL-padded character(len=32) record
std::string trim_name(const char *p) {
  size_t n = 0;
  while (n < ECCKD_NAME_LEN && p[n] != '\0') ++n;
  while (n > 0 && p[n - 1] == ' ') --n;
  return std::string(p, n);
}

// Grow-only device arena used by the ECCKD_HOST flavours.
struct Arena {
  std::mutex mu;
  void *p = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return 0;
    if (p) { HIPCHK(hipFree(p)); p = nullptr; bytes = 0; }
    HIPCHK(hipMalloc(&p, need));
    bytes = need;
    return 0;
  }
};
Arena g_solver_arena[16];

// Scratch rings of the solvers (rte_sw always; rte_lw beyond 96 layers), stream-ordered: a launch takes the
// block that belongs to (device, stream).  Work on one stream is ordered, so consecutive calls on a stream
// reuse its block without any synchronisation, and calls on different streams never share one -- the call
// stays asynchronous and can be captured in a HIP graph.
// Lifetime rules:
//  * A call holds a ScratchLease from the moment it asks for a block until it has launched its kernels.  A block that
//    has become too small is freed at once only when nothing can still refer to it: no other call holds a lease on this
//    device (another host thread between "got the pointer" and "launched"), the stream has drained, and the block was
//    never handed to a captured call.  Otherwise it is retired and released by ecckd_release_scratch() or at exit.
//  * A block handed out while its stream was being captured belongs to that graph from then on: the next eager call on
//    the stream gets a fresh block (the graph may be replayed on any stream, concurrently with eager calls).
//  * A caller that wants no allocation inside the library at all hands its own buffer over with
//    ecckd_set_stream_scratch(); ecckd_release_scratch() leaves such entries alone.
struct ScratchPool {
  std::mutex mu;
  struct Block { void *p = nullptr; size_t bytes = 0; bool caller_owned = false; bool captured = false; };
  std::map<hipStream_t, Block> live;
  std::vector<void *> retired;
  int leases = 0;             // calls between acquiring a block and having launched on it
};
ScratchPool g_scratch_pool[16];
ScratchPool g_flag_pool[16];   // one small block per (device, stream): the counter of the gas-optics spread probe ("gas_slab_f32" = auto)

struct ScratchLease {
  ScratchPool *pool = nullptr;
  ScratchLease() = default;
  ScratchLease(const ScratchLease &) = delete;
  ScratchLease &operator=(const ScratchLease &) = delete;
  void take(ScratchPool &p) {   // (p.mu held by the caller)
    if (!pool) { pool = &p; ++p.leases; }
  }
  ~ScratchLease() {
    if (!pool) return;
    std::lock_guard<std::mutex> lock(pool->mu);
    --pool->leases;
  }
};

bool stream_is_capturing(hipStream_t stream) {
  hipStreamCaptureStatus c = hipStreamCaptureStatusNone;
  return stream && hipStreamIsCapturing(stream, &c) == hipSuccess && c != hipStreamCaptureStatusNone;
}

// Returns the scratch block of (device, stream) with at least `need` bytes in *out; `lease` keeps it safe from other
// host threads until the caller has launched (it must outlive the launches of the call).
int stream_scratch(int device, hipStream_t stream, size_t need, void **out, ScratchLease &lease, ScratchPool *pools = g_scratch_pool) {
  ScratchPool &pool = pools[device];
  std::lock_guard<std::mutex> lock(pool.mu);
  lease.take(pool);
  ScratchPool::Block &b = pool.live[stream];
  const bool capturing = stream_is_capturing(stream);
  if (b.captured && !capturing && !b.caller_owned) {   // the block belongs to a captured graph: eager calls move on
    pool.retired.push_back(b.p);
    b = ScratchPool::Block{};
  }
  if (b.bytes >= need) {
    if (capturing && !b.caller_owned) b.captured = true;
    *out = b.p;
    return 0;
  }
  if (b.caller_owned)
    return fail("ecckd: the scratch buffer set with ecckd_set_stream_scratch is too small for this call (" +
                std::to_string(need) + " bytes needed: ecckd_rte_lw_scratch_bytes / ecckd_rte_sw_scratch_bytes, or "
                "ecckd_rte_lw_tail_scratch_bytes / ecckd_rte_sw_tail_scratch_bytes with the tail splits)");
  if (capturing)
    return fail("ecckd: this call needs " + std::to_string(need) + " bytes of solver scratch on a stream that is being "
                "captured; run the call once on this stream before the capture, or hand a buffer over with "
                "ecckd_set_stream_scratch (no allocation happens inside a capture)");
  void *p = nullptr;
  size_t want = need + need / 4;   // head room: a slightly larger shape on the same stream does not reallocate
  if (hipMalloc(&p, want) != hipSuccess) {
    (void)hipGetLastError();       // the refused request must not surface as the next launch's error
    want = need;
    HIPCHK(hipMalloc(&p, want));
  }
  if (b.p) {   // the outgrown block: see the lifetime rules above
    if (!b.captured && pool.leases == 1 && hipStreamQuery(stream) == hipSuccess) (void)hipFree(b.p);
    else pool.retired.push_back(b.p);
    (void)hipGetLastError();       // hipStreamQuery's hipErrorNotReady is not an error of this call
  }
  b.p = p; b.bytes = want; b.caller_owned = false; b.captured = false;
  *out = p;
  return 0;
}

// Same block, for a use the call can do without (the tail split of rte_lw): nullptr instead of an error when the block
// cannot be provided (caller-owned block too small, stream being captured, allocation refused).
void *stream_scratch_optional(int device, hipStream_t stream, size_t need, ScratchLease &lease, ScratchPool *pools = g_scratch_pool) {
  {
    ScratchPool &pool = pools[device];
    std::lock_guard<std::mutex> lock(pool.mu);
    const auto it = pool.live.find(stream);
    const bool capturing = stream_is_capturing(stream);
    const bool have = it != pool.live.end() && it->second.bytes >= need;
    if (!have) {
      if (it != pool.live.end() && it->second.caller_owned) return nullptr;
      if (capturing) return nullptr;
    }
  }
  void *p = nullptr;
  if (stream_scratch(device, stream, need, &p, lease, pools)) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

// SIMDs of the device = waves of the register-resident LW solver that run at a time (one wave owns a SIMD)
int simd_slots(int device) {
  static std::atomic<int> cached[16];
  int v = cached[device].load();
  if (v > 0) return v;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) return 0;
  cached[device].store(4 * cus);
  return 4 * cus;
}

size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// Single precision: the *_f32 entry points set this for the duration of the call; every staging
// helper below sizes its copies with esz() and the kernels are launched with f32 = 1.  Data pointers
// keep their `double *` static type on the way through (they are only passed on, never indexed).
thread_local int g_f32 = 0;
thread_local int g_shared_levels = 0;
thread_local const double *g_inc_flux = nullptr;   // set by ecckd_rte_lw_inc_flux around ecckd_rte_lw
// set by the *_byband entry points around the per-band solver calls: band of every g-point of the sub-range
thread_local int g_band_override = -1;
// set by ecckd_sw_fluxes around ecckd_rte_sw: the solver derives ssa / g / toa itself (RteSwArgs::derive)
struct SwDerive { const double *plev, *rayleigh, *solar, *toa_scale; double gw; };
thread_local const SwDerive *g_sw_derive = nullptr;
thread_local double *g_sw_partials = nullptr;   // ... and the room for the solver's partial sums inside the caller's scratch block   // set by ecckd_rte_lw_shared_levels around ecckd_rte_lw
// ecckd_gas_optics_plan(): when set, gas_optical_depth_dev() records its decisions here and launches nothing
struct PlanRecord {
  const int *is_scalar = nullptr;   // per gas of the list: its mole fraction would be passed as one number
  int npass = 0, first_fused = 0, planck_fused = 0;
  ecckd::FusedPlan first;
};
thread_local PlanRecord *g_plan = nullptr;
size_t esz() { return g_f32 ? sizeof(float) : sizeof(double); }
struct F32Scope {
  F32Scope() { g_f32 = 1; }
  ~F32Scope() { g_f32 = 0; }
};

// Arithmetic mode (ecckd_set_arithmetic): 0 = fast (fused kernel, re-associated FMAs),
// 1 = reference order (kernels_tau.hip + kernels_planck.hip, bit-faithful expression order).
std::atomic<int> g_arith{0};

#ifndef ECCKD_LW_DEFAULT_SOLVER
#define ECCKD_LW_DEFAULT_SOLVER 0
#endif
#ifndef ECCKD_GAS_SLAB_F32_DEFAULT
#define ECCKD_GAS_SLAB_F32_DEFAULT 2
#endif
// Version switches of the (un-pinned) RTE-RRTMGP solvers, ecckd_set_solver_option.  Process-wide, read once
// per call; the defaults are the v1.5-era forms the oracle restates.
struct SolverOptions {
  std::atomic<double> lw_tau_thresh{0.};        // <= 0: sqrt(epsilon) of the working precision
  std::atomic<int> lw_series_terms{2};
  std::atomic<int> lw_inc_flux_isotropic{0};
  std::atomic<double> sw_k_floor{1.e-12};
  std::atomic<int> sw_dir_clamp{0};
  // implementation choices (same results to ~1e-16 relative): which fp64 / 60-layer longwave solver runs
  std::atomic<int> lw_solver{ECCKD_LW_DEFAULT_SOLVER};   // 0 register-resident (kernels_rte_lw.hip), 1 layer-split
  std::atomic<int> lw_split_seg{10};
  // ... and whether the call-constant gases of a pass share one slab slot (merge_scalar_gases(); ~1e-16 relative on tau)
  std::atomic<int> gas_merge_scalars{1};
  std::atomic<int> lw_tail_split{1};
  std::atomic<int> sw_tail_split{1};
  std::atomic<int> gas_slab_f32{ECCKD_GAS_SLAB_F32_DEFAULT};   // fp64 gas optics over the float32 image of the tables in LDS
  std::atomic<int> sw_solver{0};   // 0 layer-systolic (kernels_rte_sw_sys.hip; up to 60 layers), 1 per-lane two-pass kernel
};
SolverOptions g_opt;

// ---- optional per-kernel timing with HIP events on the launch stream (ecckd_prof_*) ----
struct ProfRec { const char *name; hipEvent_t start, stop; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof;

struct ProfScope {
  hipStream_t s;
  ProfRec r{};
  bool on;
  ProfScope(const char *name, hipStream_t stream) : s(stream), on(g_prof_on) {
    if (!on) return;
    r.name = name;
    if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(r.start, s);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.stop, s);
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_prof.push_back(r);
  }
};

// Bump allocator over an Arena block.
struct Bump {
  char *base;
  size_t off = 0;
  explicit Bump(void *b) : base(static_cast<char *>(b)) {}
  double *take(size_t nelem) {
    double *r = reinterpret_cast<double *>(base + off);
    off += align256(nelem * esz());
    return r;
  }
};

struct GasDesc {   // gas_desc as it crosses the ABI
  int ngas;
  const char *names;
  const double *const *vmr;
  const long long *cs, *ls;
  const double *scalar;
};

size_t vmr_extent(const GasDesc &gd, int j, int ncol, int nlay) {
  if (!gd.vmr || !gd.vmr[j]) return 0;
  const long long cs = gd.cs ? gd.cs[j] : 0, ls = gd.ls ? gd.ls[j] : 0;
  return (size_t)(1 + (long long)(ncol - 1) * cs + (long long)(nlay - 1) * ls);
}

// Planck side of a longwave call, for the fused kernel.
struct PlanckSide {
  const double *tlev, *tsfc;
  double *lay_source, *lev_inc, *lev_dec, *sfc_source;
};

// gas_optical_depth (src/gas_optics_ecckd.f90:323-376) on device pointers.  `sw` selects the
// gas_optics_ext epilogue (:455-460).  When `pl` is given and the fast arithmetic mode is on, the
// Planck sources (:407-424) are produced by the same launch and *planck_done is set.
int gas_optical_depth_dev(const ecckd_model *m, int ncol, int nlay, const double *plev,
                          const double *tlay, const GasDesc &gd, double *tau, bool sw, double *ssa,
                          double *g, const PlanckSide *pl, bool *planck_done, hipStream_t stream) {
  using namespace ecckd;
  if (planck_done) *planck_done = false;
  std::vector<SeqGas> seq;
  bool first_calc = true;   // :347
  for (int j = 0; j < gd.ngas; ++j) {   // :348
    const std::string name = trim_name(gd.names + (size_t)j * ECCKD_NAME_LEN);
    size_t i = 0;
    for (; i < m->gas.size(); ++i)
      if (m->gas[i].name == name) break;   // :349-357
    if (i >= m->gas.size()) continue;      // :358-364 unknown gas: silently skipped
    const ecckd_model::Gas &t = m->gas[i];
    if (t.composite_only && !first_calc) continue;   // :365-367
    SeqGas e{};
    e.coef = g_f32 ? reinterpret_cast<const double *>(m->dbuf32 + t.dev_off) : m->dbuf + t.dev_off;
    e.vmr = gd.vmr ? gd.vmr[j] : nullptr;
    e.cs = gd.cs ? gd.cs[j] : 0;
    e.ls = gd.ls ? gd.ls[j] : 0;
    e.scalar = gd.scalar ? gd.scalar[j] : 0.;
    if (g_plan) {   // no data in a plan call: an array gas gets a (never dereferenced) non-null marker, a scalar one 1.0
      const bool sc = g_plan->is_scalar && g_plan->is_scalar[j];
      e.vmr = sc ? nullptr : reinterpret_cast<const double *>(sizeof(double));
      e.scalar = 1.;
    }
    e.ref = t.ref;
    e.code = t.code;
    e.nv = t.nv;
    e.clamp = t.has_negative ? 1 : 0;
    if (t.code == ECCKD_LOOK_UP_TABLE) {
      e.mf0 = t.mole_fraction[0];
      e.log_mf0 = std::log(t.mole_fraction[0]);                          // :156
      e.d_log_vmr = std::log(t.mole_fraction[1] / t.mole_fraction[0]);   // :154-155
    }
    seq.push_back(e);
    if (t.composite_only) first_calc = false;   // :371-373
  }

  // Split the sequence into passes holding at most one look_up_table gas and kTauPassGases
  // gases each; later passes start from the tau already stored, so the summation order of :370
  // is preserved exactly in the reference-order mode.
  const bool fast = g_arith.load() == 0;
  size_t pos = 0;
  bool first_pass = true;
  do {
    FusedArgs fa{};
    TauArgs &a = fa.tau;
    a.ncol = ncol; a.nlay = nlay; a.ng = m->ng; a.np = m->np; a.nt = m->nt;
    a.plev = plev; a.tlay = tlay;
    a.temperature = g_f32 ? reinterpret_cast<const double *>(m->dbuf32 + m->off_temperature) : m->dbuf + m->off_temperature;
    a.zero = g_f32 ? reinterpret_cast<const double *>(m->dbuf32 + m->off_zero) : m->dbuf + m->off_zero;
    a.lp0 = m->log_pressure[0];                               // :104
    a.dlp = m->log_pressure[1] - m->log_pressure[0];          // :105
    a.dt = m->temperature[m->np] - m->temperature[0];         // :106 T(1,2)-T(1,1)
    // :107  1./(gravity*0.001*dry_air_molar_mass) with default-real literals (:51-52)
    a.gw = 1. / ((double)9.80665f * (double)0.001f * (double)28.970f);
    if (g_f32) a.gw = (double)(1.f / (9.80665f * 0.001f * 28.970f));   // the same expression with wp = float
    a.lut = -1;
    a.merge_slot = -1;
    a.nmerge = 0;
    a.nbil = 0;
    a.nseq = 0;
    while (pos < seq.size()) {
      SeqGas e = seq[pos];
      if (a.nseq >= kTauPassGases) break;
      if (e.code == ECCKD_LOOK_UP_TABLE) {
        if (a.lut >= 0) break;
        a.lut = a.nseq;
      } else {
        e.slot = a.nbil;
        a.bil_seq[a.nbil++] = a.nseq;
      }
      a.seq[a.nseq++] = e;
      ++pos;
    }
    a.accumulate = first_pass ? 0 : 1;
    a.tau = tau;
    const bool last = pos >= seq.size();
    if (sw && last) {
      a.rayleigh = g_f32 ? reinterpret_cast<const double *>(m->dbuf32 + m->off_rayleigh) : m->dbuf + m->off_rayleigh;
      a.ssa = ssa;
      a.g = g;
    }
    if (fast) {
      fa.mode = (sw && last) ? 2 : 0;
      // (fp64 only: in single precision the two-slot instantiation measured 5 % slower than the seven-slot one it replaces)
      if (g_opt.gas_merge_scalars.load() && !g_f32) merge_scalar_gases(a, g_f32);
      const int nv_lut = a.lut >= 0 ? a.seq[a.lut].nv : 0;
      // Planck sources ride along with the first pass when the table, or a window of it, fits next
      // to >= 3 slab rows
      int pass_clamp = 0;
      for (int k = 0; k < a.nseq; ++k) pass_clamp |= a.seq[k].clamp;
      if (pl && first_pass && !sw &&
          fused_planck_rows(a.ng, a.np, a.nt, a.nbil, nv_lut, m->ntp, pass_clamp, g_f32) > 0) {
        fa.mode = 1;
        fa.ntp = m->ntp;
        fa.planck = g_f32 ? reinterpret_cast<const double *>(m->dbuf32 + m->off_planck) : m->dbuf + m->off_planck;
        fa.pt0 = m->temperature_planck[0];                                  // :272
        fa.pdt = m->temperature_planck[1] - m->temperature_planck[0];      // :271
        fa.tlev = pl->tlev; fa.tsfc = pl->tsfc;
        fa.lay_source = pl->lay_source; fa.lev_source_inc = pl->lev_inc;
        fa.lev_source_dec = pl->lev_dec; fa.sfc_source = pl->sfc_source;
        if (planck_done) *planck_done = true;
      }
      fa.f32 = g_f32;
      // fp64 over the float32 image of the tables in LDS ("gas_slab_f32": 0 never, 1 always, 2 where the probe finds the
      // columns spread over many pressure rows); the probe's counter is a word that belongs to this stream
      ScratchLease flag_lease;
      fa.slab32 = (!g_f32 && !g_plan && m->f32_exact && fa.mode == 1) ? g_opt.gas_slab_f32.load() : 0;
      if (fa.slab32 == 2) fa.choose_buf = static_cast<int *>(stream_scratch_optional(m->device, stream, 256, flag_lease, g_flag_pool));
      if (g_f32 && ((fa.mode != 1 && fa.mode != 2) || !last || !first_pass))
        return fail("ecckd: single precision is implemented for one-pass gas optics (fused longwave, shortwave); this "
                    "model / gas list needs the multi-pass or unfused path");
      if (g_plan) {
        FusedPlan fp;
        HIPCHK(prepare_gas_fused(fa, fp));
        if (g_plan->npass == 0) { g_plan->first_fused = 1; g_plan->planck_fused = fa.mode == 1; g_plan->first = fp; }
        ++g_plan->npass;
      } else {
        ProfScope prof(fa.mode == 1 ? (g_f32 ? "gas_lw_fused_f32" : "gas_lw_fused") : "tau", stream);
        HIPCHK(launch_gas_fused(fa, stream));
      }
    } else {
      if (g_f32) return fail("ecckd: single precision needs the fast arithmetic mode (ecckd_set_arithmetic(0))");
      if (g_plan) {
        ++g_plan->npass;
      } else {
        ProfScope prof("tau", stream);
        HIPCHK(launch_tau(a, stream));
      }
    }
    first_pass = false;
  } while (pos < seq.size());
  return 0;
}

int check_model(const ecckd_model *m) {
  if (!m) return fail("ecckd: null model");
  if (!m->finalized) return fail("ecckd: model is not finalized");
  if (m->device < 0) return fail("ecckd: host-only model (device -1): no GPU, no compute (there is no CPU fallback)");
  return 0;
}

int check_dims(int ncol, int nlay) {
  if (ncol < 0 || nlay < 1) return fail("ecckd: bad ncol/nlay");
  return 0;
}
int check_gas_optics_dims(int ncol, int nlay) {
  if (check_dims(ncol, nlay)) return 1;
  // the gas-optics kernels address one g-plane pair of a column-fastest array with 32-bit byte offsets
  if (((size_t)ncol * (size_t)nlay + (size_t)ncol) * sizeof(double) >= (size_t)0xFFFFFFF0u)
    return fail("ecckd: ncol*(nlay+1) too large for one call (limit 2^29 elements): split the column range");
  return 0;
}

// Copies gas_desc data arrays to the arena and returns the device-side description.
struct StagedGases {
  std::vector<const double *> ptr;
  GasDesc gd;
};
size_t staged_gas_bytes(const GasDesc &gd, int ncol, int nlay) {
  size_t b = 0;
  for (int j = 0; j < gd.ngas; ++j) b += align256(vmr_extent(gd, j, ncol, nlay) * esz());
  return b;
}
int stage_gases(const GasDesc &gd, int ncol, int nlay, Bump &bump, hipStream_t s, StagedGases &out) {
  out.ptr.assign(gd.ngas, nullptr);
  for (int j = 0; j < gd.ngas; ++j) {
    const size_t n = vmr_extent(gd, j, ncol, nlay);
    if (!n) continue;
    double *d = bump.take(n);
    HIPCHK(hipMemcpyAsync(d, gd.vmr[j], n * esz(), hipMemcpyHostToDevice, s));
    out.ptr[j] = d;
  }
  out.gd = gd;
  out.gd.vmr = out.ptr.data();
  return 0;
}

int h2d(double *d, const double *h, size_t n, hipStream_t s) {
  HIPCHK(hipMemcpyAsync(d, h, n * esz(), hipMemcpyHostToDevice, s));
  return 0;
}
int d2h(double *h, const double *d, size_t n, hipStream_t s) {
  HIPCHK(hipMemcpyAsync(h, d, n * esz(), hipMemcpyDeviceToHost, s));
  return 0;
}

int fill_band_map(int ngpt, int nband, const int *band2gpt, unsigned char *gpt2band) {
  if (ngpt < 1 || ngpt > 256) return fail("ecckd: ngpt must be in 1..256");
  if (nband < 1 || nband > 255 || !band2gpt) return fail("ecckd: bad band description");
  std::memset(gpt2band, 0, 256);
  std::vector<int> seen(ngpt, 0);
  for (int b = 0; b < nband; ++b) {
    const int lo = band2gpt[2 * b], hi = band2gpt[2 * b + 1];
    if (lo < 1 || hi > ngpt || lo > hi) return fail("ecckd: band2gpt out of range");
    for (int gpt = lo; gpt <= hi; ++gpt) { gpt2band[gpt - 1] = (unsigned char)b; seen[gpt - 1] = 1; }
  }
  for (int i = 0; i < ngpt; ++i)
    if (!seen[i]) return fail("ecckd: band2gpt does not cover every g-point");
  return 0;
}

}  // namespace

extern "C" {

const char *ecckd_last_error(void) { return g_err.c_str(); }

const char *ecckd_build_info(void) {
  return "rte-ecckd hot path for MI355X: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off, fp64";
}

// ------------------------------- device memory for hosts without a HIP binding ---------------------

int ecckd_device_malloc(int device, size_t bytes, void **ptr) {
  if (!ptr) return fail("ecckd_device_malloc: null output");
  *ptr = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail("ecckd: no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail("ecckd_device_malloc: bad device ordinal");
  HIPCHK(hipSetDevice(device));
  if (bytes == 0) return 0;
  HIPCHK(hipMalloc(ptr, bytes));
  return 0;
}

int ecckd_device_free(int device, void *ptr) {
  if (!ptr) return 0;
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipFree(ptr));
  return 0;
}

int ecckd_device_memcpy(int device, void *dst, const void *src, size_t bytes, int to_device) {
  if (bytes == 0) return 0;
  if (!dst || !src) return fail("ecckd_device_memcpy: null pointer");
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipMemcpy(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
  return 0;
}

// ------------------------------- kernel timing hooks -------------------------------------

int ecckd_set_arithmetic(int mode) {
  if (mode != 0 && mode != 1) return fail("ecckd_set_arithmetic: mode must be 0 (fast) or 1 (reference order)");
  g_arith.store(mode);
  return 0;
}

int ecckd_get_arithmetic(void) { return g_arith.load(); }

int ecckd_set_solver_option(const char *name, double value) {
  if (!name) return fail("ecckd_set_solver_option: null name");
  const std::string n(name);
  if (!(value - value == 0.)) return fail("ecckd_set_solver_option: value must be finite");
  if (n == "lw_tau_thresh") g_opt.lw_tau_thresh.store(value);                       // <= 0 restores the default
  else if (n == "lw_series_terms") {
    if (value != 2. && value != 3.) return fail("ecckd_set_solver_option: lw_series_terms must be 2 or 3");
    g_opt.lw_series_terms.store((int)value);
  } else if (n == "lw_inc_flux_isotropic") g_opt.lw_inc_flux_isotropic.store(value != 0. ? 1 : 0);
  else if (n == "sw_k_floor") {
    if (!(value > 0.)) return fail("ecckd_set_solver_option: sw_k_floor must be > 0");
    g_opt.sw_k_floor.store(value);
  } else if (n == "sw_dir_clamp") g_opt.sw_dir_clamp.store(value != 0. ? 1 : 0);
  else if (n == "lw_solver") {
    if (value != 0. && value != 1.) return fail("ecckd_set_solver_option: lw_solver must be 0 (register-resident) or 1 (layer-split)");
    g_opt.lw_solver.store((int)value);
  } else if (n == "lw_split_seg") {
    if (value != 10. && value != 12. && value != 15.) return fail("ecckd_set_solver_option: lw_split_seg must be 10, 12 or 15");
    g_opt.lw_split_seg.store((int)value);
  } else if (n == "gas_merge_scalars") g_opt.gas_merge_scalars.store(value != 0. ? 1 : 0);
  else if (n == "lw_tail_split") g_opt.lw_tail_split.store(value != 0. ? 1 : 0);
  else if (n == "sw_tail_split") g_opt.sw_tail_split.store(value != 0. ? 1 : 0);
  else if (n == "sw_solver") {
    if (value != 0. && value != 1.) return fail("ecckd_set_solver_option: sw_solver must be 0 (layer-systolic) or 1 (two-pass per lane)");
    g_opt.sw_solver.store((int)value);
  }
  else if (n == "gas_slab_f32") {
    if (value != 0. && value != 1. && value != 2.) return fail("ecckd_set_solver_option: gas_slab_f32 must be 0 (never), 1 (always) or 2 (auto)");
    g_opt.gas_slab_f32.store((int)value);
  }
  else return fail("ecckd_set_solver_option: unknown option '" + n + "' (lw_tau_thresh, lw_series_terms, "
                   "lw_inc_flux_isotropic, sw_k_floor, sw_dir_clamp, lw_solver, lw_split_seg, gas_merge_scalars, lw_tail_split, "
                   "sw_tail_split, sw_solver, gas_slab_f32)");
  return 0;
}

int ecckd_get_solver_option(const char *name, double *value) {
  if (!name || !value) return fail("ecckd_get_solver_option: null argument");
  const std::string n(name);
  if (n == "lw_tau_thresh") { const double t = g_opt.lw_tau_thresh.load(); *value = t > 0. ? t : std::sqrt(2.220446049250313e-16); }
  else if (n == "lw_series_terms") *value = g_opt.lw_series_terms.load();
  else if (n == "lw_inc_flux_isotropic") *value = g_opt.lw_inc_flux_isotropic.load();
  else if (n == "sw_k_floor") *value = g_opt.sw_k_floor.load();
  else if (n == "sw_dir_clamp") *value = g_opt.sw_dir_clamp.load();
  else if (n == "lw_solver") *value = g_opt.lw_solver.load();
  else if (n == "lw_split_seg") *value = g_opt.lw_split_seg.load();
  else if (n == "gas_merge_scalars") *value = g_opt.gas_merge_scalars.load();
  else if (n == "lw_tail_split") *value = g_opt.lw_tail_split.load();
  else if (n == "sw_tail_split") *value = g_opt.sw_tail_split.load();
  else if (n == "sw_solver") *value = g_opt.sw_solver.load();
  else if (n == "gas_slab_f32") *value = g_opt.gas_slab_f32.load();
  else return fail("ecckd_get_solver_option: unknown option '" + n + "'");
  return 0;
}

size_t ecckd_rte_lw_scratch_bytes(int ncol, int nlay, int ngpt) {
  return ncol > 0 && nlay > 0 ? ecckd::rte_lw_scratch_bytes(ncol, nlay, ngpt) : 0;
}
size_t ecckd_rte_sw_scratch_bytes(int ncol, int nlay, int ngpt) {
  return ncol > 0 && nlay > 0 ? ecckd::rte_sw_scratch_bytes(ncol, nlay, ngpt) : 0;
}

size_t ecckd_rte_sw_tail_scratch_bytes(int device, int ncol, int nlay, int ngpt) {
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0 || device < 0 || device >= 16 || !g_opt.sw_tail_split.load()) return 0;
  ecckd::RteSwArgs a{};
  a.ncol = ncol; a.nlay = nlay; a.ng = ngpt;
  if (g_opt.sw_solver.load() == 0 && ecckd::rte_sw_sys_applies(a)) return ecckd::rte_sw_sys_plan(a, simd_slots(device) / 4);
  long first = -1;
  size_t at = 0;
  return ecckd::rte_sw_tail_plan(a, &first, &at);
}
size_t ecckd_rte_lw_tail_scratch_bytes(int device, int ncol, int nlay, int ngpt, int n_gauss_angles, int single_precision) {
  if (ncol <= 0 || nlay <= 0 || ngpt <= 0 || device < 0 || device >= 16 || !g_opt.lw_tail_split.load()) return 0;
  if (n_gauss_angles < 1 || n_gauss_angles > 4) return 0;
  ecckd::RteLwArgs a{};
  a.ncol = ncol; a.nlay = nlay; a.ng = ngpt; a.nmus = n_gauss_angles; a.f32 = single_precision ? 1 : 0;
  a.use_split = g_opt.lw_solver.load();
  long first = -1;
  return ecckd::rte_lw_tail_plan(a, simd_slots(device), &first);
}

int ecckd_set_stream_scratch(int device, void *stream, void *buffer, size_t bytes) {
  if (device < 0 || device >= 16) return fail("ecckd_set_stream_scratch: bad device ordinal");
  if ((buffer == nullptr) != (bytes == 0)) return fail("ecckd_set_stream_scratch: buffer and size must both be given, or neither");
  ScratchPool &pool = g_scratch_pool[device];
  std::lock_guard<std::mutex> lock(pool.mu);
  ScratchPool::Block &b = pool.live[static_cast<hipStream_t>(stream)];
  if (b.p && !b.caller_owned) pool.retired.push_back(b.p);
  b.p = buffer; b.bytes = bytes; b.caller_owned = buffer != nullptr; b.captured = false;
  return 0;
}

static int release_pool(ScratchPool &pool, int device);
int ecckd_release_scratch(int device) {
  if (device < 0 || device >= 16) return fail("ecckd_release_scratch: bad device ordinal");
  if (release_pool(g_flag_pool[device], device)) return 1;
  return release_pool(g_scratch_pool[device], device);
}
static int release_pool(ScratchPool &pool, int device) {
  std::lock_guard<std::mutex> lock(pool.mu);
  if (pool.live.empty() && pool.retired.empty()) return 0;
  HIPCHK(hipSetDevice(device));
  if (pool.leases > 0) return fail("ecckd_release_scratch: another thread is inside a solver call on this device");
  HIPCHK(hipDeviceSynchronize());   // nothing in flight may still use a block
  // (graphs captured with library-owned scratch hold these addresses: destroy them before releasing -- see the header)
  for (auto it = pool.live.begin(); it != pool.live.end();) {
    if (it->second.caller_owned) { ++it; continue; }   // the caller's buffer stays registered
    if (it->second.p) (void)hipFree(it->second.p);
    it = pool.live.erase(it);
  }
  for (void *p : pool.retired) (void)hipFree(p);
  pool.retired.clear();
  return 0;
}

int ecckd_prof_enable(int on) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  g_prof_on = on != 0;
  return 0;
}

int ecckd_prof_report(int max_kernels, char *names, double *total_ms, long long *launches) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  int n = 0;
  for (ProfRec &r : g_prof) {
    float ms = 0.f;
    if (hipEventSynchronize(r.stop) == hipSuccess) (void)hipEventElapsedTime(&ms, r.start, r.stop);
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
    int k = 0;
    for (; k < n; ++k)
      if (std::strncmp(names + (size_t)k * ECCKD_NAME_LEN, r.name, ECCKD_NAME_LEN) == 0) break;
    if (k == n) {
      if (n >= max_kernels) continue;
      std::memset(names + (size_t)n * ECCKD_NAME_LEN, 0, ECCKD_NAME_LEN);
      std::strncpy(names + (size_t)n * ECCKD_NAME_LEN, r.name, ECCKD_NAME_LEN - 1);
      total_ms[n] = 0.;
      launches[n] = 0;
      ++n;
    }
    total_ms[k] += ms;
    launches[k] += 1;
  }
  g_prof.clear();
  return n;
}

// ------------------------------- model construction -------------------------------------

int ecckd_model_begin(int ng, int np, int nt, const double *log_pressure, const double *temperature,
                      ecckd_model_t **model) {
  if (!model) return fail("ecckd_model_begin: null output");
  *model = nullptr;
  if (ng < 1 || np < 2 || nt < 2 || !log_pressure || !temperature)
    return fail("ecckd_model_begin: bad dimensions or null tables");
  ecckd_model *m = new ecckd_model;
  m->ng = ng; m->np = np; m->nt = nt;
  m->log_pressure.assign(log_pressure, log_pressure + np);
  m->temperature.assign(temperature, temperature + (size_t)np * nt);
  // default band structure: a single band over all g-points
  m->nband = 1;
  m->band2gpt = {1, ng};
  m->band_lims_wvn = {0., 0.};
  *model = m;
  return 0;
}

int ecckd_model_set_planck(ecckd_model_t *m, int ntp, const double *temperature_planck,
                           const double *planck_function) {
  if (!m || m->finalized) return fail("ecckd_model_set_planck: model missing or frozen");
  if (ntp < 2 || !temperature_planck || !planck_function) return fail("ecckd_model_set_planck: bad table");
  m->ntp = ntp;
  m->temperature_planck.assign(temperature_planck, temperature_planck + ntp);
  m->planck_function.assign(planck_function, planck_function + (size_t)m->ng * ntp);
  m->has_planck = true;
  return 0;
}

int ecckd_model_set_solar(ecckd_model_t *m, const double *solar_irradiance, const double *rayleigh) {
  if (!m || m->finalized) return fail("ecckd_model_set_solar: model missing or frozen");
  if (!solar_irradiance || !rayleigh) return fail("ecckd_model_set_solar: null table");
  m->solar_irradiance.assign(solar_irradiance, solar_irradiance + m->ng);
  m->rayleigh.assign(rayleigh, rayleigh + m->ng);
  m->total_solar_irradiance = 0.;
  for (double s : m->solar_irradiance) m->total_solar_irradiance += s;   // mo_load_coefficients.F90:89
  m->has_solar = true;
  return 0;
}

int ecckd_model_set_bands(ecckd_model_t *m, int nband, const double *band_lims_wvn, const int *band2gpt) {
  if (!m || m->finalized) return fail("ecckd_model_set_bands: model missing or frozen");
  unsigned char tmp[256];
  if (fill_band_map(m->ng, nband, band2gpt, tmp)) return 1;
  m->nband = nband;
  m->band2gpt.assign(band2gpt, band2gpt + 2 * (size_t)nband);
  if (band_lims_wvn) m->band_lims_wvn.assign(band_lims_wvn, band_lims_wvn + 2 * (size_t)nband);
  else m->band_lims_wvn.assign(2 * (size_t)nband, 0.);
  return 0;
}

int ecckd_model_add_gas(ecckd_model_t *m, const char *name, int code, int composite_only, int nv,
                        const double *mole_fraction, double reference_mole_fraction,
                        const double *coefficient) {
  if (!m || m->finalized) return fail("ecckd_model_add_gas: model missing or frozen");
  if (!name || !coefficient) return fail("ecckd_model_add_gas: null argument");
  if (m->gas.size() >= ECCKD_MAX_GASES) return fail("ecckd_model_add_gas: more than 16 gases");
  if (code < 0 || code > 3) return fail(std::string("load_and_init_ecckd: bad concentration code for ") + name);
  ecckd_model::Gas g;
  g.name = trim_name(name);
  g.code = code;
  g.composite_only = composite_only ? 1 : 0;
  g.ref = reference_mole_fraction;
  if (code == ECCKD_LOOK_UP_TABLE) {
    if (nv < 2 || !mole_fraction) return fail("ecckd_model_add_gas: look_up_table gas needs mole_fraction(nv>=2)");
    g.nv = nv;
    g.mole_fraction.assign(mole_fraction, mole_fraction + nv);
  } else {
    g.nv = 1;
  }
  g.coef.assign(coefficient, coefficient + (size_t)m->ng * m->np * m->nt * g.nv);
  m->gas.push_back(std::move(g));
  return 0;
}

int ecckd_model_finalize(ecckd_model_t *m, int device) {
  if (!m) return fail("ecckd_model_finalize: null model");
  if (m->finalized) return fail("ecckd_model_finalize: already finalized");
  {
    const std::string bad = ecckd::validate_model(*m);
    if (!bad.empty()) return fail("ecckd_model_finalize: " + bad);
  }
  if (device == -1) {   // host-only model: getters work, every compute call fails
    m->device = -1;
    m->finalized = true;
    return 0;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail("ecckd: no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev || device >= 16) return fail("ecckd_model_finalize: bad device ordinal");
  HIPCHK(hipSetDevice(device));
  // layout of the single device buffer
  std::vector<double> host;
  auto put = [&](const std::vector<double> &v) {
    size_t off = host.size();
    host.insert(host.end(), v.begin(), v.end());
    while (host.size() % 32) host.push_back(0.);
    return off;
  };
  m->off_zero = put(std::vector<double>(32, 0.));
  m->off_temperature = put(m->temperature);
  if (m->has_planck) m->off_planck = put(m->planck_function);
  if (m->has_solar) { m->off_rayleigh = put(m->rayleigh); m->off_solar = put(m->solar_irradiance); }
  for (size_t i = 0; i < m->gas.size(); ++i) {
    ecckd_model::Gas &g = m->gas[i];
    g.has_negative = false;
    for (double c : g.coef)
      if (!(c >= 0.)) { g.has_negative = true; break; }   // negatives or NaN: keep the per-g clamp
    bool shared = false;
    for (size_t k = 0; k < i && !shared; ++k) {   // o2/n2 carry identical composite tables
      const ecckd_model::Gas &o = m->gas[k];
      if (o.coef.size() == g.coef.size() &&
          std::memcmp(o.coef.data(), g.coef.data(), g.coef.size() * sizeof(double)) == 0) {
        g.dev_off = o.dev_off;
        shared = true;
      }
    }
    if (!shared) g.dev_off = put(g.coef);
  }
  HIPCHK(hipMalloc(reinterpret_cast<void **>(&m->dbuf), host.size() * sizeof(double)));
  HIPCHK(hipMemcpy(m->dbuf, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice));
  {   // the same image rounded to float, for the single-precision entry points
    std::vector<float> hostf(host.begin(), host.end());
    // ... and whether that rounding changed anything: ecCKD's files hold float32 variables (widened exactly by the reader,
    // example/rfmip-rad-irf/mo_simple_netcdf.F90:44-142), so a fp64 kernel may stage them in LDS as float32 ("gas_slab_f32")
    m->f32_exact = true;
    for (size_t i = 0; i < host.size() && m->f32_exact; ++i) m->f32_exact = (double)hostf[i] == host[i];
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&m->dbuf32), hostf.size() * sizeof(float)));
    HIPCHK(hipMemcpy(m->dbuf32, hostf.data(), hostf.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  HIPCHK(hipStreamCreateWithFlags(&m->host_stream, hipStreamNonBlocking));
  m->device = device;
  m->finalized = true;
  return 0;
}

int ecckd_model_load(const char *filename, int device, ecckd_model_t **model) {
  if (!model || !filename) return fail("ecckd_model_load: null argument");
  *model = nullptr;
  ecckd_model *m = new ecckd_model;
  try {
    ecckd::load_and_init(*m, filename);
  } catch (const std::exception &e) {
    delete m;
    return fail(e.what());
  }
  if (ecckd_model_finalize(m, device)) {
    delete m;
    return 1;
  }
  *model = m;
  return 0;
}

void ecckd_model_destroy(ecckd_model_t *m) {
  if (!m) return;
  if (m->finalized && m->device >= 0) {
    (void)hipSetDevice(m->device);
    if (m->host_stream) (void)hipStreamDestroy(m->host_stream);
    if (m->arena) (void)hipFree(m->arena);
    if (m->dbuf) (void)hipFree(m->dbuf);
    if (m->dbuf32) (void)hipFree(m->dbuf32);
  }
  delete m;
}

// ------------------------------------- getters -------------------------------------------

int ecckd_model_get_ngpt(const ecckd_model_t *m) { return m ? m->ng : 0; }
int ecckd_model_get_nband(const ecckd_model_t *m) { return m ? m->nband : 0; }
int ecckd_model_get_ngas(const ecckd_model_t *m) { return m ? (int)m->gas.size() : 0; }
int ecckd_model_get_gas_name(const ecckd_model_t *m, int index, char *name) {
  if (!m || !name || index < 0 || index >= (int)m->gas.size()) return fail("ecckd_model_get_gas_name: bad index");
  std::memset(name, 0, ECCKD_NAME_LEN);
  std::strncpy(name, m->gas[index].name.c_str(), ECCKD_NAME_LEN - 1);
  return 0;
}
int ecckd_model_source_is_internal(const ecckd_model_t *m) { return m && m->has_planck; }
int ecckd_model_source_is_external(const ecckd_model_t *m) { return m && m->has_solar; }
double ecckd_model_get_press_min(const ecckd_model_t *m) { return m ? std::exp(m->log_pressure.front()) : 0.; }
double ecckd_model_get_press_max(const ecckd_model_t *m) { return m ? std::exp(m->log_pressure.back()) : 0.; }
double ecckd_model_get_temp_min(const ecckd_model_t *m) {
  if (!m) return 0.;
  double v = m->temperature[0];
  for (double t : m->temperature) v = t < v ? t : v;
  return v;
}
double ecckd_model_get_temp_max(const ecckd_model_t *m) {
  if (!m) return 0.;
  double v = m->temperature[0];
  for (double t : m->temperature) v = t > v ? t : v;
  return v;
}
double ecckd_model_get_total_solar_irradiance(const ecckd_model_t *m) { return m ? m->total_solar_irradiance : 0.; }
int ecckd_model_get_band2gpt(const ecckd_model_t *m, int *band2gpt) {
  if (!m || !band2gpt) return fail("ecckd_model_get_band2gpt: null argument");
  std::memcpy(band2gpt, m->band2gpt.data(), m->band2gpt.size() * sizeof(int));
  return 0;
}
int ecckd_model_get_band_lims_wvn(const ecckd_model_t *m, double *lims) {
  if (!m || !lims) return fail("ecckd_model_get_band_lims_wvn: null argument");
  std::memcpy(lims, m->band_lims_wvn.data(), m->band_lims_wvn.size() * sizeof(double));
  return 0;
}
int ecckd_model_get_device(const ecckd_model_t *m) { return m ? m->device : -1; }

// ------------------------------------ gas optics -----------------------------------------

static int gas_optics_lw_dev(const ecckd_model *m, int ncol, int nlay, const double *plev,
                             const double *tlay, const double *tsfc, const double *tlev,
                             const GasDesc &gd, double *tau, double *lay_source, double *lev_inc,
                             double *lev_dec, double *sfc_source, hipStream_t stream) {
  const PlanckSide pl{tlev, tsfc, lay_source, lev_inc, lev_dec, sfc_source};
  bool planck_done = false;
  if (gas_optical_depth_dev(m, ncol, nlay, plev, tlay, gd, tau, false, nullptr, nullptr, &pl, &planck_done, stream))
    return 1;   // :401
  if (planck_done) return 0;
  if (g_f32) return fail("ecckd: single precision is implemented for the fused longwave gas optics only");
  ecckd::PlanckArgs p{};
  p.ncol = ncol; p.nlay = nlay; p.ng = m->ng; p.ntp = m->ntp;
  p.planck = m->dbuf + m->off_planck;
  p.t0 = m->temperature_planck[0];                                  // :272
  p.dt = m->temperature_planck[1] - m->temperature_planck[0];      // :271
  p.tlay = tlay; p.tlev = tlev; p.tsfc = tsfc;
  p.lay_source = lay_source; p.lev_source_inc = lev_inc; p.lev_source_dec = lev_dec;
  p.sfc_source = sfc_source;
  {
    ProfScope prof("planck", stream);
    // :407-424; the fast arithmetic mode takes the kernel with paired 16-byte stores (same bits)
    if (g_arith.load() == 0) HIPCHK(ecckd::launch_planck_pair(p, 0, stream));
    else HIPCHK(ecckd::launch_planck(p, stream));
  }
  return 0;
}

int ecckd_planck_sources(const ecckd_model_t *m, int ncol, int nlay, const double *tlay, const double *tlev,
                         const double *tsfc, double *lay_source, double *lev_source_inc, double *lev_source_dec,
                         double *sfc_source, int memspace, void *stream) {
  if (check_model(m) || check_gas_optics_dims(ncol, nlay)) return 1;
  if (!m->has_planck) return fail("ecckd_planck_sources: model has no Planck table (shortwave model?)");
  if (memspace != ECCKD_DEVICE) return fail("ecckd_planck_sources: device arrays only (ECCKD_DEVICE)");
  if (!tlay || !tsfc || !lay_source || !sfc_source) return fail("ecckd_planck_sources: null argument");
  if (tlev && (!lev_source_inc || !lev_source_dec)) return fail("ecckd_planck_sources: null level sources");
  HIPCHK(hipSetDevice(m->device));
  if (ncol == 0) return 0;
  ecckd::PlanckArgs p{};
  p.ncol = ncol; p.nlay = nlay; p.ng = m->ng; p.ntp = m->ntp;
  p.planck = m->dbuf + m->off_planck;
  p.t0 = m->temperature_planck[0];                                  // :272
  p.dt = m->temperature_planck[1] - m->temperature_planck[0];      // :271
  p.tlay = tlay; p.tlev = tlev; p.tsfc = tsfc;
  p.lay_source = lay_source; p.lev_source_inc = lev_source_inc; p.lev_source_dec = lev_source_dec;
  p.sfc_source = sfc_source;
  ProfScope prof("planck", static_cast<hipStream_t>(stream));
  if (g_arith.load() == 0) HIPCHK(ecckd::launch_planck_pair(p, 0, static_cast<hipStream_t>(stream)));
  else HIPCHK(ecckd::launch_planck(p, static_cast<hipStream_t>(stream)));
  return 0;
}

int ecckd_gas_optics_plan(const ecckd_model_t *m, int ncol, int nlay, int single_precision, int ngas,
                          const char *gas_names, int *plan) {
  int p[ECCKD_PLAN_LEN];
  if (ecckd_gas_optics_plan_ex(m, ncol, nlay, single_precision, ngas, gas_names, nullptr, ECCKD_PLAN_LEN, p)) return 1;
  if (!plan) return fail("ecckd_gas_optics_plan: null argument");
  for (int i = 0; i < 8; ++i) plan[i] = p[i];
  return 0;
}

int ecckd_gas_optics_plan_ex(const ecckd_model_t *m, int ncol, int nlay, int single_precision, int ngas,
                             const char *gas_names, const int *vmr_is_scalar, int nplan, int *plan_out) {
  if (!m) return fail("ecckd: null model");
  if (!m->finalized) return fail("ecckd: model is not finalized");
  if (check_gas_optics_dims(ncol, nlay)) return 1;
  if (!plan_out || nplan < 1 || (ngas > 0 && !gas_names)) return fail("ecckd_gas_optics_plan: null argument");
  int plan[ECCKD_PLAN_LEN] = {0};
  PlanRecord rec;
  rec.is_scalar = vmr_is_scalar;
  const GasDesc gd{ngas, gas_names, nullptr, nullptr, nullptr, nullptr};
  const PlanckSide pl{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const bool lw = m->has_planck;
  struct Guard {
    explicit Guard(PlanRecord *r, int f32) { g_plan = r; g_f32 = f32; }
    ~Guard() { g_plan = nullptr; g_f32 = 0; }
  } guard(&rec, single_precision ? 1 : 0);
  bool planck_done = false;
  if (gas_optical_depth_dev(m, ncol, nlay, nullptr, nullptr, gd, nullptr, !lw, nullptr, nullptr, lw ? &pl : nullptr,
                            &planck_done, nullptr))
    return 1;
  plan[0] = rec.npass;
  plan[1] = rec.first_fused;
  plan[2] = rec.planck_fused;
  plan[3] = rec.first.slab_rows;
  plan[4] = rec.first.planck_rows;
  plan[5] = rec.first.col_chunks;
  plan[6] = (int)rec.first.lds_bytes;
  plan[7] = rec.first.GC;
  plan[8] = rec.first.NB;
  plan[9] = rec.first.merged;
  for (int i = 0; i < nplan && i < ECCKD_PLAN_LEN; ++i) plan_out[i] = plan[i];
  return 0;
}

int ecckd_gas_optics_lw(const ecckd_model_t *m, int ncol, int nlay, const double *plev,
                        const double *tlay, const double *tsfc, const double *tlev, int ngas,
                        const char *gas_names, const double *const *vmr,
                        const long long *vmr_col_stride, const long long *vmr_lay_stride,
                        const double *vmr_scalar, double *tau, double *lay_source,
                        double *lev_source_inc, double *lev_source_dec, double *sfc_source,
                        int memspace, void *stream) {
  if (check_model(m) || check_gas_optics_dims(ncol, nlay)) return 1;
  if (!m->has_planck) return fail("ecckd_gas_optics_lw: model has no Planck table (shortwave model?)");
  if (!plev || !tlay || !tsfc || !tau || !lay_source || !sfc_source || (ngas > 0 && !gas_names))
    return fail("ecckd_gas_optics_lw: null argument");
  if (tlev && (!lev_source_inc || !lev_source_dec)) return fail("ecckd_gas_optics_lw: null level sources");
  HIPCHK(hipSetDevice(m->device));
  const GasDesc gd{ngas, gas_names, vmr, vmr_col_stride, vmr_lay_stride, vmr_scalar};
  const size_t n2 = (size_t)ncol * nlay, n2l = (size_t)ncol * (nlay + 1), n3 = n2 * m->ng;
  if (ncol == 0) return tlev ? 0 : fail("tlev is required for ecckd");

  if (memspace == ECCKD_DEVICE) {
    if (gas_optics_lw_dev(m, ncol, nlay, plev, tlay, tsfc, tlev, gd, tau, lay_source, lev_source_inc,
                          lev_source_dec, sfc_source, static_cast<hipStream_t>(stream)))
      return 1;
    return tlev ? 0 : fail("tlev is required for ecckd");   // :414-417
  }
  if (memspace != ECCKD_HOST && memspace != ECCKD_MIXED) return fail("ecckd: bad memspace");
  // ECCKD_MIXED: inputs are host arrays (staged here), the (ncol,nlay,ngpt) / (ncol,ngpt) outputs are the caller's
  // device buffers: nothing but ~4 KB per column crosses the bus
  const bool mixed = memspace == ECCKD_MIXED;

  ecckd_model *mm = const_cast<ecckd_model *>(m);
  std::lock_guard<std::mutex> lock(mm->mu);
  hipStream_t s = mm->host_stream;
  size_t need = align256(n2l * esz()) * 2 + align256(n2 * esz()) + align256((size_t)ncol * esz()) +
                staged_gas_bytes(gd, ncol, nlay);
  if (!mixed) need += align256(n3 * esz()) * 4 + align256((size_t)ncol * m->ng * esz());
  if (need > mm->arena_bytes) {
    HIPCHK(hipStreamSynchronize(s));
    if (mm->arena) { HIPCHK(hipFree(mm->arena)); mm->arena = nullptr; mm->arena_bytes = 0; }
    HIPCHK(hipMalloc(&mm->arena, need));
    mm->arena_bytes = need;
  }
  Bump b(mm->arena);
  double *d_plev = b.take(n2l), *d_tlev = b.take(n2l), *d_tlay = b.take(n2), *d_tsfc = b.take(ncol);
  if (h2d(d_plev, plev, n2l, s) || h2d(d_tlay, tlay, n2, s) || h2d(d_tsfc, tsfc, ncol, s)) return 1;
  if (tlev && h2d(d_tlev, tlev, n2l, s)) return 1;
  StagedGases sg;
  if (stage_gases(gd, ncol, nlay, b, s, sg)) return 1;
  double *d_tau = tau, *d_lay = lay_source, *d_inc = lev_source_inc, *d_dec = lev_source_dec, *d_sfc = sfc_source;
  if (!mixed) {
    d_tau = b.take(n3); d_lay = b.take(n3); d_inc = b.take(n3); d_dec = b.take(n3);
    d_sfc = b.take((size_t)ncol * m->ng);
  }
  if (gas_optics_lw_dev(m, ncol, nlay, d_plev, d_tlay, d_tsfc, tlev ? d_tlev : nullptr, sg.gd, d_tau,
                        d_lay, d_inc, d_dec, d_sfc, s))
    return 1;
  if (!mixed) {
    if (d2h(tau, d_tau, n3, s) || d2h(lay_source, d_lay, n3, s) ||
        d2h(sfc_source, d_sfc, (size_t)ncol * m->ng, s))
      return 1;
    if (tlev && (d2h(lev_source_inc, d_inc, n3, s) || d2h(lev_source_dec, d_dec, n3, s))) return 1;
  }
  HIPCHK(hipStreamSynchronize(s));   // (mixed too: the solver call that follows runs on another stream)
  return tlev ? 0 : fail("tlev is required for ecckd");
}

int ecckd_gas_optics_lw_f32(const ecckd_model_t *m, int ncol, int nlay, const float *plev, const float *tlay,
                            const float *tsfc, const float *tlev, int ngas, const char *gas_names,
                            const float *const *vmr, const long long *vmr_col_stride,
                            const long long *vmr_lay_stride, const double *vmr_scalar, float *tau,
                            float *lay_source, float *lev_source_inc, float *lev_source_dec,
                            float *sfc_source, int memspace, void *stream) {
  F32Scope scope;
  auto c = [](const float *p) { return reinterpret_cast<const double *>(p); };
  auto w = [](float *p) { return reinterpret_cast<double *>(p); };
  return ecckd_gas_optics_lw(m, ncol, nlay, c(plev), c(tlay), c(tsfc), c(tlev), ngas, gas_names,
                             reinterpret_cast<const double *const *>(vmr), vmr_col_stride, vmr_lay_stride,
                             vmr_scalar, w(tau), w(lay_source), w(lev_source_inc), w(lev_source_dec),
                             w(sfc_source), memspace, stream);
}

static int gas_optics_sw_dev(const ecckd_model *m, int ncol, int nlay, const double *plev,
                             const double *tlay, const GasDesc &gd, double *tau, double *ssa,
                             double *g, double *toa_src, hipStream_t stream) {
  const bool two_stream = ssa && g;
  if (gas_optical_depth_dev(m, ncol, nlay, plev, tlay, gd, tau, true, two_stream ? ssa : nullptr,
                            two_stream ? g : nullptr, nullptr, nullptr, stream))   // :449-460
    return 1;
  if (!two_stream) return 0;   // caller reports :461-463 after tau has been written
  HIPCHK(ecckd::launch_toa_src(g_f32 ? reinterpret_cast<const double *>(m->dbuf32 + m->off_solar) : m->dbuf + m->off_solar, ncol,
                               m->ng, toa_src, g_f32, stream));   // :468-472
  return 0;
}

int ecckd_gas_optics_sw(const ecckd_model_t *m, int ncol, int nlay, const double *plev,
                        const double *tlay, int ngas, const char *gas_names,
                        const double *const *vmr, const long long *vmr_col_stride,
                        const long long *vmr_lay_stride, const double *vmr_scalar, double *tau,
                        double *ssa, double *g, double *toa_src, int memspace, void *stream) {
  if (check_model(m) || check_gas_optics_dims(ncol, nlay)) return 1;
  if (!m->has_solar) return fail("ecckd_gas_optics_sw: model has no solar table (longwave model?)");
  if (!plev || !tlay || !tau || (ngas > 0 && !gas_names)) return fail("ecckd_gas_optics_sw: null argument");
  const bool two_stream = ssa && g;
  if (two_stream && !toa_src) return fail("ecckd_gas_optics_sw: null toa_src");
  HIPCHK(hipSetDevice(m->device));
  const GasDesc gd{ngas, gas_names, vmr, vmr_col_stride, vmr_lay_stride, vmr_scalar};
  const size_t n2 = (size_t)ncol * nlay, n2l = (size_t)ncol * (nlay + 1), n3 = n2 * m->ng;
  static const char *kNot2str = "shortwave must use ty_optical_props_2str";   // :462
  if (ncol == 0) return two_stream ? 0 : fail(kNot2str);

  if (memspace == ECCKD_DEVICE) {
    if (gas_optics_sw_dev(m, ncol, nlay, plev, tlay, gd, tau, ssa, g, toa_src, static_cast<hipStream_t>(stream)))
      return 1;
    return two_stream ? 0 : fail(kNot2str);
  }
  if (memspace != ECCKD_HOST && memspace != ECCKD_MIXED) return fail("ecckd: bad memspace");
  const bool mixed = memspace == ECCKD_MIXED;   // tau, ssa, g: caller's device buffers; toa_src and the inputs: host

  ecckd_model *mm = const_cast<ecckd_model *>(m);
  std::lock_guard<std::mutex> lock(mm->mu);
  hipStream_t s = mm->host_stream;
  size_t need = align256(n2l * esz()) + align256(n2 * esz()) + staged_gas_bytes(gd, ncol, nlay) +
                align256((size_t)ncol * m->ng * esz());
  if (!mixed) need += align256(n3 * esz()) * 3;
  if (need > mm->arena_bytes) {
    HIPCHK(hipStreamSynchronize(s));
    if (mm->arena) { HIPCHK(hipFree(mm->arena)); mm->arena = nullptr; mm->arena_bytes = 0; }
    HIPCHK(hipMalloc(&mm->arena, need));
    mm->arena_bytes = need;
  }
  Bump b(mm->arena);
  double *d_plev = b.take(n2l), *d_tlay = b.take(n2);
  if (h2d(d_plev, plev, n2l, s) || h2d(d_tlay, tlay, n2, s)) return 1;
  StagedGases sg;
  if (stage_gases(gd, ncol, nlay, b, s, sg)) return 1;
  double *d_toa = b.take((size_t)ncol * m->ng);
  double *d_tau = tau, *d_ssa = ssa, *d_g = g;
  if (!mixed) { d_tau = b.take(n3); d_ssa = b.take(n3); d_g = b.take(n3); }
  if (gas_optics_sw_dev(m, ncol, nlay, d_plev, d_tlay, sg.gd, d_tau, two_stream ? d_ssa : nullptr,
                        two_stream ? d_g : nullptr, d_toa, s))
    return 1;
  if (!mixed) {
    if (d2h(tau, d_tau, n3, s)) return 1;
    if (two_stream && (d2h(ssa, d_ssa, n3, s) || d2h(g, d_g, n3, s))) return 1;
  }
  if (two_stream && d2h(toa_src, d_toa, (size_t)ncol * m->ng, s)) return 1;
  HIPCHK(hipStreamSynchronize(s));
  return two_stream ? 0 : fail(kNot2str);
}

// -------------------------------------- solvers ------------------------------------------

// Gauss-Jacobi-5 quadrature of RTE-RRTMGP's mo_rte_lw (secants and weights for 1..4 angles;
// ecckd_rfmip_lw.F90:40-44 selects 1 or 3).
static const double kGaussDs[4][4] = {{1.66, 0., 0., 0.},
                                      {1.18350343, 2.81649655, 0., 0.},
                                      {1.09719858, 1.69338507, 4.70941630, 0.},
                                      {1.06056257, 1.38282560, 2.40148179, 7.15513024}};
static const double kGaussWts[4][4] = {{0.5, 0., 0., 0.},
                                       {0.3180413817, 0.1819586183, 0., 0.},
                                       {0.2009319137, 0.2292411064, 0.0698269799, 0.},
                                       {0.1355069134, 0.2034645680, 0.1298475476, 0.0311809710}};

static int check_device(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail("ecckd: no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev || device >= 16) return fail("ecckd: bad device ordinal");
  HIPCHK(hipSetDevice(device));
  return 0;
}

int ecckd_rte_lw(int device, int ncol, int nlay, int ngpt, int top_at_1, int n_gauss_angles,
                 const double *tau, const double *lay_source, const double *lev_source_inc,
                 const double *lev_source_dec, const double *sfc_source, int nband,
                 const int *band2gpt, const double *sfc_emis, double *flux_up, double *flux_dn,
                 int memspace, void *stream) {
  if (check_dims(ncol, nlay)) return 1;
  if (n_gauss_angles < 1 || n_gauss_angles > 4) return fail("rte_lw: have to ask for at least one quadrature point and no more than 4");
  if (!tau || !lay_source || !lev_source_inc || !lev_source_dec || !sfc_source || !sfc_emis || !flux_up || !flux_dn)
    return fail("ecckd_rte_lw: null argument");
  ecckd::RteLwArgs a{};
  if (g_band_override >= 0) std::memset(a.gpt2band, g_band_override, sizeof a.gpt2band);
  else if (fill_band_map(ngpt, nband, band2gpt, a.gpt2band)) return 1;
  if (check_device(device)) return 1;
  if (ncol == 0) return 0;
  a.ncol = ncol; a.nlay = nlay; a.ng = ngpt; a.top_at_1 = top_at_1 ? 1 : 0; a.nmus = n_gauss_angles;
  a.nband = nband;
  a.f32 = g_f32;
  a.shared_levels = g_shared_levels;
  {
    const double t = g_opt.lw_tau_thresh.load();
    a.tau_thresh = t > 0. ? t : (g_f32 ? std::sqrt(1.1920928955078125e-07) : std::sqrt(2.220446049250313e-16));   // sqrt(epsilon(1._wp))
    a.series3 = g_opt.lw_series_terms.load() == 3;
    a.inc_isotropic = g_opt.lw_inc_flux_isotropic.load();
    a.use_split = g_opt.lw_solver.load();
    a.split_seg = g_opt.lw_split_seg.load();
  }
  for (int k = 0; k < n_gauss_angles; ++k) {
    a.Ds[k] = kGaussDs[n_gauss_angles - 1][k];
    a.wts[k] = kGaussWts[n_gauss_angles - 1][k];
  }
  const size_t n3 = (size_t)ncol * nlay * ngpt, n2l = (size_t)ncol * (nlay + 1);
  const size_t scratch = ecckd::rte_lw_scratch_bytes(ncol, nlay, ngpt);
  const hipStream_t launch_stream = memspace == ECCKD_DEVICE ? static_cast<hipStream_t>(stream) : nullptr;
  ScratchLease lease;   // (held until the kernels of this call have been launched)
  if (scratch) {   // more than 96 layers: stream-ordered scratch ring, no synchronisation (see ScratchPool)
    void *sp = nullptr;
    if (stream_scratch(device, launch_stream, scratch, &sp, lease)) return 1;
    a.scratch = static_cast<double *>(sp);
  } else if (g_opt.lw_tail_split.load()) {   // tail tiles one g-pair per wave (rte_lw_tail_plan); optional, same bits
    long first = -1;
    const size_t need = ecckd::rte_lw_tail_plan(a, simd_slots(device), &first);
    if (need) {
      if (void *sp = stream_scratch_optional(device, launch_stream, need, lease)) {
        a.partials = static_cast<double *>(sp);
        a.tail_first = first;
      }
    }
  }
  if (memspace == ECCKD_DEVICE) {
    a.tau = tau; a.lay_source = lay_source; a.lev_source_inc = lev_source_inc;
    a.lev_source_dec = lev_source_dec; a.sfc_source = sfc_source; a.sfc_emis = sfc_emis;
    a.flux_up = flux_up; a.flux_dn = flux_dn;
    a.inc_flux = g_inc_flux;
    {
      ProfScope prof("rte_lw", launch_stream);
      HIPCHK(ecckd::launch_rte_lw(a, launch_stream));
    }
    return 0;
  }
  if (memspace != ECCKD_HOST && memspace != ECCKD_MIXED) return fail("ecckd: bad memspace");
  // ECCKD_MIXED: tau, the three source arrays and sfc_source are device buffers (what gas_optics left there);
  // sfc_emis / inc_flux come from the host and the fluxes go back to it
  const bool mixed = memspace == ECCKD_MIXED;
  Arena &ar = g_solver_arena[device];
  std::lock_guard<std::mutex> lock(ar.mu);
  const size_t need = (mixed ? 0 : align256(n3 * esz()) * 4 + align256((size_t)ncol * ngpt * esz())) +
                      align256((size_t)ncol * ngpt * esz()) + align256((size_t)ncol * nband * esz()) +
                      align256(n2l * esz()) * 2;
  if (ar.ensure(need)) return 1;
  Bump b(ar.p);
  hipStream_t s = nullptr;
  if (mixed) {
    a.tau = tau; a.lay_source = lay_source; a.lev_source_inc = lev_source_inc; a.lev_source_dec = lev_source_dec;
    a.sfc_source = sfc_source;
  } else {
    double *d_tau = b.take(n3), *d_lay = b.take(n3), *d_inc = b.take(n3), *d_dec = b.take(n3);
    double *d_sfc = b.take((size_t)ncol * ngpt);
    if (h2d(d_tau, tau, n3, s) || h2d(d_lay, lay_source, n3, s) || h2d(d_inc, lev_source_inc, n3, s) ||
        h2d(d_dec, lev_source_dec, n3, s) || h2d(d_sfc, sfc_source, (size_t)ncol * ngpt, s))
      return 1;
    a.tau = d_tau; a.lay_source = d_lay; a.lev_source_inc = d_inc; a.lev_source_dec = d_dec; a.sfc_source = d_sfc;
  }
  double *d_emis = b.take((size_t)ncol * nband), *d_up = b.take(n2l), *d_dn = b.take(n2l);
  double *d_incf = b.take((size_t)ncol * ngpt);
  if (h2d(d_emis, sfc_emis, (size_t)ncol * nband, s)) return 1;
  if (g_inc_flux && h2d(d_incf, g_inc_flux, (size_t)ncol * ngpt, s)) return 1;
  a.sfc_emis = d_emis; a.flux_up = d_up; a.flux_dn = d_dn;
  a.inc_flux = g_inc_flux ? d_incf : nullptr;
  HIPCHK(ecckd::launch_rte_lw(a, s));
  if (d2h(flux_up, d_up, n2l, s) || d2h(flux_dn, d_dn, n2l, s)) return 1;
  HIPCHK(hipStreamSynchronize(s));
  return 0;
}

int ecckd_rte_lw_f32(int device, int ncol, int nlay, int ngpt, int top_at_1, int n_gauss_angles, const float *tau,
                     const float *lay_source, const float *lev_source_inc, const float *lev_source_dec,
                     const float *sfc_source, int nband, const int *band2gpt, const float *sfc_emis,
                     float *flux_up, float *flux_dn, int memspace, void *stream) {
  F32Scope scope;
  auto c = [](const float *p) { return reinterpret_cast<const double *>(p); };
  auto w = [](float *p) { return reinterpret_cast<double *>(p); };
  return ecckd_rte_lw(device, ncol, nlay, ngpt, top_at_1, n_gauss_angles, c(tau), c(lay_source), c(lev_source_inc),
                      c(lev_source_dec), c(sfc_source), nband, band2gpt, c(sfc_emis), w(flux_up), w(flux_dn),
                      memspace, stream);
}

int ecckd_rte_lw_shared_levels(int device, int ncol, int nlay, int ngpt, int top_at_1, int n_gauss_angles,
                               const double *tau, const double *lay_source, const double *lev_source_inc,
                               const double *lev_source_dec, const double *sfc_source, int nband,
                               const int *band2gpt, const double *sfc_emis, double *flux_up, double *flux_dn,
                               int memspace, void *stream) {
  struct Scope {
    Scope() { g_shared_levels = 1; }
    ~Scope() { g_shared_levels = 0; }
  } scope;
  return ecckd_rte_lw(device, ncol, nlay, ngpt, top_at_1, n_gauss_angles, tau, lay_source, lev_source_inc,
                      lev_source_dec, sfc_source, nband, band2gpt, sfc_emis, flux_up, flux_dn, memspace, stream);
}

int ecckd_rte_lw_inc_flux(int device, int ncol, int nlay, int ngpt, int top_at_1, int n_gauss_angles,
                          const double *tau, const double *lay_source, const double *lev_source_inc,
                          const double *lev_source_dec, const double *sfc_source, int nband,
                          const int *band2gpt, const double *sfc_emis, const double *inc_flux, double *flux_up,
                          double *flux_dn, int memspace, void *stream) {
  struct Scope {
    explicit Scope(const double *p) { g_inc_flux = p; }
    ~Scope() { g_inc_flux = nullptr; }
  } scope(inc_flux);
  return ecckd_rte_lw(device, ncol, nlay, ngpt, top_at_1, n_gauss_angles, tau, lay_source, lev_source_inc,
                      lev_source_dec, sfc_source, nband, band2gpt, sfc_emis, flux_up, flux_dn, memspace, stream);
}

int ecckd_rte_sw(int device, int ncol, int nlay, int ngpt, int top_at_1, const double *tau,
                 const double *ssa, const double *g, const double *mu0, const double *toa_flux,
                 int nband, const int *band2gpt, const double *sfc_alb_dir,
                 const double *sfc_alb_dif, double *flux_up, double *flux_dn, double *flux_dir,
                 int memspace, void *stream) {
  if (check_dims(ncol, nlay)) return 1;
  if (!tau || !mu0 || !sfc_alb_dir || !sfc_alb_dif || !flux_up || !flux_dn || (!g_sw_derive && (!ssa || !g || !toa_flux)))
    return fail("ecckd_rte_sw: null argument");
  ecckd::RteSwArgs a{};
  if (g_band_override >= 0) std::memset(a.gpt2band, g_band_override, sizeof a.gpt2band);
  else if (fill_band_map(ngpt, nband, band2gpt, a.gpt2band)) return 1;
  if (check_device(device)) return 1;
  if (ncol == 0) return 0;
  a.ncol = ncol; a.nlay = nlay; a.ng = ngpt; a.top_at_1 = top_at_1 ? 1 : 0; a.nband = nband;
  a.exact_division = g_arith.load() != 0;
  a.k_floor = g_opt.sw_k_floor.load();
  // (fast arithmetic mode: sw_sqrt() of sw_two_stream.hpp takes normal numbers; a subnormal floor changes nothing a flux can show)
  if (!a.exact_division && a.k_floor < 2.2250738585072014e-308) a.k_floor = 2.2250738585072014e-308;
  a.dir_clamp = g_opt.sw_dir_clamp.load();
  a.f32 = g_f32;
  if (g_sw_derive) {   // ecckd_sw_fluxes: ssa / g / toa derived inside the solver (RteSwArgs::derive)
    a.derive = 1;
    a.plev = g_sw_derive->plev; a.rayleigh = g_sw_derive->rayleigh; a.solar = g_sw_derive->solar; a.gw = g_sw_derive->gw;
    a.toa_scale = g_sw_derive->toa_scale;
  }
  const size_t n3 = (size_t)ncol * nlay * ngpt, n2l = (size_t)ncol * (nlay + 1);
  const hipStream_t launch_stream = memspace == ECCKD_DEVICE ? static_cast<hipStream_t>(stream) : nullptr;
  ScratchLease lease;   // (held until the kernels of this call have been launched)
  const int cus = simd_slots(device) / 4;
  a.use_sys = g_opt.sw_solver.load() == 0 && ecckd::rte_sw_sys_applies(a) && cus > 0;
  if ((a.f32 || a.derive) && !a.use_sys)
    return fail("ecckd_rte_sw: single precision and the fused shortwave path need the layer-systolic solver (sw_solver = 0, at most 60 layers)");
  if (a.use_sys) {
    // layer-systolic solver: no scratch ring; the g-point chunks of the last part-empty round go through partial sums
    if (g_opt.sw_tail_split.load()) {
      const size_t need = ecckd::rte_sw_sys_plan(a, cus);
      if (need) {
        if (g_sw_derive) a.partials = g_sw_partials;   // (ecckd_sw_fluxes sized its block with ecckd_rte_sw_tail_scratch_bytes)
        else if (void *sp = stream_scratch_optional(device, launch_stream, need, lease)) a.partials = static_cast<double *>(sp);
        if (!a.partials) { a.sys_tail_first = -1; a.sys_gchunk = 0; }
      }
    }
  } else {
  const size_t scratch = ecckd::rte_sw_scratch_bytes(ncol, nlay, ngpt);
  if (g_opt.sw_tail_split.load()) {   // tail tiles one g-point group per wave (rte_sw_tail_plan); optional, same bits
    long first = -1;
    size_t partials_at = 0;
    const size_t need = ecckd::rte_sw_tail_plan(a, &first, &partials_at);
    if (need) {
      if (void *sp = stream_scratch_optional(device, launch_stream, need > scratch ? need : scratch, lease)) {
        a.scratch = static_cast<double *>(sp);
        a.partials = reinterpret_cast<double *>(static_cast<char *>(sp) + partials_at);
        a.tail_first = first;
      }
    }
  }
  if (scratch && a.tail_first < 0) {   // stream-ordered scratch ring, no synchronisation (see ScratchPool)
    void *sp = nullptr;
    if (stream_scratch(device, launch_stream, scratch, &sp, lease)) return 1;
    a.scratch = static_cast<double *>(sp);
  }
  }
  auto launch = [&](hipStream_t st) { return a.use_sys ? ecckd::launch_rte_sw_sys(a, cus, st) : ecckd::launch_rte_sw(a, st); };
  if (memspace == ECCKD_DEVICE) {
    a.tau = tau; a.ssa = ssa; a.g = g; a.mu0 = mu0; a.toa = toa_flux;
    a.alb_dir = sfc_alb_dir; a.alb_dif = sfc_alb_dif;
    a.flux_up = flux_up; a.flux_dn = flux_dn; a.flux_dir = flux_dir;
    {
      ProfScope prof("rte_sw", launch_stream);
      HIPCHK(launch(launch_stream));
    }
    return 0;
  }
  if (memspace != ECCKD_HOST && memspace != ECCKD_MIXED) return fail("ecckd: bad memspace");
  const bool mixed = memspace == ECCKD_MIXED;   // tau, ssa, g: device buffers; everything else host
  Arena &ar = g_solver_arena[device];
  std::lock_guard<std::mutex> lock(ar.mu);
  const size_t need = (mixed ? 0 : align256(n3 * esz()) * 3) + align256((size_t)ncol * esz()) +
                      align256((size_t)ncol * ngpt * esz()) + align256((size_t)ncol * nband * esz()) * 2 +
                      align256(n2l * esz()) * 3;
  if (ar.ensure(need)) return 1;
  Bump b(ar.p);
  hipStream_t s = nullptr;
  if (mixed) {
    a.tau = tau; a.ssa = ssa; a.g = g;
  } else {
    double *d_tau = b.take(n3), *d_ssa = b.take(n3), *d_g = b.take(n3);
    if (h2d(d_tau, tau, n3, s) || h2d(d_ssa, ssa, n3, s) || h2d(d_g, g, n3, s)) return 1;
    a.tau = d_tau; a.ssa = d_ssa; a.g = d_g;
  }
  double *d_mu0 = b.take(ncol);
  double *d_toa = b.take((size_t)ncol * ngpt), *d_ad = b.take((size_t)ncol * nband), *d_af = b.take((size_t)ncol * nband);
  double *d_up = b.take(n2l), *d_dn = b.take(n2l), *d_dir = b.take(n2l);
  if (h2d(d_mu0, mu0, ncol, s) || h2d(d_toa, toa_flux, (size_t)ncol * ngpt, s) ||
      h2d(d_ad, sfc_alb_dir, (size_t)ncol * nband, s) || h2d(d_af, sfc_alb_dif, (size_t)ncol * nband, s))
    return 1;
  a.mu0 = d_mu0; a.toa = d_toa; a.alb_dir = d_ad; a.alb_dif = d_af;
  a.flux_up = d_up; a.flux_dn = d_dn; a.flux_dir = flux_dir ? d_dir : nullptr;
  HIPCHK(launch(s));
  if (d2h(flux_up, d_up, n2l, s) || d2h(flux_dn, d_dn, n2l, s)) return 1;
  if (flux_dir && d2h(flux_dir, d_dir, n2l, s)) return 1;
  HIPCHK(hipStreamSynchronize(s));
  return 0;
}

// ---- RTE-RRTMGP's kernel-level interfaces: spectral fluxes (ncol,nlay+1,ngpt), sum_broadband ----

namespace {
void fill_gpt_options(ecckd::RteGptArgs &a) {
  const double t = g_opt.lw_tau_thresh.load();
  a.tau_thresh = t > 0. ? t : std::sqrt(2.220446049250313e-16);
  a.series3 = g_opt.lw_series_terms.load() == 3;
  a.inc_isotropic = g_opt.lw_inc_flux_isotropic.load();
  a.k_floor = g_opt.sw_k_floor.load();
  a.dir_clamp = g_opt.sw_dir_clamp.load();
}
}  // namespace

int ecckd_lw_solver_noscat_gpt(int device, int ncol, int nlay, int ngpt, int top_at_1, int nmus, const double *Ds,
                               const double *weights, const double *tau, const double *lay_source,
                               const double *lev_source_inc, const double *lev_source_dec, const double *sfc_emis,
                               const double *sfc_src, const double *inc_flux, double *gpt_flux_up, double *gpt_flux_dn,
                               int memspace, void *stream) {
  if (check_dims(ncol, nlay)) return 1;
  if (ngpt < 1) return fail("ecckd_lw_solver_noscat_gpt: bad ngpt");
  if (nmus < 1 || nmus > 4 || !Ds || !weights) return fail("ecckd_lw_solver_noscat_gpt: 1..4 quadrature angles with Ds and weights");
  if (!tau || !lay_source || !lev_source_inc || !lev_source_dec || !sfc_emis || !sfc_src || !gpt_flux_up || !gpt_flux_dn)
    return fail("ecckd_lw_solver_noscat_gpt: null argument");
  if (check_device(device)) return 1;
  if (ncol == 0) return 0;
  ecckd::RteGptArgs a{};
  a.ncol = ncol; a.nlay = nlay; a.ng = ngpt; a.top_at_1 = top_at_1 ? 1 : 0; a.nmus = nmus;
  for (int k = 0; k < nmus; ++k) { a.Ds[k] = Ds[k]; a.wts[k] = weights[k]; }
  fill_gpt_options(a);
  const size_t n3 = (size_t)ncol * nlay * ngpt, n2 = (size_t)ncol * ngpt, nf = (size_t)ncol * (nlay + 1) * ngpt;
  if (memspace == ECCKD_DEVICE) {
    a.tau = tau; a.lay_source = lay_source; a.lev_source_inc = lev_source_inc; a.lev_source_dec = lev_source_dec;
    a.sfc_emis = sfc_emis; a.sfc_src = sfc_src; a.inc_flux = inc_flux; a.flux_up = gpt_flux_up; a.flux_dn = gpt_flux_dn;
    ProfScope prof("lw_gpt", static_cast<hipStream_t>(stream));
    HIPCHK(ecckd::launch_lw_gpt(a, static_cast<hipStream_t>(stream)));
    return 0;
  }
  if (memspace != ECCKD_HOST) return fail("ecckd: bad memspace");
  Arena &ar = g_solver_arena[device];
  std::lock_guard<std::mutex> lock(ar.mu);
  if (ar.ensure(align256(n3 * 8) * 4 + align256(n2 * 8) * 3 + align256(nf * 8) * 2)) return 1;
  Bump b(ar.p);
  double *d_tau = b.take(n3), *d_lay = b.take(n3), *d_inc = b.take(n3), *d_dec = b.take(n3);
  double *d_emis = b.take(n2), *d_src = b.take(n2), *d_incf = b.take(n2), *d_up = b.take(nf), *d_dn = b.take(nf);
  hipStream_t s = nullptr;
  if (h2d(d_tau, tau, n3, s) || h2d(d_lay, lay_source, n3, s) || h2d(d_inc, lev_source_inc, n3, s) ||
      h2d(d_dec, lev_source_dec, n3, s) || h2d(d_emis, sfc_emis, n2, s) || h2d(d_src, sfc_src, n2, s))
    return 1;
  if (inc_flux && h2d(d_incf, inc_flux, n2, s)) return 1;
  a.tau = d_tau; a.lay_source = d_lay; a.lev_source_inc = d_inc; a.lev_source_dec = d_dec; a.sfc_emis = d_emis;
  a.sfc_src = d_src; a.inc_flux = inc_flux ? d_incf : nullptr; a.flux_up = d_up; a.flux_dn = d_dn;
  HIPCHK(ecckd::launch_lw_gpt(a, s));
  if (d2h(gpt_flux_up, d_up, nf, s) || d2h(gpt_flux_dn, d_dn, nf, s)) return 1;
  HIPCHK(hipStreamSynchronize(s));
  return 0;
}

int ecckd_sw_solver_2stream_gpt(int device, int ncol, int nlay, int ngpt, int top_at_1, const double *tau, const double *ssa,
                                const double *g, const double *mu0, const double *flux_dir_top, const double *inc_flux_dif,
                                const double *sfc_alb_dir, const double *sfc_alb_dif, double *gpt_flux_up, double *gpt_flux_dn,
                                double *gpt_flux_dir, int memspace, void *stream) {
  if (check_dims(ncol, nlay)) return 1;
  if (ngpt < 1) return fail("ecckd_sw_solver_2stream_gpt: bad ngpt");
  if (!tau || !ssa || !g || !mu0 || !flux_dir_top || !sfc_alb_dir || !sfc_alb_dif || !gpt_flux_up || !gpt_flux_dn)
    return fail("ecckd_sw_solver_2stream_gpt: null argument");
  if (check_device(device)) return 1;
  if (ncol == 0) return 0;
  ecckd::RteGptArgs a{};
  a.ncol = ncol; a.nlay = nlay; a.ng = ngpt; a.top_at_1 = top_at_1 ? 1 : 0; a.nmus = 1;
  fill_gpt_options(a);
  const size_t n3 = (size_t)ncol * nlay * ngpt, n2 = (size_t)ncol * ngpt, nf = (size_t)ncol * (nlay + 1) * ngpt;
  if (memspace == ECCKD_DEVICE) {
    a.tau = tau; a.ssa = ssa; a.g = g; a.mu0 = mu0; a.fdir_top = flux_dir_top; a.inc_dif = inc_flux_dif;
    a.alb_dir = sfc_alb_dir; a.alb_dif = sfc_alb_dif; a.flux_up = gpt_flux_up; a.flux_dn = gpt_flux_dn; a.flux_dir = gpt_flux_dir;
    ProfScope prof("sw_gpt", static_cast<hipStream_t>(stream));
    HIPCHK(ecckd::launch_sw_gpt(a, static_cast<hipStream_t>(stream)));
    return 0;
  }
  if (memspace != ECCKD_HOST) return fail("ecckd: bad memspace");
  Arena &ar = g_solver_arena[device];
  std::lock_guard<std::mutex> lock(ar.mu);
  if (ar.ensure(align256(n3 * 8) * 3 + align256((size_t)ncol * 8) + align256(n2 * 8) * 4 + align256(nf * 8) * 3)) return 1;
  Bump b(ar.p);
  double *d_tau = b.take(n3), *d_ssa = b.take(n3), *d_g = b.take(n3), *d_mu0 = b.take(ncol);
  double *d_top = b.take(n2), *d_dif = b.take(n2), *d_ad = b.take(n2), *d_af = b.take(n2);
  double *d_up = b.take(nf), *d_dn = b.take(nf), *d_dir = b.take(nf);
  hipStream_t s = nullptr;
  if (h2d(d_tau, tau, n3, s) || h2d(d_ssa, ssa, n3, s) || h2d(d_g, g, n3, s) || h2d(d_mu0, mu0, ncol, s) ||
      h2d(d_top, flux_dir_top, n2, s) || h2d(d_ad, sfc_alb_dir, n2, s) || h2d(d_af, sfc_alb_dif, n2, s))
    return 1;
  if (inc_flux_dif && h2d(d_dif, inc_flux_dif, n2, s)) return 1;
  a.tau = d_tau; a.ssa = d_ssa; a.g = d_g; a.mu0 = d_mu0; a.fdir_top = d_top; a.inc_dif = inc_flux_dif ? d_dif : nullptr;
  a.alb_dir = d_ad; a.alb_dif = d_af; a.flux_up = d_up; a.flux_dn = d_dn; a.flux_dir = gpt_flux_dir ? d_dir : nullptr;
  HIPCHK(ecckd::launch_sw_gpt(a, s));
  if (d2h(gpt_flux_up, d_up, nf, s) || d2h(gpt_flux_dn, d_dn, nf, s)) return 1;
  if (gpt_flux_dir && d2h(gpt_flux_dir, d_dir, nf, s)) return 1;
  HIPCHK(hipStreamSynchronize(s));
  return 0;
}

int ecckd_sum_broadband(int device, int ncol, int nlev, int ngpt, const double *spectral_flux, double *broadband_flux,
                        int memspace, void *stream) {
  if (ncol < 0 || nlev < 1 || ngpt < 1) return fail("ecckd_sum_broadband: bad dimensions");
  if (!spectral_flux || !broadband_flux) return fail("ecckd_sum_broadband: null argument");
  const size_t n = (size_t)ncol * nlev;
  if (n == 0) return 0;
  if (memspace == ECCKD_HOST) {   // a sum over ngpt planes: not worth a round trip over the bus
    for (size_t i = 0; i < n; ++i) {
      double acc = 0.;
      for (int k = 0; k < ngpt; ++k) acc += spectral_flux[(size_t)k * n + i];
      broadband_flux[i] = acc;
    }
    return 0;
  }
  if (memspace != ECCKD_DEVICE) return fail("ecckd: bad memspace");
  if (check_device(device)) return 1;
  HIPCHK(ecckd::launch_sum_planes(spectral_flux, ngpt, n, broadband_flux, 0, static_cast<hipStream_t>(stream)));
  return 0;
}

// ---- fused longwave: tau-only gas optics + solver that recomputes the Planck sources ----

int ecckd_gas_optics_lw_tau(const ecckd_model_t *m, int ncol, int nlay, const double *plev, const double *tlay, int ngas,
                            const char *gas_names, const double *const *vmr, const long long *vmr_col_stride,
                            const long long *vmr_lay_stride, const double *vmr_scalar, double *tau, int memspace,
                            void *stream) {
  if (check_model(m) || check_gas_optics_dims(ncol, nlay)) return 1;
  if (memspace != ECCKD_DEVICE) return fail("ecckd_gas_optics_lw_tau: device arrays only (ECCKD_DEVICE); ecckd_lw_fluxes takes host arrays");
  if (!plev || !tlay || !tau || (ngas > 0 && !gas_names)) return fail("ecckd_gas_optics_lw_tau: null argument");
  HIPCHK(hipSetDevice(m->device));
  if (ncol == 0) return 0;
  const GasDesc gd{ngas, gas_names, vmr, vmr_col_stride, vmr_lay_stride, vmr_scalar};
  return gas_optical_depth_dev(m, ncol, nlay, plev, tlay, gd, tau, false, nullptr, nullptr, nullptr, nullptr,
                               static_cast<hipStream_t>(stream));
}

// Scratch (in doubles) the fused solver needs besides tau: none at 60 layers (the Planck sources are recomputed inside the
// layer-split solver); any other layer count takes the general route -- Planck kernel into scratch, then the
// register-resident solver with shared level sources -- and needs room for the sources and that solver's ring.
static bool fused_lw_kernels_apply(const ecckd_model *m, int nlay) {
  return nlay == 60 && ecckd::rte_lw_planck_fits(m->ng, m->ntp);   // (a 64-g Planck table does not fit: general route)
}
static size_t fused_scratch_doubles(const ecckd_model *m, int ncol, int nlay) {
  if (fused_lw_kernels_apply(m, nlay)) return 0;
  const size_t n3 = (size_t)ncol * nlay * m->ng;
  return 3 * n3 + (size_t)ncol * m->ng + 32 + ecckd::rte_lw_scratch_bytes(ncol, nlay, m->ng) / sizeof(double);
}

static int rte_lw_fused_dev(const ecckd_model *m, int ncol, int nlay, int top_at_1, int n_gauss_angles, const double *tau,
                            const double *tlay, const double *tlev, const double *tsfc, const double *sfc_emis,
                            const double *inc_flux, double *flux_up, double *flux_dn, double *scratch, hipStream_t stream) {
  ecckd::RteLwArgs a{};
  if (fill_band_map(m->ng, m->nband, m->band2gpt.data(), a.gpt2band)) return 1;
  a.ncol = ncol; a.nlay = nlay; a.ng = m->ng; a.top_at_1 = top_at_1 ? 1 : 0; a.nmus = n_gauss_angles;
  a.nband = m->nband;
  {
    const double t = g_opt.lw_tau_thresh.load();
    a.tau_thresh = t > 0. ? t : std::sqrt(2.220446049250313e-16);
    a.series3 = g_opt.lw_series_terms.load() == 3;
    a.inc_isotropic = g_opt.lw_inc_flux_isotropic.load();
    a.use_split = 1;
    a.split_seg = g_opt.lw_split_seg.load();
  }
  for (int k = 0; k < n_gauss_angles; ++k) {
    a.Ds[k] = kGaussDs[n_gauss_angles - 1][k];
    a.wts[k] = kGaussWts[n_gauss_angles - 1][k];
  }
  a.tau = tau; a.sfc_emis = sfc_emis; a.inc_flux = inc_flux; a.flux_up = flux_up; a.flux_dn = flux_dn;
  if (fused_lw_kernels_apply(m, nlay)) {
    ProfScope prof("rte_lw_fused", stream);
    HIPCHK(ecckd::launch_rte_lw_planck(a, m->dbuf + m->off_planck, m->ntp, m->temperature_planck[0],
                                       m->temperature_planck[1] - m->temperature_planck[0], tlay, tlev, tsfc, stream));
    return 0;
  }
  // general route (any layer count): sources through scratch, one value per level
  if (!scratch) return fail("ecckd: internal: the fused longwave solver needs scratch for this layer count");
  const size_t n3 = (size_t)ncol * nlay * m->ng;
  double *lay = scratch, *inc = lay + n3, *dec = inc + n3, *sfc = dec + n3, *ring = sfc + (((size_t)ncol * m->ng + 31) & ~(size_t)31);
  ecckd::PlanckArgs p{};
  p.ncol = ncol; p.nlay = nlay; p.ng = m->ng; p.ntp = m->ntp;
  p.planck = m->dbuf + m->off_planck;
  p.t0 = m->temperature_planck[0];
  p.dt = m->temperature_planck[1] - m->temperature_planck[0];
  p.tlay = tlay; p.tlev = tlev; p.tsfc = tsfc;
  p.lay_source = lay; p.lev_source_inc = inc; p.lev_source_dec = dec; p.sfc_source = sfc;
  {
    ProfScope prof("planck", stream);
    HIPCHK(ecckd::launch_planck(p, stream));
  }
  a.lay_source = lay; a.lev_source_inc = inc; a.lev_source_dec = dec; a.sfc_source = sfc;
  a.shared_levels = inc_flux ? 0 : 1;
  a.use_split = 0;
  a.scratch = ecckd::rte_lw_scratch_bytes(ncol, nlay, m->ng) ? ring : nullptr;
  ProfScope prof("rte_lw", stream);
  HIPCHK(ecckd::launch_rte_lw(a, stream));
  return 0;
}

int ecckd_rte_lw_fused(const ecckd_model_t *m, int ncol, int nlay, int top_at_1, int n_gauss_angles, const double *tau,
                       const double *tlay, const double *tlev, const double *tsfc, const double *sfc_emis,
                       const double *inc_flux, double *flux_up, double *flux_dn, int memspace, void *stream) {
  if (check_model(m) || check_dims(ncol, nlay)) return 1;
  if (!m->has_planck) return fail("ecckd_rte_lw_fused: model has no Planck table (shortwave model?)");
  if (memspace != ECCKD_DEVICE) return fail("ecckd_rte_lw_fused: device arrays only (ECCKD_DEVICE); ecckd_lw_fluxes takes host arrays");
  if (n_gauss_angles < 1 || n_gauss_angles > 4) return fail("rte_lw: have to ask for at least one quadrature point and no more than 4");
  if (!tau || !tlay || !tsfc || !sfc_emis || !flux_up || !flux_dn) return fail("ecckd_rte_lw_fused: null argument");
  if (!tlev) return fail("tlev is required for ecckd");
  HIPCHK(hipSetDevice(m->device));
  if (ncol == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  void *sp = nullptr;
  ScratchLease lease;
  const size_t extra = fused_scratch_doubles(m, ncol, nlay);
  if (extra && stream_scratch(m->device, st, extra * sizeof(double), &sp, lease)) return 1;
  return rte_lw_fused_dev(m, ncol, nlay, top_at_1, n_gauss_angles, tau, tlay, tlev, tsfc, sfc_emis, inc_flux, flux_up,
                          flux_dn, static_cast<double *>(sp), st);
}

int ecckd_lw_fluxes(const ecckd_model_t *m, int ncol, int nlay, const double *plev, const double *tlay, const double *tsfc,
                    const double *tlev, int ngas, const char *gas_names, const double *const *vmr,
                    const long long *vmr_col_stride, const long long *vmr_lay_stride, const double *vmr_scalar, int top_at_1,
                    int n_gauss_angles, const double *sfc_emis, const double *inc_flux, double *flux_up, double *flux_dn,
                    int memspace, void *stream) {
  if (check_model(m) || check_gas_optics_dims(ncol, nlay)) return 1;
  if (!m->has_planck) return fail("ecckd_lw_fluxes: model has no Planck table (shortwave model?)");
  if (n_gauss_angles < 1 || n_gauss_angles > 4) return fail("rte_lw: have to ask for at least one quadrature point and no more than 4");
  if (!plev || !tlay || !tsfc || !sfc_emis || !flux_up || !flux_dn || (ngas > 0 && !gas_names)) return fail("ecckd_lw_fluxes: null argument");
  if (!tlev) return fail("tlev is required for ecckd");
  if (g_arith.load() != 0) return fail("ecckd_lw_fluxes: needs the fast arithmetic mode (ecckd_set_arithmetic(0))");
  HIPCHK(hipSetDevice(m->device));
  if (ncol == 0) return 0;
  const GasDesc gd{ngas, gas_names, vmr, vmr_col_stride, vmr_lay_stride, vmr_scalar};
  const size_t n2 = (size_t)ncol * nlay, n2l = (size_t)ncol * (nlay + 1), n3 = n2 * m->ng;
  if (memspace == ECCKD_DEVICE) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    void *tau_p = nullptr;   // tau lives in the stream's scratch block between the two kernels
    ScratchLease lease;
    const size_t extra = fused_scratch_doubles(m, ncol, nlay);
    if (stream_scratch(m->device, st, (n3 + 32 + extra) * sizeof(double), &tau_p, lease)) return 1;
    double *d_tau = static_cast<double *>(tau_p);
    if (gas_optical_depth_dev(m, ncol, nlay, plev, tlay, gd, d_tau, false, nullptr, nullptr, nullptr, nullptr, st)) return 1;
    return rte_lw_fused_dev(m, ncol, nlay, top_at_1, n_gauss_angles, d_tau, tlay, tlev, tsfc, sfc_emis, inc_flux, flux_up,
                            flux_dn, extra ? d_tau + ((n3 + 31) & ~(size_t)31) : nullptr, st);
  }
  if (memspace != ECCKD_HOST) return fail("ecckd: bad memspace");
  ecckd_model *mm = const_cast<ecckd_model *>(m);
  std::lock_guard<std::mutex> lock(mm->mu);
  hipStream_t s = mm->host_stream;
  const size_t need = align256(n2l * 8) * 4 + align256(n2 * 8) + align256((size_t)ncol * 8) + staged_gas_bytes(gd, ncol, nlay) +
                      align256((size_t)ncol * m->nband * 8) + align256((size_t)ncol * m->ng * 8) + align256(n3 * 8) +
                      align256(fused_scratch_doubles(m, ncol, nlay) * 8);
  if (need > mm->arena_bytes) {
    HIPCHK(hipStreamSynchronize(s));
    if (mm->arena) { HIPCHK(hipFree(mm->arena)); mm->arena = nullptr; mm->arena_bytes = 0; }
    HIPCHK(hipMalloc(&mm->arena, need));
    mm->arena_bytes = need;
  }
  Bump b(mm->arena);
  double *d_plev = b.take(n2l), *d_tlev = b.take(n2l), *d_tlay = b.take(n2), *d_tsfc = b.take(ncol);
  double *d_up = b.take(n2l), *d_dn = b.take(n2l), *d_emis = b.take((size_t)ncol * m->nband), *d_incf = b.take((size_t)ncol * m->ng);
  if (h2d(d_plev, plev, n2l, s) || h2d(d_tlay, tlay, n2, s) || h2d(d_tsfc, tsfc, ncol, s) || h2d(d_tlev, tlev, n2l, s) ||
      h2d(d_emis, sfc_emis, (size_t)ncol * m->nband, s))
    return 1;
  if (inc_flux && h2d(d_incf, inc_flux, (size_t)ncol * m->ng, s)) return 1;
  StagedGases sg;
  if (stage_gases(gd, ncol, nlay, b, s, sg)) return 1;
  double *d_tau = b.take(n3);
  const size_t extra = fused_scratch_doubles(m, ncol, nlay);
  double *d_extra = extra ? b.take(extra) : nullptr;
  if (gas_optical_depth_dev(m, ncol, nlay, d_plev, d_tlay, sg.gd, d_tau, false, nullptr, nullptr, nullptr, nullptr, s)) return 1;
  if (rte_lw_fused_dev(m, ncol, nlay, top_at_1, n_gauss_angles, d_tau, d_tlay, d_tlev, d_tsfc, d_emis, inc_flux ? d_incf : nullptr,
                       d_up, d_dn, d_extra, s))
    return 1;
  if (d2h(flux_up, d_up, n2l, s) || d2h(flux_dn, d_dn, n2l, s)) return 1;
  HIPCHK(hipStreamSynchronize(s));
  return 0;
}

// ---- single-precision flavours of the shortwave pair and of the incident-flux longwave solver ----

int ecckd_gas_optics_sw_f32(const ecckd_model_t *m, int ncol, int nlay, const float *plev, const float *tlay, int ngas,
                            const char *gas_names, const float *const *vmr, const long long *vmr_col_stride,
                            const long long *vmr_lay_stride, const double *vmr_scalar, float *tau, float *ssa, float *g,
                            float *toa_src, int memspace, void *stream) {
  F32Scope scope;
  auto c = [](const float *p) { return reinterpret_cast<const double *>(p); };
  auto w = [](float *p) { return reinterpret_cast<double *>(p); };
  return ecckd_gas_optics_sw(m, ncol, nlay, c(plev), c(tlay), ngas, gas_names, reinterpret_cast<const double *const *>(vmr),
                             vmr_col_stride, vmr_lay_stride, vmr_scalar, w(tau), w(ssa), w(g), w(toa_src), memspace, stream);
}

int ecckd_rte_sw_f32(int device, int ncol, int nlay, int ngpt, int top_at_1, const float *tau, const float *ssa, const float *g,
                     const float *mu0, const float *toa_flux, int nband, const int *band2gpt, const float *sfc_alb_dir,
                     const float *sfc_alb_dif, float *flux_up, float *flux_dn, float *flux_dir, int memspace, void *stream) {
  F32Scope scope;
  auto c = [](const float *p) { return reinterpret_cast<const double *>(p); };
  auto w = [](float *p) { return reinterpret_cast<double *>(p); };
  return ecckd_rte_sw(device, ncol, nlay, ngpt, top_at_1, c(tau), c(ssa), c(g), c(mu0), c(toa_flux), nband, band2gpt,
                      c(sfc_alb_dir), c(sfc_alb_dif), w(flux_up), w(flux_dn), w(flux_dir), memspace, stream);
}

int ecckd_rte_lw_inc_flux_f32(int device, int ncol, int nlay, int ngpt, int top_at_1, int n_gauss_angles, const float *tau,
                              const float *lay_source, const float *lev_source_inc, const float *lev_source_dec,
                              const float *sfc_source, int nband, const int *band2gpt, const float *sfc_emis,
                              const float *inc_flux, float *flux_up, float *flux_dn, int memspace, void *stream) {
  F32Scope scope;
  auto c = [](const float *p) { return reinterpret_cast<const double *>(p); };
  auto w = [](float *p) { return reinterpret_cast<double *>(p); };
  return ecckd_rte_lw_inc_flux(device, ncol, nlay, ngpt, top_at_1, n_gauss_angles, c(tau), c(lay_source), c(lev_source_inc),
                               c(lev_source_dec), c(sfc_source), nband, band2gpt, c(sfc_emis), c(inc_flux), w(flux_up),
                               w(flux_dn), memspace, stream);
}

// ---- fused shortwave: total optical depth only between the kernels (SURVEY 8(f) rank 4 for the shortwave) ----

int ecckd_sw_fluxes(const ecckd_model_t *m, int ncol, int nlay, const double *plev, const double *tlay, int ngas,
                    const char *gas_names, const double *const *vmr, const long long *vmr_col_stride,
                    const long long *vmr_lay_stride, const double *vmr_scalar, int top_at_1, const double *mu0,
                    const double *toa_scale, const double *sfc_alb_dir, const double *sfc_alb_dif, double *flux_up,
                    double *flux_dn, double *flux_dir, int memspace, void *stream) {
  if (check_model(m) || check_gas_optics_dims(ncol, nlay)) return 1;
  if (!m->has_solar) return fail("ecckd_sw_fluxes: model has no solar table (longwave model?)");
  if (!plev || !tlay || !mu0 || !sfc_alb_dir || !sfc_alb_dif || !flux_up || !flux_dn || (ngas > 0 && !gas_names))
    return fail("ecckd_sw_fluxes: null argument");
  if (g_arith.load() != 0) return fail("ecckd_sw_fluxes: needs the fast arithmetic mode (ecckd_set_arithmetic(0))");
  if (nlay > 60 || g_opt.sw_solver.load() != 0)
    return fail("ecckd_sw_fluxes: needs the layer-systolic shortwave solver (sw_solver = 0, at most 60 layers)");
  HIPCHK(hipSetDevice(m->device));
  if (ncol == 0) return 0;
  const GasDesc gd{ngas, gas_names, vmr, vmr_col_stride, vmr_lay_stride, vmr_scalar};
  const size_t n2 = (size_t)ncol * nlay, n2l = (size_t)ncol * (nlay + 1), n3 = n2 * m->ng;
  // :107 / :314 with default-real literals (:51-52), as gas_optical_depth_dev computes it
  double gw = 1. / ((double)9.80665f * (double)0.001f * (double)28.970f);
  if (g_f32) gw = (double)(1.f / (9.80665f * 0.001f * 28.970f));
  auto tabs = [&](size_t off) { return g_f32 ? reinterpret_cast<const double *>(m->dbuf32 + off) : m->dbuf + off; };
  struct Scope {
    explicit Scope(const SwDerive *d) { g_sw_derive = d; }
    ~Scope() { g_sw_derive = nullptr; }
  };
  if (memspace == ECCKD_DEVICE) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    void *tau_p = nullptr;   // tau lives in the stream's scratch block between the two kernels; the solver's partial sums behind it
    ScratchLease lease;
    const size_t tau_bytes = align256(n3 * esz());
    const size_t tail = ecckd_rte_sw_tail_scratch_bytes(m->device, ncol, nlay, m->ng);
    if (stream_scratch(m->device, st, tau_bytes + tail, &tau_p, lease)) return 1;
    double *d_tau = static_cast<double *>(tau_p);
    // gas_optics_ext's tau (:449-456) without ssa / g: the total optical depth, gases + Rayleigh
    if (gas_optical_depth_dev(m, ncol, nlay, plev, tlay, gd, d_tau, true, nullptr, nullptr, nullptr, nullptr, st)) return 1;
    const SwDerive dv{plev, tabs(m->off_rayleigh), tabs(m->off_solar), toa_scale, gw};
    Scope scope(&dv);
    g_sw_partials = tail ? reinterpret_cast<double *>(static_cast<char *>(tau_p) + tau_bytes) : nullptr;
    const int rc = ecckd_rte_sw(m->device, ncol, nlay, m->ng, top_at_1, d_tau, nullptr, nullptr, mu0, nullptr, m->nband,
                                m->band2gpt.data(), sfc_alb_dir, sfc_alb_dif, flux_up, flux_dn, flux_dir, ECCKD_DEVICE, stream);
    g_sw_partials = nullptr;
    return rc;
  }
  if (memspace != ECCKD_HOST) return fail("ecckd: bad memspace");
  ecckd_model *mm = const_cast<ecckd_model *>(m);
  std::lock_guard<std::mutex> lock(mm->mu);
  hipStream_t s = mm->host_stream;
  const size_t tail = ecckd_rte_sw_tail_scratch_bytes(m->device, ncol, nlay, m->ng);
  const size_t need = align256(n2l * esz()) * 4 + align256(n2 * esz()) + align256((size_t)ncol * esz()) * 2 +
                      staged_gas_bytes(gd, ncol, nlay) + align256((size_t)ncol * m->nband * esz()) * 2 + align256(n3 * esz()) +
                      align256(tail);
  if (need > mm->arena_bytes) {
    HIPCHK(hipStreamSynchronize(s));
    if (mm->arena) { HIPCHK(hipFree(mm->arena)); mm->arena = nullptr; mm->arena_bytes = 0; }
    HIPCHK(hipMalloc(&mm->arena, need));
    mm->arena_bytes = need;
  }
  Bump b(mm->arena);
  double *d_plev = b.take(n2l), *d_tlay = b.take(n2), *d_mu0 = b.take(ncol), *d_scale = b.take(ncol);
  double *d_up = b.take(n2l), *d_dn = b.take(n2l), *d_dir = b.take(n2l);
  double *d_ad = b.take((size_t)ncol * m->nband), *d_af = b.take((size_t)ncol * m->nband);
  if (h2d(d_plev, plev, n2l, s) || h2d(d_tlay, tlay, n2, s) || h2d(d_mu0, mu0, ncol, s) ||
      h2d(d_ad, sfc_alb_dir, (size_t)ncol * m->nband, s) || h2d(d_af, sfc_alb_dif, (size_t)ncol * m->nband, s))
    return 1;
  if (toa_scale && h2d(d_scale, toa_scale, ncol, s)) return 1;
  StagedGases sg;
  if (stage_gases(gd, ncol, nlay, b, s, sg)) return 1;
  double *d_tau = b.take(n3);
  double *d_part = tail ? b.take((tail + esz() - 1) / esz()) : nullptr;
  if (gas_optical_depth_dev(m, ncol, nlay, d_plev, d_tlay, sg.gd, d_tau, true, nullptr, nullptr, nullptr, nullptr, s)) return 1;
  const SwDerive dv{d_plev, tabs(m->off_rayleigh), tabs(m->off_solar), toa_scale ? d_scale : nullptr, gw};
  Scope scope(&dv);
  g_sw_partials = d_part;
  const int rc = ecckd_rte_sw(m->device, ncol, nlay, m->ng, top_at_1, d_tau, nullptr, nullptr, d_mu0, nullptr, m->nband,
                              m->band2gpt.data(), d_ad, d_af, d_up, d_dn, flux_dir ? d_dir : nullptr, ECCKD_DEVICE, s);
  g_sw_partials = nullptr;
  if (rc) return 1;
  if (d2h(flux_up, d_up, n2l, s) || d2h(flux_dn, d_dn, n2l, s)) return 1;
  if (flux_dir && d2h(flux_dir, d_dir, n2l, s)) return 1;
  HIPCHK(hipStreamSynchronize(s));
  return 0;
}

int ecckd_sw_fluxes_f32(const ecckd_model_t *m, int ncol, int nlay, const float *plev, const float *tlay, int ngas,
                        const char *gas_names, const float *const *vmr, const long long *vmr_col_stride,
                        const long long *vmr_lay_stride, const double *vmr_scalar, int top_at_1, const float *mu0,
                        const float *toa_scale, const float *sfc_alb_dir, const float *sfc_alb_dif, float *flux_up,
                        float *flux_dn, float *flux_dir, int memspace, void *stream) {
  F32Scope scope;
  auto c = [](const float *p) { return reinterpret_cast<const double *>(p); };
  auto w = [](float *p) { return reinterpret_cast<double *>(p); };
  return ecckd_sw_fluxes(m, ncol, nlay, c(plev), c(tlay), ngas, gas_names, reinterpret_cast<const double *const *>(vmr),
                         vmr_col_stride, vmr_lay_stride, vmr_scalar, top_at_1, c(mu0), c(toa_scale), c(sfc_alb_dir),
                         c(sfc_alb_dif), w(flux_up), w(flux_dn), w(flux_dir), memspace, stream);
}

// ---- spectral (per-band) fluxes: ty_fluxes_byband of RTE-RRTMGP ----
namespace {
struct BandScope {
  explicit BandScope(int b) { g_band_override = b; }
  ~BandScope() { g_band_override = -1; }
};
// out(:) = sum over planes of planes(:, b), on the device or on the host
int sum_planes(const double *planes, int nplanes, size_t n, double *out, int memspace, void *stream) {
  if (memspace == ECCKD_DEVICE) {
    HIPCHK(ecckd::launch_sum_planes(planes, nplanes, n, out, g_f32, static_cast<hipStream_t>(stream)));
    return 0;
  }
  if (g_f32) {   // (the single-precision flavours: float data behind the double pointers)
    const float *pf = reinterpret_cast<const float *>(planes);
    float *of = reinterpret_cast<float *>(out);
    for (size_t i = 0; i < n; ++i) {
      float acc = 0.f;
      for (int b = 0; b < nplanes; ++b) acc += pf[(size_t)b * n + i];
      of[i] = acc;
    }
    return 0;
  }
  for (size_t i = 0; i < n; ++i) {
    double acc = 0.;
    for (int b = 0; b < nplanes; ++b) acc += planes[(size_t)b * n + i];
    out[i] = acc;
  }
  return 0;
}
// p + n elements of the precision of the call (float data sits behind the double pointers of the _f32 flavours)
inline const double *el(const double *p, size_t n) {
  return g_f32 ? reinterpret_cast<const double *>(reinterpret_cast<const float *>(p) + n) : p + n;
}
inline double *elw(double *p, size_t n) { return g_f32 ? reinterpret_cast<double *>(reinterpret_cast<float *>(p) + n) : p + n; }
}  // namespace

int ecckd_rte_lw_byband(int device, int ncol, int nlay, int ngpt, int top_at_1, int n_gauss_angles,
                        const double *tau, const double *lay_source, const double *lev_source_inc,
                        const double *lev_source_dec, const double *sfc_source, int nband, const int *band2gpt,
                        const double *sfc_emis, double *bnd_flux_up, double *bnd_flux_dn, double *flux_up,
                        double *flux_dn, int memspace, void *stream) {
  if (check_dims(ncol, nlay)) return 1;
  unsigned char map[256];
  if (fill_band_map(ngpt, nband, band2gpt, map)) return 1;
  if (!bnd_flux_up || !bnd_flux_dn) return fail("ecckd_rte_lw_byband: null argument");
  const size_t n3 = (size_t)ncol * nlay, n2l = (size_t)ncol * (nlay + 1);
  for (int b = 0; b < nband; ++b) {   // one solver pass per band over its (contiguous) g-points
    const int g0 = band2gpt[2 * b] - 1, n = band2gpt[2 * b + 1] - g0;
    const int one_band[2] = {1, n};
    BandScope scope(b);
    if (ecckd_rte_lw(device, ncol, nlay, n, top_at_1, n_gauss_angles, el(tau, n3 * g0), el(lay_source, n3 * g0),
                     el(lev_source_inc, n3 * g0), el(lev_source_dec, n3 * g0), el(sfc_source, (size_t)ncol * g0), nband,
                     one_band, sfc_emis, elw(bnd_flux_up, n2l * b), elw(bnd_flux_dn, n2l * b), memspace, stream))
      return 1;
  }
  if (ncol == 0) return 0;
  if (flux_up && sum_planes(bnd_flux_up, nband, n2l, flux_up, memspace, stream)) return 1;
  if (flux_dn && sum_planes(bnd_flux_dn, nband, n2l, flux_dn, memspace, stream)) return 1;
  return 0;
}

int ecckd_rte_sw_byband(int device, int ncol, int nlay, int ngpt, int top_at_1, const double *tau,
                        const double *ssa, const double *g, const double *mu0, const double *toa_flux, int nband,
                        const int *band2gpt, const double *sfc_alb_dir, const double *sfc_alb_dif,
                        double *bnd_flux_up, double *bnd_flux_dn, double *bnd_flux_dir, double *flux_up,
                        double *flux_dn, double *flux_dir, int memspace, void *stream) {
  if (check_dims(ncol, nlay)) return 1;
  unsigned char map[256];
  if (fill_band_map(ngpt, nband, band2gpt, map)) return 1;
  if (!bnd_flux_up || !bnd_flux_dn) return fail("ecckd_rte_sw_byband: null argument");
  if (flux_dir && !bnd_flux_dir) return fail("ecckd_rte_sw_byband: flux_dir needs bnd_flux_dir");
  const size_t n3 = (size_t)ncol * nlay, n2l = (size_t)ncol * (nlay + 1);
  for (int b = 0; b < nband; ++b) {
    const int g0 = band2gpt[2 * b] - 1, n = band2gpt[2 * b + 1] - g0;
    const int one_band[2] = {1, n};
    BandScope scope(b);
    if (ecckd_rte_sw(device, ncol, nlay, n, top_at_1, el(tau, n3 * g0), el(ssa, n3 * g0), el(g, n3 * g0), mu0,
                     el(toa_flux, (size_t)ncol * g0), nband, one_band, sfc_alb_dir, sfc_alb_dif, elw(bnd_flux_up, n2l * b),
                     elw(bnd_flux_dn, n2l * b), bnd_flux_dir ? elw(bnd_flux_dir, n2l * b) : nullptr, memspace, stream))
      return 1;
  }
  if (ncol == 0) return 0;
  if (flux_up && sum_planes(bnd_flux_up, nband, n2l, flux_up, memspace, stream)) return 1;
  if (flux_dn && sum_planes(bnd_flux_dn, nband, n2l, flux_dn, memspace, stream)) return 1;
  if (flux_dir && sum_planes(bnd_flux_dir, nband, n2l, flux_dir, memspace, stream)) return 1;
  return 0;
}

// Single-precision flavours of the per-band solvers (the reference is precision-generic through `wp`).
int ecckd_rte_lw_byband_f32(int device, int ncol, int nlay, int ngpt, int top_at_1, int n_gauss_angles, const float *tau,
                            const float *lay_source, const float *lev_source_inc, const float *lev_source_dec,
                            const float *sfc_source, int nband, const int *band2gpt, const float *sfc_emis,
                            float *bnd_flux_up, float *bnd_flux_dn, float *flux_up, float *flux_dn, int memspace, void *stream) {
  F32Scope scope;
  auto c = [](const float *p) { return reinterpret_cast<const double *>(p); };
  auto w = [](float *p) { return reinterpret_cast<double *>(p); };
  return ecckd_rte_lw_byband(device, ncol, nlay, ngpt, top_at_1, n_gauss_angles, c(tau), c(lay_source), c(lev_source_inc),
                             c(lev_source_dec), c(sfc_source), nband, band2gpt, c(sfc_emis), w(bnd_flux_up), w(bnd_flux_dn),
                             w(flux_up), w(flux_dn), memspace, stream);
}

int ecckd_rte_sw_byband_f32(int device, int ncol, int nlay, int ngpt, int top_at_1, const float *tau, const float *ssa,
                            const float *g, const float *mu0, const float *toa_flux, int nband, const int *band2gpt,
                            const float *sfc_alb_dir, const float *sfc_alb_dif, float *bnd_flux_up, float *bnd_flux_dn,
                            float *bnd_flux_dir, float *flux_up, float *flux_dn, float *flux_dir, int memspace, void *stream) {
  F32Scope scope;
  auto c = [](const float *p) { return reinterpret_cast<const double *>(p); };
  auto w = [](float *p) { return reinterpret_cast<double *>(p); };
  return ecckd_rte_sw_byband(device, ncol, nlay, ngpt, top_at_1, c(tau), c(ssa), c(g), c(mu0), c(toa_flux), nband, band2gpt,
                             c(sfc_alb_dir), c(sfc_alb_dif), w(bnd_flux_up), w(bnd_flux_dn), w(bnd_flux_dir), w(flux_up),
                             w(flux_dn), w(flux_dir), memspace, stream);
}

}  // extern "C"
