// kernels_gas_fused.hip -- gas optics in one pass: optical depth of all gases AND (longwave) the
// three Planck source arrays, or (shortwave) the Rayleigh/ssa/g epilogue.
//
// Replaces, for one call of gas_optics_int / gas_optics_ext (src/gas_optics_ecckd.f90:381-473):
// calculate_optical_depth x ngas + the tau accumulation (:64-241, :348-374),
// calculate_planck_function x 3 (:245-289, :407-424) or the Rayleigh epilogue (:293-319, :455-460).
//
// Why one kernel: the interpolation is bound by fp64 VALU issue (61 fp64 operations and 21 16-byte
// LDS reads per cell pair), the Planck sources by HBM stores (24 B/cell).  Run back to back they
// take the sum; fused, most of the arithmetic hides under the stores (32 B/cell written: tau,
// lay_source, lev_source_inc, lev_source_dec).
//
// Arithmetic ("fast" mode, the default): the four (eight) interpolation weights of a cell are
// multiplied out once per (column,layer) and every coefficient costs one FMA,
//     od_gas(g) = w_gas * (a00*c00 + a10*c10 + a01*c01 + a11*c11),   a_pt = tw_t * pw_p,
// the look_up_table gas is summed first and the other gases in gas_desc order.  That is the
// reference's formula re-associated: tau agrees with the reference-order kernels
// (kernels_tau.hip, kept as the bit-faithful mode) to a few ulp; the stated test tolerance is
// 1e-12 relative.  Per-gas clamping of negative optical depths (:234-238) is kept exactly:
// tables without negative entries (all ecCKD files) clamp the weight instead, tables with
// negative entries take the ANYCLAMP instantiation.  Planck sources are computed in the
// reference's order and are bit-identical.
//
// Mapping: lane -> column, block = kBlock columns of ONE layer (grid.y), LDS = slab of R pressure
// rows of every active table + the Planck table (or a window of it), row strides == 2 (mod 4)
// doubles.  Per segment of kSeg tiles a pre-pass finds the pressure-row (and Planck-row) range and
// places the slab; segments spanning more rows than the slab holds are walked once per slab
// position; waves with a lane outside the staged rows read the tables from global memory.
#include <cstdlib>
#include <type_traits>

#include "kernels.hpp"
#include "wave_pair.hpp"

namespace ecckd {
namespace {

#ifndef ECCKD_FUSED_BLOCK
#define ECCKD_FUSED_BLOCK 512
#endif
constexpr int kBlock = ECCKD_FUSED_BLOCK;   // threads of a block = columns of a tile in the longwave mode; the others take kBlockSw
// The shortwave and tau-only modes (no Planck items: 162-185 VGPRs at two waves per SIMD) fit three waves per SIMD with few
// or no spills (168 VGPRs; 6-28 spilled registers in fp64, outside the loops; none in single precision): blocks of 768
// threads.  Same box, profiles/r03_ab_gas_blocks_f32.txt, r03_ab_gas_blocks2.txt: shortwave gas optics 1.29 -> 1.10 ms in
// fp64 and 1.17 -> 0.89 ms in fp32 at 1e5 columns, tau-only 7.38 -> 7.10 ms at 1e6.  The longwave shape cannot afford the
// registers (102 spilled: +10 %; single precision: no gain) and keeps 512.
#ifndef ECCKD_FUSED_BLOCK_SW
#define ECCKD_FUSED_BLOCK_SW 768
#endif
constexpr int kBlockSw = ECCKD_FUSED_BLOCK_SW;
#ifndef ECCKD_FUSED_SEG
#define ECCKD_FUSED_SEG 8
#endif
#ifndef ECCKD_FUSED_SEG_MAX
#define ECCKD_FUSED_SEG_MAX 64
#endif
// Tiles between two slab-range checks (pre-pass + block barriers).  kSeg sizes the grid (enough blocks for small
// column counts); the segment a block actually walks is TauArgs::seg = its whole tile range up to kSegMax -- measured at
// 1e6 columns: 4 / 8 / 16 / 32 tiles per segment = 13.51 / 13.27 / 13.06 / 12.92 ms.
constexpr int kSeg = ECCKD_FUSED_SEG;
constexpr int kSegMax = ECCKD_FUSED_SEG_MAX;

// Compile-time loop: f(integral_constant<int, I>) for I = I0 .. N-1, as straight-line code.  The
// item pipeline below must be fully unrolled (its buffer indices and item kinds are static);
// `#pragma unroll` silently gives up on a body this large.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Order of the items of a chunk.  Item INDICES are: [0, NLI) look_up_table gas (g-pair p, vmr plane h: index 2p + h),
// [NLI, NLI+NBI) bilinear slots, then NPI Planck items (g-pair p: index p; first layer only: NP2 + p).  Every g-pair ends
// in stores: tau once its last slot is in, three or four source planes per Planck item.  ECCKD_FUSED_INTERLEAVE:
//   0  slot-major: all look_up_table items, all bilinear items (slot outer, g-pair inner), all Planck items -- the stores
//      of a chunk come as one burst of 16-20 KiB per wave at its end;
//   1  (default) the same with the Planck items spread between the bilinear items: -0.4 % in fp64, -5.5 % in fp32;
//   2  g-pair-major: per g-pair its look_up_table items, its slots, its Planck item(s) -- four stores every tenth item;
//      the slot addresses stay live across the pairs and the 8-g-point double instantiation spills 29 VGPRs: measured
//      equal to 1 in fp64 and 2 % slower in fp32 (round 2, same box).
// The arithmetic per g-point is the same in every order (look_up_table gas first, then the slots in order).
// seq_item(pos) = index of the item at position pos of the chunk's sequence; bil_slot / bil_pair decode a bilinear index.
#ifndef ECCKD_FUSED_INTERLEAVE
#define ECCKD_FUSED_INTERLEAVE 1
#endif
constexpr int seq_item(int pos, int NLI, int NBI, int NPI, int NP2) {
  if (ECCKD_FUSED_INTERLEAVE == 2) {
    const int NB = NBI / NP2, npl = NPI / NP2, per_pair = 2 + NB + npl;
    const int p = pos / per_pair, r = pos % per_pair;
    if (r < 2) return 2 * p + r;
    if (r < 2 + NB) return NLI + p * NB + (r - 2);
    return NLI + NBI + (r - 2 - NB) * NP2 + p;
  }
  if (ECCKD_FUSED_INTERLEAVE == 0 || NPI == 0 || pos < NLI) return pos;
  int r = pos - NLI, b = 0, k = 0;   // r-th item after the look_up_table items
  for (int q = 0;; ++q) {
    // Planck item k follows bilinear item ((2k+1)*NBI)/(2*NPI) - 1
    const bool planck_next = k < NPI && b >= ((2 * k + 1) * NBI) / (2 * NPI);
    if (q == r) return planck_next ? NLI + NBI + k : NLI + b;
    if (planck_next) ++k; else ++b;
  }
}
constexpr int bil_slot(int b, int NB, int NP2) { return ECCKD_FUSED_INTERLEAVE == 2 ? b % NB : b / NP2; }
constexpr int bil_pair(int b, int NB, int NP2) { return ECCKD_FUSED_INTERLEAVE == 2 ? b / NB : b % NP2; }

template <typename real> __device__ __forceinline__ real selmin(real a, real b) { return a < b ? a : b; }
template <typename real> __device__ __forceinline__ real selmax(real a, real b) { return a > b ? a : b; }

struct FLayout {
  int tb, red, bil, SB, lut, SL, pl, SP, total;
};

// NB  = bilinear slots the kernel reads per row (>= nbil; the slab is followed by a pad so that the
//       zero-weight slots read finite data).
// ngp = g-points per row in LDS: ng rounded up to the chunk size GC (the tail stays zero).
// ntp = Planck rows held in LDS (the whole table, or the window FusedArgs::pw).
// Offsets are in elements of the LDS storage type; `wide` = sizeof(arithmetic type) / sizeof(storage type) (2 for the
// fp32 image of an fp64 call): the reduction scratch holds arithmetic-type values.
__host__ __device__ inline FLayout f_layout(int ngp, int np, int nt, int nbil, int NB, int nv_lut, int R, int ntp, int wide = 1, int waves = kBlock / 64) {
  FLayout L;
  L.tb = 0;
  L.red = (np + 1) & ~1;
  L.bil = L.red + 4 * waves * wide;
  L.SB = nbil > 0 ? row_stride(nbil * ngp) : 2;
  L.lut = L.bil + R * nt * L.SB + NB * ngp;
  L.SL = nv_lut > 0 ? row_stride(ngp) : 2;
  L.pl = L.lut + (nv_lut > 0 ? R * nt * nv_lut * L.SL : 0) + ngp;
  L.SP = row_stride(ngp);
  L.total = L.pl + (ntp > 0 ? ntp * L.SP : 0);
  return L;
}

template <typename real> struct PPoint { int ip0; real pw0, pw1; };
template <typename real> __device__ __forceinline__ PPoint<real> pressure_point(real p0, real p1, real lp0, const UDivT<real> &dlp, int np) {
  const real log_pressure = log(real(0.5) * (p1 + p0));                      // :120
  real pressure_index = udiv(log_pressure - lp0, dlp);
  pressure_index = real(1) + selmax(real(0), selmin(pressure_index, (real)np - real(1.0001)));
  PPoint<real> r;
  r.ip0 = (int)pressure_index;
  r.pw1 = pressure_index - r.ip0;
  r.pw0 = real(1) - r.pw1;
  return r;
}

enum { MODE_TAU = 0, MODE_LW = 1, MODE_SW = 2 };
#ifndef ECCKD_FUSED_BLOCK_LW_ONLY   // (A/B builds: every mode in the 512-thread blocks of the longwave mode)
constexpr int fused_block(int mode) { return mode == MODE_LW ? kBlock : kBlockSw; }
#else
constexpr int fused_block(int) { return kBlock; }
#endif

// "gas_slab_f32" = auto.  The fp64 slab holds R = 3 pressure rows next to the Planck table, the float32 image R = 8; the
// widening costs 9 % where 3 rows do (measured, round 3: 13.6 against 12.6 ms at 1e6 columns) and saves a factor 3.4
// where the columns of a wave are spread over more rows than a slab position serves (surface pressures of 50-103 kPa
// shuffled over the columns: 11.1 -> 3.3 ms at 2e5 columns).  spread_probe_kernel counts, at the bottom layer of the
// call (where the spread is largest), the waves of 64 columns whose pressure indices span three values or more; more than
// one wave in kSpreadOneIn -> the float32 image.  Both forms give the same bits, so the choice only moves time.
constexpr int kSpreadOneIn = 24;
template <typename real>
__global__ void __launch_bounds__(256) spread_probe_kernel(const real *plev, int ncol, int nlay, real lp0, UDiv dlp_h, int np, int *count) {
  // both ends of the arrays (blockIdx.y): the call does not say which one is the surface, and the top contributes nothing
  const real *plev0 = plev + (long)ncol * (blockIdx.y ? nlay - 1 : 0), *plev1 = plev0 + ncol;
  const UDivT<real> dlp = make_udiv_t<real>(dlp_h);
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  int ip = -1;
  if (c < ncol) ip = pressure_point<real>(plev0[c], plev1[c], lp0, dlp, np).ip0;
  int lo = ip < 0 ? 0x7fffffff : ip, hi = ip;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o));
    hi = max(hi, __shfl_xor(hi, o));
  }
  if ((threadIdx.x & 63) == 0 && hi - lo >= 2) atomicAdd(count, 1);
}

// sreal: type of the tables in LDS.  = real, or float under a double kernel: every table of the ecCKD files is float32 on
// disk (widened exactly on read, mo_simple_netcdf.F90:44-142), so the slab and the Planck table can be staged as the
// float32 they are -- half the LDS -- and widened again (v_cvt_f64_f32, exact) when they are used: the same bits.  The
// host checks that every value is float32-representable (FusedArgs::slab32).
#ifndef ECCKD_F64_WAVES
#define ECCKD_F64_WAVES 2
#endif
#ifndef ECCKD_F32_WAVES
#define ECCKD_F32_WAVES 2
#endif
template <typename real, int GC, int NB, bool FULL, bool ANYCLAMP, int MODE, typename sreal = real, int BLOCK = kBlock>
__global__ void __launch_bounds__(BLOCK, (BLOCK > 512 ? 3 : sizeof(real) == 4 ? ECCKD_F32_WAVES : ECCKD_F64_WAVES)) gas_fused_kernel(const FusedArgs a) {
  constexpr int kBlock = BLOCK, kWaves = BLOCK / 64;   // (this instantiation's, not the file's defaults)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  sreal *lds = reinterpret_cast<sreal *>(lds_raw);
  typedef sreal double2_t __attribute__((ext_vector_type(2)));   // (two consecutive g-points as they sit in LDS)
  typedef __attribute__((address_space(3))) const volatile sreal lds_cvd;
  typedef __attribute__((address_space(3))) const volatile double2_t lds_cvd2;
  lds_cvd *lv = (lds_cvd *)lds;
  // two consecutive g-points in one ds_read_b128 (ds_read_b64 for a float32 image)
  // (at a BYTE address plus a compile-time element offset, the immediate of the ds_read)
  typedef __attribute__((address_space(3))) const volatile char lds_cvc;
  auto ld2b = [&](int bytes, int elem) -> double2_t { return *(lds_cvd2 *)((lds_cvc *)lv + bytes + elem * (int)sizeof(sreal)); };
  constexpr int WIDE = (int)(sizeof(real) / sizeof(sreal));
  if (a.choose) {   // "gas_slab_f32" = auto: both forms are launched, the spread probe's count says which one works
    const bool want32 = (long)a.choose[0] * kSpreadOneIn > (long)a.choose_total;
    if (want32 != (WIDE == 2)) return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wcol = (tid & ~63) + wave_column(lane);   // column of this thread inside a tile
  const int j = blockIdx.y;
  const TauArgs &t = a.tau;
  const int ncol = t.ncol, nlay = t.nlay, ng = t.ng, np = t.np, nt = t.nt, R = t.R;
  const int nv_lut = t.lut >= 0 ? t.seq[t.lut].nv : 0;
  const int ntp = MODE == MODE_LW ? a.ntp : 0;
  const int PW = MODE == MODE_LW ? a.pw : 0;   // Planck rows staged in LDS: ntp (whole table) or a window
  const int ngp = (ng + GC - 1) / GC * GC;
  const FLayout L = f_layout(ngp, np, nt, t.nbil, NB, nv_lut, R, PW, WIDE, kWaves);
  real *redd = reinterpret_cast<real *>(lds + L.red);
  // The argument structs carry `double` pointers and scalars; in the single-precision
  // instantiation the pointers address float data (host side casts) and the scalars are rounded.
  auto P = [](const double *p) { return reinterpret_cast<const real *>(p); };
  auto Q = [](double *p) { return reinterpret_cast<real *>(p); };
  const UDivT<real> ud_dlp = make_udiv_t<real>(a.ud_dlp), ud_dt = make_udiv_t<real>(a.ud_dt);
  const UDivT<real> ud_dlv = make_udiv_t<real>(a.ud_dlv), ud_pdt = make_udiv_t<real>(a.ud_pdt);
  const real lp0 = (real)t.lp0, gw = (real)t.gw, pt0 = (real)a.pt0;

  // Everything a zero-weight slot (unused bilinear slot, absent look_up_table gas) can read must
  // be finite: clear the whole allocation once, the staged rows overwrite their part.
  for (int i = tid; i < L.total; i += kBlock) lds[i] = sreal(0);
  __syncthreads();
  for (int i = tid; i < np; i += kBlock) lds[L.tb + i] = (sreal)P(t.temperature)[i];
  int pw_lo = -1;   // first table row of the staged Planck window (-1: nothing staged yet)
  // per-column inputs: wave-uniform row pointers (this block's layer) + 32-bit per-lane byte offsets
  typedef __attribute__((address_space(1))) const char gcchar_t;
  typedef __attribute__((address_space(1))) const real greal_t;
  gcchar_t *slot_base[NB + 1];
  unsigned slot_cs[NB + 1];
#pragma unroll
  for (int s = 0; s <= NB; ++s) {
    const SlotArgs &e = a.slot[s < NB ? s : kTauPassGases];
    slot_base[s] = (gcchar_t *)(P(e.vmr) + (long)j * e.ls);
    slot_cs[s] = e.cs_bytes;
  }

  const long ntiles = ((long)ncol + kBlock - 1) / kBlock;
  const long t_begin = ntiles * blockIdx.x / gridDim.x;
  const long t_end = ntiles * (blockIdx.x + 1) / gridDim.x;
  int slab_lo = -1;
  const real *plev0 = P(t.plev) + (long)ncol * j, *plev1 = P(t.plev) + (long)ncol * (j + 1);
  const real pi = (real)3.14159265359f, rpi = real(1) / pi;   // :53

  const int seg_len = t.seg;
  for (long seg = t_begin; seg < t_end; seg += seg_len) {
    const long seg_end = seg + seg_len < t_end ? seg + seg_len : t_end;
    // ---- pre-pass: range of p0+p1 over the segment; the pressure index is monotone in it ----
    // ... and, when only a window of the Planck table is staged, the range of the temperatures that
    // index it (layer j and its two levels); the table row is monotone in T.
    const bool windowed = MODE == MODE_LW && PW < ntp;
    real smin = real(3.0e38), smax = -real(3.0e38), tmin = real(3.0e38), tmax = -real(3.0e38);
    for (long tile = seg; tile < seg_end; ++tile) {
      const long c = tile * kBlock + tid;
      if (c < ncol) {
        const real sp = plev1[c] + plev0[c];
        smin = selmin(smin, sp);
        smax = selmax(smax, sp);
        if (windowed) {
          const real tl = P(t.tlay)[c + (long)ncol * j];
          tmin = selmin(tmin, tl);
          tmax = selmax(tmax, tl);
          if (a.tlev) {
            const real ta = P(a.tlev)[c + (long)ncol * j], tb = P(a.tlev)[c + (long)ncol * (j + 1)];
            tmin = selmin(tmin, selmin(ta, tb));
            tmax = selmax(tmax, selmax(ta, tb));
          }
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      smin = selmin(smin, __shfl_xor(smin, o));
      smax = selmax(smax, __shfl_xor(smax, o));
      if (windowed) {
        tmin = selmin(tmin, __shfl_xor(tmin, o));
        tmax = selmax(tmax, __shfl_xor(tmax, o));
      }
    }
    __syncthreads();   // the previous segment's LDS reads are done
    if (lane == 0) { redd[4 * wave] = smin; redd[4 * wave + 1] = smax; redd[4 * wave + 2] = tmin; redd[4 * wave + 3] = tmax; }
    __syncthreads();
    smin = redd[0]; smax = redd[1]; tmin = redd[2]; tmax = redd[3];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) {
      smin = selmin(smin, redd[4 * w]); smax = selmax(smax, redd[4 * w + 1]);
      tmin = selmin(tmin, redd[4 * w + 2]); tmax = selmax(tmax, redd[4 * w + 3]);
    }
    if (MODE == MODE_LW) {
      // rows [pw_lo, pw_lo + PW) of the Planck table; a lane needs rows r and r + 1
      int want = pw_lo;
      if (!windowed) {
        want = 0;
      } else if (tmin <= tmax) {
        const int rlo = planck_point<real>(tmin, pt0, ud_pdt, ntp).row, rhi = planck_point<real>(tmax, pt0, ud_pdt, ntp).row + 1;
        if (pw_lo < 0 || rlo < pw_lo || rhi > pw_lo + PW - 1) {
          // centre the range if it fits, else start at its low end (the rest takes the slow path)
          const int slack = PW - (rhi - rlo + 1);
          want = rlo - (slack > 0 ? slack / 2 : 0);
          want = want < 0 ? 0 : (want > ntp - PW ? ntp - PW : want);
        }
      } else if (pw_lo < 0) {
        want = 0;
      }
      if (want != pw_lo) {   // block-uniform: every thread holds the same reduced range
        pw_lo = want;
        for (int q = tid; q < PW * ng; q += kBlock) {
          const int r = q / ng, g = q - r * ng;
          lds[L.pl + r * L.SP + g] = (sreal)P(a.planck)[(long)(pw_lo + r) * ng + g];
        }
      }
    }
    int ipmin = 1, ipmax = 0;
    if (smin <= smax) {
      // same expression as pressure_point(): log(0.5*(p1+p0)); 0 + s == s exactly
      ipmin = pressure_point<real>(real(0), smin, lp0, ud_dlp, np).ip0;
      ipmax = pressure_point<real>(real(0), smax, lp0, ud_dlp, np).ip0;
    }
    // Entry `o` of the merged table (TauArgs::merge_*).  The slab staging and the tables-from-global-memory path both
    // go through this one expression: a column gets the same bits whichever path its wave takes.
    auto merged_coef = [&](long o) -> real {
      real v = (real)t.merge_mult[0] * P(t.seq[t.merge_seq[0]].coef)[o];
      for (int k = 1; k < t.nmerge; ++k) v = fma((real)t.merge_mult[k], P(t.seq[t.merge_seq[k]].coef)[o], v);
      return v;
    };
    auto stage_slab = [&](int lo) {   // pressure rows [lo, lo + R) of every active table -> LDS
      slab_lo = lo;
      const int rows_b = R * nt;
      const int items_b = rows_b * t.nbil;
      for (int q = wave; q < items_b; q += kWaves) {
        const int s = q % t.nbil, rb = q / t.nbil;
        const int ipl = rb % R, it = rb / R;
        const long row = (long)ng * ((slab_lo + ipl) + (long)np * it);
        // a row holds [g-point chunk][slot][GC g-points]: inside a chunk every (slot, g-point) is a compile-time
        // offset from the four corner addresses of the cell
        sreal *dst = lds + L.bil + rb * L.SB + s * GC;
        const int cs = t.nbil * GC;
        if (s == t.merge_slot) {   // sum_k mult_k * coefficient_k: the call-constant gases as one table
          // (never with a float32 image under a double kernel: the sums are not float32 numbers -- the host keeps the two apart)
          for (int g = lane; g < ng; g += 64) dst[(g / GC) * cs + g % GC] = (sreal)merged_coef(row + g);
        } else {
          const real *src = P(t.seq[t.bil_seq[s]].coef) + row;
          for (int g = lane; g < ng; g += 64) dst[(g / GC) * cs + g % GC] = (sreal)src[g];
        }
      }
      if (t.lut >= 0) {
        const real *coef = P(t.seq[t.lut].coef);
        const int rows_l = rows_b * nv_lut;
        for (int q = wave; q < rows_l; q += kWaves) {
          const int ipl = q % R, itv = q / R;
          const real *src = coef + (long)ng * ((slab_lo + ipl) + (long)np * itv);
          sreal *dst = lds + L.lut + q * L.SL;
          for (int g = lane; g < ng; g += 64) dst[g] = (sreal)src[g];
        }
      }
    };
    // A slab serves R - 1 consecutive values of the pressure index.  When the columns of the
    // segment span more (orography), the segment is walked once per slab POSITION: a wave handles
    // a tile in the pass whose position holds all of its lanes; a wave whose lanes straddle two
    // positions handles it once, in the pass of its lowest lane, through the slow path.
    const int span = R >= 2 ? R - 1 : 1;
    const int npos = (R >= 2 && ipmin <= ipmax) ? (ipmax - ipmin + span) / span : 1;
    for (int pos = 0; pos < npos; ++pos) {
    const int pos_lo = npos > 1 ? ipmin + pos * span : -0x40000000;   // pressure indices of this pass
    const int pos_hi = npos > 1 ? pos_lo + span - 1 : 0x40000000;
    if (pos > 0) __syncthreads();   // the previous pass's LDS reads are done
    if (R >= 2 && ipmin <= ipmax) {
      if (npos > 1) {
        const int lo = min(pos_lo - 1, np - R);
        if (lo != slab_lo) stage_slab(lo);
      } else if (!(slab_lo >= 0 && ipmin - 1 >= slab_lo && ipmax <= slab_lo + R - 1)) {
        stage_slab(min(ipmin - 1, np - R));
      }
    }
    __syncthreads();

    // The per-column inputs of a tile, one round of global loads.  (Loading them a tile ahead, before
    // the stores of the tile in flight -- vector memory operations retire in order, so a load issued
    // after a tile's 64 stores waits for all of them -- measured +1-2 % as long as the 26 extra VGPRs
    // did not spill; with the slab-position logic in the kernel they do, and it measures -3 %.)
    real nx_p0, nx_p1, nx_T, nx_W[NB], nx_vlut, nx_Tl0 = real(0), nx_Tl1 = real(0);
    auto load_inputs = [&](long tile) {
      const long c = tile * kBlock + wcol;
      const unsigned cc32 = (unsigned)(c < ncol ? c : (long)ncol - 1);
      const unsigned co = cc32 * (unsigned)sizeof(real);
      auto at = [&](const real *row) { return *reinterpret_cast<greal_t *>((gcchar_t *)row + co); };
      nx_p0 = at(plev0); nx_p1 = at(plev1);
      nx_T = at(P(t.tlay) + (long)ncol * j);
#pragma unroll
      for (int s = 0; s < NB; ++s) nx_W[s] = *reinterpret_cast<greal_t *>(slot_base[s] + cc32 * slot_cs[s]);
      nx_vlut = *reinterpret_cast<greal_t *>(slot_base[NB] + cc32 * slot_cs[NB]);
      if (MODE == MODE_LW && a.tlev) {
        nx_Tl0 = at(P(a.tlev) + (long)ncol * j);
        nx_Tl1 = at(P(a.tlev) + (long)ncol * (j + 1));
      }
    };

    for (long tile = seg; tile < seg_end; ++tile) {
      const long c = tile * kBlock + wcol;   // (see wave_column)
      const bool valid = c < ncol;
      const bool upper = lane >= 32;        // this lane stores plane g+1 of its column pair
      const long cc = c < ncol ? c : (long)ncol - 1;
      // ---- setup: one round of global loads ----
      load_inputs(tile);
      const real p0 = nx_p0, p1 = nx_p1, Tlayer = nx_T;
      real W[NB];        // per-slot vmr, then weight (:143-149); 0 for unused slots
#pragma unroll
      for (int s = 0; s < NB; ++s) W[s] = nx_W[s];
      real vlut = nx_vlut;
      const real Tl0 = nx_Tl0, Tl1 = nx_Tl1;

      const PPoint<real> pp = pressure_point<real>(p0, p1, lp0, ud_dlp, np);
      const int ip0 = pp.ip0;
      // Lanes this pass is responsible for: existing columns whose pressure index belongs to the slab
      // position of the pass (with a single position: every existing column).
      // (a lane whose pressure index lies outside the range the pre-pass found -- only a NaN pressure does
      // that: min/max skip it -- is assigned to the nearest position and sends its wave down the slow path:
      // its column comes out NaN like the oracle's, and no lane is ever dropped)
      const int ipq = ipmin <= ipmax ? (ip0 < ipmin ? ipmin : (ip0 > ipmax ? ipmax : ip0)) : ip0;
      const bool own = valid && ipq >= pos_lo && ipq <= pos_hi;
      if (!__any(own)) {   // nothing of this wave in this pass
        continue;
      }
      const bool lowest = !__any(valid && ipq < pos_lo);   // this is the pass of the wave's lowest lane
      const bool stray_wave = __any(valid && (ipq != ip0 || ipmin > ipmax));
      // (lanes of other passes are carried along with a clamped row: finite values, never stored)
      int ipl = ip0 - 1 - slab_lo;
      const bool inslab = (R >= 2) && slab_lo >= 0 && ipl >= 0 && ipl + 1 <= R - 1;
      ipl = ipl < 0 ? 0 : (ipl > R - 2 ? (R >= 2 ? R - 2 : 0) : ipl);

      const real t0 = pp.pw0 * (real)lds[L.tb + ip0 - 1] + pp.pw1 * (real)lds[L.tb + ip0];   // :131-132
      real temperature_index = udiv(Tlayer - t0, ud_dt);
      temperature_index = real(1) + selmax(real(0), selmin(temperature_index, (real)nt - real(1.0001)));
      const int it0 = (int)temperature_index;
      const real tw1 = temperature_index - it0;
      const real tw0 = real(1) - tw1;
      const real dp = p1 - p0;
      const real simple_weight = gw * dp;   // :143

      // corner weights, multiplied out once per cell
      const real a00 = tw0 * pp.pw0, a10 = tw0 * pp.pw1, a01 = tw1 * pp.pw0, a11 = tw1 * pp.pw1;
      real l000 = real(0), l100 = real(0), l010 = real(0), l110 = real(0), l001 = real(0), l101 = real(0), l011 = real(0), l111 = real(0);
      int iv0 = 1;
      if (t.lut >= 0) {   // :153-163
        const SeqGas &e = t.seq[t.lut];
        vlut = fma((real)a.slot[kTauPassGases].alpha, vlut, (real)a.slot[kTauPassGases].beta);
        const real log_vmr = log(selmax(vlut, (real)e.mf0));
        real vmr_index = udiv(log_vmr - (real)e.log_mf0, ud_dlv);
        vmr_index = real(1) + selmax(real(0), selmin(vmr_index, (real)e.nv - real(1.001)));
        iv0 = (int)vmr_index;
        const real vw1 = vmr_index - iv0, vw0 = real(1) - vw1;
        real wl = simple_weight * vlut;   // :148
        if (!ANYCLAMP) wl = wl < real(0) ? real(0) : wl;
        const real u0 = ANYCLAMP ? vw0 : wl * vw0, u1 = ANYCLAMP ? vw1 : wl * vw1;
        l000 = u0 * a00; l100 = u0 * a10; l010 = u0 * a01; l110 = u0 * a11;
        l001 = u1 * a00; l101 = u1 * a10; l011 = u1 * a01; l111 = u1 * a11;
        if (ANYCLAMP) vlut = wl;   // keep the weight; applied (and clamped) per g-point
      }
#pragma unroll
      for (int s = 0; s < NB; ++s) {
        const SlotArgs &e = a.slot[s];
        real x = simple_weight * fma((real)e.alpha, W[s], (real)e.beta);   // :143-149, see SlotArgs
        if (!ANYCLAMP) x = x < real(0) ? real(0) : x;   // od<0 -> 0 (:234-238) == weight<0 -> 0 for tables >= 0
        W[s] = x;
      }
      PlPoint<real> qlay{0, 0, real(0), real(0)}, ql0{0, 0, real(0), real(0)}, ql1{0, 0, real(0), real(0)};
      bool inwin = true;
      if (MODE == MODE_LW) {
        qlay = planck_point<real>(Tlayer, pt0, ud_pdt, ntp);
        // level j+1 serves lev_source_inc(:,j,:) AND lev_source_dec(:,j+1,:) (the same level: :419-424), so a
        // block evaluates ONE level per layer; the top level (lev_source_dec of layer 1) is the extra
        // work of the blocks of the first layer.
        ql0 = (a.tlev && j == 0) ? planck_point<real>(Tl0, pt0, ud_pdt, ntp) : qlay;
        ql1 = a.tlev ? planck_point<real>(Tl1, pt0, ud_pdt, ntp) : qlay;
        const int rmin = min(qlay.row, min(ql0.row, ql1.row)), rmax = max(qlay.row, max(ql0.row, ql1.row));
        inwin = rmin >= pw_lo && rmax + 1 <= pw_lo + PW - 1;
        qlay.off = (qlay.row - pw_lo) * L.SP;
        ql0.off = (ql0.row - pw_lo) * L.SP;
        ql1.off = (ql1.row - pw_lo) * L.SP;
      }
      const real moles = dp * gw;   // :313-314 (SW)
      // A wave with a lane that no slab position can serve (outside the Planck window, or no slab at
      // all) goes through the tables-from-global-memory path, all its lanes at once, in the pass of
      // its lowest lane.  Every other wave runs the item pipeline in each pass it owns lanes of:
      // all 64 lanes owned -> paired 16-byte stores, else the owned lanes store on their own.
      const bool slow_wave = __any(valid && !(inwin && R >= 2 && slab_lo >= 0)) || stray_wave;
      const bool fast = !slow_wave;
      const bool active = own && inslab;
      const bool masked_wave = !__all(active);
      const bool mine = slow_wave ? (lowest && valid) : active;   // lanes whose results this pass stores
      const int ip0_ = ip0, it0_ = it0, iv0_ = iv0;

      if (fast) {
        // The g-point work of a chunk is a static sequence of "items" of at most four 16-byte LDS
        // reads (two consecutive g-points each):
        //   look_up_table gas : (g-pair, vmr plane h)   4 corner reads, 8 FMAs  -> partial / final od
        //   bilinear slot     : (slot, g-pair)          4 corner reads, 10 FMAs
        //   Planck sources    : (g-pair)                layer + level j+1 (4 reads); first layer only: level j (2)
        // software-pipelined by hand: the reads of item i+1 are issued before the arithmetic of item
        // i.  Reads are volatile (kept in program order, never paired into ds_read2_b64) and every
        // item ends in an empty asm that pins its results, otherwise instruction selection floats
        // all arithmetic below all reads.  Four reads per item (not eight) keep the two read buffers
        // at 32 VGPRs.
        static_assert(GC % 4 == 0, "chunks are made of g-point pairs");
        constexpr int NP2 = GC / 2;                                  // g-pairs per chunk
        constexpr int NLI = 2 * NP2, NBI = NB * NP2;
        // LDS addresses (in elements): one register per corner row of the cell -- 4 for the bilinear slots, 8 for the
        // look_up_table gas, 4 (6 in the first layer) for the Planck rows -- advanced once per chunk; everything
        // inside a chunk is an immediate offset of the ds_read.  (Until round 2 the slot offset s*ngp was a run-time
        // value: 80 v_add_u32 per chunk of eight g-points.)
        const int ob = L.bil + (ipl + R * (it0 - 1)) * L.SB;
        const int ol = t.lut >= 0 ? L.lut + (ipl + R * ((it0 - 1) + nt * (iv0 - 1))) * L.SL : L.bil;
        const int dPb = L.SB, dTb = R * L.SB;
        const int dPl = t.lut >= 0 ? L.SL : 0, dTl = t.lut >= 0 ? R * L.SL : 0, dVl = t.lut >= 0 ? R * nt * L.SL : 0;
        constexpr int ES = (int)sizeof(sreal);   // (the registers hold byte addresses)
        int ab[4] = {ES * ob, ES * (ob + dPb), ES * (ob + dTb), ES * (ob + dTb + dPb)};
        int al[8] = {ES * ol, ES * (ol + dPl), ES * (ol + dTl), ES * (ol + dTl + dPl),
                     ES * (ol + dVl), ES * (ol + dVl + dPl), ES * (ol + dVl + dTl), ES * (ol + dVl + dTl + dPl)};
        int ap[6] = {ES * (L.pl + qlay.off), ES * (L.pl + qlay.off + L.SP), ES * (L.pl + ql1.off), ES * (L.pl + ql1.off + L.SP),
                     ES * (L.pl + ql0.off), ES * (L.pl + ql0.off + L.SP)};
        const int cstride = ES * t.nbil * GC;   // a chunk of the bilinear row: [slot][GC]
        // Output addressing: a uniform plane pointer (SGPRs) plus ONE per-lane 32-bit byte
        // offset shared by all four arrays -- the lower half-wave writes column pairs of plane g, the upper
        // half-wave those of plane g+1 (see store_pair).  launch_gas_fused() checks that it fits 32 bits.
        const unsigned plane = (unsigned)ncol * (unsigned)nlay;
        const unsigned coff = (unsigned)sizeof(real) * (unsigned)cc;
        const unsigned voff = (unsigned)sizeof(real) * ((unsigned)(c - (upper ? 1 : 0)) + (upper ? plane : 0u));
        // running output pointers: (column 0, layer j, g-point 0), advanced by store_pair
        const long plane2 = 2L * plane;
        real *w_tau = Q(t.tau) + (long)ncol * j, *w_ssa = MODE == MODE_SW && t.ssa ? Q(t.ssa) + (long)ncol * j : nullptr;
        real *w_g = MODE == MODE_SW && t.ssa ? Q(t.g) + (long)ncol * j : nullptr;
        real *w_lay = MODE == MODE_LW ? Q(a.lay_source) + (long)ncol * j : nullptr;
        const bool has_next = j + 1 < nlay;
        real *w_dec0 = MODE == MODE_LW && a.tlev ? Q(a.lev_source_dec) + (long)ncol * j : nullptr;          // first layer
        real *w_decn = MODE == MODE_LW && a.tlev ? Q(a.lev_source_dec) + (long)ncol * (j + 1) : nullptr;    // has_next
        real *w_inc = MODE == MODE_LW && a.tlev ? Q(a.lev_source_inc) + (long)ncol * j : nullptr;
        // Two copies of the chunk loop: with every lane owned (the common case) the stores are the
        // unconditional paired ones; the masked copy is for ragged waves and waves split between
        // slab positions.  A run-time flag instead costs a branch per store and ~2 % of the kernel.
        auto chunks = [&](auto masked_c, auto first_c) __attribute__((always_inline)) {
        constexpr bool masked = decltype(masked_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;   // blocks of the first layer: the top level too
        constexpr int NPI = (MODE == MODE_LW) ? (FIRST ? 2 * NP2 : NP2) : 0;
        constexpr int NIT = NLI + NBI + NPI;
        for (int gb = 0; gb < ngp; gb += GC) {
          real acc[GC];
          // (never in the longwave mode: the Planck sources ride with the FIRST pass only.  Compiled out there, because
          // with these loads in the chunk loop the compiler keeps an s_waitcnt vmcnt(0) at the join below, at the top of
          // EVERY chunk, and on the common path that wait is for the previous chunk's stores to be acknowledged:
          // the stores of a chunk then never overlap the arithmetic of the next.  Found in the disassembly late in round 2.)
          if (MODE != MODE_LW && t.accumulate) {   // second and later passes of a model with more gases than one pass takes
            typedef __attribute__((address_space(1))) const char gcchar;
            typedef __attribute__((address_space(1))) const real greal;
#pragma unroll
            for (int g = 0; g < GC; ++g) {
              acc[g] = real(0);
              if (FULL || gb + g < ng) {
                // uniform plane pointer + one 32-bit per-lane offset (see store_pair)
                gcchar *pl = (gcchar *)(P(t.tau) + (long)ncol * (j + (long)nlay * (gb + g)));
                asm volatile("" : "+s"(pl));
                acc[g] = *reinterpret_cast<greal *>(pl + coff);
              }
            }
          } else {
#pragma unroll
            for (int g = 0; g < GC; ++g) acc[g] = real(0);
          }
          // (pinned: otherwise the address arithmetic is re-derived per read from the row indices)
          asm volatile("" : "+v"(ab[0]), "+v"(ab[1]), "+v"(ab[2]), "+v"(ab[3]));
          asm volatile("" : "+v"(al[0]), "+v"(al[1]), "+v"(al[2]), "+v"(al[3]), "+v"(al[4]), "+v"(al[5]), "+v"(al[6]), "+v"(al[7]));
          if (MODE == MODE_LW) asm volatile("" : "+v"(ap[0]), "+v"(ap[1]), "+v"(ap[2]), "+v"(ap[3]));
          if (MODE == MODE_LW && FIRST) asm volatile("" : "+v"(ap[4]), "+v"(ap[5]));
          double2_t buf[2][4];
          real lutp[2] = {real(0), real(0)};
          static_for<0, NIT + 1>([&](auto pos_c) __attribute__((always_inline)) {
            constexpr int pos = decltype(pos_c)::value;                       // position in the sequence
            constexpr int it = pos < NIT ? seq_item(pos, NLI, NBI, NPI, NP2) : NIT;    // item read at this position
            // ---------------- issue the reads of item `it` ----------------
            if constexpr (it < NLI) {                       // look_up_table gas: g-points 2*pr, 2*pr+1, vmr plane h
              constexpr int g = 2 * (it / 2), h = it & 1;
              double2_t *b = buf[pos & 1];
              b[0] = ld2b(al[4 * h], g); b[1] = ld2b(al[4 * h + 1], g); b[2] = ld2b(al[4 * h + 2], g); b[3] = ld2b(al[4 * h + 3], g);
            } else if constexpr (it < NLI + NBI) {          // one bilinear slot, one g-pair
              constexpr int so = bil_slot(it - NLI, NB, NP2) * GC + 2 * bil_pair(it - NLI, NB, NP2);
              double2_t *b = buf[pos & 1];
              b[0] = ld2b(ab[0], so); b[1] = ld2b(ab[1], so); b[2] = ld2b(ab[2], so); b[3] = ld2b(ab[3], so);
            } else if constexpr (it < NIT) {                // Planck sources of one g-pair
              constexpr int k = it - NLI - NBI;
              double2_t *b = buf[pos & 1];
              if constexpr (k < NP2) {                      // layer and level j+1
                constexpr int g = 2 * k;
                b[0] = ld2b(ap[0], g); b[1] = ld2b(ap[1], g);
                b[2] = ld2b(ap[2], g); b[3] = ld2b(ap[3], g);
              } else {                                      // first layer: level j
                constexpr int g = 2 * (k - NP2);
                b[0] = ld2b(ap[4], g); b[1] = ld2b(ap[5], g);
              }
            }
            // ---------------- arithmetic of item `it - 1` ----------------
            if constexpr (pos >= 1) {
              constexpr int pi_ = seq_item(pos - 1, NLI, NBI, NPI, NP2);       // the item read at the previous position
              const double2_t *b = buf[(pos - 1) & 1];
              if constexpr (pi_ < NLI) {
                const int g0 = 2 * (pi_ / 2);
                if ((pi_ & 1) == 0) {
#pragma unroll
                  for (int q = 0; q < 2; ++q) {
                    real v = l000 * (real)b[0][q];
                    v = fma(l100, (real)b[1][q], v); v = fma(l010, (real)b[2][q], v); v = fma(l110, (real)b[3][q], v);
                    asm volatile("" : "+v"(v));
                    lutp[q] = v;
                  }
                } else {
#pragma unroll
                  for (int q = 0; q < 2; ++q) {
                    real v = lutp[q];
                    v = fma(l001, (real)b[0][q], v); v = fma(l101, (real)b[1][q], v); v = fma(l011, (real)b[2][q], v);
                    v = fma(l111, (real)b[3][q], v);
                    if (ANYCLAMP) { v = vlut * v; v = v < real(0) ? real(0) : v; }
                    acc[g0 + q] = acc[g0 + q] + v;
                    asm volatile("" : "+v"(acc[g0 + q]));
                  }
                }
              } else if constexpr (pi_ < NLI + NBI) {
                constexpr int s = bil_slot(pi_ - NLI, NB, NP2), g0 = 2 * bil_pair(pi_ - NLI, NB, NP2);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                  real v = a00 * (real)b[0][q];
                  v = fma(a10, (real)b[1][q], v); v = fma(a01, (real)b[2][q], v); v = fma(a11, (real)b[3][q], v);
                  if (ANYCLAMP) { v = W[s] * v; v = v < real(0) ? real(0) : v; acc[g0 + q] = acc[g0 + q] + v; }
                  else acc[g0 + q] = fma(W[s], v, acc[g0 + q]);
                  asm volatile("" : "+v"(acc[g0 + q]));
                }
                // tau is complete: of the whole chunk after its last bilinear item, or (g-pair-major order) of this
                // g-pair after its last slot
                constexpr bool pair_major = ECCKD_FUSED_INTERLEAVE == 2;
                if (pair_major ? s == NB - 1 : pi_ == NLI + NBI - 1) {
#pragma unroll
                  for (int g = pair_major ? g0 : 0; g < (pair_major ? g0 + 2 : GC); g += 2) {
                    // planes of this pair that exist: both, or (g-point counts that are not a multiple of the chunk) only the
                    // first one -- then the half-wave that holds it stores alone -- or none
                    const int npl = FULL ? 2 : ng - (gb + g);
                    if (FULL || npl >= 1) {
                      if (MODE == MODE_SW) {
                        const real r0 = moles * P(t.rayleigh)[gb + g], r1 = moles * P(t.rayleigh)[gb + g + 1];   // :316
                        const real t0_ = acc[g] + r0, t1_ = acc[g + 1] + r1;                                 // :456
                        store_pair<real>(w_tau, plane2, voff, coff, t0_, t1_, masked, active, npl, upper);
                        if (t.ssa) {                                                                          // :459-460
                          store_pair<real>(w_ssa, plane2, voff, coff, r0 / t0_, r1 / t1_, masked, active, npl, upper);
                          store_pair<real>(w_g, plane2, voff, coff, real(0), real(0), masked, active, npl, upper);
                        }
                      } else {
                        store_pair<real>(w_tau, plane2, voff, coff, acc[g], acc[g + 1], masked, active, npl, upper);
                      }
                    }
                  }
                }
              } else {
                constexpr int k = pi_ - NLI - NBI;
                constexpr int g = 2 * (k < NP2 ? k : k - NP2);
                const int npl = FULL ? 2 : ng - (gb + g);   // planes of this pair that exist (see the tau stores)
                if constexpr (k < NP2) {
                  real vl[2], v1[2];
#pragma unroll
                  for (int q = 0; q < 2; ++q) {
                    vl[q] = div_pi(qlay.w0 * (real)b[0][q] + qlay.w1 * (real)b[1][q], pi, rpi);
                    v1[q] = div_pi(ql1.w0 * (real)b[2][q] + ql1.w1 * (real)b[3][q], pi, rpi);
                  }
                  if (FULL || npl >= 1) {
                    store_pair<real>(w_lay, plane2, voff, coff, vl[0], vl[1], masked, active, npl, upper);
                    if (a.tlev) {                                                    // :423-424
                      store_pair<real>(w_inc, plane2, voff, coff, v1[0], v1[1], masked, active, npl, upper);
                      if (has_next) store_pair<real>(w_decn, plane2, voff, coff, v1[0], v1[1], masked, active, npl, upper);
                    }
                  }
                } else {
                  real v0[2];
#pragma unroll
                  for (int q = 0; q < 2; ++q) v0[q] = div_pi(ql0.w0 * (real)b[0][q] + ql0.w1 * (real)b[1][q], pi, rpi);
                  if (a.tlev && (FULL || npl >= 1)) store_pair<real>(w_dec0, plane2, voff, coff, v0[0], v0[1], masked, active, npl, upper);
                }
              }
            }
          });
#pragma unroll
          for (int q = 0; q < 4; ++q) ab[q] += cstride;
#pragma unroll
          for (int q = 0; q < 8; ++q) al[q] += ES * GC;
#pragma unroll
          for (int q = 0; q < 6; ++q) ap[q] += ES * GC;
        }
        };
        if (j == 0) {
          if (masked_wave) chunks(std::true_type{}, std::true_type{}); else chunks(std::false_type{}, std::true_type{});
        } else {
          if (masked_wave) chunks(std::true_type{}, std::false_type{}); else chunks(std::false_type{}, std::false_type{});
        }
      } else if (lowest) {
        // ---- a lane of this wave is outside the staged rows: tables from global memory ----
        // (the indices go through an opaque asm: otherwise the 64-bit address arithmetic of this
        // rare path is speculated above the branch and paid by every tile)
        int ip0 = ip0_, it0 = it0_, iv0 = iv0_;
        asm volatile("" : "+v"(ip0), "+v"(it0), "+v"(iv0));
        for (int g = 0; g < ng; ++g) {
          const long o = cc + (long)ncol * (j + (long)nlay * g);
          real acc = t.accumulate ? P(t.tau)[o] : real(0);
          if (t.lut >= 0) {
            const SeqGas &e = t.seq[t.lut];
            const real *cp = P(e.coef) + (long)ng * ((ip0 - 1) + (long)np * ((it0 - 1) + (long)nt * (iv0 - 1))) + g;
            const long dP = ng, dT = (long)ng * np, dV = (long)ng * np * nt;
            real v = l000 * cp[0];
            v = fma(l100, cp[dP], v);
            v = fma(l010, cp[dT], v);
            v = fma(l110, cp[dT + dP], v);
            v = fma(l001, cp[dV], v);
            v = fma(l101, cp[dV + dP], v);
            v = fma(l011, cp[dV + dT], v);
            v = fma(l111, cp[dV + dT + dP], v);
            if (ANYCLAMP) { v = vlut * v; v = v < real(0) ? real(0) : v; }
            acc = acc + v;
          }
#pragma unroll
          for (int s = 0; s < NB; ++s) {
            if (s < t.nbil) {
              const long o00 = (long)ng * ((ip0 - 1) + (long)np * (it0 - 1)) + g;
              const long dP = ng, dT = (long)ng * np;
              real c00, c10, c01, c11;
              if (s == t.merge_slot) {
                c00 = merged_coef(o00); c10 = merged_coef(o00 + dP); c01 = merged_coef(o00 + dT); c11 = merged_coef(o00 + dT + dP);
              } else {
                const real *cp = P(t.seq[t.bil_seq[s]].coef) + o00;
                c00 = cp[0]; c10 = cp[dP]; c01 = cp[dT]; c11 = cp[dT + dP];
              }
              real v = a00 * c00;
              v = fma(a10, c10, v);
              v = fma(a01, c01, v);
              v = fma(a11, c11, v);
              if (ANYCLAMP) { v = W[s] * v; v = v < real(0) ? real(0) : v; acc = acc + v; }
              else acc = fma(W[s], v, acc);
            }
          }
          if (valid) {
            if (MODE == MODE_SW) {
              const real ray = moles * P(t.rayleigh)[g];
              const real tt = acc + ray;
              Q(t.tau)[o] = tt;
              if (t.ssa) { Q(t.ssa)[o] = ray / tt; Q(t.g)[o] = real(0); }
            } else {
              Q(t.tau)[o] = acc;
            }
            if (MODE == MODE_LW) {
              const real *pg = P(a.planck) + g;   // table rows from global memory: any row, staged or not
              Q(a.lay_source)[o] = div_pi(qlay.w0 * pg[(long)qlay.row * ng] + qlay.w1 * pg[(long)(qlay.row + 1) * ng], pi, rpi);
              if (a.tlev) {   // level j+1 -> inc of this layer and dec of the next; the top level by the first layer
                const real v1 = div_pi(ql1.w0 * pg[(long)ql1.row * ng] + ql1.w1 * pg[(long)(ql1.row + 1) * ng], pi, rpi);
                Q(a.lev_source_inc)[o] = v1;
                if (j + 1 < nlay) Q(a.lev_source_dec)[o + ncol] = v1;
                if (j == 0)
                  Q(a.lev_source_dec)[o] = div_pi(ql0.w0 * pg[(long)ql0.row * ng] + ql0.w1 * pg[(long)(ql0.row + 1) * ng], pi, rpi);
              }
            }
          }
        }
      }

      // ---- surface source (:408-413), by the blocks of the first layer ----
      if (MODE == MODE_LW && j == 0 && mine) {
        // (table rows from global memory: the surface temperature is not part of the window range)
        const PlPoint<real> qs = planck_point<real>(P(a.tsfc)[c], pt0, ud_pdt, ntp);
        const real *p0r = P(a.planck) + (long)qs.row * ng, *p1r = p0r + ng;
        for (int g = 0; g < ng; ++g)
          Q(a.sfc_source)[c + (long)ncol * g] = div_pi(qs.w0 * p0r[g] + qs.w1 * p1r[g], pi, rpi);
      }
    }
    }   // slab positions
  }
}

template <typename real, int GC, int NB, bool FULL, bool ANYCLAMP, int MODE, typename sreal = real>
hipError_t launch_one(const FusedArgs &a, size_t lds_bytes, hipStream_t s) {
  constexpr int kBlock = fused_block(MODE);
  auto k = gas_fused_kernel<real, GC, NB, FULL, ANYCLAMP, MODE, sreal, kBlock>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(a.tau.col_chunks, a.tau.nlay), dim3(kBlock), lds_bytes, s, a);
  return hipGetLastError();
}

int pick_nb(int nbil) {
  if (nbil <= 2) return 2;
  if (nbil <= 5) return 5;
  if (nbil <= 7) return 7;
  return kTauPassGases;
}

template <typename real, int MODE>
hipError_t launch_mode(const FusedArgs &a, size_t lds, int NBsel, bool anyclamp, hipStream_t s) {
  const int ng = a.tau.ng;
  if (anyclamp && ng % 4 == 0) return launch_one<real, 4, kTauPassGases, true, true, MODE>(a, lds, s);
  if (anyclamp) return launch_one<real, 4, kTauPassGases, false, true, MODE>(a, lds, s);
#ifndef ECCKD_FUSED_NOGC8   // (experiment: chunks of four g-points only)
  if (NBsel == 2 && ng % 8 == 0) return launch_one<real, 8, 2, true, false, MODE>(a, lds, s);
#endif
  if (NBsel == 2 && ng % 4 == 0) return launch_one<real, 4, 2, true, false, MODE>(a, lds, s);
  if (NBsel == 2) return launch_one<real, 4, 2, false, false, MODE>(a, lds, s);
#ifndef ECCKD_FUSED_NOGC8
  if (NBsel == 7 && ng % 8 == 0) return launch_one<real, 8, 7, true, false, MODE>(a, lds, s);
#endif
  if (NBsel == 7 && ng % 4 == 0) return launch_one<real, 4, 7, true, false, MODE>(a, lds, s);
  if (NBsel == 5 && ng % 4 == 0) return launch_one<real, 4, 5, true, false, MODE>(a, lds, s);
  if (NBsel == 5) return launch_one<real, 4, 5, false, false, MODE>(a, lds, s);
  if (ng % 4 == 0) return launch_one<real, 4, kTauPassGases, true, false, MODE>(a, lds, s);
  return launch_one<real, 4, kTauPassGases, false, false, MODE>(a, lds, s);
}

// (FULL = true instantiations -- g-point count a multiple of the chunk, true for both longwave tables -- carry no
// per-g-point bounds checks and keep everything in registers; the FULL = false ones spill 13-108 VGPRs and only
// serve tables whose g-point count is not a multiple of 4, e.g. the 27 g-points of the shortwave table with
// more than 7 gases.)
// (GC, NB) of the instantiation launch_mode() will pick
// float32 image of the tables under a double kernel: instantiated for the full-chunk longwave shape of the ecCKD files
bool slab32_applies(int mode, int ng, int nbil, bool anyclamp) {
  return mode == MODE_LW && !anyclamp && ng % 8 == 0 && pick_nb(nbil) == 7;
}

void pick_shape(int ng, int nbil, bool anyclamp, int *GC, int *NB) {
  const int nb = pick_nb(nbil);
#ifdef ECCKD_FUSED_NOGC8
  if (ng % 8 == 0) ng += 4;
#endif
  if (anyclamp) { *GC = 4; *NB = kTauPassGases; }
  else if (nb == 2) { *GC = ng % 8 == 0 ? 8 : 4; *NB = 2; }
  else if (nb == 7 && ng % 8 == 0) { *GC = 8; *NB = 7; }
  else if (nb == 7 && ng % 4 == 0) { *GC = 4; *NB = 7; }
  else if (nb == 5) { *GC = 4; *NB = 5; }
  else { *GC = 4; *NB = kTauPassGases; }
}

// LDS a block may take: all of a CU's (one block per CU), or ECCKD_FUSED_LDS_KB for experiments with several blocks per CU
size_t lds_budget(int f32 = 0) {
#ifdef ECCKD_F32_LDS_KB      // (experiment: two single-precision blocks per CU)
  if (f32 == 1) return (size_t)ECCKD_F32_LDS_KB * 1024;
#endif
  (void)f32;
#ifdef ECCKD_FUSED_LDS_KB
  return (size_t)ECCKD_FUSED_LDS_KB * 1024;
#else
  return (size_t)kLdsBudget;
#endif
}

}  // namespace

// Rows of the LDS slab for a fused launch that keeps `pl_rows` rows of the Planck table in LDS, or 0
// if it does not fit with at least `min_rows`.
// f32: 0 double arithmetic and tables, 1 float arithmetic and tables, 2 double arithmetic over the float32 image of the
// tables (FusedArgs::slab32).
int fused_slab_rows(int ng, int np, int nt, int nbil, int nv_lut, int pl_rows, int min_rows, int anyclamp, int f32, int block) {
  if (block <= 0) block = kBlock;
  int GC, NB;
  pick_shape(ng, nbil, anyclamp != 0, &GC, &NB);
  const int ngp = (ng + GC - 1) / GC * GC;
  const size_t esz = f32 ? sizeof(float) : sizeof(double);
  const size_t budget = lds_budget(f32);
  int R = 0;
  for (int r = 2; r <= np; ++r) {
    if (esz * (size_t)f_layout(ngp, np, nt, nbil, NB, nv_lut, r, pl_rows, f32 == 2 ? 2 : 1, block / 64).total <= budget) R = r;
    else break;
  }
#ifdef ECCKD_FUSED_MAXROWS   // (experiments: cap the slab)
  if (R > ECCKD_FUSED_MAXROWS) R = ECCKD_FUSED_MAXROWS;
#endif
  return R >= min_rows ? R : 0;
}

// Planck rows to stage for a fused longwave launch: the whole table if it fits next to >= 3 slab
// rows, else the largest window of the list that does (a segment of 4096 columns of one layer
// rarely spans more than ~70 K = 70 rows of the 1 K table); 0 = does not fit at all.
// ECCKD_PLANCK_WINDOW=<rows> forces a window (tests of the out-of-window path).
int fused_planck_rows(int ng, int np, int nt, int nbil, int nv_lut, int ntp, int anyclamp, int f32) {
  if (ntp < 2) return 0;
  if (const char *e = getenv("ECCKD_PLANCK_WINDOW")) {
    const int w = atoi(e);
    if (w >= 2 && w < ntp && fused_slab_rows(ng, np, nt, nbil, nv_lut, w, 3, anyclamp, f32) > 0) return w;
  }
  if (fused_slab_rows(ng, np, nt, nbil, nv_lut, ntp, 3, anyclamp, f32) > 0) return ntp;
  for (int w : {128, 96, 64, 48, 32, 16})
    if (w < ntp && fused_slab_rows(ng, np, nt, nbil, nv_lut, w, 3, anyclamp, f32) > 0) return w;
  return 0;
}

UDiv make_udiv(double d, int f32) {
  UDiv u;
  if (f32) d = (double)(float)d;   // the kernel works with the rounded divisor
  u.d = d;
  u.r = f32 ? (double)(1.f / (float)d) : 1. / d;   // correctly rounded reciprocal in the working precision
  unsigned long long bits;
  static_assert(sizeof(bits) == sizeof(d), "");
  __builtin_memcpy(&bits, &d, 8);
  bool all_ones = (bits & 0xFFFFFFFFFFFFFULL) == 0xFFFFFFFFFFFFFULL;
  if (f32) {
    const float df = (float)d;
    unsigned int b32;
    __builtin_memcpy(&b32, &df, 4);
    all_ones = (b32 & 0x7FFFFFu) == 0x7FFFFFu;
  }
  u.exact = (d == d) && d != 0. && (d - d == 0.) && !all_ones && (u.r - u.r == 0.) ? 1 : 0;
  return u;
}

// The gases of a pass whose mole fraction is one number for the whole call: scalar entries of gas_desc (get_vmr
// broadcasts them, src/gas_optics_ecckd.f90:351) and the none_ composite (:213-221).  Their optical depths are
//     od_k = simple_weight * m_k * bilinear(coefficient_k),   m_k = vmr_k | vmr_k - reference_k | 1,
// so their sum is simple_weight * bilinear(sum_k m_k * coefficient_k): one table, one slab slot, four LDS reads and five
// FMAs per g-point pair instead of that per gas.  The per-gas clamp od_k < 0 -> 0 (:234-238) is kept exactly as long as
// every od_k of a cell has the sign of simple_weight, i.e. for tables without negative entries and m_k >= 0 -- a gas
// below its reference concentration, a table with negative entries or a non-finite value keeps its own slot.
int merge_scalar_gases(TauArgs &t, int f32) {
  t.merge_slot = -1;
  t.nmerge = 0;
  int cand[kMaxSeq], ncand = 0;
  double mult[kMaxSeq];
  for (int s = 0; s < t.nbil; ++s) {
    const SeqGas &e = t.seq[t.bil_seq[s]];
    if (e.clamp) continue;
    double m;
    if (e.code == 0) m = 1.;
    else if (e.vmr) continue;
    else if (e.code == 3) m = f32 ? (double)((float)e.scalar - (float)e.ref) : e.scalar - e.ref;
    else m = f32 ? (double)(float)e.scalar : e.scalar;
    if (!(m >= 0.) || !(m - m == 0.)) continue;
    cand[ncand] = t.bil_seq[s];
    mult[ncand++] = m;
  }
  if (ncand < 2) return 0;
  int keep[kMaxSeq], nkeep = 0;
  for (int s = 0; s < t.nbil; ++s) {
    bool merged = false;
    for (int k = 0; k < ncand; ++k) merged |= cand[k] == t.bil_seq[s];
    if (!merged) keep[nkeep++] = t.bil_seq[s];
  }
  for (int s = 0; s < nkeep; ++s) { t.bil_seq[s] = keep[s]; t.seq[keep[s]].slot = s; }
  t.merge_slot = nkeep;
  t.bil_seq[nkeep] = cand[0];   // (never read for the merged slot)
  t.nbil = nkeep + 1;
  t.nmerge = ncand;
  for (int k = 0; k < ncand; ++k) { t.merge_seq[k] = cand[k]; t.merge_mult[k] = mult[k]; t.seq[cand[k]].slot = nkeep; }
  return ncand;
}

// Host-side decisions of a fused launch (slots, slab rows, Planck window, grid): everything but the
// launch itself, so that ecckd_gas_optics_plan() can report them without a GPU.
hipError_t prepare_gas_fused(FusedArgs &a, FusedPlan &plan) {
  TauArgs &t = a.tau;
  plan = FusedPlan{};
  a.ud_dlp = make_udiv(t.dlp, a.f32);
  a.ud_dt = make_udiv(t.dt, a.f32);
  a.ud_dlv = make_udiv(t.lut >= 0 ? t.seq[t.lut].d_log_vmr : 1., a.f32);
  a.ud_pdt = make_udiv(a.mode == MODE_LW ? a.pdt : 1., a.f32);
  const size_t esz = a.f32 ? sizeof(float) : sizeof(double);
  // working-precision arithmetic for the constants folded on the host
  auto sub = [&](double x, double y) { return a.f32 ? (double)((float)x - (float)y) : x - y; };
  for (int s = 0; s <= kTauPassGases; ++s) {
    const int k = s < kTauPassGases ? (s < t.nbil ? t.bil_seq[s] : -1) : t.lut;
    SlotArgs &o = a.slot[s];
    o = SlotArgs{t.zero, 0, 0u, 0., 0.};   // unused slot or scalar gas: the load reads a zero word of the model
    if (s < kTauPassGases && s == t.merge_slot) {
      o.beta = 1.;                         // the multipliers are in the merged table: weight = simple_weight
    } else if (k >= 0) {
      const SeqGas &e = t.seq[k];
      const bool arr = e.vmr != nullptr;
      if (arr) {
        if (e.cs < 0 || (unsigned long long)e.cs * (unsigned long long)(t.ncol > 0 ? t.ncol : 1) * esz >= 0xFFFFFFF0ull)
          return hipErrorInvalidValue;   // column stride of a vmr array beyond 32-bit byte offsets
        o.vmr = e.vmr; o.ls = e.ls; o.cs_bytes = (unsigned)((unsigned long long)e.cs * esz);
      }
      switch (e.code) {
        case 0: o.alpha = 0.; o.beta = 1.; break;                                     // none_: weight = simple_weight (:213-221)
        case 3: o.alpha = arr ? 1. : 0.; o.beta = arr ? -e.ref : sub(e.scalar, e.ref); break;   // relative_linear (:146)
        default: o.alpha = arr ? 1. : 0.; o.beta = arr ? 0. : e.scalar; break;        // linear, look_up_table (:148)
      }
    }
  }
  if (t.ncol <= 0 || t.nlay <= 0) { plan.empty = 1; return hipSuccess; }
  if (t.nseq > kTauPassGases) return hipErrorInvalidValue;
  // store_pair() addresses a plane pair with a 32-bit byte offset
  if (((size_t)t.ncol * (size_t)t.nlay + (size_t)t.ncol) * (a.f32 ? sizeof(float) : sizeof(double)) >= (size_t)0xFFFFFFF0u)
    return hipErrorInvalidValue;
  const int nv_lut = t.lut >= 0 ? t.seq[t.lut].nv : 0;
  bool anyclamp = false;
  for (int k = 0; k < t.nseq; ++k) anyclamp |= t.seq[k].clamp != 0;
  if (a.mode == MODE_LW) {
    a.pw = fused_planck_rows(t.ng, t.np, t.nt, t.nbil, nv_lut, a.ntp, anyclamp, a.f32);
    if (a.pw < 2) return hipErrorInvalidValue;   // the caller checks fused_planck_rows() first
  } else {
    a.pw = 0;
  }
  int GC, NB;
  pick_shape(t.ng, t.nbil, anyclamp, &GC, &NB);
  const int ngp = (t.ng + GC - 1) / GC * GC;
  // float32 image of the tables under a double kernel: only the instantiations built for it (slab32_applies)
  if (a.slab32 && !(a.f32 == 0 && slab32_applies(a.mode, t.ng, t.nbil, anyclamp) && t.merge_slot < 0)) a.slab32 = 0;
  const int store = a.f32 ? 1 : (a.slab32 ? 2 : 0);
  if (a.mode == MODE_LW && store == 2) a.pw = fused_planck_rows(t.ng, t.np, t.nt, t.nbil, nv_lut, a.ntp, anyclamp, 2);
  const int block = fused_block(a.mode);   // threads of a block = columns of a tile
  t.R = fused_slab_rows(t.ng, t.np, t.nt, t.nbil, nv_lut, a.pw, 0, anyclamp, store, block);
  const size_t lds = (store ? sizeof(float) : sizeof(double)) *
                     (size_t)f_layout(ngp, t.np, t.nt, t.nbil, NB, nv_lut, t.R, a.pw, store == 2 ? 2 : 1, block / 64).total;
  if (lds > lds_budget(store)) return hipErrorInvalidValue;
  // one block per CU (LDS-bound): a block count that is a multiple of the 256 CUs keeps the last
  // round of blocks full
  const long ntiles = ((long)t.ncol + block - 1) / block;
  long chunks = 1;
  while ((chunks * t.nlay) % 256 != 0 && chunks < 256) ++chunks;
  while (chunks * 2 * kSeg <= ntiles && chunks * t.nlay < 2048) chunks *= 2;
  if (chunks * kSeg > ntiles) chunks = (ntiles + kSeg - 1) / kSeg;
  if (chunks < 1) chunks = 1;
  t.col_chunks = (int)chunks;
  {
    long per_block = (ntiles + chunks - 1) / chunks;
    t.seg = (int)(per_block < kSeg ? kSeg : (per_block > kSegMax ? kSegMax : per_block));
  }
  if (a.f32 && a.mode == MODE_TAU) return hipErrorNotSupported;   // single precision: the one-pass longwave and shortwave paths
  plan.lds_bytes = lds;
  plan.anyclamp = anyclamp ? 1 : 0;
  plan.GC = GC; plan.NB = NB; plan.merged = t.merge_slot >= 0 ? t.nmerge : 0;
  plan.slab_rows = t.R; plan.planck_rows = a.pw; plan.col_chunks = t.col_chunks;
  return hipSuccess;
}

hipError_t launch_gas_fused(FusedArgs &a, hipStream_t s) {
  FusedPlan plan;
  if (a.slab32 == 2) {   // auto: spread probe, then both forms; the one the probe does not choose returns at once
    FusedArgs b = a;
    b.slab32 = 1;
    FusedPlan pb;
    hipError_t eb = prepare_gas_fused(b, pb);
    if (eb == hipSuccess && !pb.empty && b.slab32 == 1 && a.choose_buf) {
      a.slab32 = 0;
      eb = prepare_gas_fused(a, plan);
      if (eb != hipSuccess) return eb;
      const TauArgs &t = a.tau;
      eb = hipMemsetAsync(a.choose_buf, 0, sizeof(int), s);
      if (eb != hipSuccess) return eb;
      const long nw = ((long)t.ncol + 63) / 64;
      hipLaunchKernelGGL(spread_probe_kernel<double>, dim3((unsigned)((t.ncol + 255) / 256), t.nlay > 1 ? 2 : 1), dim3(256), 0, s,
                         t.plev, t.ncol, t.nlay, t.lp0, a.ud_dlp, t.np, a.choose_buf);
      eb = hipGetLastError();
      if (eb != hipSuccess) return eb;
      a.choose = b.choose = a.choose_buf;
      a.choose_total = b.choose_total = (int)(nw > 0x7fffffff ? 0x7fffffff : nw);
      eb = launch_mode<double, MODE_LW>(a, plan.lds_bytes, pick_nb(t.nbil), plan.anyclamp != 0, s);
      if (eb != hipSuccess) return eb;
      return launch_one<double, 8, 7, true, false, MODE_LW, float>(b, pb.lds_bytes, s);
    }
    a.slab32 = 0;   // no float32 form for this shape (or no room for the probe's counter): the fp64 slab
  }
  const hipError_t e = prepare_gas_fused(a, plan);
  if (e != hipSuccess || plan.empty) return e;
  const TauArgs &t = a.tau;
  const bool anyclamp = plan.anyclamp != 0;
  if (a.slab32 && !a.f32) return launch_one<double, 8, 7, true, false, MODE_LW, float>(a, plan.lds_bytes, s);
  if (a.f32 && a.mode == MODE_SW) return launch_mode<float, MODE_SW>(a, plan.lds_bytes, pick_nb(t.nbil), anyclamp, s);
  if (a.f32) return launch_mode<float, MODE_LW>(a, plan.lds_bytes, pick_nb(t.nbil), anyclamp, s);
  if (a.mode == MODE_LW) return launch_mode<double, MODE_LW>(a, plan.lds_bytes, pick_nb(t.nbil), anyclamp, s);
  if (a.mode == MODE_SW) return launch_mode<double, MODE_SW>(a, plan.lds_bytes, pick_nb(t.nbil), anyclamp, s);
  return launch_mode<double, MODE_TAU>(a, plan.lds_bytes, pick_nb(t.nbil), anyclamp, s);
}

}  // namespace ecckd
