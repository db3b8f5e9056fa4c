// kernels_tau.hip -- optical depth of all gases, fused: tau(col,lay,gpt) written once.
//
// Replaces calculate_optical_depth + the accumulation loop of gas_optical_depth
// (src/gas_optics_ecckd.f90:64-241, :323-376) and, for the shortwave, the Rayleigh/ssa
// epilogue of gas_optics_ext (:293-319, :455-460).
//
// Mapping (gfx950): lane -> column (coalesced 512 B wave stores into the column-fastest
// output), block = 1024 columns of ONE layer, grid.y = layer.  At a fixed layer the columns of
// a tile differ in pressure index by at most a few rows, so the block stages a slab of R
// consecutive pressure rows of every active table in LDS (rows padded to an odd number of
// doubles: lanes that differ in (ip,it,iv) fall on different banks) and the g-point loop reads
// coefficients with ds_read_b64 at immediate offsets.  A wave that has a lane outside the
// staged rows falls back to reading the tables from global memory (L2) -- slower, never wrong.
//
// Arithmetic follows the reference expression order exactly (compiled with -ffp-contract=off),
// accumulation over gases is in gas_desc order, so tau differs from the reference only through
// the device log().
#include "kernels.hpp"

namespace ecckd {

namespace {

#ifndef ECCKD_TAU_BLOCK
#define ECCKD_TAU_BLOCK 768
#endif
#ifndef ECCKD_TAU_GC
#define ECCKD_TAU_GC (ECCKD_TAU_BLOCK >= 1024 ? 8 : (ECCKD_TAU_BLOCK > 512 ? 16 : 32))
#endif
#ifndef ECCKD_TAU_SPAN
#define ECCKD_TAU_SPAN 4
#endif
#ifndef ECCKD_TAU_VOLATILE
#define ECCKD_TAU_VOLATILE 1
#endif
constexpr int kTauBlock = ECCKD_TAU_BLOCK;
constexpr int kTauWaves = kTauBlock / 64;
// g-points whose LDS reads and arithmetic the scheduler may interleave (bounds live registers)
constexpr int kSpan = ECCKD_TAU_SPAN;
constexpr int kSpanLut = ECCKD_TAU_SPAN > 1 ? ECCKD_TAU_SPAN / 2 : 1;
constexpr int kSeg = 8;      // tiles between two slab-range checks (block barriers)
constexpr int kPass = kTauPassGases;

// The oracle's (and hence our) min/max: plain selects, so NaN handling is identical.
__device__ __forceinline__ double selmin(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double selmax(double a, double b) { return a > b ? a : b; }

__host__ __device__ inline int odd_up(int n) { return n | 1; }

struct SlabLayout {
  int tb;      // T(:,1) base profile, np doubles
  int red;     // reduction scratch (2*kTauWaves ints)
  int bil;     // bilinear slab
  int SB;      // bilinear row stride (doubles)
  int lut;     // look_up_table slab
  int SL;      // lut row stride (doubles)
  int total;   // doubles
};

__host__ __device__ inline SlabLayout slab_layout(int ng, int np, int nt, int nbil, int nv_lut,
                                                  int R) {
  SlabLayout L;
  L.tb = 0;
  L.red = (np + 1) & ~1;
  L.bil = L.red + kTauWaves;   // 2*kTauWaves ints = kTauWaves doubles
  L.SB = nbil > 0 ? odd_up(nbil * ng) : 0;
  L.lut = L.bil + R * nt * L.SB;
  L.SL = nv_lut > 0 ? odd_up(ng) : 0;
  L.total = L.lut + R * nt * nv_lut * L.SL;
  return L;
}

// One pressure interpolation point, src/gas_optics_ecckd.f90:120-128.
struct PPoint { int ip0; double pw0, pw1; };
__device__ __forceinline__ PPoint pressure_point(double p0, double p1, double lp0, double dlp, int np) {
  const double log_pressure = log(0.5 * (p1 + p0));
  double pressure_index = (log_pressure - lp0) / dlp;
  pressure_index = 1. + selmax(0., selmin(pressure_index, (double)np - 1.0001));
  PPoint r;
  r.ip0 = (int)pressure_index;   // 1-based
  r.pw1 = pressure_index - r.ip0;
  r.pw0 = 1. - r.pw1;
  return r;
}

// FULL: ng is a multiple of GC, so the unrolled g loops carry no `gb+g < ng` predicates (each
// predicate would otherwise end a basic block and serialise LDS latency against the arithmetic).
//
// Structure per block (one layer, a chunk of column tiles):
//   segment of kSeg tiles:  pre-pass over the segment's columns -> [min,max] pressure row,
//                           (re)stage the slab if it does not cover them          [3 barriers]
//     tile:                 setup: ONE round of global loads (plev, tlay, every gas's vmr),
//                           indices, weights of all gases of the pass into registers
//                           main:  for g-chunk, for gas (runtime loop, weight picked by a select
//                           chain): LDS reads + lerp, accumulate; store tau       [no barrier]
// so that inside a segment the waves drift apart and hide each other's setup latency.
template <int GC, bool FULL, bool ANYCLAMP, bool SW>
__global__ void __launch_bounds__(kTauBlock) tau_kernel(const TauArgs a) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = blockIdx.y;
  const int ncol = a.ncol, nlay = a.nlay, ng = a.ng, np = a.np, nt = a.nt, R = a.R;
  const int nv_lut = a.lut >= 0 ? a.seq[a.lut].nv : 0;
  const SlabLayout L = slab_layout(ng, np, nt, a.nbil, nv_lut, R);
  int *red = reinterpret_cast<int *>(lds + L.red);
#if ECCKD_TAU_VOLATILE
  typedef __attribute__((address_space(3))) const volatile double lds_cvd;
#else
  typedef __attribute__((address_space(3))) const double lds_cvd;
#endif
  lds_cvd *lds_v = (lds_cvd *)lds;

  for (int i = tid; i < np; i += kTauBlock) lds[L.tb + i] = a.temperature[i];

  const long ntiles = ((long)ncol + kTauBlock - 1) / kTauBlock;
  const long t_begin = ntiles * blockIdx.x / gridDim.x;
  const long t_end = ntiles * (blockIdx.x + 1) / gridDim.x;
  int slab_lo = -1;   // 0-based first staged pressure row; -1 = nothing staged
  const double *plev0 = a.plev + (long)ncol * j, *plev1 = a.plev + (long)ncol * (j + 1);

  for (long seg = t_begin; seg < t_end; seg += kSeg) {
    const long seg_end = seg + kSeg < t_end ? seg + kSeg : t_end;
    // ---- pre-pass: pressure-row range of the segment ----
    int vmin = np, vmax = 0;
    for (long tile = seg; tile < seg_end; ++tile) {
      const long c = tile * kTauBlock + tid;
      if (c < ncol) {
        const int ip0 = pressure_point(plev0[c], plev1[c], a.lp0, a.dlp, np).ip0;
        vmin = min(vmin, ip0);
        vmax = max(vmax, ip0);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      vmin = min(vmin, __shfl_xor(vmin, o));
      vmax = max(vmax, __shfl_xor(vmax, o));
    }
    __syncthreads();   // previous segment's LDS reads (and red[] reads) are done
    if (lane == 0) { red[2 * wave] = vmin; red[2 * wave + 1] = vmax; }
    __syncthreads();
    int ipmin = red[0], ipmax = red[1];
#pragma unroll
    for (int w = 1; w < kTauWaves; ++w) { ipmin = min(ipmin, red[2 * w]); ipmax = max(ipmax, red[2 * w + 1]); }

    if (R >= 2 && ipmin <= ipmax && !(slab_lo >= 0 && ipmin - 1 >= slab_lo && ipmax <= slab_lo + R - 1)) {
      slab_lo = min(ipmin - 1, np - R);
      // stage: one (row, gas) item per wave-iteration, lanes over g
      const int rows_b = R * nt;
      const int items_b = rows_b * a.nbil;
      for (int q = wave; q < items_b; q += kTauWaves) {
        const int s = q % a.nbil, rb = q / a.nbil;
        const int ipl = rb % R, it = rb / R;
        const double *src = a.seq[a.bil_seq[s]].coef + (long)ng * ((slab_lo + ipl) + (long)np * it);
        double *dst = lds + L.bil + rb * L.SB + s * ng;
        for (int g = lane; g < ng; g += 64) dst[g] = src[g];
      }
      if (a.lut >= 0) {
        const double *coef = a.seq[a.lut].coef;
        const int rows_l = rows_b * nv_lut;
        for (int q = wave; q < rows_l; q += kTauWaves) {
          const int ipl = q % R, itv = q / R;   // itv = it + nt*iv
          const double *src = coef + (long)ng * ((slab_lo + ipl) + (long)np * itv);
          double *dst = lds + L.lut + q * L.SL;
          for (int g = lane; g < ng; g += 64) dst[g] = src[g];
        }
      }
    }
    __syncthreads();

    for (long tile = seg; tile < seg_end; ++tile) {
      const long c = tile * kTauBlock + tid;
      const bool valid = c < ncol;
      const long cc = valid ? c : (long)ncol - 1;
      // ---- setup: one round of global loads ----
      const double p0 = plev0[cc], p1 = plev1[cc];
      const double T = a.tlay[cc + (long)ncol * j];
      double w[kPass];   // per-gas weight (:143-149); for the LUT gas also the trilinear weight
#pragma unroll
      for (int k = 0; k < kPass; ++k) {
        w[k] = 0.;
        if (k < a.nseq) {
          const SeqGas &e = a.seq[k];
          w[k] = e.vmr ? e.vmr[cc * e.cs + j * e.ls] : e.scalar;   // vmr for now
        }
      }
      const PPoint pp = pressure_point(p0, p1, a.lp0, a.dlp, np);
      const int ip0 = pp.ip0;
      const double pw0 = pp.pw0, pw1 = pp.pw1;
      const int ipl = ip0 - 1 - slab_lo;
      const bool inslab = (R >= 2) && slab_lo >= 0 && ipl >= 0 && ipl + 1 <= R - 1;
      const bool fast = __all(inslab);   // wave-uniform

      // :131-140 temperature interpolation point
      const double t0 = pw0 * lds[L.tb + ip0 - 1] + pw1 * lds[L.tb + ip0];
      double temperature_index = (T - t0) / a.dt;
      temperature_index = 1. + selmax(0., selmin(temperature_index, (double)nt - 1.0001));
      const int it0 = (int)temperature_index;   // 1-based
      const double tw1 = temperature_index - it0;
      const double tw0 = 1. - tw1;

      const double dp = p1 - p0;
      const double simple_weight = a.gw * dp;   // :143

      // look_up_table gas of this pass: vmr interpolation point, :153-163
      int iv0 = 1;
      double vw0 = 1., vw1 = 0.;
#pragma unroll
      for (int k = 0; k < kPass; ++k) {
        if (k == a.lut) {
          const SeqGas &e = a.seq[k];
          const double log_vmr = log(selmax(w[k], e.mf0));
          double vmr_index = (log_vmr - e.log_mf0) / e.d_log_vmr;
          vmr_index = 1. + selmax(0., selmin(vmr_index, (double)e.nv - 1.001));
          iv0 = (int)vmr_index;   // 1-based
          vw1 = vmr_index - iv0;
          vw0 = 1. - vw1;
        }
      }
      // vmr -> weight.  Tables without negative entries need no per-g clamp: a negative weight
      // gives od <= 0, which :234-238 turns into 0, exactly what a zero weight produces.
#pragma unroll
      for (int k = 0; k < kPass; ++k) {
        if (k < a.nseq) {
          const SeqGas &e = a.seq[k];
          double x = e.code == 3 ? simple_weight * (w[k] - e.ref)
                                 : (e.code == 0 ? simple_weight : simple_weight * w[k]);
          if (!ANYCLAMP) x = x < 0. ? 0. : x;
          w[k] = x;
        }
      }

      for (int gb = 0; gb < ng; gb += GC) {
        double acc[GC];
        if (a.accumulate) {
#pragma unroll
          for (int g = 0; g < GC; ++g)
            acc[g] = (FULL || gb + g < ng) ? a.tau[cc + (long)ncol * (j + (long)nlay * (gb + g))] : 0.;
        } else {
#pragma unroll
          for (int g = 0; g < GC; ++g) acc[g] = 0.;   // :346
        }

        for (int k = 0; k < a.nseq; ++k) {   // gas_desc order, :348
          double wk = w[0];
#pragma unroll
          for (int i = 1; i < kPass; ++i) wk = (k == i) ? w[i] : wk;
          if (k == a.lut) {
            // ---- look_up_table gas, tri-linear, :167-178 ----
#define ECCKD_TRILINEAR(LD, o000, dP, dT, dV, SPAN)                                                \
  _Pragma("unroll") for (int g = 0; g < GC; ++g) {                                             \
    if (FULL || gb + g < ng) {                                                                 \
      const double c000 = LD(o000 + g), c100 = LD(o000 + dP + g);                              \
      const double c010 = LD(o000 + dT + g), c110 = LD(o000 + dT + dP + g);                    \
      const double c001 = LD(o000 + dV + g), c101 = LD(o000 + dV + dP + g);                    \
      const double c011 = LD(o000 + dV + dT + g), c111 = LD(o000 + dV + dT + dP + g);          \
      double od = wk * (vw0 * (tw0 * (pw0 * c000 + pw1 * c100) +                               \
                               tw1 * (pw0 * c010 + pw1 * c110)) +                              \
                        vw1 * (tw0 * (pw0 * c001 + pw1 * c101) +                               \
                               tw1 * (pw0 * c011 + pw1 * c111)));                              \
      if (ANYCLAMP) od = od < 0. ? 0. : od;                                                    \
      acc[g] = acc[g] + od;                                                                    \
    }                                                                                          \
    if (FULL && g % (SPAN) == (SPAN) - 1) __builtin_amdgcn_sched_barrier(0);                   \
  }
            if (fast) {
              const int o = L.lut + (ipl + R * ((it0 - 1) + nt * (iv0 - 1))) * L.SL + gb;
              const int dP = L.SL, dT = R * L.SL, dV = R * nt * L.SL;
// volatile (optional): keeps single ds_read_b64 instead of paired ds_read2_b64
#define LDS_LD(x) (lds_v[x])
              ECCKD_TRILINEAR(LDS_LD, o, dP, dT, dV, kSpanLut)
            } else {
              const double *cp = a.seq[k].coef +
                                 (long)ng * ((ip0 - 1) + (long)np * ((it0 - 1) + (long)nt * (iv0 - 1))) + gb;
              const long dP = ng, dT = (long)ng * np, dV = (long)ng * np * nt;
#define GLB_LD(x) cp[x]
              ECCKD_TRILINEAR(GLB_LD, 0L, dP, dT, dV, 1)
            }
          } else {
            // ---- bi-linear gases (:198-203 with weight, :216-221 with simple_weight) ----
#define ECCKD_BILINEAR(LD, o00, dP, dT, SPAN)                                                      \
  _Pragma("unroll") for (int g = 0; g < GC; ++g) {                                             \
    if (FULL || gb + g < ng) {                                                                 \
      const double c00 = LD(o00 + g), c10 = LD(o00 + dP + g);                                  \
      const double c01 = LD(o00 + dT + g), c11 = LD(o00 + dT + dP + g);                        \
      double od = wk * (tw0 * (pw0 * c00 + pw1 * c10) + tw1 * (pw0 * c01 + pw1 * c11));        \
      if (ANYCLAMP) od = od < 0. ? 0. : od;                                                    \
      acc[g] = acc[g] + od;                                                                    \
    }                                                                                          \
    if (FULL && g % (SPAN) == (SPAN) - 1) __builtin_amdgcn_sched_barrier(0);                   \
  }
            if (fast) {
              const int o = L.bil + (ipl + R * (it0 - 1)) * L.SB + a.seq[k].slot * ng + gb;
              const int dP = L.SB, dT = R * L.SB;
              ECCKD_BILINEAR(LDS_LD, o, dP, dT, kSpan)
            } else {
              const double *cp = a.seq[k].coef + (long)ng * ((ip0 - 1) + (long)np * (it0 - 1)) + gb;
              const long dP = ng, dT = (long)ng * np;
              ECCKD_BILINEAR(GLB_LD, 0L, dP, dT, 1)
            }
          }
        }

        if (valid) {
          if (SW) {
            const double moles = dp * a.gw;   // :313-314
#pragma unroll
            for (int g = 0; g < GC; ++g) {
              if (FULL || gb + g < ng) {
                const long o = c + (long)ncol * (j + (long)nlay * (gb + g));
                const double ray = moles * a.rayleigh[gb + g];   // :316
                const double t = acc[g] + ray;                    // :456
                a.tau[o] = t;
                if (a.ssa) { a.ssa[o] = ray / t; a.g[o] = 0.; }   // :459-460
              }
            }
          } else {
#pragma unroll
            for (int g = 0; g < GC; ++g)
              if (FULL || gb + g < ng) a.tau[c + (long)ncol * (j + (long)nlay * (gb + g))] = acc[g];
          }
        }
      }
    }
  }
}

template <int GC, bool FULL, bool ANYCLAMP, bool SW>
hipError_t launch_one(const TauArgs &a, size_t lds_bytes, hipStream_t s) {
  auto k = tau_kernel<GC, FULL, ANYCLAMP, SW>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  dim3 grid(a.col_chunks, a.nlay);
  hipLaunchKernelGGL(k, grid, dim3(kTauBlock), lds_bytes, s, a);
  return hipGetLastError();
}

template <int GC, bool FULL>
hipError_t launch_gc(const TauArgs &a, size_t lds, bool anyclamp, hipStream_t s) {
  const bool sw = a.rayleigh != nullptr;
  if (anyclamp)
    return sw ? launch_one<GC, FULL, true, true>(a, lds, s) : launch_one<GC, FULL, true, false>(a, lds, s);
  return sw ? launch_one<GC, FULL, false, true>(a, lds, s) : launch_one<GC, FULL, false, false>(a, lds, s);
}

}  // namespace

size_t tau_lds_bytes(int ng, int np, int nt, int nbil, int nv_lut, int R) {
  return sizeof(double) * (size_t)slab_layout(ng, np, nt, nbil, nv_lut, R).total;
}

int tau_slab_rows(int ng, int np, int nt, int nbil, int nv_lut) {
  int R = 0;
  for (int r = 2; r <= np; ++r) {
    if (tau_lds_bytes(ng, np, nt, nbil, nv_lut, r) <= (size_t)kLdsBudget) R = r; else break;
  }
  return R;
}

hipError_t launch_tau(TauArgs &a, hipStream_t s) {
  if (a.ncol <= 0 || a.nlay <= 0) return hipSuccess;
  const int nv_lut = a.lut >= 0 ? a.seq[a.lut].nv : 0;
  a.R = tau_slab_rows(a.ng, a.np, a.nt, a.nbil, nv_lut);
  const size_t lds = tau_lds_bytes(a.ng, a.np, a.nt, a.nbil, nv_lut, a.R);
  bool anyclamp = false;
  for (int k = 0; k < a.nseq; ++k) anyclamp |= a.seq[k].clamp != 0;
  // one block per CU (LDS-bound): aim at ~4 blocks per CU over the whole grid for balance
  const long ntiles = ((long)a.ncol + kTauBlock - 1) / kTauBlock;
  long chunks = (4L * 256 + a.nlay - 1) / a.nlay;
  if (chunks > ntiles) chunks = ntiles;
  if (chunks < 1) chunks = 1;
  a.col_chunks = (int)chunks;
  if (a.nseq > kTauPassGases) return hipErrorInvalidValue;
  // g-points per register chunk: ECCKD_TAU_GC (32 -> exact instantiations for 32/36/27 g-points)
#if ECCKD_TAU_GC == 8
  if (a.ng % 8 == 0) return launch_gc<8, true>(a, lds, anyclamp, s);
  if (a.ng % 9 == 0) return launch_gc<9, true>(a, lds, anyclamp, s);
  return launch_gc<8, false>(a, lds, anyclamp, s);
#elif ECCKD_TAU_GC == 16
  if (a.ng % 16 == 0) return launch_gc<16, true>(a, lds, anyclamp, s);
  if (a.ng % 18 == 0) return launch_gc<18, true>(a, lds, anyclamp, s);
  if (a.ng % 9 == 0) return launch_gc<9, true>(a, lds, anyclamp, s);
  return launch_gc<16, false>(a, lds, anyclamp, s);
#else
  if (a.ng % 32 == 0) return launch_gc<32, true>(a, lds, anyclamp, s);
  if (a.ng % 36 == 0) return launch_gc<36, true>(a, lds, anyclamp, s);
  if (a.ng % 27 == 0) return launch_gc<27, true>(a, lds, anyclamp, s);
  if (a.ng % 16 == 0) return launch_gc<16, true>(a, lds, anyclamp, s);
  return launch_gc<16, false>(a, lds, anyclamp, s);
#endif
}

}  // namespace ecckd
