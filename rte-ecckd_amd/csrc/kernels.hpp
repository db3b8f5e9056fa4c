// kernels.hpp -- launch interface between the C ABI (capi.cpp) and the gfx950 kernels.
// All pointers are device pointers on the current device; every array is column-major with the
// column index fastest (the reference's layout, src/gas_optics_ecckd.f90:69-72).
#pragma once
#include <hip/hip_runtime.h>

namespace ecckd {

constexpr int kMaxSeq = 16;          // src/gas_optics_ecckd.f90:24 (at most 16 tables)
constexpr int kTauPassGases = 10;    // gases one tau launch accumulates (more -> another pass)
constexpr int kLdsBudget = 160 * 1024;

// One entry of the per-call gas sequence: a table of the model matched to a gas of gas_desc,
// in gas_desc order (src/gas_optics_ecckd.f90:348-374).
struct SeqGas {
  const double *coef;     // (ng,np,nt,nv) table in device memory
  const double *vmr;      // device pointer or nullptr -> scalar
  long long cs, ls;       // vmr(i,l) = vmr[i*cs + l*ls]
  double scalar;
  double ref;             // reference_mole_fraction (relative_linear)
  double mf0;             // mole_fraction(1)                       (look_up_table)
  double log_mf0;         // log(mole_fraction(1))                  (host libm, as the reference)
  double d_log_vmr;       // log(mole_fraction(2)/mole_fraction(1)) (host libm)
  int code;               // ECCKD_NONE / LINEAR / LOOK_UP_TABLE / RELATIVE_LINEAR
  int nv;
  int slot;               // bilinear: slot inside a slab row; LUT: unused
  int clamp;              // 1 if the table holds negative coefficients -> per-g clamp needed
};

struct TauArgs {
  int ncol, nlay, ng, np, nt;
  const double *plev, *tlay;
  const double *temperature;   // (np,nt) device; only T(:,1) is used (:131-132)
  const double *zero;          // a zero word of the model's device image (loads of unused / scalar gas slots)
  double lp0, dlp, dt, gw;     // :104-107
  int nseq;
  SeqGas seq[kMaxSeq];
  int nbil;                    // bilinear gases in this pass (slab row holds nbil*ng values)
  int bil_seq[kMaxSeq];        // slab slot -> index into seq
  int lut;                     // index into seq of the look_up_table gas, or -1
  // Merged slot (fused kernel, fast arithmetic; merge_scalar_gases()): the gases of this pass whose mole fraction is one
  // number for the whole call -- scalar entries of gas_desc and the none_ composite -- share ONE slab slot holding
  // sum_k merge_mult[k] * coefficient_k, built while the slab is staged; its weight is simple_weight alone.
  int merge_slot;              // slab slot of the merged table, or -1
  int nmerge;                  // gases in it (0 or >= 2)
  int merge_seq[kMaxSeq];      // their indices into seq, in gas_desc order
  double merge_mult[kMaxSeq];  // vmr, vmr - reference, or 1 (none_): >= 0, finite, rounded to the working precision
  int accumulate;              // start from the tau already in memory (later passes)
  double *tau;
  // shortwave epilogue (src/gas_optics_ecckd.f90:455-460); rayleigh == nullptr for LW
  const double *rayleigh;
  double *ssa, *g;
  // launch geometry chosen by the host
  int R;                       // pressure rows of the LDS slab
  int col_chunks;              // grid.x; each block walks tiles chunk by chunk
  int seg;                     // fused kernel: tiles per segment (slab-range pre-pass + barriers once per segment)
};

// Division by a wave-uniform constant, d with its reciprocal (kernels_gas_fused.hip: udiv()).
struct UDiv { double d, r; int exact; };

// Per-slot view of the gases of a pass for the fused kernel: slot s < nbil is the s-th bilinear
// gas, slot kTauPassGases is the look_up_table gas.  `vmr` is ALWAYS a valid device address (the
// load is issued unconditionally, all slots in one round; slots without an array read TauArgs::zero, a
// word the model owns, so that a NaN in the caller's arrays stays in its own column).  The mole fraction that enters the
// weight is fma(alpha, loaded, beta) with alpha in {0, 1}, which spells every case of
// src/gas_optics_ecckd.f90:143-149 exactly: array/linear (1, 0); array/relative_linear (1, -ref);
// scalar/linear (0, scalar); scalar/relative_linear (0, scalar - ref); none_ (0, 1); unused (0, 0).
struct SlotArgs {
  const double *vmr;
  long long ls;                // layer stride, elements
  unsigned cs_bytes;           // column stride, bytes (32-bit: checked on the host)
  double alpha, beta;
};

// Fused gas-optics launch (kernels_gas_fused.hip): the tau arguments plus the Planck side.
struct FusedArgs {
  TauArgs tau;
  UDiv ud_dlp, ud_dt, ud_dlv, ud_pdt;          // filled by launch_gas_fused
  SlotArgs slot[kTauPassGases + 1];            // filled by launch_gas_fused
  int f32;                     // 1: every data pointer addresses float arrays (LW fused path only)
  int slab32;                  // fp64 call: stage the tables in LDS as the float32 they are (model checked: every value is
                               // float32-representable): 1 always, 2 where the columns are spread over many pressure rows
                               // (spread probe); prepare_gas_fused clears it where no such instantiation exists
  int *choose_buf;             // slab32 == 2: one int of device memory owned by the call's stream (the probe's counter), or null
  const int *choose;           // (set by launch_gas_fused) the probe's counter, or null: unconditional launch
  int choose_total;            // (set by launch_gas_fused) waves of 64 columns in the call
  int mode;                    // 0 tau only, 1 longwave (tau + Planck sources), 2 shortwave epilogue
  int ntp;
  int pw;                      // Planck rows staged in LDS (ntp, or a window); filled by launch_gas_fused
  const double *planck;        // (ng,ntp) device
  double pt0, pdt;             // temperature_planck(1), (2)-(1)
  const double *tlev, *tsfc;   // tlev may be nullptr
  double *lay_source, *lev_source_inc, *lev_source_dec, *sfc_source;
};

struct PlanckArgs {
  int ncol, nlay, ng, ntp;
  const double *planck;        // (ng,ntp) device
  double t0, dt;               // temperature_planck(1), (2)-(1)   (:271-272)
  const double *tlay, *tlev, *tsfc;    // tlev may be nullptr
  double *lay_source, *lev_source_inc, *lev_source_dec, *sfc_source;
  int lev_chunks;              // grid.y
};

struct RteLwArgs {
  int ncol, nlay, ng, top_at_1, nmus;
  double Ds[4], wts[4];
  const double *tau, *lay_source, *lev_source_inc, *lev_source_dec, *sfc_source;
  const double *sfc_emis;      // (nband,ncol)
  int nband;
  unsigned char gpt2band[256]; // 0-based band of each g-point
  double *flux_up, *flux_dn;
  double *scratch;             // generic-nlay path only
  int f32;                     // 1: the data pointers address float arrays
  int shared_levels;           // 1: lev_source_inc(:,l,:) == lev_source_dec(:,l+1,:): each level is read once
  const double *inc_flux;      // (ncol,ng) incident diffuse flux at the top of the domain, or nullptr (none)
  // version switches of the un-pinned RTE-RRTMGP solver (ecckd_set_solver_option)
  double tau_thresh;           // lw_source_noscat: series below this optical depth (sqrt(epsilon) of the precision)
  int series3;                 // 0: tau*(0.5 - tau/3) (v1.5), 1: tau*(0.5 + tau*(-1/3 + tau/8))
  int inc_isotropic;           // 0: I_dn(top) = inc_flux/(2 pi w_k) per angle, 1: inc_flux/pi
  // implementation choice for fp64 / 60 layers (ecckd_set_solver_option "lw_solver", "lw_split_seg")
  int use_split;               // 1: layer-split solver (kernels_rte_lw_split.hip), 0: register-resident solver
  int split_seg;               // layers per wave of the layer-split solver: 10, 12 or 15
  // tail split (rte_lw_tail_plan): tiles from tail_first on are solved one g-pair iteration per wave; -1: none
  long tail_first = -1;
  double *partials = nullptr;  // [tail tile][iteration][dn, up][nlay+1][columns per tile]
};

struct RteSwArgs {
  int ncol, nlay, ng, top_at_1;
  const double *tau, *ssa, *g, *mu0, *toa;
  const double *alb_dir, *alb_dif;   // (nband,ncol)
  int nband;
  unsigned char gpt2band[256];
  double *flux_up, *flux_dn, *flux_dir;   // flux_dir may be nullptr
  double *scratch;
  int exact_division;          // 1 (reference-order arithmetic mode): IEEE `/`; 0: reciprocal + Newton steps
  // version switches (ecckd_set_solver_option)
  double k_floor;              // lower bound of (gamma1-gamma2)(gamma1+gamma2) under the square root (1e-12)
  int dir_clamp;               // 1: Rdir/Tdir energy clamps of later RTE-RRTMGP releases
  // tail split (rte_sw_tail_plan): tiles from tail_first on are solved one g-point group per wave; -1: none
  long tail_first = -1;
  double *partials = nullptr;  // [tail tile][group][up, dn, dir][nlay+1][columns per tile]
  int f32 = 0;                 // 1: the data pointers address float arrays (layer-systolic solver only)
  // Layer-systolic solver (kernels_rte_sw_sys.hip; rte_sw_sys_applies()): tiles of 64 columns; tiles from sys_tail_first
  // on are solved in chunks of sys_gchunk g-points per block, the chunk sums go through `partials`
  // ([tail tile][chunk][up, dn, dir][nlay+1][64]) and are added in chunk order by rte_sw_tail_reduce.
  int use_sys = 0;
  long sys_tail_first = -1;
  int sys_gchunk = 0;
  int sys_npark = 0;           // (set by the launcher) lower waves that park the next g-point's coefficients in LDS
  unsigned sys_park_at = 0;    // (set by the launcher) byte offset of the parking areas in the block's LDS
  // Fused shortwave path (ecckd_sw_fluxes): `tau` is the total optical depth and nothing else is read per cell --
  // ssa = (moles*rayleigh(g))/tau with moles = (plev(l+1)-plev(l))*gw, g = 0, toa = solar(g): gas_optics_ext's own
  // expressions (src/gas_optics_ecckd.f90:313-317,455-472).  derive != 0 selects it; ssa / g / toa are then unused.
  int derive = 0;
  const double *plev = nullptr, *rayleigh = nullptr, *solar = nullptr;   // plev(ncol,nlay+1) device; (ng) device tables
  const double *toa_scale = nullptr;   // (ncol) or null: toa(i,g) = solar(g)*toa_scale(i), the driver's TSI rescaling (ecckd_rfmip_sw.F90:126-133)
  double gw = 0.;
};

// Spectral-output solvers (kernels_rte_gpt.hip): RTE-RRTMGP's kernel-level interfaces.  LW uses tau, lay_source,
// lev_source_*, sfc_emis(ncol,ng), sfc_src(ncol,ng), inc_flux(ncol,ng)|null, Ds/wts; SW uses tau, ssa, g, mu0(ncol),
// fdir_top(ncol,ng) (direct flux at the top = toa*mu0), inc_dif(ncol,ng)|null, alb_dir/alb_dif(ncol,ng).
// flux_up / flux_dn / flux_dir are (ncol,nlay+1,ng).
struct RteGptArgs {
  int ncol, nlay, ng, top_at_1, nmus;
  double Ds[4], wts[4];
  const double *tau, *lay_source, *lev_source_inc, *lev_source_dec, *sfc_emis, *sfc_src, *inc_flux;
  const double *ssa, *g, *mu0, *fdir_top, *inc_dif, *alb_dir, *alb_dif;
  double *flux_up, *flux_dn, *flux_dir;
  double tau_thresh, k_floor;
  int series3, inc_isotropic, dir_clamp;
};
hipError_t launch_lw_gpt(const RteGptArgs &a, hipStream_t s);
hipError_t launch_sw_gpt(const RteGptArgs &a, hipStream_t s);

// Host-side helpers -------------------------------------------------------------------------
int tau_slab_rows(int ng, int np, int nt, int nbil, int nv_lut);   // R that fits LDS (>= 0)
size_t tau_lds_bytes(int ng, int np, int nt, int nbil, int nv_lut, int R);
size_t rte_lw_scratch_bytes(int ncol, int nlay, int ng);
size_t rte_lw_tail_plan(const RteLwArgs &a, int slots, long *tail_first);
size_t rte_sw_scratch_bytes(int ncol, int nlay, int ng);
size_t rte_sw_tail_plan(const RteSwArgs &a, long *tail_first, size_t *partials_at);

hipError_t launch_tau(TauArgs &a, hipStream_t s);
// What prepare_gas_fused() decided for one pass.
struct FusedPlan {
  int empty = 0;               // ncol == 0: nothing to launch
  size_t lds_bytes = 0;
  int anyclamp = 0, GC = 0, NB = 0;
  int merged = 0;              // gases sharing the merged slot
  int slab_rows = 0, planck_rows = 0, col_chunks = 0;
};
hipError_t prepare_gas_fused(FusedArgs &a, FusedPlan &plan);
// Folds the call-constant gases of a pass into one slot (TauArgs::merge_*); rewrites nbil / bil_seq.  Returns the number
// of gases merged (0: nothing changed).  Only for the fused kernel (the reference-order kernels take each gas alone).
int merge_scalar_gases(TauArgs &t, int f32);
int fused_slab_rows(int ng, int np, int nt, int nbil, int nv_lut, int pl_rows, int min_rows, int anyclamp, int f32, int block = 0);   // block: threads per block (0: the default 512)
int fused_planck_rows(int ng, int np, int nt, int nbil, int nv_lut, int ntp, int anyclamp, int f32);
hipError_t launch_gas_fused(FusedArgs &a, hipStream_t s);
hipError_t launch_planck(PlanckArgs &a, hipStream_t s);
// the Planck sources as a stand-alone fast kernel (paired 16-byte stores)
hipError_t launch_planck_pair(const PlanckArgs &a, int f32, hipStream_t s);
size_t planck_pair_lds_bytes(int ng, int ntp, int f32);
UDiv make_udiv(double d, int f32);
hipError_t launch_toa_src(const double *solar, int ncol, int ng, double *toa_src, int f32, hipStream_t s);
// out(i) = sum_b planes(i, b): broadband from per-band fluxes
hipError_t launch_sum_planes(const double *planes, int nplanes, size_t n, double *out, int f32, hipStream_t s);
hipError_t launch_rte_lw(const RteLwArgs &a, hipStream_t s);
// layer-split form (kernels_rte_lw_split.hip): fp64, 60 layers
bool rte_lw_split_applies(const RteLwArgs &a);
bool rte_lw_planck_fits(int ng, int ntp);   // the Planck-recomputing form: does the model's table fit in LDS?
hipError_t launch_rte_lw_split(const RteLwArgs &a, hipStream_t s);
// ... with the Planck sources recomputed in the solver from tlay(ncol,nlay), tlev(ncol,nlay+1), tsfc(ncol) and the
// model's table planck(ng,ntp) (a.lay_source / lev_source_* / sfc_source are not read)
hipError_t launch_rte_lw_planck(const RteLwArgs &a, const double *planck, int ntp, double t0, double dt, const double *tlay,
                                const double *tlev, const double *tsfc, hipStream_t s);
hipError_t launch_rte_sw(const RteSwArgs &a, hipStream_t s);
// layer-systolic shortwave solver (kernels_rte_sw_sys.hip): any precision, nlay <= 60
bool rte_sw_sys_applies(const RteSwArgs &a);
size_t rte_sw_sys_plan(RteSwArgs &a, int cus);   // fills sys_tail_first / sys_gchunk; returns the bytes of `partials` (0: none)
hipError_t launch_rte_sw_sys(const RteSwArgs &a, int cus, hipStream_t s);
// partial sums of the tail tiles -> fluxes, in chunk order (kernels_rte_sw.hip)
hipError_t launch_rte_sw_tail_reduce(const RteSwArgs &a, int nchunks, int cw, long tail_first, long ntail, hipStream_t s);

}  // namespace ecckd
