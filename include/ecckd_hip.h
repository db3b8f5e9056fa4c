/*
 * ecckd_hip.h -- C ABI of librte_ecckd_hip.so: the MI355X (gfx950) implementation of the
 * rte-ecckd hot path (ecCKD gas optics + RTE LW/SW flux solvers).
 *
 * This is the drop-in boundary.  Every entry point is `extern "C"`, takes plain pointers and
 * sizes, and is what a Fortran `iso_c_binding` interface block (or ctypes) binds; the
 * reference-side binding is shown in INTEGRATION.md and shipped in
 * rte-ecckd_amd/fortran/gas_optics_ecckd.F90.  Citations `file:line` are into the reference
 * repository (earth-system-radiation/rte-ecckd).
 *
 * Conventions
 *   - All arrays are contiguous, Fortran column-major, column index fastest, fp64:
 *     plev(ncol,nlay+1), tlay(ncol,nlay), tau(ncol,nlay,ngpt), flux(ncol,nlay+1) ...
 *   - `memspace` says where the DATA arrays live: ECCKD_HOST (the library stages them
 *     through device buffers it owns, and synchronises before returning) or ECCKD_DEVICE
 *     (pointers are device pointers on the model's GPU; the call is asynchronous on
 *     `stream`).  Small descriptor arrays (gas names, pointer tables, strides, Ds/weights,
 *     band2gpt) are always host memory.
 *     ECCKD_DEVICE calls never synchronise and never allocate once a (shape, stream) pair has been
 *     seen: they can be captured in a HIP graph after one warm-up call on the capturing stream (the
 *     solvers' scratch rings are per stream, see ecckd_set_stream_scratch).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - Return value: 0 on success, non-zero on error; the message (same texts as the
 *     reference's character(len=128) results, src/gas_optics_ecckd.f90:331,393,442) is
 *     returned by ecckd_last_error().  There is no CPU fallback: without a usable GPU every
 *     compute entry point fails with an error.
 *   - Re-entrancy: a model is immutable after ecckd_model_finalize (the reference's
 *     `intent(in) :: this`, :385,:434); concurrent calls on different streams are safe in
 *     ECCKD_DEVICE mode.  ECCKD_HOST mode uses a per-model staging arena and is serialised
 *     by an internal mutex.
 */
#ifndef ECCKD_HIP_H
#define ECCKD_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ECCKD_HOST 0
#define ECCKD_DEVICE 1
/* ECCKD_MIXED: the big arrays live on the device, the small ones on the host -- the calling convention of a
 * host model (Fortran, say) that keeps optical_props / sources in HBM between gas_optics and rte_* but holds
 * its atmosphere and wants its fluxes in host memory.  Device pointers: tau, ssa, g, lay_source,
 * lev_source_inc, lev_source_dec, sfc_source.  Host pointers, staged by the library: plev, tlay, tlev, tsfc,
 * gas arrays, toa_src / toa_flux, mu0, sfc_emis, albedos, inc_flux, fluxes.  The call synchronises before it
 * returns.  About 3.4 KB per column cross the bus instead of 34 KB (ECCKD_HOST).  Accepted by
 * ecckd_gas_optics_lw / _sw, ecckd_rte_lw (and _shared_levels, _inc_flux) and ecckd_rte_sw. */
#define ECCKD_MIXED 2

/* concentration_dependence_code values, src/gas_optics_ecckd.f90:54-57 */
#define ECCKD_NONE 0
#define ECCKD_LINEAR 1
#define ECCKD_LOOK_UP_TABLE 2
#define ECCKD_RELATIVE_LINEAR 3

#define ECCKD_MAX_GASES 16   /* src/gas_optics_ecckd.f90:24-25 */
#define ECCKD_NAME_LEN 32    /* character(len=32) gas names, :25 */

typedef struct ecckd_model ecckd_model_t; /* replaces type(ty_gas_optics_ecckd), :23-48 */

/* Message of the most recent failing call on this thread (never NULL). */
const char *ecckd_last_error(void);

/* ---------------------------------------------------------------------------------------
 * Model construction.  Two routes, both ending in an immutable device-resident model:
 *   (1) ecckd_model_load  = load_and_init, example/rfmip-rad-irf/mo_load_coefficients.F90:19-146
 *       (own netCDF-3 classic reader; no libnetcdf needed);
 *   (2) ecckd_model_begin / _set_* / _add_gas / _finalize = filling the public members of
 *       ty_gas_optics_ecckd directly (src/gas_optics_ecckd.f90:24-36), for hosts that
 *       already hold the tables.
 * --------------------------------------------------------------------------------------- */

/* load_and_init(ecckd, filename, available_gases): available_gases is accepted and ignored
 * by the reference (mo_load_coefficients.F90:19,23) and therefore has no parameter here. */
int ecckd_model_load(const char *filename, int device, ecckd_model_t **model);

/* log_pressure(np) [ln Pa] (:27), temperature(np,nt) [K] (:34); ng = size(gpoint_fraction,2)
 * (:26, only its extent is ever used, :110). */
int ecckd_model_begin(int ng, int np, int nt, const double *log_pressure,
                      const double *temperature, ecckd_model_t **model);
/* planck_function(ng,ntp) (:30), temperature_planck(ntp) (:35)  -> source_is_internal */
int ecckd_model_set_planck(ecckd_model_t *model, int ntp, const double *temperature_planck,
                           const double *planck_function);
/* solar_irradiance(ng) (:33), rayleigh_molar_scattering_coeff(ng) (:31) -> source_is_external */
int ecckd_model_set_solar(ecckd_model_t *model, const double *solar_irradiance,
                          const double *rayleigh_molar_scattering_coeff);
/* ty_optical_props%init(band_lims_wvn(2,nband), band2gpt(2,nband)) as called at
 * mo_load_coefficients.F90:74; band2gpt is 1-based, inclusive. */
int ecckd_model_set_bands(ecckd_model_t *model, int nband, const double *band_lims_wvn,
                          const int *band2gpt);
/* One AbsorptionTable (:13-19) + its name (:25).  coefficient is (ng,np,nt,nv); nv = 1 and
 * mole_fraction = NULL unless code == ECCKD_LOOK_UP_TABLE. */
int ecckd_model_add_gas(ecckd_model_t *model, const char *name, int concentration_dependence_code,
                        int composite_only, int nv, const double *mole_fraction,
                        double reference_mole_fraction, const double *coefficient);
/* Upload the tables to GPU `device` (HIP ordinal) and freeze the model.  device == -1 makes a
 * host-only model (also accepted by ecckd_model_load): the getters work, compute calls fail. */
int ecckd_model_finalize(ecckd_model_t *model, int device);
void ecckd_model_destroy(ecckd_model_t *model);

/* Type-bound getters of ty_gas_optics_ecckd (:38-45, :477-553) and of its parent
 * (get_ngpt/get_nband, used at ecckd_rfmip_sw.F90:78-79). */
int ecckd_model_get_ngpt(const ecckd_model_t *model);
int ecckd_model_get_nband(const ecckd_model_t *model);
int ecckd_model_get_ngas(const ecckd_model_t *model);                       /* :477-483 */
/* name is ECCKD_NAME_LEN bytes, NUL-terminated; index is 0-based. */
int ecckd_model_get_gas_name(const ecckd_model_t *model, int index, char *name);   /* :507-513 */
int ecckd_model_source_is_internal(const ecckd_model_t *model);             /* :487-493 */
int ecckd_model_source_is_external(const ecckd_model_t *model);             /* :497-503 */
double ecckd_model_get_press_min(const ecckd_model_t *model);               /* :517-523 */
double ecckd_model_get_press_max(const ecckd_model_t *model);               /* :527-533 */
double ecckd_model_get_temp_min(const ecckd_model_t *model);                /* :537-543 */
double ecckd_model_get_temp_max(const ecckd_model_t *model);                /* :547-553 */
double ecckd_model_get_total_solar_irradiance(const ecckd_model_t *model);  /* :36 */
/* band2gpt(2,nband), 1-based inclusive; band_lims_wvn(2,nband) */
int ecckd_model_get_band2gpt(const ecckd_model_t *model, int *band2gpt);
int ecckd_model_get_band_lims_wvn(const ecckd_model_t *model, double *band_lims_wvn);
int ecckd_model_get_device(const ecckd_model_t *model);

/* ---------------------------------------------------------------------------------------
 * gas_optics.  `gas_desc` (type(ty_gas_concs)) crosses the boundary as:
 *   ngas          gas_desc%get_num_gases()                         (:340)
 *   gas_names     ngas records of ECCKD_NAME_LEN chars, blank- or NUL-padded, in
 *                 gas_desc%get_gas_names() order                  (:342; order matters: :348)
 *   vmr[j]        data pointer of gas j in `memspace`, or NULL to use vmr_scalar[j]
 *   vmr_col_stride[j], vmr_lay_stride[j]
 *                 element strides so that get_vmr's broadcast (:351) is
 *                 vmr(i,l) = vmr[j][i*col_stride + l*lay_stride]:
 *                 (1,ncol) for a (ncol,nlay) array, (0,1) for a profile, (1,0) per column
 *   vmr_scalar[j] value used when vmr[j] == NULL (a ty_gas_concs scalar)
 * Gases unknown to the model are skipped, composite-only gases contribute the composite
 * table once (:358-373).
 * --------------------------------------------------------------------------------------- */

/* gas_optics_int (:381-426): tau, lay_source, lev_source_inc, lev_source_dec are
 * (ncol,nlay,ngpt), sfc_source is (ncol,ngpt).  `play` and `col_dry` are unused by the
 * reference (:386,:394) and have no parameter.  tlev == NULL reproduces the reference:
 * tau, lay_source and sfc_source are written, then the call fails with
 * "tlev is required for ecckd" (:414-417).  Unlike the reference (:266-269 via :407) the
 * caller's lay_source is never reallocated. */
int ecckd_gas_optics_lw(const ecckd_model_t *model, int ncol, int nlay, const double *plev,
                        const double *tlay, const double *tsfc, const double *tlev, int ngas,
                        const char *gas_names, const double *const *vmr,
                        const long long *vmr_col_stride, const long long *vmr_lay_stride,
                        const double *vmr_scalar, double *tau, double *lay_source,
                        double *lev_source_inc, double *lev_source_dec, double *sfc_source,
                        int memspace, void *stream);

/* Single-precision flavour (a host built with RTE-RRTMGP's RTE_USE_SP, i.e. wp = real32): every
 * data array is float, arithmetic is float.  Implemented for the fused fast longwave path (one
 * pass, Planck table next to >= 3 slab rows: true for all ecCKD files in single precision); other
 * cases fail with a message.  vmr_scalar stays double (values, not arrays). */
int ecckd_gas_optics_lw_f32(const ecckd_model_t *model, int ncol, int nlay, const float *plev,
                            const float *tlay, const float *tsfc, const float *tlev, int ngas,
                            const char *gas_names, const float *const *vmr,
                            const long long *vmr_col_stride, const long long *vmr_lay_stride,
                            const double *vmr_scalar, float *tau, float *lay_source,
                            float *lev_source_inc, float *lev_source_dec, float *sfc_source,
                            int memspace, void *stream);

/* gas_optics_ext (:431-473): tau, ssa, g are (ncol,nlay,ngpt); toa_src is (ncol,ngpt).
 * ssa == NULL or g == NULL stands for an optical_props that is not ty_optical_props_2str:
 * tau (gas + Rayleigh) is written and the call fails with
 * "shortwave must use ty_optical_props_2str" (:461-463). */
int ecckd_gas_optics_sw(const ecckd_model_t *model, int ncol, int nlay, const double *plev,
                        const double *tlay, int ngas, const char *gas_names,
                        const double *const *vmr, const long long *vmr_col_stride,
                        const long long *vmr_lay_stride, const double *vmr_scalar, double *tau,
                        double *ssa, double *g, double *toa_src, int memspace, void *stream);

/* ---------------------------------------------------------------------------------------
 * RTE solvers (RTE-RRTMGP rte_lw / rte_sw as called at ecckd_rfmip_lw.F90:130-135 and
 * ecckd_rfmip_sw.F90:148-154) with the broadband g-point reduction
 * (ty_fluxes_broadband%reduce) fused in.  flux_up, flux_dn are (ncol,nlay+1).
 * --------------------------------------------------------------------------------------- */

/* No-scattering LW with Gauss-Jacobi quadrature, n_gauss_angles in 1..4.  sfc_emis is
 * (nband,ncol) as in rte_lw; band2gpt(2,nband) (1-based, inclusive) expands it to g-points
 * (ecckd_rfmip_lw.F90:112-116).  ngpt <= 256. */
int ecckd_rte_lw(int device, int ncol, int nlay, int ngpt, int top_at_1, int n_gauss_angles,
                 const double *tau, const double *lay_source, const double *lev_source_inc,
                 const double *lev_source_dec, const double *sfc_source, int nband,
                 const int *band2gpt, const double *sfc_emis, double *flux_up, double *flux_dn,
                 int memspace, void *stream);

/* ecckd_rte_lw for level sources that hold ONE value per level:
 *     lev_source_inc(:,l,:) == lev_source_dec(:,l+1,:)   for l = 1 .. nlay-1.
 * That is what ecckd_gas_optics_lw writes (src/gas_optics_ecckd.f90:419-424: both arrays are slices
 * of one (ncol,nlay+1,ngpt) buffer), but it is NOT part of RTE-RRTMGP's rte_lw contract (RRTMGP's
 * own gas optics fills the two arrays with different values), so the caller has to assert it by
 * calling this entry point.  The solver then reads each level once (24 instead of 32 B/cell): the
 * array that holds the far edge of the layers in walking order (lev_source_inc when top_at_1) in
 * full, the other one for the first layer only.  Same arithmetic and the same fluxes, bit for bit,
 * as ecckd_rte_lw when the assertion holds.  Same arguments. */
int ecckd_rte_lw_shared_levels(int device, int ncol, int nlay, int ngpt, int top_at_1,
                               int n_gauss_angles, const double *tau, const double *lay_source,
                               const double *lev_source_inc, const double *lev_source_dec,
                               const double *sfc_source, int nband, const int *band2gpt,
                               const double *sfc_emis, double *flux_up, double *flux_dn,
                               int memspace, void *stream);

/* ecckd_rte_lw with the optional incident diffuse flux at the top of the domain: rte_lw's `inc_flux(ncol,ngpt)`
 * argument [RTE-RRTMGP; no reference call site passes it: ecckd_rfmip_lw.F90:130-135].  I_dn(top) =
 * inc_flux/(2 pi w_k) for quadrature angle k (see the solver option
 * lw_inc_flux_isotropic).  inc_flux == NULL is ecckd_rte_lw.  fp64, generic level sources. */
int ecckd_rte_lw_inc_flux(int device, int ncol, int nlay, int ngpt, int top_at_1, int n_gauss_angles,
                          const double *tau, const double *lay_source, const double *lev_source_inc,
                          const double *lev_source_dec, const double *sfc_source, int nband,
                          const int *band2gpt, const double *sfc_emis, const double *inc_flux,
                          double *flux_up, double *flux_dn, int memspace, void *stream);

/* Single-precision flavour of ecckd_rte_lw. */
int ecckd_rte_lw_f32(int device, int ncol, int nlay, int ngpt, int top_at_1, int n_gauss_angles,
                     const float *tau, const float *lay_source, const float *lev_source_inc,
                     const float *lev_source_dec, const float *sfc_source, int nband,
                     const int *band2gpt, const float *sfc_emis, float *flux_up, float *flux_dn,
                     int memspace, void *stream);

/* ---------------------------------------------------------------------------------------
 * RTE-RRTMGP's KERNEL-level solver interfaces [RTE-ext: mo_rte_solver_kernels.F90 / mo_fluxes_broadband_kernels.F90 of the
 * v1.5 era; that library is not part of the reference tree -- reference Makefile:19,33 links it]: spectral fluxes
 * (ncol,nlay+1,ngpt), per-g-point boundary conditions (ncol,ngpt), quadrature passed in, the g-point sum left to
 * sum_broadband.  include/rte_kernels_hip.h + librte_kernels_hip.so export them under RTE-RRTMGP's own bind(C)
 * names and by-reference argument lists, so that they can stand in for RTE's kernel objects at link time.  These
 * are compatibility kernels (one thread per column and g-point); rte_lw / rte_sw callers get the fused solvers.
 *   inc_flux (LW) / inc_flux_dif (SW): diffuse flux incident at the top, (ncol,ngpt) or NULL
 *   flux_dir_top (SW): direct flux at the top of the domain, inc_flux*mu0, (ncol,ngpt)
 * --------------------------------------------------------------------------------------- */
int ecckd_lw_solver_noscat_gpt(int device, int ncol, int nlay, int ngpt, int top_at_1, int nmus, const double *Ds,
                               const double *weights, const double *tau, const double *lay_source,
                               const double *lev_source_inc, const double *lev_source_dec,
                               const double *sfc_emis, const double *sfc_src, const double *inc_flux,
                               double *gpt_flux_up, double *gpt_flux_dn, int memspace, void *stream);
int ecckd_sw_solver_2stream_gpt(int device, int ncol, int nlay, int ngpt, int top_at_1, const double *tau,
                                const double *ssa, const double *g, const double *mu0,
                                const double *flux_dir_top, const double *inc_flux_dif,
                                const double *sfc_alb_dir, const double *sfc_alb_dif, double *gpt_flux_up,
                                double *gpt_flux_dn, double *gpt_flux_dir, int memspace, void *stream);
int ecckd_sum_broadband(int device, int ncol, int nlev, int ngpt, const double *spectral_flux,
                        double *broadband_flux, int memspace, void *stream);

/* ---------------------------------------------------------------------------------------
 * Fused longwave path (no counterpart call in the reference, whose block loop calls
 * gas_optics and rte_lw back to back with nothing reading tau or the sources in between:
 * ecckd_rfmip_lw.F90:120-135).  The three source arrays are pure functions of tlay / tlev / tsfc and the model's
 * Planck table (src/gas_optics_ecckd.f90:407-424), so a host that only needs fluxes does not have to move them:
 * gas optics writes tau only (8 B/cell) and the solver recomputes the sources while it reads tau (8 B/cell) --
 * 16 instead of 64 B per (column, layer, g-point) between the two kernels.  Same arithmetic per cell (sources bit
 * identical to ecckd_gas_optics_lw); fast arithmetic mode, fp64.  60 layers take the fused kernels; any other layer
 * count is served by the general route (Planck kernel into library scratch, then the register-resident solver): the same
 * results at the API path's rate.  This path is bound by fp64 issue, not by HBM, and bench.py reports it apart from the
 * API-boundary roofline ("fused_lw").
 *   ecckd_gas_optics_lw_tau  gas_optical_depth (:323-376) alone: tau(ncol,nlay,ngpt)              ECCKD_DEVICE
 *   ecckd_rte_lw_fused       rte_lw on tau + temperatures; sfc_emis(nband,ncol), inc_flux(ncol,ngpt) or NULL  ECCKD_DEVICE
 *   ecckd_lw_fluxes          both, tau in library-owned stream-ordered scratch      ECCKD_DEVICE or ECCKD_HOST
 * --------------------------------------------------------------------------------------- */
int ecckd_gas_optics_lw_tau(const ecckd_model_t *model, int ncol, int nlay, const double *plev, const double *tlay,
                            int ngas, const char *gas_names, const double *const *vmr,
                            const long long *vmr_col_stride, const long long *vmr_lay_stride,
                            const double *vmr_scalar, double *tau, int memspace, void *stream);
int ecckd_rte_lw_fused(const ecckd_model_t *model, int ncol, int nlay, int top_at_1, int n_gauss_angles,
                       const double *tau, const double *tlay, const double *tlev, const double *tsfc,
                       const double *sfc_emis, const double *inc_flux, double *flux_up, double *flux_dn,
                       int memspace, void *stream);
int ecckd_lw_fluxes(const ecckd_model_t *model, int ncol, int nlay, const double *plev, const double *tlay,
                    const double *tsfc, const double *tlev, int ngas, const char *gas_names,
                    const double *const *vmr, const long long *vmr_col_stride, const long long *vmr_lay_stride,
                    const double *vmr_scalar, int top_at_1, int n_gauss_angles, const double *sfc_emis,
                    const double *inc_flux, double *flux_up, double *flux_dn, int memspace, void *stream);

/* Two-stream + adding SW.  mu0(ncol), toa_flux(ncol,ngpt), sfc_alb_dir/dif(nband,ncol).
 * flux_dn includes the direct beam; flux_dir (ncol,nlay+1) may be NULL. */
int ecckd_rte_sw(int device, int ncol, int nlay, int ngpt, int top_at_1, const double *tau,
                 const double *ssa, const double *g, const double *mu0, const double *toa_flux,
                 int nband, const int *band2gpt, const double *sfc_alb_dir,
                 const double *sfc_alb_dif, double *flux_up, double *flux_dn, double *flux_dir,
                 int memspace, void *stream);

/* Single-precision flavours of the shortwave pair and of ecckd_rte_lw_inc_flux (a host built with RTE-RRTMGP's
 * RTE_USE_SP: wp = real32, src/gas_optics_ecckd.f90:6).  Every data array is float, arithmetic is float, the g-point
 * sums are accumulated in double and rounded once.  ecckd_gas_optics_sw_f32: one-pass gas lists (all ecCKD files);
 * ecckd_rte_sw_f32: the layer-systolic solver (at most 60 layers). */
int ecckd_gas_optics_sw_f32(const ecckd_model_t *model, int ncol, int nlay, const float *plev, const float *tlay,
                            int ngas, const char *gas_names, const float *const *vmr,
                            const long long *vmr_col_stride, const long long *vmr_lay_stride,
                            const double *vmr_scalar, float *tau, float *ssa, float *g, float *toa_src,
                            int memspace, void *stream);
int ecckd_rte_sw_f32(int device, int ncol, int nlay, int ngpt, int top_at_1, const float *tau, const float *ssa,
                     const float *g, const float *mu0, const float *toa_flux, int nband, const int *band2gpt,
                     const float *sfc_alb_dir, const float *sfc_alb_dif, float *flux_up, float *flux_dn,
                     float *flux_dir, int memspace, void *stream);
int ecckd_rte_lw_inc_flux_f32(int device, int ncol, int nlay, int ngpt, int top_at_1, int n_gauss_angles,
                              const float *tau, const float *lay_source, const float *lev_source_inc,
                              const float *lev_source_dec, const float *sfc_source, int nband,
                              const int *band2gpt, const float *sfc_emis, const float *inc_flux, float *flux_up,
                              float *flux_dn, int memspace, void *stream);

/* ---------------------------------------------------------------------------------------
 * Fused shortwave path (SURVEY.md section 8(f) rank 4 for the shortwave; no counterpart call in the reference, whose
 * block loop calls gas_optics and rte_sw back to back: ecckd_rfmip_sw.F90:118-154).  gas_optics_ext derives ssa, g and
 * toa_src from plev and two small tables (src/gas_optics_ecckd.f90:455-472: ssa = tau_rayleigh/tau with
 * tau_rayleigh = (plev(l+1)-plev(l))*global_weight*rayleigh_molar_scattering_coeff(g) (:313-317), g = 0,
 * toa_src = solar_irradiance(g)), so a host that only needs fluxes moves the TOTAL optical depth alone: gas optics
 * writes tau (8 B/cell) and the solver evaluates those same expressions while it reads it (8 B/cell) -- 16 instead of
 * 48 B per (column, layer, g-point).  Same arithmetic per cell: fluxes bit-identical to ecckd_gas_optics_sw +
 * ecckd_rte_sw.  tau lives in library-owned stream-ordered scratch.  Fast arithmetic mode, layer-systolic solver
 * (at most 60 layers); ECCKD_DEVICE or ECCKD_HOST.
 *   toa_scale(ncol) or NULL: toa(i,g) = solar_irradiance(g)*toa_scale(i) -- the drivers' rescaling to the file's total
 *   solar irradiance (ecckd_rfmip_sw.F90:126-133); mu0(ncol), sfc_alb_dir/dif(nband,ncol) as ecckd_rte_sw;
 *   flux_dir may be NULL.
 * bench.py reports it apart from the API-boundary roofline ("fused_sw").
 * --------------------------------------------------------------------------------------- */
int ecckd_sw_fluxes(const ecckd_model_t *model, int ncol, int nlay, const double *plev, const double *tlay, int ngas,
                    const char *gas_names, const double *const *vmr, const long long *vmr_col_stride,
                    const long long *vmr_lay_stride, const double *vmr_scalar, int top_at_1, const double *mu0,
                    const double *toa_scale, const double *sfc_alb_dir, const double *sfc_alb_dif, double *flux_up,
                    double *flux_dn, double *flux_dir, int memspace, void *stream);
int ecckd_sw_fluxes_f32(const ecckd_model_t *model, int ncol, int nlay, const float *plev, const float *tlay, int ngas,
                        const char *gas_names, const float *const *vmr, const long long *vmr_col_stride,
                        const long long *vmr_lay_stride, const double *vmr_scalar, int top_at_1, const float *mu0,
                        const float *toa_scale, const float *sfc_alb_dir, const float *sfc_alb_dif, float *flux_up,
                        float *flux_dn, float *flux_dir, int memspace, void *stream);

/* Spectral (per-band) fluxes: what RTE-RRTMGP callers get by passing a ty_fluxes_byband to rte_lw /
 * rte_sw instead of the ty_fluxes_broadband the reference drivers use (ecckd_rfmip_lw.F90:108-109).
 * bnd_flux_*(ncol,nlay+1,nband) = sum over the g-points of each band (one solver pass per band over its
 * contiguous g-points); flux_up / flux_dn / flux_dir (ncol,nlay+1) are optional (NULL) and hold the sum over
 * bands.  Other arguments as ecckd_rte_lw / ecckd_rte_sw. */
int ecckd_rte_lw_byband(int device, int ncol, int nlay, int ngpt, int top_at_1, int n_gauss_angles,
                        const double *tau, const double *lay_source, const double *lev_source_inc,
                        const double *lev_source_dec, const double *sfc_source, int nband,
                        const int *band2gpt, const double *sfc_emis, double *bnd_flux_up,
                        double *bnd_flux_dn, double *flux_up, double *flux_dn, int memspace, void *stream);
int ecckd_rte_sw_byband(int device, int ncol, int nlay, int ngpt, int top_at_1, const double *tau,
                        const double *ssa, const double *g, const double *mu0, const double *toa_flux,
                        int nband, const int *band2gpt, const double *sfc_alb_dir,
                        const double *sfc_alb_dif, double *bnd_flux_up, double *bnd_flux_dn,
                        double *bnd_flux_dir, double *flux_up, double *flux_dn, double *flux_dir,
                        int memspace, void *stream);
/* The same in single precision (float arrays throughout; the band sums are taken in float). */
int ecckd_rte_lw_byband_f32(int device, int ncol, int nlay, int ngpt, int top_at_1, int n_gauss_angles, const float *tau,
                            const float *lay_source, const float *lev_source_inc, const float *lev_source_dec,
                            const float *sfc_source, int nband, const int *band2gpt, const float *sfc_emis,
                            float *bnd_flux_up, float *bnd_flux_dn, float *flux_up, float *flux_dn, int memspace, void *stream);
int ecckd_rte_sw_byband_f32(int device, int ncol, int nlay, int ngpt, int top_at_1, const float *tau, const float *ssa,
                            const float *g, const float *mu0, const float *toa_flux, int nband, const int *band2gpt,
                            const float *sfc_alb_dir, const float *sfc_alb_dif, float *bnd_flux_up, float *bnd_flux_dn,
                            float *bnd_flux_dir, float *flux_up, float *flux_dn, float *flux_dir, int memspace, void *stream);

/* ---------------------------------------------------------------------------------------
 * Device memory for host languages without a HIP binding of their own (the Fortran shim's device-resident
 * twins of ty_optical_props / ty_source_func_lw own their buffers through these).  to_device: 1 = host to
 * device, 0 = device to host; synchronous.
 * --------------------------------------------------------------------------------------- */
int ecckd_device_malloc(int device, size_t bytes, void **ptr);
int ecckd_device_free(int device, void *ptr);
int ecckd_device_memcpy(int device, void *dst, const void *src, size_t bytes, int to_device);

/* ---------------------------------------------------------------------------------------
 * calculate_planck_function x 3 alone (src/gas_optics_ecckd.f90:245-289 as gas_optics_int applies it, :407-424):
 * lay_source(ncol,nlay,ngpt) from tlay, lev_source_inc / lev_source_dec from tlev(ncol,nlay+1) (tlev may be NULL: the
 * level sources are then left alone), sfc_source(ncol,ngpt) from tsfc.  Device arrays (ECCKD_DEVICE), fp64,
 * asynchronous on `stream`.  The same kernel ecckd_gas_optics_lw runs beside its optical-depth kernel.
 * --------------------------------------------------------------------------------------- */
int ecckd_planck_sources(const ecckd_model_t *model, int ncol, int nlay, const double *tlay, const double *tlev,
                         const double *tsfc, double *lay_source, double *lev_source_inc, double *lev_source_dec,
                         double *sfc_source, int memspace, void *stream);

/* ---------------------------------------------------------------------------------------
 * Launch plan of a gas_optics call (no counterpart in the reference; works on host-only models,
 * device -1, and launches nothing): how the library would run gas_optics for this model, gas list
 * (names only; LW if the model has a Planck table, else SW), size, precision and the current
 * arithmetic mode.
 *   plan[0] kernel passes over the gas list (<= 10 gases and one look_up_table gas per pass)
 *   plan[1] 1: first pass is the fused kernel, 0: reference-order tau kernel
 *   plan[2] 1: the Planck sources ride in that pass (LW), 0: separate Planck kernel / SW
 *   plan[3] pressure rows of the LDS slab      plan[4] Planck-table rows staged in LDS (ntp = whole table)
 *   plan[5] column chunks (grid.x)             plan[6] LDS bytes per block
 *   plan[7] g-points per chunk of the item pipeline
 * --------------------------------------------------------------------------------------- */
int ecckd_gas_optics_plan(const ecckd_model_t *model, int ncol, int nlay, int single_precision,
                          int ngas, const char *gas_names, int *plan);
/* The same with the shape of gas_desc: vmr_is_scalar[j] != 0 says gas j would be passed as one number (a null
 * pointer: every gas is an array).  Writes min(nplan, ECCKD_PLAN_LEN) entries; beyond the eight above:
 *   plan[8] bilinear slots of the kernel instantiation (2, 5, 7 or 10)
 *   plan[9] gases folded into the merged slot ("gas_merge_scalars" below; 0: none) */
#define ECCKD_PLAN_LEN 10
int ecckd_gas_optics_plan_ex(const ecckd_model_t *model, int ncol, int nlay, int single_precision,
                             int ngas, const char *gas_names, const int *vmr_is_scalar, int nplan, int *plan);

/* ---------------------------------------------------------------------------------------
 * Arithmetic mode of gas_optics (process-wide, atomic; read once per call).
 *   0 (default) fast: one fused kernel per call; the interpolation weights of a cell are
 *               multiplied out once and each coefficient costs one FMA.  Same formula as
 *               src/gas_optics_ecckd.f90:167-221, re-associated: tau differs from mode 1 by a
 *               few ulp.  Planck sources are identical in both modes.
 *   1           reference order: every product and sum in the order the Fortran expressions
 *               spell, no FMA contraction, gases accumulated in gas_desc order (:370).  tau then
 *               differs from an IEEE evaluation of the reference only through the device log().
 *   The shortwave solver follows the mode too.  Mode 1: IEEE division, the device library's sqrt and exp, the adding
 *   recurrences operation by operation in the restated order.  Mode 0 (fp64): reciprocals as v_rcp_f64 + one third-order
 *   step, sqrt and exp without the library's range handling (~1 ulp), multiply-add pairs of the recurrences fused, the
 *   pair a wave hands upwards from a division-free form of the adding recurrence, the flux recurrence pre-multiplied --
 *   fluxes within 1e-11 W m-2 of mode 1 (tests/test_gpu_round3.py: optical depths from 1e-12 to inf, NaN columns).
 *   Mode 0 raises "sw_k_floor" to the smallest normal double if it is set below.
 *   The longwave solver is the same in both modes; its division and exp are written out (csrc/lw_layer.hpp): the division
 *   gives the bits of `/` for optical depths x secant below 1e290 and ~1e-290 instead of less beyond (up to inf).
 * --------------------------------------------------------------------------------------- */
int ecckd_set_arithmetic(int mode);
int ecckd_get_arithmetic(void);

/* ---------------------------------------------------------------------------------------
 * Version switches of the solvers (process-wide; read once per call).  RTE-RRTMGP is an un-pinned dependency
 * of the reference (.github/workflows/continuous-integration.yml:98-102 checks out its default branch), and a few
 * details of rte_lw / rte_sw changed between releases.  The defaults are the v1.5-era forms; a host linked
 * against a later RTE-RRTMGP selects the matching forms here.  bench.py prints the active values.  PARITY UNPINNED: both
 * forms of every switch restate RTE-RRTMGP from the published sources -- the library is not in the reference tree and
 * the reference holds no fixture for it; the tests check the HIP solvers against this repository's
 * CPU restatement with the same switch, nothing more.
 *   "lw_tau_thresh"          lw_source_noscat uses the series below this tau*D (default sqrt(epsilon(1._wp));
 *                            later releases: sqrt(sqrt(epsilon)));  <= 0 restores the default
 *   "lw_series_terms"        2: tau*(0.5 - tau/3) (default);  3: tau*(0.5 + tau*(-1/3 + tau/8))
 *   "lw_inc_flux_isotropic"  0: I_dn(top) = inc_flux/(2 pi w_k) per angle (default);
 *                            1: inc_flux/pi (flux_dn(top) == inc_flux with any number of angles)
 *   "sw_k_floor"             k = sqrt(max((gamma1-gamma2)(gamma1+gamma2), sw_k_floor)), default 1e-12
 *   "sw_dir_clamp"           1: Rdir = max(0,min(Rdir,1-Tnoscat)), Tdir = max(0,min(Tdir,1-Tnoscat-Rdir))
 *                            (v1.6+); 0: no clamp (default)
 * Implementation choices through the same call (results agree to ~1e-16 relative; bench.py prints them too):
 *   "lw_solver"              fp64, 60 layers: 0 register-resident solver (one wave per SIMD), 1 layer-split solver
 *                            (waves of a block share a tile and take 10-15 layers each; three waves per SIMD)
 *   "lw_split_seg"           layers per wave of the layer-split solver: 10 (default), 12 or 15
 *   "lw_tail_split"          register-resident solver: 1 (default) the tiles beyond the last full round of waves (one
 *                            wave per SIMD) are solved one g-point pair per wave and summed in g-point order by a
 *                            second small kernel -- bit-identical fluxes, no idle SIMDs in the last round (1e5 columns:
 *                            3 125 tiles on 1 024 SIMDs).  Needs up to 64 MiB of stream scratch, taken only when it
 *                            can be had without an error (not inside a graph capture that has not seen the call
 *                            before, not beyond a caller-owned buffer): 0 switches it off.  A host that hands its own
 *                            block over (ecckd_set_stream_scratch) sizes it with ecckd_rte_lw_tail_scratch_bytes /
 *                            ecckd_rte_sw_tail_scratch_bytes (+ ecckd_rte_*_scratch_bytes where that is not 0)
 *   "sw_tail_split"          the same for rte_sw: 1 (default), 0 off; bit-identical fluxes.  Layer-systolic solver: the
 *                            tiles beyond the last full round of blocks (one block per CU; every tile of a call of
 *                            less than 16 384 columns) one g-point per block, up to 128 MiB of partial sums.  Two-pass
 *                            solver: calls that do not fill one round of its persistent grid, one g-point group per wave
 *   "sw_solver"              0 (default): layer-systolic solver (kernels_rte_sw_sys.hip: the two-stream coefficients are
 *                            computed once and stay in registers between the sweeps, the layers of a column are spread
 *                            over the waves of a block; at most 60 layers, no scratch ring); 1: two-pass kernel (any
 *                            layer count; reads tau / ssa / g twice, scratch ring).  The same arithmetic per (column,
 *                            g-point); the g-point sums are ordered differently (sequential / shuffle tree)
 *   "gas_merge_scalars"      fast arithmetic mode, fp64: 1 (default) the gases of gas_desc given as ONE number for the call
 *                            (vmr pointer NULL + vmr_scalar; get_vmr broadcasts them, src/gas_optics_ecckd.f90:351) and
 *                            the none_ composite share one table sum_k m_k*coefficient_k, m_k = vmr | vmr - reference | 1,
 *                            built on the fly: tau = ... + simple_weight * bilinear(merged table) instead of one
 *                            interpolation per gas (same real-number formula; the per-gas clamp :234-238 is kept
 *                            because only gases with m_k >= 0 and tables without negative entries are merged);
 *                            0: every gas interpolated on its own
 * --------------------------------------------------------------------------------------- */
int ecckd_set_solver_option(const char *name, double value);
int ecckd_get_solver_option(const char *name, double *value);

/* ---------------------------------------------------------------------------------------
 * Solver scratch (no counterpart in the reference).  ecckd_rte_sw (always) and ecckd_rte_lw (more than 96
 * layers) keep per-wave rings in global memory.  By default the library owns one block per (device, stream),
 * allocated at the first call that needs it and reused by every later call on that stream without any
 * synchronisation; inside a stream capture nothing is allocated (a call that would have to fails with a
 * message).  ecckd_set_stream_scratch hands a caller-owned device buffer over for the calls on `stream`
 * (buffer == NULL, bytes == 0 takes it back); ecckd_*_scratch_bytes say how much a shape needs (0: none);
 * ecckd_release_scratch synchronises the device and frees every library-owned block (caller-owned buffers stay
 * registered).  A block that was handed to a captured call belongs to that graph: the next eager call on the stream
 * takes a fresh one, so a graph may be replayed on any stream; graphs captured back to back on one stream share a
 * block (replay those on one stream), and a graph must be destroyed before ecckd_release_scratch.
 * --------------------------------------------------------------------------------------- */
size_t ecckd_rte_lw_scratch_bytes(int ncol, int nlay, int ngpt);
size_t ecckd_rte_sw_scratch_bytes(int ncol, int nlay, int ngpt);   /* (two-pass solver: "sw_solver" = 1, or more than 60 layers) */
/* What the tail splits ("lw_tail_split", "sw_tail_split") of a call of this shape would take on top, with the
 * solver options as they are now (0: the call would not split). */
size_t ecckd_rte_lw_tail_scratch_bytes(int device, int ncol, int nlay, int ngpt, int n_gauss_angles, int single_precision);
size_t ecckd_rte_sw_tail_scratch_bytes(int device, int ncol, int nlay, int ngpt);
int ecckd_set_stream_scratch(int device, void *stream, void *buffer, size_t bytes);
int ecckd_release_scratch(int device);

/* ---------------------------------------------------------------------------------------
 * Measurement hooks (no counterpart in the reference, which has no timers: SURVEY.md §5).
 * While enabled, every kernel launch is bracketed by HIP events recorded on the stream the
 * kernel is launched on.  ecckd_prof_report waits for the recorded events, sums the elapsed
 * time per kernel name ("gas_lw_fused", "tau", "planck", "rte_lw", "rte_sw"), clears the records and returns
 * the number of distinct kernels; names is max_kernels records of ECCKD_NAME_LEN bytes.
 * --------------------------------------------------------------------------------------- */
int ecckd_prof_enable(int on);
int ecckd_prof_report(int max_kernels, char *names, double *total_ms, long long *launches);

/* Library / build identification ("gfx950", compile flags); never NULL. */
const char *ecckd_build_info(void);

#ifdef __cplusplus
}
#endif
#endif
