/*
 * ecckd_oracle.h -- CPU restatement (plain C, fp64) of the rte-ecckd hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped path (librte_ecckd_hip.so)
 * never links, loads or calls anything in this directory.
 *
 * What it restates (citations are into /root/reference, read as text only):
 *   gas optics  src/gas_optics_ecckd.f90:51-57,64-241,245-289,293-319,323-376,381-473
 *   rte_lw      RTE-RRTMGP v1.5-era mo_rte_solver_kernels (lw_solver_noscat_GaussQuad,
 *               lw_source_noscat, lw_transport_noscat, sum_broadband).  That library is an
 *               un-vendored, un-pinned dependency (reference Makefile:19,33;
 *               .github/workflows/continuous-integration.yml:98-112) and is absent from
 *               /root/reference; the published algorithm is restated and anchored on the
 *               reference call sites example/rfmip-rad-irf/ecckd_rfmip_lw.F90:130-135 and
 *               ecckd_rfmip_sw.F90:148-154.
 *   rte_sw      same library: sw_two_stream, sw_source_2str, adding.
 *
 * Pinning status:
 *   gas optics  pinned by the six known-answer values SURVEY.md §8(a) records from a run
 *               of the unmodified reference module (tests/test_oracle_kat.py).
 *   solvers     PARITY UNPINNED: the reference holds no fixtures for fluxes and the solver
 *               source is not in /root/reference; only analytic known-answer tests apply.
 */
#ifndef ECCKD_ORACLE_H
#define ECCKD_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MAX_GASES 16   /* src/gas_optics_ecckd.f90:24-25 */

enum { ORACLE_NONE = 0, ORACLE_LINEAR = 1, ORACLE_LOOK_UP_TABLE = 2, ORACLE_RELATIVE_LINEAR = 3 };

/* AbsorptionTable, src/gas_optics_ecckd.f90:13-19.  coefficient is Fortran (ng,np,nt,nv),
 * i.e. g fastest. */
typedef struct {
  char name[32];
  const double *coefficient;
  int nv;                         /* 1 unless look_up_table */
  int composite_only;
  int concentration_dependence_code;
  const double *mole_fraction;    /* (nv) for look_up_table, else NULL */
  double reference_mole_fraction;
} oracle_gas_t;

/* ty_gas_optics_ecckd, src/gas_optics_ecckd.f90:23-48 */
typedef struct {
  int ng, np, nt, ntp;
  int num_gases;
  const double *log_pressure;                     /* (np) */
  const double *temperature;                      /* (np,nt) */
  const double *planck_function;                  /* (ng,ntp) or NULL */
  const double *temperature_planck;               /* (ntp)    or NULL */
  const double *solar_irradiance;                 /* (ng)     or NULL */
  const double *rayleigh_molar_scattering_coeff;  /* (ng)     or NULL */
  oracle_gas_t gas[ORACLE_MAX_GASES];
} oracle_model_t;

/* ty_gas_concs as the hot path sees it: names in gas_desc order and a vmr per gas that
 * get_vmr broadcasts to (ncol,nlay).  vmr(i,j) = ptr[i*col_stride + j*lay_stride]. */
typedef struct {
  int ngas;
  const char *const *names;
  const double *const *vmr;
  const long *col_stride;
  const long *lay_stride;
} oracle_gas_concs_t;

/* src/gas_optics_ecckd.f90:64-241.  od is (ncol,nlay,ng). */
void oracle_calculate_optical_depth(const oracle_model_t *m, int gas, int ncol, int nlay,
                                    const double *plev, const double *tlay,
                                    const double *layer_vmr, double *od);
/* src/gas_optics_ecckd.f90:245-289.  planck is (ncol,nlev,ng). */
void oracle_calculate_planck_function(const oracle_model_t *m, int ncol, int nlev,
                                      const double *temperature, double *planck);
/* src/gas_optics_ecckd.f90:293-319 */
void oracle_calculate_rayleigh_optical_depth(const oracle_model_t *m, int ncol, int nlay,
                                             const double *plev, double *od);
/* src/gas_optics_ecckd.f90:323-376.  Returns 0; errmsg (128 bytes) empty on success. */
int oracle_gas_optical_depth(const oracle_model_t *m, int ncol, int nlay, const double *plev,
                             const double *tlay, const oracle_gas_concs_t *gc, double *tau,
                             char *errmsg);
/* src/gas_optics_ecckd.f90:381-426.  tlev may be NULL (-> "tlev is required for ecckd"
 * after tau, lay_source and sfc_source have been written, as the reference does). */
int oracle_gas_optics_int(const oracle_model_t *m, int ncol, int nlay, const double *plev,
                          const double *tlay, const double *tsfc, const oracle_gas_concs_t *gc,
                          const double *tlev, double *tau, double *lay_source,
                          double *lev_source_inc, double *lev_source_dec, double *sfc_source,
                          char *errmsg);
/* src/gas_optics_ecckd.f90:431-473.  ssa/g may be NULL (-> "shortwave must use
 * ty_optical_props_2str"). */
int oracle_gas_optics_ext(const oracle_model_t *m, int ncol, int nlay, const double *plev,
                          const double *tlay, const oracle_gas_concs_t *gc, double *tau,
                          double *ssa, double *g, double *toa_src, char *errmsg);

/* RTE-RRTMGP rte_lw, no scattering, Gauss quadrature with nmus in 1..4; sfc_emis_gpt is
 * (ncol,ng) (already expanded from bands); fluxes are (ncol,nlay+1) broadband. */
void oracle_rte_lw(int ncol, int nlay, int ng, int top_at_1, int nmus, const double *tau,
                   const double *lay_source, const double *lev_source_inc,
                   const double *lev_source_dec, const double *sfc_emis_gpt,
                   const double *sfc_source, double *flux_up, double *flux_dn);
/* Version-sensitive details of the (un-pinned) solvers as switches; defaults = the v1.5-era forms:
 *   lw_tau_thresh          optical depth below which lw_source_noscat uses the series (sqrt(epsilon))
 *   lw_series_terms        2: tau*(0.5 - tau/3); 3: tau*(0.5 + tau*(-1/3 + tau/8)) (later releases)
 *   lw_inc_flux_isotropic  0: I_dn(top) = inc_flux/(2 pi w_k) per angle; 1: inc_flux/pi
 *   sw_k_floor             lower bound of (gamma1-gamma2)(gamma1+gamma2) under the square root (1e-12)
 *   sw_dir_clamp           1: Rdir = max(0,min(Rdir,1-Tnoscat)), Tdir = max(0,min(Tdir,1-Tnoscat-Rdir)) (v1.6+) */
typedef struct {
  double lw_tau_thresh;
  int lw_series_terms;
  int lw_inc_flux_isotropic;
  double sw_k_floor;
  int sw_dir_clamp;
} oracle_solver_options_t;
void oracle_default_solver_options(oracle_solver_options_t *o);
/* oracle_rte_lw + incident diffuse flux inc_flux(ncol,ng) at the top (NULL: none) + switches */
void oracle_rte_lw_opt(int ncol, int nlay, int ng, int top_at_1, int nmus, const double *tau,
                       const double *lay_source, const double *lev_source_inc,
                       const double *lev_source_dec, const double *sfc_emis_gpt,
                       const double *sfc_source, const double *inc_flux,
                       const oracle_solver_options_t *opt, double *flux_up, double *flux_dn);
void oracle_rte_sw_opt(int ncol, int nlay, int ng, int top_at_1, const double *tau,
                       const double *ssa, const double *g, const double *mu0, const double *toa,
                       const double *sfc_alb_dir_gpt, const double *sfc_alb_dif_gpt,
                       const oracle_solver_options_t *opt, double *flux_up, double *flux_dn,
                       double *flux_dir);
/* The same with the spectral fluxes (ncol,nlay+1,ng) that RTE-RRTMGP's kernels lw_solver_noscat_GaussQuad /
 * sw_solver_2stream return before sum_broadband (any output may be NULL), and the shortwave's diffuse incident
 * flux inc_flux_dif(ncol,ng) (NULL: none). */
void oracle_rte_lw_gpt(int ncol, int nlay, int ng, int top_at_1, int nmus, const double *tau,
                       const double *lay_source, const double *lev_source_inc,
                       const double *lev_source_dec, const double *sfc_emis_gpt,
                       const double *sfc_source, const double *inc_flux,
                       const oracle_solver_options_t *opt, double *flux_up, double *flux_dn,
                       double *gpt_flux_up, double *gpt_flux_dn);
void oracle_rte_sw_gpt(int ncol, int nlay, int ng, int top_at_1, const double *tau,
                       const double *ssa, const double *g, const double *mu0, const double *toa,
                       const double *inc_flux_dif, const double *sfc_alb_dir_gpt,
                       const double *sfc_alb_dif_gpt, const oracle_solver_options_t *opt,
                       double *flux_up, double *flux_dn, double *flux_dir, double *gpt_flux_up,
                       double *gpt_flux_dn, double *gpt_flux_dir);

/* RTE-RRTMGP rte_sw, two-stream + adding; albedos are (ncol,ng). flux_dn includes direct. */
void oracle_rte_sw(int ncol, int nlay, int ng, int top_at_1, const double *tau,
                   const double *ssa, const double *g, const double *mu0, const double *toa,
                   const double *sfc_alb_dir_gpt, const double *sfc_alb_dif_gpt,
                   double *flux_up, double *flux_dn, double *flux_dir);

/* bench.py cpu_baseline leg: the LW pair gas_optics_int + rte_lw run block by block over
 * columns the way example/rfmip-rad-irf/ecckd_rfmip_lw.F90:107-136 does, with `nthreads`
 * OpenMP threads each taking whole blocks.  Returns 0 on success. */
int oracle_lw_pipeline(const oracle_model_t *m, int ncol, int nlay, int block, int nthreads,
                       const double *plev, const double *tlay, const double *tlev,
                       const double *tsfc, const oracle_gas_concs_t *gc, const double *sfc_emis,
                       int nmus, double *flux_up, double *flux_dn);

#ifdef __cplusplus
}
#endif
#endif
