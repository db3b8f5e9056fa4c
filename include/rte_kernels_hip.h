/*
 * rte_kernels_hip.h -- librte_kernels_hip.so: RTE-RRTMGP's solver KERNELS under their own bind(C) names, implemented
 * on the MI355X by librte_ecckd_hip.so (ecckd_lw_solver_noscat_gpt, ecckd_sw_solver_2stream_gpt, ecckd_sum_broadband).
 *
 * What it replaces [RTE-ext -- the library is an un-vendored dependency of the reference (Makefile:19,33;
 * .github/workflows/continuous-integration.yml:98-112), so these signatures are restated from the public v1.5-era
 * sources mo_rte_solver_kernels.F90 / mo_fluxes_broadband_kernels.F90 and cannot be checked against a file in
 * /root/reference]:
 *
 *   subroutine lw_solver_noscat_GaussQuad(ncol, nlay, ngpt, top_at_1, nmus, Ds, weights, tau, lay_source,
 *                lev_source_inc, lev_source_dec, sfc_emis, sfc_src, flux_up, flux_dn) bind(C, name="lw_solver_noscat_GaussQuad")
 *   subroutine sw_solver_2stream(ncol, nlay, ngpt, top_at_1, tau, ssa, g, mu0, sfc_alb_dir, sfc_alb_dif,
 *                flux_up, flux_dn, flux_dir) bind(C, name="sw_solver_2stream")
 *   subroutine sum_broadband(ncol, nlev, ngpt, spectral_flux, broadband_flux) bind(C, name="sum_broadband")
 *   subroutine net_broadband(ncol, nlev, flux_dn, flux_up, broadband_flux_net) bind(C, name="net_broadband_precalc")
 *
 * Fortran bind(C) without VALUE passes every argument by reference; `top_at_1` is logical(wl), a C _Bool when RTE-RRTMGP
 * is built with -DRTE_USE_CBOOL (the flag of the reference's CI, continuous-integration.yml:15).  Arrays are host
 * arrays, column-major: tau(ncol,nlay,ngpt), sfc_emis / sfc_src / sfc_alb_*(ncol,ngpt), mu0(ncol),
 * flux_*(ncol,nlay+1,ngpt).  On entry flux_dn(:,top,:) holds the diffuse incident flux and (shortwave)
 * flux_dir(:,top,:) the direct one, top = 1 or nlay+1 by top_at_1 -- RTE's apply_BC convention.
 * The library runs on HIP device 0 unless ECCKD_RTE_KERNELS_DEVICE is set; a failure prints the message of
 * ecckd_last_error() to stderr and stops the process (the kernels have no status argument), like RTE's own
 * stop_on_err convention.
 */
#ifndef RTE_KERNELS_HIP_H
#define RTE_KERNELS_HIP_H
#include <stdbool.h>
#ifdef __cplusplus
extern "C" {
#endif

void lw_solver_noscat_GaussQuad(const int *ncol, const int *nlay, const int *ngpt, const bool *top_at_1, const int *nmus,
                                const double *Ds, const double *weights, const double *tau, const double *lay_source,
                                const double *lev_source_inc, const double *lev_source_dec, const double *sfc_emis,
                                const double *sfc_src, double *flux_up, double *flux_dn);
void sw_solver_2stream(const int *ncol, const int *nlay, const int *ngpt, const bool *top_at_1, const double *tau,
                       const double *ssa, const double *g, const double *mu0, const double *sfc_alb_dir,
                       const double *sfc_alb_dif, double *flux_up, double *flux_dn, double *flux_dir);
void sum_broadband(const int *ncol, const int *nlev, const int *ngpt, const double *spectral_flux, double *broadband_flux);
void net_broadband_precalc(const int *ncol, const int *nlev, const double *flux_dn, const double *flux_up,
                           double *broadband_flux_net);

#ifdef __cplusplus
}
#endif
#endif
