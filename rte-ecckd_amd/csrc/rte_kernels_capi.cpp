// rte_kernels_capi.cpp -- librte_kernels_hip.so: RTE-RRTMGP's kernel-level bind(C) entry points (include/rte_kernels_hip.h)
// over the C ABI of librte_ecckd_hip.so.  Marshalling only: boundary conditions are lifted out of the flux arrays the
// way RTE's apply_BC put them there, everything else is passed through.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/ecckd_hip.h"
#include "../../include/rte_kernels_hip.h"

namespace {
int device() {
  const char *e = std::getenv("ECCKD_RTE_KERNELS_DEVICE");
  return e ? std::atoi(e) : 0;
}
void stop_on_err(int rc, const char *who) {
  if (!rc) return;
  std::fprintf(stderr, "%s: %s\n", who, ecckd_last_error());
  std::exit(1);
}
// (ncol,ngpt) plane of flux(ncol,nlev,ngpt) at level `lev`
std::vector<double> level_plane(const double *flux, int ncol, int nlev, int ngpt, int lev) {
  std::vector<double> out((size_t)ncol * ngpt);
  for (int g = 0; g < ngpt; ++g)
    for (int i = 0; i < ncol; ++i) out[i + (size_t)ncol * g] = flux[i + (size_t)ncol * (lev + (size_t)nlev * g)];
  return out;
}
}  // namespace

extern "C" {

void lw_solver_noscat_GaussQuad(const int *ncol, const int *nlay, const int *ngpt, const bool *top_at_1, const int *nmus,
                                const double *Ds, const double *weights, const double *tau, const double *lay_source,
                                const double *lev_source_inc, const double *lev_source_dec, const double *sfc_emis,
                                const double *sfc_src, double *flux_up, double *flux_dn) {
  const int top = *top_at_1 ? 0 : *nlay;
  const std::vector<double> inc = level_plane(flux_dn, *ncol, *nlay + 1, *ngpt, top);   // apply_BC left it there
  stop_on_err(ecckd_lw_solver_noscat_gpt(device(), *ncol, *nlay, *ngpt, *top_at_1 ? 1 : 0, *nmus, Ds, weights, tau, lay_source,
                                         lev_source_inc, lev_source_dec, sfc_emis, sfc_src, inc.data(), flux_up, flux_dn,
                                         ECCKD_HOST, nullptr),
              "lw_solver_noscat_GaussQuad");
}

void sw_solver_2stream(const int *ncol, const int *nlay, const int *ngpt, const bool *top_at_1, const double *tau,
                       const double *ssa, const double *g, const double *mu0, const double *sfc_alb_dir,
                       const double *sfc_alb_dif, double *flux_up, double *flux_dn, double *flux_dir) {
  const int top = *top_at_1 ? 0 : *nlay;
  const std::vector<double> dir_top = level_plane(flux_dir, *ncol, *nlay + 1, *ngpt, top);
  const std::vector<double> dif_top = level_plane(flux_dn, *ncol, *nlay + 1, *ngpt, top);
  stop_on_err(ecckd_sw_solver_2stream_gpt(device(), *ncol, *nlay, *ngpt, *top_at_1 ? 1 : 0, tau, ssa, g, mu0, dir_top.data(),
                                          dif_top.data(), sfc_alb_dir, sfc_alb_dif, flux_up, flux_dn, flux_dir, ECCKD_HOST,
                                          nullptr),
              "sw_solver_2stream");
}

void sum_broadband(const int *ncol, const int *nlev, const int *ngpt, const double *spectral_flux, double *broadband_flux) {
  stop_on_err(ecckd_sum_broadband(device(), *ncol, *nlev, *ngpt, spectral_flux, broadband_flux, ECCKD_HOST, nullptr),
              "sum_broadband");
}

void net_broadband_precalc(const int *ncol, const int *nlev, const double *flux_dn, const double *flux_up,
                           double *broadband_flux_net) {
  const size_t n = (size_t)*ncol * *nlev;
  for (size_t i = 0; i < n; ++i) broadband_flux_net[i] = flux_dn[i] - flux_up[i];
}

}  // extern "C"
