/*
 * ecckd_oracle.c -- CPU restatement of the rte-ecckd hot path (see ecckd_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked into, loaded by or called from the product path.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp (oracle/Makefile).  -ffp-contract=off keeps
 * every a*b+c as two roundings, which is what the reference's Fortran expressions give on a
 * baseline x86-64 build (SURVEY.md §8(a) row 0 probe).
 *
 * Array layout everywhere: Fortran column-major, column index fastest.
 */
#include "ecckd_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* src/gas_optics_ecckd.f90:51-53 -- default-real (f32) literals widened to double. */
static const double gravity = (double)9.80665f;
static const double dry_air_molar_mass = (double)28.970f;
static const double pi_f32 = (double)3.14159265359f;

static double global_weight(void) {
  /* :107  1./(gravity*0.001*dry_air_molar_mass), 0.001 also an f32 literal */
  return 1. / (gravity * (double)0.001f * dry_air_molar_mass);
}

/* Temporaries of the routines below (the reference's allocate / deallocate of optical_depth, layer_vmr, buffer ...:
 * src/gas_optics_ecckd.f90:112-115,375,413,425,465).  Inside oracle_lw_pipeline every thread owns one arena that it
 * re-uses for all its column blocks; elsewhere this is malloc / free.  (Round 2 timed the all-cores leg with ten
 * malloc / free pairs per block inside the parallel loop: blocks of 8 columns are 0.1-0.5 MB each, beyond glibc's mmap
 * threshold, so every one of them was an mmap / munmap system call and 256 threads queued on the address-space lock.) */
static _Thread_local char *tl_arena = NULL;
static _Thread_local size_t tl_cap = 0, tl_off = 0;
static void *tmp_alloc(size_t n) {
  if (tl_arena) {
    const size_t a = (n + 63) & ~(size_t)63;
    if (tl_off + a <= tl_cap) {
      void *p = tl_arena + tl_off;
      tl_off += a;
      return p;
    }
  }
  return malloc(n);
}
static void tmp_free(void *p) {
  if (tl_arena && (char *)p >= tl_arena && (char *)p < tl_arena + tl_cap) return;   /* released with the block */
  free(p);
}

static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }

/* ---- src/gas_optics_ecckd.f90:64-241 (logarithmic_interpolation = .false., the only
 * value the reference ever passes, :368-369) ---- */
void oracle_calculate_optical_depth(const oracle_model_t *m, int gas, int ncol, int nlay,
                                    const double *plev, const double *tlay,
                                    const double *layer_vmr, double *od) {
  const oracle_gas_t *a = &m->gas[gas];
  const int ng = m->ng, np = m->np, nt = m->nt;
  const double log_p_0 = m->log_pressure[0];                       /* :104 */
  const double d_log_p = m->log_pressure[1] - m->log_pressure[0];  /* :105 */
  const double dt = m->temperature[np] - m->temperature[0];        /* :106 T(1,2)-T(1,1) */
  const double gw = global_weight();
  const long cstride = (long)ncol * nlay; /* stride between consecutive g in od */
  const double *c = a->coefficient;
#define COEF(ip, it, iv) (c + (long)ng * (((ip)-1) + (long)np * (((it)-1) + (long)nt * ((iv)-1))))

  for (int j = 0; j < nlay; ++j) {
    for (int i = 0; i < ncol; ++i) {
      const double p1 = plev[i + (long)ncol * (j + 1)], p0 = plev[i + (long)ncol * j];
      /* :120-128 */
      double log_pressure = log(0.5 * (p1 + p0));
      double pressure_index = (log_pressure - log_p_0) / d_log_p;
      pressure_index = 1. + dmax(0., dmin(pressure_index, (double)np - 1.0001));
      int ip0 = (int)pressure_index;
      double pw1 = pressure_index - ip0;
      double pw0 = 1. - pw1;
      /* :131-140 */
      double t0 = pw0 * m->temperature[ip0 - 1] + pw1 * m->temperature[ip0];
      double temperature_index = (tlay[i + (long)ncol * j] - t0) / dt;
      temperature_index = 1. + dmax(0., dmin(temperature_index, (double)nt - 1.0001));
      int it0 = (int)temperature_index;
      double tw1 = temperature_index - it0;
      double tw0 = 1. - tw1;
      /* :143-149 */
      double simple_weight = gw * (p1 - p0);
      double vmr = layer_vmr[i + (long)ncol * j];
      double weight;
      if (a->concentration_dependence_code == ORACLE_RELATIVE_LINEAR)
        weight = simple_weight * (vmr - a->reference_mole_fraction);
      else
        weight = simple_weight * vmr;

      double *o = od + i + (long)ncol * j;
      if (a->concentration_dependence_code == ORACLE_LOOK_UP_TABLE) {
        /* :153-163 */
        double log_vmr = log(dmax(vmr, a->mole_fraction[0]));
        double d_log_vmr = log(a->mole_fraction[1] / a->mole_fraction[0]);
        double vmr_index = (log_vmr - log(a->mole_fraction[0])) / d_log_vmr;
        vmr_index = 1. + dmax(0., dmin(vmr_index, (double)a->nv - 1.001));
        int iv0 = (int)vmr_index;
        double vw1 = vmr_index - iv0;
        double vw0 = 1. - vw1;
        const double *c000 = COEF(ip0, it0, iv0), *c100 = COEF(ip0 + 1, it0, iv0);
        const double *c010 = COEF(ip0, it0 + 1, iv0), *c110 = COEF(ip0 + 1, it0 + 1, iv0);
        const double *c001 = COEF(ip0, it0, iv0 + 1), *c101 = COEF(ip0 + 1, it0, iv0 + 1);
        const double *c011 = COEF(ip0, it0 + 1, iv0 + 1),
                     *c111 = COEF(ip0 + 1, it0 + 1, iv0 + 1);
        for (int k = 0; k < ng; ++k) { /* :167-178 */
          o[k * cstride] =
              weight * (vw0 * (tw0 * (pw0 * c000[k] + pw1 * c100[k]) +
                               tw1 * (pw0 * c010[k] + pw1 * c110[k])) +
                        vw1 * (tw0 * (pw0 * c001[k] + pw1 * c101[k]) +
                               tw1 * (pw0 * c011[k] + pw1 * c111[k])));
        }
      } else {
        const double *c00 = COEF(ip0, it0, 1), *c10 = COEF(ip0 + 1, it0, 1);
        const double *c01 = COEF(ip0, it0 + 1, 1), *c11 = COEF(ip0 + 1, it0 + 1, 1);
        /* :198-203 (linear, relative_linear: weight) / :216-221 (none_: simple_weight) */
        const double w = (a->concentration_dependence_code == ORACLE_LINEAR ||
                          a->concentration_dependence_code == ORACLE_RELATIVE_LINEAR)
                             ? weight
                             : simple_weight;
        for (int k = 0; k < ng; ++k) {
          o[k * cstride] = w * (tw0 * (pw0 * c00[k] + pw1 * c10[k]) +
                                tw1 * (pw0 * c01[k] + pw1 * c11[k]));
        }
      }
      /* :234-238 remove negative optical depths */
      for (int k = 0; k < ng; ++k)
        if (o[k * cstride] < 0.) o[k * cstride] = 0.;
    }
  }
#undef COEF
}

/* ---- src/gas_optics_ecckd.f90:245-289 ---- */
void oracle_calculate_planck_function(const oracle_model_t *m, int ncol, int nlev,
                                      const double *temperature, double *planck) {
  const int ng = m->ng, ntp = m->ntp;
  const double dt = m->temperature_planck[1] - m->temperature_planck[0]; /* :271 */
  const double t0 = m->temperature_planck[0];                            /* :272 */
  const long cstride = (long)ncol * nlev;
  const double *B = m->planck_function;
  for (int j = 0; j < nlev; ++j) {
    for (int i = 0; i < ncol; ++i) {
      const double T = temperature[i + (long)ncol * j];
      double *o = planck + i + (long)ncol * j;
      double temperature_index = (T - t0) / dt; /* :275 */
      if (temperature_index >= 0) {
        temperature_index = 1. + temperature_index;
        int it0 = (int)temperature_index;
        if (it0 > ntp - 1) it0 = ntp - 1; /* :278 */
        double w1 = temperature_index - it0;
        double w0 = 1. - w1;
        const double *b0 = B + (long)ng * (it0 - 1), *b1 = B + (long)ng * it0;
        for (int k = 0; k < ng; ++k) o[k * cstride] = w0 * b0[k] + w1 * b1[k]; /* :281-282 */
      } else {
        const double r = T / t0;
        for (int k = 0; k < ng; ++k) o[k * cstride] = r * B[k]; /* :284 */
      }
    }
  }
  const long n = cstride * ng;
  for (long q = 0; q < n; ++q) planck[q] = planck[q] / pi_f32; /* :288 */
}

/* ---- src/gas_optics_ecckd.f90:293-319 ---- */
void oracle_calculate_rayleigh_optical_depth(const oracle_model_t *m, int ncol, int nlay,
                                             const double *plev, double *od) {
  const double gw = global_weight(); /* :314 (1./(gravity*0.001*dry_air_molar_mass)) */
  const long n2 = (long)ncol * nlay;
  double *moles = (double *)tmp_alloc(sizeof(double) * n2);
  for (int j = 0; j < nlay; ++j)
    for (int i = 0; i < ncol; ++i)
      moles[i + (long)ncol * j] =
          (plev[i + (long)ncol * (j + 1)] - plev[i + (long)ncol * j]) * gw; /* :313-314 */
  for (int k = 0; k < m->ng; ++k)
    for (long q = 0; q < n2; ++q)
      od[q + n2 * k] = moles[q] * m->rayleigh_molar_scattering_coeff[k]; /* :316 */
  tmp_free(moles);
}

/* trim() comparison of two blank-padded Fortran names == strcmp of the C strings. */
static int same_name(const char *a, const char *b) { return strcmp(a, b) == 0; }

/* ---- src/gas_optics_ecckd.f90:323-376 ---- */
int oracle_gas_optical_depth(const oracle_model_t *m, int ncol, int nlay, const double *plev,
                             const double *tlay, const oracle_gas_concs_t *gc, double *tau,
                             char *errmsg) {
  const long n2 = (long)ncol * nlay, n3 = n2 * m->ng;
  double *layer_vmr = (double *)tmp_alloc(sizeof(double) * n2);
  double *od = (double *)tmp_alloc(sizeof(double) * n3);
  if (errmsg) errmsg[0] = 0;
  for (long q = 0; q < n3; ++q) tau[q] = 0.; /* :346 */
  int first_calc = 1;
  for (int j = 0; j < gc->ngas; ++j) { /* :348 gas_desc order */
    int i;
    for (i = 0; i < m->num_gases; ++i) {
      if (same_name(m->gas[i].name, gc->names[j])) {
        /* :351 get_vmr broadcasts the stored vmr to (ncol,nlay) */
        for (int l = 0; l < nlay; ++l)
          for (int c = 0; c < ncol; ++c)
            layer_vmr[c + (long)ncol * l] =
                gc->vmr[j][c * gc->col_stride[j] + l * gc->lay_stride[j]];
        break;
      }
    }
    if (i >= m->num_gases) continue;                             /* :358-364 unknown gas */
    if (m->gas[i].composite_only && !first_calc) continue;       /* :365-367 */
    oracle_calculate_optical_depth(m, i, ncol, nlay, plev, tlay, layer_vmr, od); /* :368 */
    for (long q = 0; q < n3; ++q) tau[q] = tau[q] + od[q];       /* :370 */
    if (m->gas[i].composite_only) first_calc = 0;                /* :371-373 */
  }
  tmp_free(layer_vmr);
  tmp_free(od);
  return 0;
}

/* ---- src/gas_optics_ecckd.f90:381-426 ---- */
int oracle_gas_optics_int(const oracle_model_t *m, int ncol, int nlay, const double *plev,
                          const double *tlay, const double *tsfc, const oracle_gas_concs_t *gc,
                          const double *tlev, double *tau, double *lay_source,
                          double *lev_source_inc, double *lev_source_dec, double *sfc_source,
                          char *errmsg) {
  const int ng = m->ng;
  oracle_gas_optical_depth(m, ncol, nlay, plev, tlay, gc, tau, errmsg);   /* :401 */
  oracle_calculate_planck_function(m, ncol, nlay, tlay, lay_source);       /* :407 */
  oracle_calculate_planck_function(m, ncol, 1, tsfc, sfc_source);          /* :408-413 */
  if (!tlev) {                                                             /* :414-417 */
    if (errmsg) strcpy(errmsg, "tlev is required for ecckd");
    return 1;
  }
  const long n2 = (long)ncol * (nlay + 1);
  double *buffer = (double *)tmp_alloc(sizeof(double) * n2 * ng);
  oracle_calculate_planck_function(m, ncol, nlay + 1, tlev, buffer);       /* :419-422 */
  for (int k = 0; k < ng; ++k)
    for (int l = 0; l < nlay; ++l)
      for (int c = 0; c < ncol; ++c) {
        const long o = c + (long)ncol * (l + (long)nlay * k);
        lev_source_inc[o] = buffer[c + (long)ncol * ((l + 1) + (long)(nlay + 1) * k)]; /* :423 */
        lev_source_dec[o] = buffer[c + (long)ncol * (l + (long)(nlay + 1) * k)];       /* :424 */
      }
  tmp_free(buffer);
  return 0;
}

/* ---- src/gas_optics_ecckd.f90:431-473 ---- */
int oracle_gas_optics_ext(const oracle_model_t *m, int ncol, int nlay, const double *plev,
                          const double *tlay, const oracle_gas_concs_t *gc, double *tau,
                          double *ssa, double *g, double *toa_src, char *errmsg) {
  const long n3 = (long)ncol * nlay * m->ng;
  oracle_gas_optical_depth(m, ncol, nlay, plev, tlay, gc, tau, errmsg);   /* :449 */
  double *od = (double *)tmp_alloc(sizeof(double) * n3);
  oracle_calculate_rayleigh_optical_depth(m, ncol, nlay, plev, od);        /* :455 */
  for (long q = 0; q < n3; ++q) tau[q] = tau[q] + od[q];                   /* :456 */
  if (!ssa || !g) {                                                        /* :461-463 */
    if (errmsg) strcpy(errmsg, "shortwave must use ty_optical_props_2str");
    tmp_free(od);
    return 1;
  }
  for (long q = 0; q < n3; ++q) { ssa[q] = od[q] / tau[q]; g[q] = 0; }     /* :459-460 */
  tmp_free(od);
  for (int j = 0; j < m->ng; ++j)                                          /* :468-472 */
    for (int i = 0; i < ncol; ++i) toa_src[i + (long)ncol * j] = m->solar_irradiance[j];
  return 0;
}

/* =====================  RTE-RRTMGP solvers (published algorithm, v1.5 era)  ===================== */

/* Gauss-Jacobi-5 secants and weights, mo_rte_lw: gauss_Ds(:,nmus), gauss_wts(:,nmus). */
static const double gauss_Ds[4][4] = {{1.66, 0., 0., 0.},
                                      {1.18350343, 2.81649655, 0., 0.},
                                      {1.09719858, 1.69338507, 4.70941630, 0.},
                                      {1.06056257, 1.38282560, 2.40148179, 7.15513024}};
static const double gauss_wts[4][4] = {{0.5, 0., 0., 0.},
                                       {0.3180413817, 0.1819586183, 0., 0.},
                                       {0.2009319137, 0.2292411064, 0.0698269799, 0.},
                                       {0.1355069134, 0.2034645680, 0.1298475476, 0.0311809710}};

/* One quadrature angle for one g-point: lw_solver_noscat = transmittance, lw_source_noscat,
 * lw_transport_noscat, intensity -> flux.  radn_* are (ncol,nlay+1). */
/* Version-sensitive details of the un-pinned RTE-RRTMGP solvers, as
 * switches; the defaults are the v1.5-era forms the rest of this file restates. */
void oracle_default_solver_options(oracle_solver_options_t *o) {
  o->lw_tau_thresh = sqrt(2.220446049250313e-16); /* sqrt(epsilon(tau)) */
  o->lw_series_terms = 2;
  o->lw_inc_flux_isotropic = 0;
  o->sw_k_floor = 1.e-12;
  o->sw_dir_clamp = 0;
}

static void lw_solver_noscat(int ncol, int nlay, int top_at_1, double D, double weight,
                             const double *tau, const double *lay_source,
                             const double *lev_source_inc, const double *lev_source_dec,
                             const double *sfc_emis, const double *sfc_src, const double *inc_flux,
                             const oracle_solver_options_t *opt, double *radn_up,
                             double *radn_dn, double *tau_loc, double *trans, double *source_dn,
                             double *source_up) {
  const double pi = acos(-1.);
  const double tau_thresh = opt->lw_tau_thresh;
  const double *lev_source_up = top_at_1 ? lev_source_dec : lev_source_inc;
  const double *lev_source_dn = top_at_1 ? lev_source_inc : lev_source_dec;
  const int top_level = top_at_1 ? 0 : nlay;
  /* flux at the top of the domain -> intensity assuming azimuthal isotropy (Appendix B.1:
   * I_dn(top) = inc_flux/(2 pi w_k); no incident flux: 0).  lw_inc_flux_isotropic: inc_flux/pi for every
   * angle, which makes flux_dn(top) == inc_flux with any number of angles. */
  for (int i = 0; i < ncol; ++i)
    radn_dn[i + (long)ncol * top_level] =
        inc_flux ? (opt->lw_inc_flux_isotropic ? inc_flux[i] / pi : inc_flux[i] / (2. * pi * weight)) : 0.;
  for (int l = 0; l < nlay; ++l)
    for (int i = 0; i < ncol; ++i) {
      const long q = i + (long)ncol * l;
      tau_loc[q] = tau[q] * D;
      trans[q] = exp(-tau_loc[q]);
      const double tl = tau_loc[q];
      const double series = opt->lw_series_terms >= 3 ? tl * (0.5 + tl * (-1. / 3. + tl * (1. / 8.)))
                                                      : tl * (0.5 - 1. / 3. * tl);
      const double fact = (tl > tau_thresh) ? (1. - trans[q]) / tl - trans[q] : series;
      source_dn[q] = (1. - trans[q]) * lev_source_dn[q] +
                     2. * fact * (lay_source[q] - lev_source_dn[q]);
      source_up[q] = (1. - trans[q]) * lev_source_up[q] +
                     2. * fact * (lay_source[q] - lev_source_up[q]);
    }
  if (top_at_1) {
    for (int l = 1; l <= nlay; ++l)
      for (int i = 0; i < ncol; ++i)
        radn_dn[i + (long)ncol * l] = trans[i + (long)ncol * (l - 1)] *
                                          radn_dn[i + (long)ncol * (l - 1)] +
                                      source_dn[i + (long)ncol * (l - 1)];
    for (int i = 0; i < ncol; ++i) {
      const double sfc_albedo = 1. - sfc_emis[i];
      const double source_sfc = sfc_emis[i] * sfc_src[i];
      radn_up[i + (long)ncol * nlay] = radn_dn[i + (long)ncol * nlay] * sfc_albedo + source_sfc;
    }
    for (int l = nlay - 1; l >= 0; --l)
      for (int i = 0; i < ncol; ++i)
        radn_up[i + (long)ncol * l] = trans[i + (long)ncol * l] * radn_up[i + (long)ncol * (l + 1)] +
                                      source_up[i + (long)ncol * l];
  } else {
    for (int l = nlay - 1; l >= 0; --l)
      for (int i = 0; i < ncol; ++i)
        radn_dn[i + (long)ncol * l] = trans[i + (long)ncol * l] * radn_dn[i + (long)ncol * (l + 1)] +
                                      source_dn[i + (long)ncol * l];
    for (int i = 0; i < ncol; ++i) {
      const double sfc_albedo = 1. - sfc_emis[i];
      const double source_sfc = sfc_emis[i] * sfc_src[i];
      radn_up[i] = radn_dn[i] * sfc_albedo + source_sfc;
    }
    for (int l = 1; l <= nlay; ++l)
      for (int i = 0; i < ncol; ++i)
        radn_up[i + (long)ncol * l] = trans[i + (long)ncol * (l - 1)] *
                                          radn_up[i + (long)ncol * (l - 1)] +
                                      source_up[i + (long)ncol * (l - 1)];
  }
  const long n2 = (long)ncol * (nlay + 1);
  for (long q = 0; q < n2; ++q) {
    radn_dn[q] = 2. * pi * weight * radn_dn[q];
    radn_up[q] = 2. * pi * weight * radn_up[q];
  }
}

void oracle_rte_lw(int ncol, int nlay, int ng, int top_at_1, int nmus, const double *tau,
                   const double *lay_source, const double *lev_source_inc,
                   const double *lev_source_dec, const double *sfc_emis_gpt,
                   const double *sfc_source, double *flux_up, double *flux_dn) {
  oracle_solver_options_t opt;
  oracle_default_solver_options(&opt);
  oracle_rte_lw_opt(ncol, nlay, ng, top_at_1, nmus, tau, lay_source, lev_source_inc, lev_source_dec, sfc_emis_gpt,
                    sfc_source, NULL, &opt, flux_up, flux_dn);
}

/* rte_lw with the optional incident diffuse flux inc_flux(ncol,ng) at the top of the domain (NULL: none)
 * and the version switches. */
void oracle_rte_lw_opt(int ncol, int nlay, int ng, int top_at_1, int nmus, const double *tau,
                       const double *lay_source, const double *lev_source_inc,
                       const double *lev_source_dec, const double *sfc_emis_gpt,
                       const double *sfc_source, const double *inc_flux,
                       const oracle_solver_options_t *opt, double *flux_up, double *flux_dn) {
  oracle_rte_lw_gpt(ncol, nlay, ng, top_at_1, nmus, tau, lay_source, lev_source_inc, lev_source_dec, sfc_emis_gpt,
                    sfc_source, inc_flux, opt, flux_up, flux_dn, NULL, NULL);
}

/* ... and with the spectral fluxes gpt_flux_up / gpt_flux_dn (ncol,nlay+1,ng) that lw_solver_noscat_GaussQuad
 * itself returns (NULL: not wanted); flux_up / flux_dn (sum_broadband of them) may be NULL too. */
void oracle_rte_lw_gpt(int ncol, int nlay, int ng, int top_at_1, int nmus, const double *tau,
                       const double *lay_source, const double *lev_source_inc,
                       const double *lev_source_dec, const double *sfc_emis_gpt,
                       const double *sfc_source, const double *inc_flux,
                       const oracle_solver_options_t *opt, double *flux_up, double *flux_dn,
                       double *gpt_flux_up, double *gpt_flux_dn) {
  const long n2 = (long)ncol * nlay, n2l = (long)ncol * (nlay + 1);
  double *gup = (double *)tmp_alloc(sizeof(double) * n2l), *gdn = (double *)tmp_alloc(sizeof(double) * n2l);
  double *rup = (double *)tmp_alloc(sizeof(double) * n2l), *rdn = (double *)tmp_alloc(sizeof(double) * n2l);
  double *w1 = (double *)tmp_alloc(sizeof(double) * n2 * 4);
  for (int k = 0; k < ng; ++k) {
    const long o3 = n2 * k, o2 = (long)ncol * k;
    lw_solver_noscat(ncol, nlay, top_at_1, gauss_Ds[nmus - 1][0], gauss_wts[nmus - 1][0], tau + o3,
                     lay_source + o3, lev_source_inc + o3, lev_source_dec + o3, sfc_emis_gpt + o2,
                     sfc_source + o2, inc_flux ? inc_flux + o2 : NULL, opt, gup, gdn, w1, w1 + n2, w1 + 2 * n2,
                     w1 + 3 * n2);
    for (int imu = 1; imu < nmus; ++imu) { /* lw_solver_noscat_GaussQuad */
      lw_solver_noscat(ncol, nlay, top_at_1, gauss_Ds[nmus - 1][imu], gauss_wts[nmus - 1][imu],
                       tau + o3, lay_source + o3, lev_source_inc + o3, lev_source_dec + o3,
                       sfc_emis_gpt + o2, sfc_source + o2, inc_flux ? inc_flux + o2 : NULL, opt, rup, rdn, w1,
                       w1 + n2, w1 + 2 * n2, w1 + 3 * n2);
      for (long q = 0; q < n2l; ++q) { gup[q] = gup[q] + rup[q]; gdn[q] = gdn[q] + rdn[q]; }
    }
    if (gpt_flux_up) memcpy(gpt_flux_up + n2l * k, gup, sizeof(double) * n2l);
    if (gpt_flux_dn) memcpy(gpt_flux_dn + n2l * k, gdn, sizeof(double) * n2l);
    if (!flux_up || !flux_dn) continue;
    /* sum_broadband: first g assigns, the rest accumulate in g order */
    if (k == 0) for (long q = 0; q < n2l; ++q) { flux_up[q] = gup[q]; flux_dn[q] = gdn[q]; }
    else for (long q = 0; q < n2l; ++q) { flux_up[q] = flux_up[q] + gup[q]; flux_dn[q] = flux_dn[q] + gdn[q]; }
  }
  tmp_free(gup); tmp_free(gdn); tmp_free(rup); tmp_free(rdn); tmp_free(w1);
}

/* sw_two_stream + sw_source_2str + adding for one g-point, one column at a time. */
void oracle_rte_sw(int ncol, int nlay, int ng, int top_at_1, const double *tau,
                   const double *ssa, const double *g, const double *mu0, const double *toa,
                   const double *sfc_alb_dir_gpt, const double *sfc_alb_dif_gpt,
                   double *flux_up, double *flux_dn, double *flux_dir) {
  oracle_solver_options_t opt;
  oracle_default_solver_options(&opt);
  oracle_rte_sw_opt(ncol, nlay, ng, top_at_1, tau, ssa, g, mu0, toa, sfc_alb_dir_gpt, sfc_alb_dif_gpt, &opt,
                    flux_up, flux_dn, flux_dir);
}

void oracle_rte_sw_opt(int ncol, int nlay, int ng, int top_at_1, const double *tau,
                       const double *ssa, const double *g, const double *mu0, const double *toa,
                       const double *sfc_alb_dir_gpt, const double *sfc_alb_dif_gpt,
                       const oracle_solver_options_t *opt, double *flux_up, double *flux_dn,
                       double *flux_dir) {
  oracle_rte_sw_gpt(ncol, nlay, ng, top_at_1, tau, ssa, g, mu0, toa, NULL, sfc_alb_dir_gpt, sfc_alb_dif_gpt, opt, flux_up,
                    flux_dn, flux_dir, NULL, NULL, NULL);
}

/* ... with the diffuse incident flux inc_flux_dif(ncol,ng) (NULL: none) and the spectral fluxes (ncol,nlay+1,ng)
 * that sw_solver_2stream itself returns (NULL: not wanted; the broadband outputs may be NULL too). */
void oracle_rte_sw_gpt(int ncol, int nlay, int ng, int top_at_1, const double *tau,
                       const double *ssa, const double *g, const double *mu0, const double *toa,
                       const double *inc_flux_dif, const double *sfc_alb_dir_gpt,
                       const double *sfc_alb_dif_gpt, const oracle_solver_options_t *opt,
                       double *flux_up, double *flux_dn, double *flux_dir, double *gpt_flux_up,
                       double *gpt_flux_dn, double *gpt_flux_dir) {
  const double eps = 2.220446049250313e-16;
  const long n2l = (long)ncol * (nlay + 1);
  double *Rdif = (double *)tmp_alloc(sizeof(double) * nlay * 9 + sizeof(double) * (nlay + 1) * 6);
  double *Tdif = Rdif + nlay, *Rdir = Tdif + nlay, *Tdir = Rdir + nlay, *Tnoscat = Tdir + nlay;
  double *src_up = Tnoscat + nlay, *src_dn = src_up + nlay, *denom = src_dn + nlay;
  double *spare = denom + nlay;
  double *albedo = spare + nlay, *src = albedo + (nlay + 1), *fdir = src + (nlay + 1);
  double *fdn = fdir + (nlay + 1), *fup = fdn + (nlay + 1);
  if (flux_up && flux_dn)
    for (long q = 0; q < n2l; ++q) { flux_up[q] = 0.; flux_dn[q] = 0.; if (flux_dir) flux_dir[q] = 0.; }
  for (int k = 0; k < ng; ++k) {
    for (int i = 0; i < ncol; ++i) {
      const double m0 = mu0[i], mu0_inv = 1. / m0;
      for (int l = 0; l < nlay; ++l) {
        const long q = i + (long)ncol * (l + (long)nlay * k);
        const double w0 = ssa[q], gg = g[q], t = tau[q];
        const double gamma1 = (8. - w0 * (5. + 3. * gg)) * .25;
        const double gamma2 = 3. * (w0 * (1. - gg)) * .25;
        const double gamma3 = (2. - 3. * m0 * gg) * .25;
        const double gamma4 = 1. - gamma3;
        const double alpha1 = gamma1 * gamma4 + gamma2 * gamma3;
        const double alpha2 = gamma1 * gamma3 + gamma2 * gamma4;
        const double kk = sqrt(dmax((gamma1 - gamma2) * (gamma1 + gamma2), opt->sw_k_floor));
        const double exp_minusktau = exp(-t * kk);
        const double exp_minus2ktau = exp_minusktau * exp_minusktau;
        double RT_term = 1. / (kk * (1. + exp_minus2ktau) + gamma1 * (1. - exp_minus2ktau));
        Rdif[l] = RT_term * gamma2 * (1. - exp_minus2ktau);
        Tdif[l] = RT_term * 2. * kk * exp_minusktau;
        Tnoscat[l] = exp(-t * mu0_inv);
        const double k_mu = kk * m0, k_gamma3 = kk * gamma3, k_gamma4 = kk * gamma4;
        const double d = 1. - k_mu * k_mu;
        RT_term = w0 * RT_term / (fabs(d) >= eps ? d : eps);
        Rdir[l] = RT_term * ((1. - k_mu) * (alpha2 + k_gamma3) -
                             (1. + k_mu) * (alpha2 - k_gamma3) * exp_minus2ktau -
                             2.0 * (k_gamma3 - alpha2 * k_mu) * exp_minusktau * Tnoscat[l]);
        Tdir[l] = -RT_term * ((1. + k_mu) * (alpha1 + k_gamma4) * Tnoscat[l] -
                              (1. - k_mu) * (alpha1 - k_gamma4) * exp_minus2ktau * Tnoscat[l] -
                              2.0 * (k_gamma4 + alpha1 * k_mu) * exp_minusktau);
        if (opt->sw_dir_clamp) { /* later RTE releases: the direct beam can neither gain energy nor go negative */
          Rdir[l] = dmax(0., dmin(Rdir[l], 1. - Tnoscat[l]));
          Tdir[l] = dmax(0., dmin(Tdir[l], 1. - Tnoscat[l] - Rdir[l]));
        }
      }
      /* sw_source_2str + adding, in "layer index from the top" coordinates */
      const int top = top_at_1 ? 0 : nlay;
      const int step = top_at_1 ? 1 : -1; /* level index moving away from the top */
      fdir[0] = toa[i + (long)ncol * k] * m0;
      for (int s = 0; s < nlay; ++s) { /* s-th layer below the top */
        const int l = top_at_1 ? s : nlay - 1 - s;
        src_up[s] = Rdir[l] * fdir[s];
        src_dn[s] = Tdir[l] * fdir[s];
        fdir[s + 1] = Tnoscat[l] * fdir[s];
      }
      albedo[nlay] = sfc_alb_dif_gpt[i + (long)ncol * k];
      src[nlay] = fdir[nlay] * sfc_alb_dir_gpt[i + (long)ncol * k];
      for (int s = nlay - 1; s >= 0; --s) {
        const int l = top_at_1 ? s : nlay - 1 - s;
        denom[s] = 1. / (1. - Rdif[l] * albedo[s + 1]);
        albedo[s] = Rdif[l] + Tdif[l] * Tdif[l] * albedo[s + 1] * denom[s];
        src[s] = src_up[s] + Tdif[l] * denom[s] * (src[s + 1] + albedo[s + 1] * src_dn[s]);
      }
      fdn[0] = inc_flux_dif ? inc_flux_dif[i + (long)ncol * k] : 0.;
      fup[0] = fdn[0] * albedo[0] + src[0];
      for (int s = 1; s <= nlay; ++s) {
        const int l = top_at_1 ? s - 1 : nlay - s;
        fdn[s] = (Tdif[l] * fdn[s - 1] + Rdif[l] * src[s] + src_dn[s - 1]) * denom[s - 1];
        fup[s] = fdn[s] * albedo[s] + src[s];
      }
      for (int s = 0; s <= nlay; ++s) {
        const long q = i + (long)ncol * (top + step * s);
        const double dn = fdn[s] + fdir[s];
        if (gpt_flux_up) gpt_flux_up[q + n2l * k] = fup[s];
        if (gpt_flux_dn) gpt_flux_dn[q + n2l * k] = dn;
        if (gpt_flux_dir) gpt_flux_dir[q + n2l * k] = fdir[s];
        if (!flux_up || !flux_dn) continue;
        if (k == 0) { flux_up[q] = fup[s]; flux_dn[q] = dn; if (flux_dir) flux_dir[q] = fdir[s]; }
        else {
          flux_up[q] = flux_up[q] + fup[s]; flux_dn[q] = flux_dn[q] + dn;
          if (flux_dir) flux_dir[q] = flux_dir[q] + fdir[s];
        }
      }
    }
  }
  tmp_free(Rdif);
}

/* ---- block loop of example/rfmip-rad-irf/ecckd_rfmip_lw.F90:107-136 ---- */
int oracle_lw_pipeline(const oracle_model_t *m, int ncol, int nlay, int block, int nthreads,
                       const double *plev, const double *tlay, const double *tlev,
                       const double *tsfc, const oracle_gas_concs_t *gc, const double *sfc_emis,
                       int nmus, double *flux_up, double *flux_dn) {
  const int ng = m->ng;
  const int nblocks = (ncol + block - 1) / block;
  int status = 0;
  /* bytes of temporaries one block needs: the block's own arrays below, gas_optical_depth (layer_vmr, optical_depth),
   * the level-source buffer of gas_optics_int, rte_lw's work arrays; with head room */
  const size_t n2b = (size_t)block * nlay, n2lb = (size_t)block * (nlay + 1);
  const size_t per_block = sizeof(double) * (n2b * ng * 4 + (size_t)block * ng * 2 + n2b * 2 + n2lb * 4 + block   /* buf */
                                             + n2b + n2b * ng                                                   /* layer_vmr, od */
                                             + n2lb * ng                                                        /* buffer */
                                             + n2lb * 4 + n2b * 4) + 64 * 32;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
  {
  char *arena = (char *)malloc(per_block);
  tl_arena = arena; tl_cap = arena ? per_block : 0;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
  for (int b = 0; b < nblocks; ++b) {
    tl_off = 0;
    const int c0 = b * block, nc = (c0 + block <= ncol) ? block : ncol - c0;
    const long n2 = (long)nc * nlay, n2l = (long)nc * (nlay + 1);
    double *buf = (double *)tmp_alloc(sizeof(double) * (n2 * ng * 4 + (long)nc * ng * 2 + n2 * 2 + n2l * 4 + nc));
    double *tau = buf, *lay = tau + n2 * ng, *inc = lay + n2 * ng, *dec = inc + n2 * ng;
    double *sfc = dec + n2 * ng, *emis = sfc + (long)nc * ng;
    double *bplev = emis + (long)nc * ng, *btlev = bplev + n2l, *btlay = btlev + n2l;
    double *fu = btlay + n2, *fd = fu + n2l, *btsfc = fd + n2l;
    /* pack the block's columns (the reference driver reads pre-blocked arrays) */
    for (int l = 0; l <= nlay; ++l)
      for (int c = 0; c < nc; ++c) {
        bplev[c + (long)nc * l] = plev[c0 + c + (long)ncol * l];
        btlev[c + (long)nc * l] = tlev[c0 + c + (long)ncol * l];
      }
    for (int l = 0; l < nlay; ++l)
      for (int c = 0; c < nc; ++c) btlay[c + (long)nc * l] = tlay[c0 + c + (long)ncol * l];
    for (int c = 0; c < nc; ++c) btsfc[c] = tsfc[c0 + c];
    /* per-block gas_concs view: offset every vmr pointer by the block start */
    const double *vp[ORACLE_MAX_GASES * 2];
    oracle_gas_concs_t bgc = *gc;
    for (int j = 0; j < gc->ngas; ++j) vp[j] = gc->vmr[j] + c0 * gc->col_stride[j];
    bgc.vmr = vp;
    char err[128];
    if (oracle_gas_optics_int(m, nc, nlay, bplev, btlay, btsfc, &bgc, btlev, tau, lay, inc, dec, sfc, err))
      status = 1;
    for (int k = 0; k < ng; ++k) /* ecckd_rfmip_lw.F90:112-116, one band */
      for (int c = 0; c < nc; ++c) emis[c + (long)nc * k] = sfc_emis[c0 + c];
    oracle_rte_lw(nc, nlay, ng, 1, nmus, tau, lay, inc, dec, emis, sfc, fu, fd);
    for (int l = 0; l <= nlay; ++l)
      for (int c = 0; c < nc; ++c) {
        flux_up[c0 + c + (long)ncol * l] = fu[c + (long)nc * l];
        flux_dn[c0 + c + (long)ncol * l] = fd[c + (long)nc * l];
      }
    tmp_free(buf);
  }
  tl_arena = NULL; tl_cap = 0; tl_off = 0;
  free(arena);
  }
  return status;
}
