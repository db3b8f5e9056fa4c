// kernels_rte_gpt.hip -- the RTE solvers with SPECTRAL (per-g-point) flux output, i.e. with the interfaces of
// RTE-RRTMGP's kernels lw_solver_noscat_GaussQuad and sw_solver_2stream (mo_rte_solver_kernels.F90, v1.5 era), which
// return flux(ncol,nlay+1,ngpt) and leave the g-point sum to sum_broadband.  These are the compatibility kernels
// behind include/rte_kernels_hip.h (a host model that calls RTE's kernel layer directly); the fast paths -- broadband
// reduction fused into the solver, kernels_rte_lw.hip / kernels_rte_sw.hip -- are what rte_lw / rte_sw callers get.
//
// One thread per (column, g-point), columns fastest (coalesced); no shared memory, no scratch:
//   longwave   the up sweep recomputes the layer transmittance and source instead of storing them;
//   shortwave  pass 1 (surface -> top) parks albedo and source of each level in the OUTPUT arrays flux_up / flux_dn,
//              pass 2 (top -> surface) reads them back and overwrites them with the fluxes.
// Arithmetic: the same expressions, in the same order, as the restatement in oracle/ (IEEE division).
#include "kernels.hpp"

namespace ecckd {
namespace {

__global__ void __launch_bounds__(256) lw_gpt_kernel(const RteGptArgs a) {
  const long n = (long)a.ncol * a.ng;
  const int nlay = a.nlay, nlev = nlay + 1;
  const long ncol = a.ncol;
  const double pi = acos(-1.);
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const long c = id % ncol, g = id / ncol;
    const long base3 = c + ncol * nlay * g, basef = c + ncol * nlev * g, base2 = c + ncol * g;
    const long lay0 = a.top_at_1 ? 0 : nlay - 1, lev0 = a.top_at_1 ? 0 : nlay, lstep = a.top_at_1 ? 1 : -1;
    const double *Bdn = a.top_at_1 ? a.lev_source_inc : a.lev_source_dec;
    const double *Bup = a.top_at_1 ? a.lev_source_dec : a.lev_source_inc;
    const double eps = a.sfc_emis[base2], sfc_src = a.sfc_src[base2];
    const double inc = a.inc_flux ? a.inc_flux[base2] : 0.;
    for (int k = 0; k < a.nmus; ++k) {
      const double D = a.Ds[k], w = a.wts[k], wfac = 2. * pi * w;
      double I = a.inc_flux ? (a.inc_isotropic ? inc / pi : inc / (2. * pi * w)) : 0.;
      auto put = [&](double *arr, int s, double v) {
        const long q = basef + ncol * (lev0 + lstep * s);
        arr[q] = k == 0 ? v : arr[q] + v;
      };
      auto cell = [&](int s, double &t, double &sdn, double &su) {
        const long q = base3 + ncol * (lay0 + lstep * s);
        const double tl = a.tau[q] * D;
        t = exp(-tl);
        const double series = a.series3 ? tl * (0.5 + tl * (-1. / 3. + tl * (1. / 8.))) : tl * (0.5 - 1. / 3. * tl);
        const double fact = tl > a.tau_thresh ? (1. - t) / tl - t : series;
        const double lay = a.lay_source[q], bdn = Bdn[q], bup = Bup[q];
        sdn = (1. - t) * bdn + 2. * fact * (lay - bdn);
        su = (1. - t) * bup + 2. * fact * (lay - bup);
      };
      for (int s = 0; s < nlay; ++s) {
        double t, sdn, su;
        cell(s, t, sdn, su);
        put(a.flux_dn, s, wfac * I);
        I = t * I + sdn;
      }
      put(a.flux_dn, nlay, wfac * I);
      double U = I * (1. - eps) + eps * sfc_src;
      for (int s = nlay - 1; s >= 0; --s) {
        double t, sdn, su;
        cell(s, t, sdn, su);
        put(a.flux_up, s + 1, wfac * U);
        U = t * U + su;
      }
      put(a.flux_up, 0, wfac * U);
    }
  }
}

__global__ void __launch_bounds__(256) sw_gpt_kernel(const RteGptArgs a) {
  const long n = (long)a.ncol * a.ng;
  const int nlay = a.nlay, nlev = nlay + 1;
  const long ncol = a.ncol;
  const double eps = 2.220446049250313e-16;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long)gridDim.x * blockDim.x) {
    const long c = id % ncol, g = id / ncol;
    const long base3 = c + ncol * nlay * g, basef = c + ncol * nlev * g, base2 = c + ncol * g;
    const long lay0 = a.top_at_1 ? 0 : nlay - 1, lev0 = a.top_at_1 ? 0 : nlay, lstep = a.top_at_1 ? 1 : -1;
    const double mu0 = a.mu0[c], mu0_inv = 1. / mu0;
    auto two_stream = [&](int s, double &Rdif, double &Tdif, double &Rdir, double &Tdir, double &Tnoscat) {
      const long q = base3 + ncol * (lay0 + lstep * s);
      const double tau = a.tau[q], w0 = a.ssa[q], gq = a.g[q];
      const double gamma1 = (8. - w0 * (5. + 3. * gq)) * .25;
      const double gamma2 = 3. * (w0 * (1. - gq)) * .25;
      const double gamma3 = (2. - 3. * mu0 * gq) * .25;
      const double gamma4 = 1. - gamma3;
      const double alpha1 = gamma1 * gamma4 + gamma2 * gamma3;
      const double alpha2 = gamma1 * gamma3 + gamma2 * gamma4;
      const double kk0 = (gamma1 - gamma2) * (gamma1 + gamma2);
      const double k = sqrt(kk0 > a.k_floor ? kk0 : a.k_floor);
      const double e1 = exp(-tau * k), e2 = e1 * e1;
      double RT = 1. / (k * (1. + e2) + gamma1 * (1. - e2));
      Rdif = RT * gamma2 * (1. - e2);
      Tdif = RT * 2. * k * e1;
      Tnoscat = exp(-tau * mu0_inv);
      const double k_mu = k * mu0, k_gamma3 = k * gamma3, k_gamma4 = k * gamma4;
      const double d = 1. - k_mu * k_mu;
      RT = w0 * RT / (fabs(d) >= eps ? d : eps);
      Rdir = RT * ((1. - k_mu) * (alpha2 + k_gamma3) - (1. + k_mu) * (alpha2 - k_gamma3) * e2 -
                   2.0 * (k_gamma3 - alpha2 * k_mu) * e1 * Tnoscat);
      Tdir = -RT * ((1. + k_mu) * (alpha1 + k_gamma4) * Tnoscat - (1. - k_mu) * (alpha1 - k_gamma4) * e2 * Tnoscat -
                    2.0 * (k_gamma4 + alpha1 * k_mu) * e1);
      if (a.dir_clamp) {
        const double lim = 1. - Tnoscat;
        Rdir = fmax(0., fmin(Rdir, lim));
        Tdir = fmax(0., fmin(Tdir, lim - Rdir));
      }
    };
    auto lev = [&](int s) { return basef + ncol * (lev0 + lstep * s); };
    // pass 1, surface -> top: albedo and the source normalised by the direct flux at its own level (it is only
    // known on the way down): parked in flux_up / flux_dn
    double albedo = a.alb_dif[base2], nsrc = a.alb_dir[base2];
    a.flux_up[lev(nlay)] = albedo;
    a.flux_dn[lev(nlay)] = nsrc;
    for (int s = nlay - 1; s >= 0; --s) {
      double Rdif, Tdif, Rdir, Tdir, Tn;
      two_stream(s, Rdif, Tdif, Rdir, Tdir, Tn);
      const double denom = 1. / (1. - Rdif * albedo);
      nsrc = Rdir + Tdif * denom * (nsrc * Tn + albedo * Tdir);
      albedo = Rdif + Tdif * Tdif * albedo * denom;
      a.flux_up[lev(s)] = albedo;
      a.flux_dn[lev(s)] = nsrc;
    }
    // pass 2, top -> surface
    double fdir = a.fdir_top[base2];
    double fdn = a.inc_dif ? a.inc_dif[base2] : 0.;
    {
      const double fup = fdn * albedo + nsrc * fdir;
      a.flux_up[lev(0)] = fup;
      a.flux_dn[lev(0)] = fdn + fdir;
      if (a.flux_dir) a.flux_dir[lev(0)] = fdir;
    }
    for (int s = 0; s < nlay; ++s) {
      const double alb_next = a.flux_up[lev(s + 1)], nsrc_next = a.flux_dn[lev(s + 1)];
      double Rdif, Tdif, Rdir, Tdir, Tn;
      two_stream(s, Rdif, Tdif, Rdir, Tdir, Tn);
      const double denom = 1. / (1. - Rdif * alb_next);
      const double fdir_next = Tn * fdir;
      const double src_next = nsrc_next * fdir_next;
      fdn = (Tdif * denom) * fdn + (Rdif * denom) * src_next + (Tdir * denom) * fdir;
      const double fup = fdn * alb_next + src_next;
      fdir = fdir_next;
      a.flux_up[lev(s + 1)] = fup;
      a.flux_dn[lev(s + 1)] = fdn + fdir;
      if (a.flux_dir) a.flux_dir[lev(s + 1)] = fdir;
    }
  }
}

}  // namespace

static unsigned gpt_blocks(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

hipError_t launch_lw_gpt(const RteGptArgs &a, hipStream_t s) {
  if (a.ncol <= 0 || a.ng <= 0) return hipSuccess;
  hipLaunchKernelGGL(lw_gpt_kernel, dim3(gpt_blocks((long)a.ncol * a.ng)), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_sw_gpt(const RteGptArgs &a, hipStream_t s) {
  if (a.ncol <= 0 || a.ng <= 0) return hipSuccess;
  hipLaunchKernelGGL(sw_gpt_kernel, dim3(gpt_blocks((long)a.ncol * a.ng)), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace ecckd
