// model.cpp -- load_and_init: ecCKD "ckd-definition" netCDF file -> ecckd_model.
// Follows example/rfmip-rad-irf/mo_load_coefficients.F90:19-203 step by step.
#include "model.hpp"

#include <cmath>
#include <stdexcept>

#include "../../include/ecckd_hip.h"
#include "cdf1.hpp"

namespace ecckd {

std::vector<std::string> tokenize(const std::string &buffer_in) {
  // mo_load_coefficients.F90:244-293.  n = len_trim(buffer); a token that starts at the very
  // last character is never closed (found_token is only set at i == n) and is dropped.
  size_t n = buffer_in.size();
  while (n > 0 && (buffer_in[n - 1] == ' ' || buffer_in[n - 1] == '\0')) --n;
  std::vector<std::string> tokens;
  bool found = false;
  size_t start = 0;
  for (size_t i = 0; i < n; ++i) {
    const char ch = buffer_in[i];
    if (found) {
      if (i == n - 1 && ch != ' ') {
        tokens.push_back(buffer_in.substr(start, i - start + 1));
        break;
      } else if (ch == ' ') {
        tokens.push_back(buffer_in.substr(start, i - start));
        found = false;
      }
    } else if (ch != ' ') {
      found = true;
      start = i;
    }
  }
  if (tokens.size() > 16) throw std::runtime_error("tokenize: more tokens than can fit in array.");
  for (auto &t : tokens)
    if (t.size() > ECCKD_NAME_LEN) throw std::runtime_error("tokenize: token is cut off.");
  return tokens;
}

namespace {

// read_gas_input_data, mo_load_coefficients.F90:149-203
ecckd_model::Gas read_gas_input_data(const CdfFile &f, const std::string &gas_name, int ng, int np,
                                     int nt) {
  ecckd_model::Gas g;
  bool lut = false;
  const std::string mf = gas_name + "_mole_fraction";
  const std::string cv = gas_name + "_molar_absorption_coeff";
  if (f.has_var(mf) && f.var(mf).shape.size() == 1) {   // :162-176
    lut = true;
    g.code = ECCKD_LOOK_UP_TABLE;
    g.mole_fraction = f.read(mf);
    g.nv = (int)g.mole_fraction.size();
    if (f.var(cv).shape.size() != 4)
      throw std::runtime_error("load_and_init_ecckd: absorption coefficient not 4d for " + gas_name);
    g.coef = f.read(cv);
  }
  if (!lut) {   // :177-202
    const int n = (int)f.read(gas_name + "_conc_dependence_code").at(0);
    if (n == 0) g.code = ECCKD_NONE;
    else if (n == 1) g.code = ECCKD_LINEAR;
    else if (n == 3) {
      g.code = ECCKD_RELATIVE_LINEAR;
      g.ref = f.read(gas_name + "_reference_mole_fraction").at(0);
    } else
      throw std::runtime_error("load_and_init_ecckd: bad concentration code for " + gas_name);
    if (f.var(cv).shape.size() != 3)
      throw std::runtime_error("load_and_init_ecckd: absorption coefficient not 3d for " + gas_name);
    g.nv = 1;
    g.coef = f.read(cv);
  }
  if (g.coef.size() != (size_t)ng * np * nt * g.nv)
    throw std::runtime_error("load_and_init_ecckd: unexpected table size for " + gas_name);
  return g;
}

}  // namespace

void load_and_init(ecckd_model &m, const std::string &filename) {
  CdfFile f(filename);
  m.log_pressure = f.read("pressure");   // :46-49
  for (double &p : m.log_pressure) p = std::log(p);
  m.np = (int)m.log_pressure.size();
  m.temperature = f.read("temperature");   // :51-53, Fortran (np,nt)
  if (f.var("temperature").shape.size() != 2) throw std::runtime_error("netcdf: temperature is not 2d");
  m.nt = (int)f.var("temperature").shape[0];

  // :55-78 bands
  std::vector<double> w1 = f.read("wavenumber1_band"), w2 = f.read("wavenumber2_band");
  m.nband = (int)w1.size();
  if (m.nband < 1 || w2.size() != w1.size()) throw std::runtime_error("load_and_init_ecckd: no bands (wavenumber1_band / wavenumber2_band) in " + filename);
  m.band_lims_wvn.resize(2 * (size_t)m.nband);
  for (int b = 0; b < m.nband; ++b) { m.band_lims_wvn[2 * b] = w1[b]; m.band_lims_wvn[2 * b + 1] = w2[b]; }
  std::vector<double> bn = f.read("band_number");
  const int ngb = (int)bn.size();
  if (ngb < 1) throw std::runtime_error("load_and_init_ecckd: empty band_number in " + filename);
  m.band2gpt.assign(2 * (size_t)m.nband, 0);
  m.band2gpt[0] = 1;
  m.band2gpt[2 * (m.nband - 1) + 1] = ngb;
  int band = 1;
  for (int i = 2; i <= ngb; ++i) {
    if ((int)bn[i - 1] + 1 > band) {
      m.band2gpt[2 * (band - 1) + 1] = i - 1;
      band += 1;
      if (band > m.nband) throw std::runtime_error("load_and_init_ecckd: band_number exceeds band count");
      m.band2gpt[2 * (band - 1)] = i;
    }
  }

  // :80-82 only size(gpoint_fraction,2) is ever used
  const CdfVar &gf = f.var("gpoint_fraction");
  if (gf.shape.size() != 2) throw std::runtime_error("netcdf: gpoint_fraction is not 2d");
  m.ng = (int)gf.shape[0];

  m.has_solar = f.has_var("solar_irradiance");   // :84
  if (m.has_solar) {
    m.solar_irradiance = f.read("solar_irradiance");
    m.total_solar_irradiance = 0.;
    for (double s : m.solar_irradiance) m.total_solar_irradiance += s;   // :89
    m.rayleigh = f.read("rayleigh_molar_scattering_coeff");
  } else {
    m.temperature_planck = f.read("temperature_planck");
    m.planck_function = f.read("planck_function");
    m.ntp = (int)m.temperature_planck.size();
    m.has_planck = true;
  }

  // :104-144 gases
  std::vector<std::string> gas = tokenize(f.text_att("constituent_id"));
  std::vector<std::string> composite_gas;
  bool uses_composite = false;
  for (const auto &g : gas)
    if (g == "composite") {
      uses_composite = true;
      composite_gas = tokenize(f.text_att("composite_constituent_id"));
      m.num_composite_gases = (int)composite_gas.size();
      break;
    }
  m.gas.clear();
  for (const auto &g : gas) {
    if (g != "composite") {
      ecckd_model::Gas t = read_gas_input_data(f, g, m.ng, m.np, m.nt);
      t.name = g;
      t.composite_only = 0;
      m.gas.push_back(t);
    }
  }
  if (uses_composite) {
    for (const auto &cg : composite_gas) {
      bool found = false;
      for (const auto &g : gas)
        if (cg == g) { found = true; break; }
      if (!found) {
        ecckd_model::Gas t = read_gas_input_data(f, "composite", m.ng, m.np, m.nt);
        t.name = cg;
        t.composite_only = 1;
        m.gas.push_back(t);
      }
    }
  }
  if (m.gas.size() > ECCKD_MAX_GASES) throw std::runtime_error("load_and_init_ecckd: more than 16 gases");
  const std::string bad = validate_model(m);
  if (!bad.empty()) throw std::runtime_error("load_and_init_ecckd: " + bad + " (" + filename + ")");
}

// Everything the kernels index without a bounds check, checked once: a short or inconsistent table is
// an error at load time, never an out-of-bounds read on the device.  (The reference trusts the file:
// mo_load_coefficients.F90 allocates from the file's own extents and would fail inside netcdf or with
// a Fortran bounds error.)
std::string validate_model(const ecckd_model &m) {
  auto finite = [](const std::vector<double> &v) {
    for (double x : v)
      if (!(x - x == 0.)) return false;
    return true;
  };
  if (m.ng < 1) return "no g-points (gpoint_fraction)";
  if (m.ng > 256) return "more than 256 g-points";
  if (m.np < 2) return "pressure grid needs at least 2 points";
  if (m.nt < 2) return "temperature grid needs at least 2 points";
  if (m.log_pressure.size() != (size_t)m.np) return "pressure has the wrong size";
  if (m.temperature.size() != (size_t)m.np * m.nt) return "temperature is not (pressure, temperature)-shaped";
  if (!finite(m.log_pressure) || !finite(m.temperature)) return "pressure/temperature grid is not finite (pressure must be > 0)";
  if (!(m.log_pressure[1] - m.log_pressure[0] > 0.)) return "pressure grid must increase";
  if (!(m.temperature[m.np] - m.temperature[0] > 0.)) return "temperature grid must increase";
  if (m.nband < 1) return "no bands";
  if (m.band2gpt.size() != 2 * (size_t)m.nband || m.band_lims_wvn.size() != 2 * (size_t)m.nband) return "band tables have the wrong size";
  int next = 1;
  for (int b = 0; b < m.nband; ++b) {   // contiguous, ascending cover of 1..ng
    if (m.band2gpt[2 * b] != next || m.band2gpt[2 * b + 1] < m.band2gpt[2 * b]) return "band_number does not map the g-points onto contiguous bands";
    next = m.band2gpt[2 * b + 1] + 1;
  }
  if (next != m.ng + 1) return "band_number does not cover every g-point (its length must equal the g-point count)";
  if (m.has_planck) {
    if (m.ntp < 2 || m.temperature_planck.size() != (size_t)m.ntp) return "temperature_planck needs at least 2 points";
    if (m.planck_function.size() != (size_t)m.ng * m.ntp) return "planck_function is not (temperature_planck, g-point)-shaped";
    if (!finite(m.temperature_planck) || !(m.temperature_planck[1] - m.temperature_planck[0] > 0.)) return "temperature_planck must increase";
  }
  if (m.has_solar) {
    if (m.solar_irradiance.size() != (size_t)m.ng) return "solar_irradiance has the wrong size";
    if (m.rayleigh.size() != (size_t)m.ng) return "rayleigh_molar_scattering_coeff has the wrong size";
  }
  for (const ecckd_model::Gas &g : m.gas) {
    if (g.code < 0 || g.code > 3) return "bad concentration code for " + g.name;
    if (g.nv < 1 || g.coef.size() != (size_t)m.ng * m.np * m.nt * g.nv) return "unexpected table size for " + g.name;
    if (g.code == ECCKD_LOOK_UP_TABLE) {
      if (g.nv < 2 || g.mole_fraction.size() != (size_t)g.nv) return "look-up-table gas " + g.name + " needs at least 2 mole fractions";
      for (int k = 0; k < g.nv; ++k)
        if (!(g.mole_fraction[k] > 0.) || (k > 0 && !(g.mole_fraction[k] > g.mole_fraction[k - 1])))
          return "mole fractions of " + g.name + " must be positive and increasing";
    }
  }
  return "";
}

}  // namespace ecckd
