// model.hpp -- host + device image of ty_gas_optics_ecckd (src/gas_optics_ecckd.f90:23-48).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

struct ecckd_model {
  // --- ty_gas_optics_ecckd members ---
  int ng = 0, np = 0, nt = 0, ntp = 0;
  std::vector<double> log_pressure;     // (np)
  std::vector<double> temperature;      // (np,nt)
  std::vector<double> planck_function;  // (ng,ntp)
  std::vector<double> temperature_planck;
  std::vector<double> solar_irradiance, rayleigh;
  double total_solar_irradiance = 0.;
  bool has_planck = false, has_solar = false;
  struct Gas {                          // AbsorptionTable (:13-19) + name (:25)
    std::string name;
    int code = 0, composite_only = 0, nv = 1;
    double ref = 0.;
    std::vector<double> mole_fraction, coef;
    bool has_negative = false;
    size_t dev_off = 0;                 // offset (doubles) into dbuf
  };
  std::vector<Gas> gas;
  int num_composite_gases = 0;
  // --- ty_optical_props (parent) band structure, set by init() ---
  int nband = 0;
  std::vector<int> band2gpt;            // (2,nband) 1-based inclusive
  std::vector<double> band_lims_wvn;    // (2,nband)
  // --- device image ---
  bool finalized = false;
  int device = -1;
  double *dbuf = nullptr;               // all tables, one allocation
  float *dbuf32 = nullptr;              // the same image in single precision (same offsets)
  bool f32_exact = false;               // every table value is a float32 number (true for files read from float32 variables)
  size_t off_temperature = 0, off_planck = 0, off_rayleigh = 0, off_solar = 0;
  size_t off_zero = 0;                  // 32 zero words: target of the loads of unused gas slots
  // --- ECCKD_HOST staging arena (grown on demand, serialised by mu) ---
  std::mutex mu;
  void *arena = nullptr;
  size_t arena_bytes = 0;
  hipStream_t host_stream = nullptr;
};

namespace ecckd {
// load_and_init (example/rfmip-rad-irf/mo_load_coefficients.F90:19-146); throws on error.
void load_and_init(ecckd_model &m, const std::string &filename);
// "" if every table has the extents the kernels assume, else what is wrong (load and builder routes)
std::string validate_model(const ecckd_model &m);
// mo_load_coefficients.F90:244-293 (including the dropped single-character last token)
std::vector<std::string> tokenize(const std::string &buffer);
}  // namespace ecckd
