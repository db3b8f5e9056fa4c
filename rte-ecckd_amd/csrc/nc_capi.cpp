// nc_capi.cpp -- C ABI of include/ecckd_nc.h over the CDF reader (host I/O plumbing for the
// RFMIP-shaped Fortran drivers; no GPU code).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ecckd_nc.h"
#include "cdf1.hpp"

namespace ecckd { int set_last_error(const std::string &msg); }

struct ecckd_nc {
  ecckd::CdfFile f;
  explicit ecckd_nc(const std::string &p) : f(p) {}
};

extern "C" {

int ecckd_nc_open(const char *path, ecckd_nc_t **file) {
  if (!path || !file) return ecckd::set_last_error("ecckd_nc_open: null argument");
  *file = nullptr;
  try {
    *file = new ecckd_nc(path);
  } catch (const std::exception &e) {
    return ecckd::set_last_error(e.what());
  }
  return 0;
}

void ecckd_nc_close(ecckd_nc_t *file) { delete file; }

int ecckd_nc_dim_size(const ecckd_nc_t *file, const char *dim, int *size) {
  if (!file || !dim || !size) return ecckd::set_last_error("ecckd_nc_dim_size: null argument");
  auto it = file->f.dims().find(dim);
  if (it == file->f.dims().end()) return ecckd::set_last_error(std::string("get_dim_size: can't find dimension ") + dim);
  *size = (int)(it->second == 0 ? file->f.numrecs() : it->second);
  return 0;
}

int ecckd_nc_var_exists(const ecckd_nc_t *file, const char *var) { return file && var && file->f.has_var(var) ? 1 : 0; }

int ecckd_nc_var_size(const ecckd_nc_t *file, const char *var, long long *n) {
  if (!file || !var || !n) return ecckd::set_last_error("ecckd_nc_var_size: null argument");
  if (!file->f.has_var(var)) return ecckd::set_last_error(std::string("can't find variable ") + var);
  long long k = 1;
  for (size_t d : file->f.var(var).shape) k *= (long long)d;
  *n = k;
  return 0;
}

int ecckd_nc_read_f64(const ecckd_nc_t *file, const char *var, double *out, long long n) {
  if (!file || !var || !out) return ecckd::set_last_error("ecckd_nc_read_f64: null argument");
  try {
    std::vector<double> v = file->f.read(var);
    if ((long long)v.size() != n)
      return ecckd::set_last_error(std::string("read_field: variable ") + var + " has an unexpected size");
    std::memcpy(out, v.data(), v.size() * sizeof(double));
  } catch (const std::exception &e) {
    return ecckd::set_last_error(e.what());
  }
  return 0;
}

int ecckd_nc_get_att_text(const ecckd_nc_t *file, const char *var, const char *att, char *buf, int buflen) {
  if (!file || !att || !buf || buflen < 1) return ecckd::set_last_error("ecckd_nc_get_att_text: null argument");
  try {
    std::string v;
    if (!var || !var[0]) {
      v = file->f.text_att(att);
    } else {
      const ecckd::CdfVar &cv = file->f.var(var);
      auto it = cv.text_atts.find(att);
      if (it == cv.text_atts.end())
        return ecckd::set_last_error(std::string("can't read attribute '") + att + "' from variable " + var);
      v = it->second;
    }
    std::memset(buf, 0, buflen);
    std::strncpy(buf, v.c_str(), buflen - 1);
  } catch (const std::exception &e) {
    return ecckd::set_last_error(e.what());
  }
  return 0;
}

int ecckd_nc_write_f64(const char *path, const char *var, const double *values, long long n) {
  if (!path || !var || !values) return ecckd::set_last_error("ecckd_nc_write_f64: null argument");
  {
    FILE *probe = std::fopen(path, "rb");
    if (!probe) return ecckd::set_last_error(std::string("unblock_and_write: can't find file ") + path);
    std::fclose(probe);
  }
  try {
    ecckd::CdfFile f(path);
    const ecckd::CdfVar &v = f.var(var);
    if (v.record) return ecckd::set_last_error(std::string("write_field: record variable not supported: ") + var);
    long long k = 1;
    for (size_t d : v.shape) k *= (long long)d;
    if (k != n) return ecckd::set_last_error(std::string("write_field: wrong size for ") + var);
    size_t ts = v.nc_type == 6 ? 8 : (v.nc_type == 5 || v.nc_type == 4 ? 4 : 0);
    if (!ts) return ecckd::set_last_error(std::string("write_field: unsupported type of ") + var);
    std::vector<unsigned char> out((size_t)n * ts);
    for (long long i = 0; i < n; ++i) {
      unsigned char *p = &out[(size_t)i * ts];
      if (v.nc_type == 6) {
        uint64_t u;
        std::memcpy(&u, &values[i], 8);
        for (int b = 0; b < 8; ++b) p[b] = (unsigned char)(u >> (56 - 8 * b));
      } else if (v.nc_type == 5) {
        float fl = (float)values[i];
        uint32_t u;
        std::memcpy(&u, &fl, 4);
        for (int b = 0; b < 4; ++b) p[b] = (unsigned char)(u >> (24 - 8 * b));
      } else {
        uint32_t u = (uint32_t)(int32_t)values[i];
        for (int b = 0; b < 4; ++b) p[b] = (unsigned char)(u >> (24 - 8 * b));
      }
    }
    FILE *fp = std::fopen(path, "r+b");
    if (!fp) return ecckd::set_last_error(std::string("unblock_and_write: can't find file ") + path);
    bool ok = std::fseek(fp, (long)v.begin, SEEK_SET) == 0 && std::fwrite(out.data(), 1, out.size(), fp) == out.size();
    ok = (std::fclose(fp) == 0) && ok;
    if (!ok) return ecckd::set_last_error(std::string("write_field: short write to ") + path);
  } catch (const std::exception &e) {
    return ecckd::set_last_error(e.what());
  }
  return 0;
}

}  // extern "C"
