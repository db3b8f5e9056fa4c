// cdf1.hpp -- minimal reader for netCDF-3 "classic" (CDF-1) and 64-bit-offset (CDF-2) files.
//
// The reference reads its look-up tables through netcdf-fortran
// (example/rfmip-rad-irf/mo_simple_netcdf.F90:8-29); neither libnetcdf nor netcdf-fortran is
// part of this build, and the ecCKD definition files are plain netCDF-3 classic, so the
// loader carries its own reader: header parse (dims, global attributes, variables), big-endian
// payloads widened to double exactly as mo_simple_netcdf.F90:44-142 does with real(wp) targets.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace ecckd {

struct CdfVar {
  std::string name;
  std::vector<int> dimids;
  std::vector<size_t> shape;   // C order (slowest first), as stored on disk
  int nc_type = 0;             // 1 byte, 2 char, 3 short, 4 int, 5 float, 6 double
  uint64_t vsize = 0, begin = 0;
  bool record = false;
  std::map<std::string, std::string> text_atts;
  std::map<std::string, std::vector<double>> num_atts;
};

class CdfFile {
 public:
  // Throws std::runtime_error with a message on any failure.
  explicit CdfFile(const std::string &path);
  bool has_var(const std::string &name) const { return vars_.count(name) != 0; }
  const CdfVar &var(const std::string &name) const;
  // Whole variable widened to double, in on-disk (C) order == Fortran order of reversed dims.
  std::vector<double> read(const std::string &name) const;
  bool has_text_att(const std::string &name) const { return gtext_.count(name) != 0; }
  const std::string &text_att(const std::string &name) const;
  const std::map<std::string, size_t> &dims() const { return dims_; }
  size_t numrecs() const { return numrecs_; }

 private:
  std::vector<unsigned char> buf_;
  std::map<std::string, size_t> dims_;
  std::vector<size_t> dimlen_;
  std::map<std::string, CdfVar> vars_;
  std::map<std::string, std::string> gtext_;
  size_t numrecs_ = 0;
  uint64_t recsize_ = 0;
};

}  // namespace ecckd
