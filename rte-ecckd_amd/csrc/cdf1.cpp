// cdf1.cpp -- see cdf1.hpp.  Format: "The NetCDF Classic Format Specification" (header =
// magic, numrecs, dim_list, gatt_list, var_list; everything big-endian, names and values padded
// to 4 bytes).
#include "cdf1.hpp"

#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace ecckd {
namespace {

struct Cursor {
  const std::vector<unsigned char> &b;
  size_t p = 0;
  explicit Cursor(const std::vector<unsigned char> &buf) : b(buf) {}
  void need(size_t n) const {
    if (p + n > b.size()) throw std::runtime_error("netcdf: truncated header");
  }
  uint32_t u32() {
    need(4);
    uint32_t v = (uint32_t)b[p] << 24 | (uint32_t)b[p + 1] << 16 | (uint32_t)b[p + 2] << 8 | b[p + 3];
    p += 4;
    return v;
  }
  uint64_t u64() {
    uint64_t hi = u32(), lo = u32();
    return hi << 32 | lo;
  }
  std::string name() {
    uint32_t n = u32();
    need(n);
    std::string s(reinterpret_cast<const char *>(&b[p]), n);
    p += (n + 3) & ~3u;
    return s;
  }
};

size_t type_size(int t) {
  switch (t) {
    case 1: case 2: return 1;
    case 3: return 2;
    case 4: case 5: return 4;
    case 6: return 8;
  }
  throw std::runtime_error("netcdf: unknown nc_type");
}

double decode(const unsigned char *p, int t) {
  switch (t) {
    case 1: return (double)(signed char)p[0];
    case 2: return (double)p[0];
    case 3: return (double)(int16_t)((uint16_t)p[0] << 8 | p[1]);
    case 4: return (double)(int32_t)((uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]);
    case 5: {
      uint32_t u = (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3];
      float f;
      std::memcpy(&f, &u, 4);
      return (double)f;
    }
    case 6: {
      uint64_t u = 0;
      for (int i = 0; i < 8; ++i) u = u << 8 | p[i];
      double d;
      std::memcpy(&d, &u, 8);
      return d;
    }
  }
  throw std::runtime_error("netcdf: unknown nc_type");
}

void read_atts(Cursor &c, std::map<std::string, std::string> &text,
               std::map<std::string, std::vector<double>> *num) {
  uint32_t tag = c.u32(), n = c.u32();
  if (tag == 0 && n == 0) return;
  if (tag != 0x0C) throw std::runtime_error("netcdf: bad attribute list tag");
  for (uint32_t i = 0; i < n; ++i) {
    std::string name = c.name();
    int t = (int)c.u32();
    uint32_t ne = c.u32();
    size_t bytes = (size_t)ne * type_size(t);
    c.need(bytes);
    if (t == 2) {
      text[name] = std::string(reinterpret_cast<const char *>(&c.b[c.p]), ne);
    } else if (num) {
      std::vector<double> v(ne);
      for (uint32_t k = 0; k < ne; ++k) v[k] = decode(&c.b[c.p + k * type_size(t)], t);
      (*num)[name] = v;
    }
    c.p += (bytes + 3) & ~(size_t)3;
  }
}

}  // namespace

CdfFile::CdfFile(const std::string &path) {
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("load_and_init_ecckd(): can't open file" + path);
  std::fseek(f, 0, SEEK_END);
  long sz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  buf_.resize(sz > 0 ? (size_t)sz : 0);
  size_t got = buf_.empty() ? 0 : std::fread(buf_.data(), 1, buf_.size(), f);
  std::fclose(f);
  if (got != buf_.size() || buf_.size() < 8) throw std::runtime_error("netcdf: short read of " + path);
  if (std::memcmp(buf_.data(), "CDF", 3) != 0 || (buf_[3] != 1 && buf_[3] != 2))
    throw std::runtime_error("netcdf: " + path + " is not a netCDF-3 classic/64-bit-offset file");
  const bool off64 = buf_[3] == 2;
  Cursor c(buf_);
  c.p = 4;
  uint32_t nr = c.u32();
  numrecs_ = nr == 0xFFFFFFFFu ? 0 : nr;
  // dim_list
  uint32_t tag = c.u32(), n = c.u32();
  if (!(tag == 0 && n == 0)) {
    if (tag != 0x0A) throw std::runtime_error("netcdf: bad dimension list tag");
    for (uint32_t i = 0; i < n; ++i) {
      std::string name = c.name();
      size_t len = c.u32();
      dims_[name] = len;
      dimlen_.push_back(len);
    }
  }
  read_atts(c, gtext_, nullptr);
  tag = c.u32();
  n = c.u32();
  if (!(tag == 0 && n == 0)) {
    if (tag != 0x0B) throw std::runtime_error("netcdf: bad variable list tag");
    for (uint32_t i = 0; i < n; ++i) {
      CdfVar v;
      v.name = c.name();
      uint32_t nd = c.u32();
      for (uint32_t k = 0; k < nd; ++k) {
        int id = (int)c.u32();
        if (id < 0 || (size_t)id >= dimlen_.size()) throw std::runtime_error("netcdf: bad dimid");
        v.dimids.push_back(id);
        if (k == 0 && dimlen_[id] == 0) { v.record = true; v.shape.push_back(numrecs_); }
        else v.shape.push_back(dimlen_[id]);
      }
      read_atts(c, v.text_atts, &v.num_atts);
      v.nc_type = (int)c.u32();
      v.vsize = c.u32();
      v.begin = off64 ? c.u64() : c.u32();
      if (v.record) recsize_ += v.vsize;
      vars_[v.name] = v;
    }
  }
}

const CdfVar &CdfFile::var(const std::string &name) const {
  auto it = vars_.find(name);
  if (it == vars_.end()) throw std::runtime_error("netcdf: can't find variable " + name);
  return it->second;
}

const std::string &CdfFile::text_att(const std::string &name) const {
  auto it = gtext_.find(name);
  if (it == gtext_.end()) throw std::runtime_error("get_global_attribute: error reading " + name);
  return it->second;
}

std::vector<double> CdfFile::read(const std::string &name) const {
  const CdfVar &v = var(name);
  const size_t ts = type_size(v.nc_type);
  size_t n = 1;
  for (size_t d : v.shape) n *= d;
  std::vector<double> out(n);
  if (!v.record) {
    if (v.begin + n * ts > buf_.size()) throw std::runtime_error("netcdf: truncated variable " + name);
    for (size_t i = 0; i < n; ++i) out[i] = decode(&buf_[v.begin + i * ts], v.nc_type);
  } else {
    const size_t per = numrecs_ ? n / numrecs_ : 0;
    for (size_t r = 0; r < numrecs_; ++r) {
      const uint64_t b = v.begin + r * recsize_;
      if (b + per * ts > buf_.size()) throw std::runtime_error("netcdf: truncated record variable " + name);
      for (size_t i = 0; i < per; ++i) out[r * per + i] = decode(&buf_[b + i * ts], v.nc_type);
    }
  }
  return out;
}

}  // namespace ecckd
