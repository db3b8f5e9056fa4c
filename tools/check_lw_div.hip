// check_lw_div.hip -- lw_div() / lw_exp() of lw_layer.hpp against `/` and the device library's exp() on 2^26 operands each.
// lw_div: x = 1 - exp(-d) (what lw_source_noscat divides), d log-uniform over [1e-8, 1e3] and over [1e3, 1e290]: the
// quotients must be the same bits.  lw_exp: arguments -d, largest difference from exp() in ulp (both are ~1 ulp routines).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/check_lw_div tools/check_lw_div.hip && /tmp/check_lw_div
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include "../rte-ecckd_amd/csrc/lw_layer.hpp"

__device__ unsigned long long mix(unsigned long long z) {
  z += 0x9e3779b97f4a7c15ULL; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31);
}
__global__ void k(double lo, double hi, unsigned long long seed, unsigned long long *out) {
  const unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
  const double u = (double)(mix(i ^ seed) >> 11) * 0x1p-53;
  const double d = exp(log(lo) + u * (log(hi) - log(lo)));
  const double t = exp(-d), x = 1. - t;
  const double a = ecckd::lw_div(x, d), b = x / d;
  if (__double_as_longlong(a) != __double_as_longlong(b)) atomicAdd(out, 1ULL);
  const double e1 = ecckd::lw_exp(-d);
  const long long dd = __double_as_longlong(e1) - __double_as_longlong(t);
  const unsigned long long ad = dd < 0 ? -dd : dd;
  if (ad) { atomicAdd(out + 1, 1ULL); atomicMax(out + 2, ad); }
}
int main() {
  unsigned long long *dev;
  if (hipMalloc(&dev, 24) != hipSuccess) return 1;
  const double ranges[3][2] = {{1e-8, 1e3}, {1e3, 1e290}, {1e-8, 700.}};
  for (auto &r : ranges) {
    (void)hipMemset(dev, 0, 24);
    hipLaunchKernelGGL(k, dim3(1 << 18), dim3(256), 0, 0, r[0], r[1], 777ULL, dev);
    unsigned long long h[3];
    (void)hipMemcpy(h, dev, 24, hipMemcpyDeviceToHost);
    printf("d in [%g, %g]: lw_div differs from `/` in %llu of %u; lw_exp differs from exp() in %llu, max %llu ulp\n", r[0], r[1], h[0], 1u << 26, h[1], h[2]);
  }
  return 0;
}
