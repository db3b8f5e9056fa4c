// check_sw_sqrt.hip -- sw_sqrt<true>() of sw_two_stream.hpp (v_rsq_f64 + Goldschmidt steps, no range handling) against the
// device library's sqrt() on 2^26 arguments, log-uniform over [1e-12, 4] (the range of the two-stream eigenvalue's
// argument) and over the whole normal range.  Prints the share of results that differ and the largest difference in ulp.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/check_sw_sqrt tools/check_sw_sqrt.hip && /tmp/check_sw_sqrt
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include "../rte-ecckd_amd/csrc/sw_two_stream.hpp"

__device__ unsigned long long mix(unsigned long long z) {
  z += 0x9e3779b97f4a7c15ULL; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31);
}
__global__ void k(double lo, double hi, unsigned long long seed, unsigned long long *ndiff, unsigned long long *maxulp) {
  const unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
  const double u = (double)(mix(i ^ seed) >> 11) * 0x1p-53;
  const double x = exp(log(lo) + u * (log(hi) - log(lo)));
  const double a = ecckd::sw_sqrt<true>(x), b = sqrt(x);
  const long long d = __double_as_longlong(a) - __double_as_longlong(b);
  const unsigned long long ad = d < 0 ? -d : d;
  if (ad) { atomicAdd(ndiff, 1ULL); atomicMax(maxulp, ad); }
}
int main() {
  unsigned long long *dev; hipMalloc(&dev, 16);
  const double ranges[3][2] = {{1e-12, 4.}, {2.3e-308, 1e-12}, {4., 1e300}};
  for (auto &r : ranges) {
    hipMemset(dev, 0, 16);
    hipLaunchKernelGGL(k, dim3(1 << 18), dim3(256), 0, 0, r[0], r[1], 12345ULL, dev, dev + 1);
    unsigned long long h[2]; hipMemcpy(h, dev, 16, hipMemcpyDeviceToHost);
    printf("[%g, %g]: %llu of %u differ from sqrt(), max %llu ulp\n", r[0], r[1], h[0], 1u << 26, h[1]);
  }
  return 0;
}
