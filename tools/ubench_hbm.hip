// Practical HBM ceilings with hand-written streaming kernels (16-byte accesses, grid-stride, 16 GiB):
// read-only, write-only, copy; plain and nontemporal.  Complements tools/hbm_ceiling.py (torch kernels).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_hbm.hip -o ubench_hbm
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT>
__global__ void rd(const d2 *x, size_t n, double *out) {
  d2 s = {0., 0.};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const d2 v = NT ? __builtin_nontemporal_load(x + i) : x[i];
    s += v;
  }
  if (s[0] + s[1] == 1.2345) out[0] = s[0];   // never true: keeps the loads
}
template <bool NT>
__global__ void wr(d2 *x, size_t n, double a) {
  const d2 v = {a, a};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (NT) __builtin_nontemporal_store(v, x + i); else x[i] = v;
  }
}
template <bool NT>
__global__ void cp(const d2 *x, d2 *y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(x + i), y + i); else y[i] = x[i];
  }
}

int main() {
  const size_t bytes = 16ull << 30, n = bytes / sizeof(d2);
  d2 *x, *y; double *out;
  hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&out, 8);
  hipMemset(x, 0, bytes); hipMemset(y, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256 * 8, 256 * 16, 256 * 32}) {
    for (int k = 0; k < 6; ++k) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        switch (k) {
          case 0: hipLaunchKernelGGL(rd<false>, dim3(blocks), dim3(256), 0, 0, x, n, out); break;
          case 1: hipLaunchKernelGGL(rd<true>, dim3(blocks), dim3(256), 0, 0, x, n, out); break;
          case 2: hipLaunchKernelGGL(wr<false>, dim3(blocks), dim3(256), 0, 0, x, n, 1.5); break;
          case 3: hipLaunchKernelGGL(wr<true>, dim3(blocks), dim3(256), 0, 0, x, n, 1.5); break;
          case 4: hipLaunchKernelGGL(cp<false>, dim3(blocks), dim3(256), 0, 0, x, y, n); break;
          case 5: hipLaunchKernelGGL(cp<true>, dim3(blocks), dim3(256), 0, 0, x, y, n); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      const char *nm[6] = {"read", "read nt", "write", "write nt", "copy", "copy nt"};
      printf("blocks %5d %-9s %.2f TB/s\n", blocks, nm[k], (k >= 4 ? 2.0 : 1.0) * bytes / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
