// Micro-benchmark: LDS floating-point atomic add rates on MI355X as the flux solvers use them
// (wave-private accumulators [level][CW], lanes cl + CW*gs; only gs == 0 contributes).
//   mode 0: all 64 lanes issue the atomic, non-owners add +0.0      (rte_lw round-1 scheme)
//   mode 1: exec-masked, owner lanes only
//   mode 2: owner lanes do a plain read / add / write
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_atomic.hip -o ubench_atomic
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename real, int MODE, int CW>
__global__ void __launch_bounds__(64) k(real *out, int iters) {
  __shared__ real acc[64 * CW];
  const int lane = threadIdx.x, cl = lane % CW, gs = lane / CW;
  const bool owner = gs == 0;
  for (int i = lane; i < 64 * CW; i += 64) acc[i] = 0;
  real v = real(1) + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 64; ++s) {
      if (MODE == 0) __hip_atomic_fetch_add(&acc[s * CW + cl], owner ? v : real(0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      if (MODE == 1) { if (owner) __hip_atomic_fetch_add(&acc[s * CW + cl], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
      if (MODE == 2) { if (owner) acc[s * CW + cl] += v; }
      v = v * real(1.0000001);
    }
  }
  real r = 0;
  for (int s = 0; s < 64; ++s) r += acc[s * CW + cl];
  out[blockIdx.x * 64 + lane] = r;
}

template <typename real, int MODE, int CW>
void run(const char *name, int wpc) {
  real *out; hipMalloc(&out, sizeof(real) * 64 * 256 * 32);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, grid = 256 * wpc;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<real, MODE, CW>), dim3(grid), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  double instr = (double)grid * iters * 64;
  printf("%-34s waves/CU %2d: %7.1f cycles per wave-instr per CU @2.4GHz\n", name, wpc, ms * 1e-3 * 2.4e9 * 256 / instr);
  hipFree(out);
}

int main() {
  for (int wpc : {4, 8}) {
    run<double, 0, 32>("f64 all-lanes(+0) CW32", wpc);
    run<double, 0, 16>("f64 all-lanes(+0) CW16", wpc);
    run<double, 1, 32>("f64 masked atomic CW32", wpc);
    run<double, 1, 16>("f64 masked atomic CW16", wpc);
    run<double, 2, 32>("f64 masked plain RMW CW32", wpc);
    run<float, 0, 32>("f32 all-lanes(+0) CW32", wpc);
    run<float, 0, 64>("f32 all-lanes CW64", wpc);
    run<float, 1, 32>("f32 masked atomic CW32", wpc);
    run<float, 1, 16>("f32 masked atomic CW16", wpc);
    run<float, 2, 32>("f32 masked plain RMW CW32", wpc);
  }
  return 0;
}
