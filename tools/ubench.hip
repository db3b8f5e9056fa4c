// Micro-benchmarks used to calibrate DESIGN.md's kernel budgets on MI355X: fp64 VALU issue rates
// (v_mul_f64 / v_add_f64 / v_fma_f64), ds_read_b64 rates with uniform and per-lane rows, at a given
// occupancy.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void valu(double *out, double a, double b, int iters) {
  double x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x * 1e-9 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) x[i] = x[i] * b;
      if (OP == 1) x[i] = x[i] + b;
      if (OP == 2) x[i] = fma(x[i], b, a);
      if (OP == 3) { x[i] = x[i] * b; x[i] = x[i] + a; }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// LDS read throughput: each lane reads rows chosen by (lane*spread) % nrows, row stride SR doubles
__global__ void lds(double *out, int iters, int spread, int nrows, int SR) {
  extern __shared__ double sm[];
  for (int i = threadIdx.x; i < nrows * SR; i += blockDim.x) sm[i] = i;
  __syncthreads();
  typedef __attribute__((address_space(3))) const volatile double lv;
  lv *p = (lv *)sm + ((threadIdx.x * spread) % nrows) * SR;
  double s = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 32; ++g) s += p[g];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  double *out;
  hipMalloc(&out, sizeof(double) * 256 * 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[4] = {"v_mul_f64", "v_add_f64", "v_fma_f64", "mul+add"};
  for (int wpc : {4, 8, 16, 32}) {          // waves per CU
    int block = 64 * (wpc < 16 ? wpc : 16), grid = 256 * (wpc / (block / 64));
    for (int op = 0; op < 4; ++op) {
      const int iters = 20000;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (op == 0) hipLaunchKernelGGL(valu<0>, dim3(grid), dim3(block), 0, 0, out, 1.0, 1.0000001, iters);
        if (op == 1) hipLaunchKernelGGL(valu<1>, dim3(grid), dim3(block), 0, 0, out, 1.0, 1.0000001, iters);
        if (op == 2) hipLaunchKernelGGL(valu<2>, dim3(grid), dim3(block), 0, 0, out, 1.0, 1.0000001, iters);
        if (op == 3) hipLaunchKernelGGL(valu<3>, dim3(grid), dim3(block), 0, 0, out, 1.0, 1.0000001, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double instr = (double)grid * block / 64 * iters * 8 * (op == 3 ? 2 : 1);   // wave-instructions
      printf("waves/CU %2d %-10s %.2f wave-instr/clk/CU @2.4GHz (%.1f T lane-ops/s)\n", wpc, names[op],
             instr / (ms * 1e-3) / 256 / 2.4e9, instr * 64 / (ms * 1e-3) / 1e12);
    }
  }
  for (int wpc : {8, 16}) {
    int block = 512, grid = 256 * wpc / 8;
    for (int spread : {0, 1, 7}) {
      const int iters = 4000, nrows = 30, SR = 225;
      hipFuncSetAttribute((const void *)lds, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(lds, dim3(grid), dim3(block), nrows * SR * 8, 0, out, iters, spread, nrows, SR);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double instr = (double)grid * block / 64 * iters * 32;
      printf("waves/CU %2d ds_read_b64 spread %d: %.2f cycles/wave-instr/CU @2.4GHz\n", wpc, spread,
             (ms * 1e-3) * 2.4e9 * 256 / instr);
    }
  }
  return 0;
}
