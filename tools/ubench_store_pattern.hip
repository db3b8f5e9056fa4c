// What HBM write bandwidth does the STORE PATTERN of gas_fused_kernel allow, with no arithmetic at all?
// The fused gas-optics kernel writes four (ncol,nlay,ng) arrays; a block of 512 threads owns 512 consecutive columns of one
// layer at a time (a "tile") and stores, per tile, one 4 KiB piece into each of 4 x 32 g-planes that lie ncol*nlay*8 B
// (480 MB at 1e6 columns) apart: 128 write streams per block, 256 blocks in flight.  This kernel issues exactly those
// stores (16 B per lane, even lanes plane g, odd lanes plane g+1, nontemporal) and nothing else, in several traversal
// orders:
//   mode 0  as the product: block (x = column chunk, y = layer) walks its contiguous tile range
//   mode 1  tiles interleaved over the chunks of a layer (concurrent blocks write adjacent 4 KiB pieces)
//   mode 2  TW tiles per plane visit: the block stores TW adjacent tiles (TW*4 KiB contiguous) into a plane before it moves
//           to the next plane (what a kernel holding TW tiles of results at once would do)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_store_pattern.hip -o build_tmp/ubench_store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d2 __attribute__((ext_vector_type(2)));

template <int TW>
__global__ void __launch_bounds__(512) pattern(double *a0, double *a1, double *a2, double *a3, int ncol, int nlay, int ng, int mode,
                                               int narr) {
  const int tid = threadIdx.x, j = blockIdx.y;
  const long ntiles = ((long)ncol + 511) / 512;
  const long plane = (long)ncol * nlay;
  const bool odd = tid & 1;
  double *arr[4] = {a0, a1, a2, a3};
  const long t_begin = ntiles * blockIdx.x / gridDim.x, t_end = ntiles * (blockIdx.x + 1) / gridDim.x;
  const long nmine = t_end - t_begin;
  for (long k = 0; k < nmine; k += TW) {
    for (int g = 0; g < ng; g += 2) {
      for (int ar = 0; ar < narr; ++ar) {
#pragma unroll
        for (int w = 0; w < TW; ++w) {
          if (k + w >= nmine) break;
          const long tile = mode == 1 ? blockIdx.x + (k + w) * gridDim.x : t_begin + k + w;
          if (tile >= ntiles) continue;
          const long c = tile * 512 + tid;
          if (c + 1 >= ncol + (odd ? 1 : 0)) continue;
          const long o = (c - (odd ? 1 : 0)) + (long)ncol * j + plane * (g + (odd ? 1 : 0));
          const d2 v = {(double)c, (double)g};
          __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(arr[ar] + o));
        }
      }
    }
  }
}

int main(int argc, char **argv) {
  const int ncol = argc > 1 ? atoi(argv[1]) : 1000000, nlay = 60, ng = 32;
  const size_t n3 = (size_t)ncol * nlay * ng;
  double *a[4];
  for (int i = 0; i < 4; ++i) { hipMalloc(&a[i], n3 * 8); hipMemset(a[i], 0, n3 * 8); }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // lds_bytes of dynamic LDS limit the blocks per CU as the product's slab does (150 KiB: one block = 8 waves per CU)
  auto run = [&](const char *name, int mode, int tw, int chunks, int narr, int lds_bytes = 0) {
    float best = 1e30f;
    hipFuncSetAttribute((const void *)pattern<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      dim3 grid(chunks, nlay), blk(512);
      if (tw == 1) hipLaunchKernelGGL(pattern<1>, grid, blk, lds_bytes, 0, a[0], a[1], a[2], a[3], ncol, nlay, ng, mode, narr);
      else if (tw == 2) hipLaunchKernelGGL(pattern<2>, grid, blk, 0, 0, a[0], a[1], a[2], a[3], ncol, nlay, ng, mode, narr);
      else if (tw == 4) hipLaunchKernelGGL(pattern<4>, grid, blk, 0, 0, a[0], a[1], a[2], a[3], ncol, nlay, ng, mode, narr);
      else hipLaunchKernelGGL(pattern<8>, grid, blk, 0, 0, a[0], a[1], a[2], a[3], ncol, nlay, ng, mode, narr);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    printf("%-46s chunks %3d arrays %d lds %3d KiB: %.2f ms  %.2f TB/s\n", name, chunks, narr, lds_bytes / 1024, best,
           narr * n3 * 8.0 / (best * 1e-3) / 1e12);
  };
  for (int chunks : {64, 128}) {
    run("product order (contiguous tile range per block)", 0, 1, chunks, 4);
    run("tiles interleaved over the blocks of a layer", 1, 1, chunks, 4);
    run("2 tiles (8 KiB) per plane visit", 2, 2, chunks, 4);
    run("4 tiles (16 KiB) per plane visit", 2, 4, chunks, 4);
    run("8 tiles (32 KiB) per plane visit", 2, 8, chunks, 4);
  }
  run("product order, ONE array (tau only)", 0, 1, 64, 1);
  // the product's occupancy: blocks per CU limited by LDS
  run("product order, two blocks per CU", 0, 1, 64, 4, 75 * 1024);
  run("product order, one block per CU", 0, 1, 64, 4, 150 * 1024);
  run("ONE array, one block per CU", 0, 1, 64, 1, 150 * 1024);
  return 0;
}
