#!/bin/bash
# Host-side code of the library (loader, CDF reader, netCDF C ABI, launch planning) under AddressSanitizer +
# UndefinedBehaviorSanitizer, on the CPU (GPU ASan is not available on the pool): builds a sanitized copy of the
# library in /tmp and runs the CPU C-ABI tests against it.
set -e
cd "$(dirname "$0")/../rte-ecckd_amd/csrc"
d=/tmp/ecckd_asan; rm -rf $d; mkdir -p $d
for f in capi.cpp model.cpp cdf1.cpp nc_capi.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -c $f -o $d/${f%.*}.o
done
for f in kernels_gas_fused.hip kernels_tau.hip kernels_planck.hip kernels_rte_lw.hip kernels_rte_sw.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c $f -o $d/${f%.*}.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fsanitize=address,undefined -Wl,-rpath,/opt/rocm/lib -o $d/librte_ecckd_hip_asan.so $d/*.o
cd ../..
rt=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
ECCKD_LIB=$d/librte_ecckd_hip_asan.so LD_PRELOAD=$rt ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_capi_host.py -x -q
