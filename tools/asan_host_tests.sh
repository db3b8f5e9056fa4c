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
# the kernels are not under test here: reuse the objects of the regular build (python -c 'import rte_ecckd_amd as p; p.build()')
for f in kernels_gas_fused kernels_tau kernels_planck kernels_rte_lw kernels_rte_lw_split kernels_rte_sw kernels_rte_sw_sys kernels_rte_gpt; do
  cp ../build/obj/$f.o $d/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fsanitize=address,undefined -Wl,-rpath,/opt/rocm/lib -o $d/librte_ecckd_hip_asan.so $d/*.o
cd ../..
rt=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
ECCKD_LIB=$d/librte_ecckd_hip_asan.so LD_PRELOAD=$rt ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_capi_host.py -x -q -k "not test_code_object_resources and not test_library_exports"
