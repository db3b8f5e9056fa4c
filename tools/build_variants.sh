#!/bin/bash
# Builds experimental variants of the library next to the product .so so that one gpurun call can
# A/B them (tools/ab.py):   tools/build_variants.sh "<extra hipcc flags>" ...
# Flags apply to kernels_gas_fused.hip, kernels_tau.hip and kernels_rte_lw.hip; a leading "ALL:"
# applies them to every source.
set -e
cd "$(dirname "$0")/../rte-ecckd_amd/csrc"
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off"
mkdir -p ../../variants_tmp /tmp/ecckd_var
i=${VSTART:-0}   # VSTART=2 numbers the variants from v3 (keeps lib_v1/v2 of an earlier source state for the A/B)
for v in "$@"; do
  i=$((i+1))
  echo "v$i: $v"
  (
    d=/tmp/ecckd_var/v$i; rm -rf $d; mkdir -p $d
    tf="$v"; of=""
    case "$v" in ALL:*) of="${v#ALL:}"; tf="${v#ALL:}";; esac
    for s in kernels_gas_fused.hip kernels_tau.hip kernels_rte_lw.hip kernels_rte_lw_split.hip kernels_rte_sw.hip kernels_rte_sw_sys.hip; do /opt/rocm/bin/hipcc $BASE $tf -c $s -o $d/${s%.*}.o & done
    for s in kernels_planck.hip kernels_rte_gpt.hip capi.cpp nc_capi.cpp model.cpp cdf1.cpp; do
      /opt/rocm/bin/hipcc $BASE $of -c $s -o $d/${s%.*}.o
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wl,-rpath,/opt/rocm/lib -o ../../variants_tmp/lib_v$i.so $d/*.o
  ) &
done
wait
