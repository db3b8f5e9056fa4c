#!/bin/bash
# address-translation counters of the LW pair at 1e6 columns (one pass): UTCL1 requests / hits / misses per kernel, UTCL2 busy
export TMPDIR=/tmp
o=gpurun_out
rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --output-format csv -d $o/r03_pmc_tlb -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-side > /dev/null 2> $o/r03_pmc_tlb.err
python tools/pmc_summary.py "synthetic 1000000 columns x 60 layers x 32 g-points, LW fsck-tol0.0161, fp64" 1000000 $o/r03_pmc_tlb > $o/r03_pmc_tlb.json
find $o -name "*agent_info.csv" -path "*r03_pmc_tlb*" -delete 2>/dev/null || true
