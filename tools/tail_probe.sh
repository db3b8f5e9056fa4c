# rte_lw / rte_sw around a full round of waves: what the tiles beyond the last full round cost
ulimit -c 0
for n in 98304 100000 131072 1000000; do
  timeout -k 10 120 python bench.py --ncol $n --steps 10 --warmup 3 --cpu-seconds 0 --no-side | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lw',$n, d['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()}, d['check_max_abs_flux_diff_vs_oracle_Wm2'])"
done
for n in 1000 98304 100000 147456; do
  timeout -k 10 120 python bench.py --mode sw --ncol $n --steps 10 --warmup 3 --cpu-seconds 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sw',$n, d['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()}, d['check_max_abs_flux_diff_vs_oracle_Wm2'])"
done
