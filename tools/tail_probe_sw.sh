# rte_sw with and without the tail split (sw_tail_split): small blocks and around whole rounds of waves
ulimit -c 0
for n in 1000 4000 16000 50000 100000; do
 for o in 1 0; do
  timeout -k 10 120 python bench.py --mode sw --ncol $n --steps 20 --warmup 3 --cpu-seconds 0 --solver-option sw_tail_split=$o | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sw ncol',$n,'sw_tail_split',$o, 'ms/step %.4f' % d['ms_per_step'], {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
 done
done
