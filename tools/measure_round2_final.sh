ulimit -c 0; export TMPDIR=/tmp; o=gpurun_out
P="--ncol 300000 --steps 2 --warmup 1 --cpu-seconds 0 --no-side"
W="synthetic 300000 columns x 60 layers x 32 g-points, LW fsck-tol0.0161, fp64"
rm -rf $o/r02_lds $o/r02_vmem1 $o/r02_vmem2
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $o/r02_lds -- python3 bench.py $P > /dev/null 2> $o/r02_lds.err && echo lds done
timeout -k 10 200 rocprofv3 --pmc SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $o/r02_vmem1 -- python3 bench.py $P > /dev/null 2> $o/r02_vmem1.err && echo vmem1 done
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $o/r02_vmem2 -- python3 bench.py $P > /dev/null 2> $o/r02_vmem2.err && echo vmem2 done
python tools/pmc_summary.py "$W" 300000 $o/r02_lds > $o/r02_pmc_lds.json
python tools/pmc_summary.py "$W" 300000 $o/r02_vmem1 $o/r02_vmem2 > $o/r02_pmc_vmem.json
python bench.py --steps 10 --warmup 2 > $o/r02_bench.json 2> $o/r02_bench.err && echo bench done
python bench.py --mode sw --ncol 100000 --steps 10 --warmup 2 > $o/r02_bench_sw.json 2> $o/r02_bench_sw.err && echo sw done
bash tools/tail_probe.sh 2>&1 | grep -v amdgpu.ids > $o/r02_tail_probe.txt
