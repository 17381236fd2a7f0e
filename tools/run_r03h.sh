cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(bash tools/run_pmc_cmd.sh ic512 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH" -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs) 2>&1 | grep -v "at::native\|rocclr" | cut -c1-260 | head -30
(bash tools/run_pmc_cmd.sh ic64 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH" -- python $R/bench.py --particles 64 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs) 2>&1 | grep -v "at::native\|rocclr" | cut -c1-260 | head -30
