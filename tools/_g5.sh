cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed" 
( for N in 144 443; do for T in 8 64; do tools/native/hmc_replay $N $T 20; done; done; tools/native/threads_bench 144 8 400 grad; tools/native/hmc_replay 2048 512 2 ) 2>&1 | grep "^{" | python -c "
import sys, json
for ln in sys.stdin:
    d=json.loads(ln); print({k:d[k] for k in d if k in ('tool','entry','n','threads','time_points','hmc_iterations_per_s','evals_per_s','mean_batch')})"
AGP_HOST_PROF=1 tools/native/hmc_replay 144 8 20 2>&1 | grep -E "HOST_PROF.*(launches|wait|staging)"
