cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( for N in 144 443; do for T in 8 16 64; do tools/native/hmc_replay $N $T 20; AGP_SPIN=0 tools/native/hmc_replay $N $T 20; done; done; tools/native/threads_bench 144 8 400 grad; tools/native/threads_bench 144 8 400; tools/native/hmc_replay 2048 512 2; tools/native/hmc_replay 2048 64 2; tools/native/hmc_replay 512 256 4 ) 2>&1 | grep "^{" | python -c "
import sys, json
for ln in sys.stdin:
    d=json.loads(ln); print({k:d[k] for k in d if k in ('tool','entry','n','threads','time_points','hmc_iterations_per_s','evals_per_s','mean_batch')})"
timeout 600 python -m pytest tests/test_gpu_soak.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | grep -E "passed|failed"
python tools/gpu_stress_threads.py 10 2>&1 | grep "^stress"
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null
