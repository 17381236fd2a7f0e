#!/bin/bash
# evidence pass: gpu tests, smoke, bench (with cpu baseline), rocprofv3 kernel stats, PMC passes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py 2>gpurun_out/bench.err | tail -1 > gpurun_out/bench_n1.json; cut -c1-300 gpurun_out/bench_n1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1
cd $R
python tools/rocprof_summary.py $(find gpurun_out/prof -name "*.db" | head -1) gpurun_out/bench_kernel_stats.txt --cmd "python bench.py --steps 5 --warmup 2 --no-cpu-baseline" | head -12
bash tools/run_pmc.sh 2>&1 | tail -5
