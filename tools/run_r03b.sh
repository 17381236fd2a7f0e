cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(time python -m pytest tests/test_gpu_lag.py -m gpu -x -q) > gpurun_out/r03b_pytest_lag.log 2>&1; tail -25 gpurun_out/r03b_pytest_lag.log
(time python -m pytest tests -m gpu -q) > gpurun_out/r03b_pytest_gpu.log 2>&1; tail -30 gpurun_out/r03b_pytest_gpu.log
python bench.py --steps 40 --warmup 5 2>gpurun_out/r03b_bench.err | tail -1 > gpurun_out/r03b_bench_n1.json; cut -c1-300 gpurun_out/r03b_bench_n1.json
python bench.py --particles 64 --no-cpu-baseline --no-extra-legs 2>>gpurun_out/r03b_bench.err | tail -1 > gpurun_out/r03b_bench_P64.json; cut -c1-300 gpurun_out/r03b_bench_P64.json
AGP_LAG=0 python bench.py --particles 64 --no-cpu-baseline --no-extra-legs 2>>gpurun_out/r03b_bench.err | tail -1 > gpurun_out/r03b_bench_P64_nolag.json; cut -c1-300 gpurun_out/r03b_bench_P64_nolag.json
