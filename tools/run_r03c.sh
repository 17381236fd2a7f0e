cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(time python -m pytest tests/test_gpu_lag.py "tests/test_gpu_parity.py::test_resampled_population_is_evaluated_once" "tests/test_gpu_parity.py::test_coalescing_of_single_particle_gradient_callers" "tests/test_gpu_configs.py::test_config3_n2048_P512_every_particle" -m gpu -x -q) > gpurun_out/r03c_pytest_lag.log 2>&1; tail -8 gpurun_out/r03c_pytest_lag.log
python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>gpurun_out/r03c_bench.err | tail -1 > gpurun_out/r03c_bench_n1.json; cut -c1-300 gpurun_out/r03c_bench_n1.json
python bench.py --particles 64 --no-cpu-baseline --no-extra-legs 2>>gpurun_out/r03c_bench.err | tail -1 > gpurun_out/r03c_bench_P64.json; cut -c1-300 gpurun_out/r03c_bench_P64.json
python tools/run_configs.py r03c 2>&1 | grep -v amdgpu | tail -8
AGP_LAG=0 python tools/run_configs.py r03c_nolag 2>&1 | grep -v amdgpu | tail -8
