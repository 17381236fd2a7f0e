cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r03a_pytest_gpu.log 2>&1; tail -15 gpurun_out/r03a_pytest_gpu.log
python bench.py --steps 20 --warmup 5 2>gpurun_out/r03a_bench.err | tail -1 > gpurun_out/r03a_bench_n1.json; cut -c1-400 gpurun_out/r03a_bench_n1.json
AGP_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 20 --warmup 2 2>gpurun_out/r03a_bench2.err | tail -1 > gpurun_out/r03a_bench_n2_shared.json; cut -c1-300 gpurun_out/r03a_bench_n2_shared.json; tail -3 gpurun_out/r03a_bench2.err
python bench.py --single-process --gpus 1 --particles 64 --steps 50 2>&1 | tail -1 | cut -c1-600
