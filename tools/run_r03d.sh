cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>gpurun_out/r03d_bench.err | tail -1 > gpurun_out/r03d_bench_n1.json; cut -c1-200 gpurun_out/r03d_bench_n1.json
for L in 10 15 40; do AGP_FUSE_MAX_US=$L python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs 2>>gpurun_out/r03d_bench.err | tail -1 > gpurun_out/r03d_bench_n1_fuse$L.json; cut -c1-200 gpurun_out/r03d_bench_n1_fuse$L.json; done
python bench.py --particles 64 --no-cpu-baseline --no-extra-legs 2>>gpurun_out/r03d_bench.err | tail -1 > gpurun_out/r03d_bench_P64.json; cut -c1-200 gpurun_out/r03d_bench_P64.json
python tools/run_configs.py r03d 2>&1 | grep -v amdgpu | tail -8
