cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_lag.py tests/test_gpu_parity.py -m gpu -x -q) > gpurun_out/r03t_pytest.log 2>&1; tail -3 gpurun_out/r03t_pytest.log
run() { tag=$1; shift; env "$@" $B 2>>gpurun_out/r03t.err | tail -1 > gpurun_out/r03t_$tag.json; python - <<PY
import json
j=json.load(open("gpurun_out/r03t_$tag.json"))
print("$tag", round(j["value"]), round(j["ms_per_step"],3), {k:round(v,2) for k,v in j["phase_ms_per_step"].items() if k!="finish_ms"}, round(j["roofline"]["frac"],3), j["config"]["not_positive_definite"])
PY
}
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs"
run P512 X=1
run P512_b X=1
B="python bench.py --particles 64 --steps 200 --warmup 5 --no-cpu-baseline --no-extra-legs"
run P64 X=1
run P64_f30 AGP_FLOW_FUSE_MAX_US=30
run P64_f15 AGP_FLOW_FUSE_MAX_US=15
run P64_f8 AGP_FLOW_FUSE_MAX_US=8
B="python bench.py --particles 64 --n-obs 1024 --steps 400 --warmup 5 --no-cpu-baseline --no-extra-legs"
run c2 X=1
python tools/gpu_launch_times.py 2048 512 2>&1 | grep -v amdgpu | tail -2
python tools/gpu_flow_trace.py 2048 64 2>&1 | grep -v amdgpu > gpurun_out/r03t_trace.txt; tail -8 gpurun_out/r03t_trace.txt
