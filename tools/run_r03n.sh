cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=$PWD/autogp.jl_amd/lib
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lag.py -m gpu -x -q) > gpurun_out/r03n_pytest.log 2>&1; tail -3 gpurun_out/r03n_pytest.log
run() { tag=$1; shift; env "$@" $B 2>>gpurun_out/r03n.err | tail -1 > gpurun_out/r03n_$tag.json; python - <<PY
import json
j=json.load(open("gpurun_out/r03n_$tag.json"))
print("$tag", round(j["value"]), round(j["ms_per_step"],3), {k:round(v,2) for k,v in j["phase_ms_per_step"].items() if k!="finish_ms"}, round(j["roofline"]["frac"],3), j["config"]["not_positive_definite"])
PY
}
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs"
run P512 X=1
run P512_dsel AUTOGP_HIP_LIB=$L/libautogp_hip_dsel.so
run P512_b X=1
run P512_dsel_b AUTOGP_HIP_LIB=$L/libautogp_hip_dsel.so
B="python bench.py --particles 64 --steps 200 --warmup 5 --no-cpu-baseline --no-extra-legs"
run P64 X=1
run P64_dsel AUTOGP_HIP_LIB=$L/libautogp_hip_dsel.so
python tools/gpu_launch_times.py 2048 512 2>&1 | grep -v amdgpu | tail -2
AUTOGP_HIP_LIB=$L/libautogp_hip_dsel.so python tools/gpu_launch_times.py 2048 512 2>&1 | grep -v amdgpu | tail -2
