cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=$PWD/autogp.jl_amd/lib
(AUTOGP_HIP_LIB=$L/libautogp_hip_kpipe.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lag.py -m gpu -x -q) > gpurun_out/r03l_pytest_kpipe.log 2>&1; tail -5 gpurun_out/r03l_pytest_kpipe.log
run() { tag=$1; shift; env "$@" $B 2>>gpurun_out/r03l.err | tail -1 > gpurun_out/r03l_$tag.json; python - <<PY
import json
j=json.load(open("gpurun_out/r03l_$tag.json"))
print("$tag", round(j["value"]), round(j["ms_per_step"],3), {k:round(v,2) for k,v in j["phase_ms_per_step"].items() if k!="finish_ms"}, round(j["roofline"]["frac"],3), j["config"]["not_positive_definite"])
PY
}
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs"
run P512 X=1
run P512_kpipe AUTOGP_HIP_LIB=$L/libautogp_hip_kpipe.so
run P512_b X=1
run P512_kpipe_b AUTOGP_HIP_LIB=$L/libautogp_hip_kpipe.so
B="python bench.py --particles 64 --steps 200 --warmup 5 --no-cpu-baseline --no-extra-legs"
run P64 X=1
run P64_kpipe AUTOGP_HIP_LIB=$L/libautogp_hip_kpipe.so
B="python bench.py --particles 64 --n-obs 1024 --steps 400 --warmup 5 --no-cpu-baseline --no-extra-legs"
run c2 X=1
run c2_kpipe AUTOGP_HIP_LIB=$L/libautogp_hip_kpipe.so
