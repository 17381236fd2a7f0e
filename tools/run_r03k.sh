cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=$PWD/autogp.jl_amd/lib
run() { tag=$1; shift; env "$@" $B 2>>gpurun_out/r03k.err | tail -1 > gpurun_out/r03k_$tag.json; python - <<PY
import json
j=json.load(open("gpurun_out/r03k_$tag.json"))
print("$tag", round(j["value"]), round(j["ms_per_step"],3), {k:round(v,2) for k,v in j["phase_ms_per_step"].items() if k!="finish_ms"}, round(j["roofline"]["frac"],3), j["config"]["not_positive_definite"])
PY
}
B="python bench.py --particles 64 --steps 200 --warmup 5 --no-cpu-baseline --no-extra-legs"
run P64 X=1
run P64_pp2 AUTOGP_HIP_LIB=$L/libautogp_hip_pp2.so
run P64_pp3 AUTOGP_HIP_LIB=$L/libautogp_hip_pp3.so
B="python bench.py --particles 64 --n-obs 1024 --steps 400 --warmup 5 --no-cpu-baseline --no-extra-legs"
run c2 X=1
run c2_pp2 AUTOGP_HIP_LIB=$L/libautogp_hip_pp2.so
run c2_pp3 AUTOGP_HIP_LIB=$L/libautogp_hip_pp3.so
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs"
run P512 X=1
run P512_pp2 AUTOGP_HIP_LIB=$L/libautogp_hip_pp2.so
run P512_pp3 AUTOGP_HIP_LIB=$L/libautogp_hip_pp3.so
run P512_flow_pp3 AGP_FLOW=1 AUTOGP_HIP_LIB=$L/libautogp_hip_pp3.so
AUTOGP_HIP_LIB=$L/libautogp_hip_pp3.so python tools/gpu_flow_trace.py 2048 64 2>&1 | grep -v amdgpu > gpurun_out/r03k_flow_trace_P64_pp3.txt; tail -9 gpurun_out/r03k_flow_trace_P64_pp3.txt
