cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=$PWD/autogp.jl_amd/lib
run() { tag=$1; shift; env "$@" $B 2>>gpurun_out/r03g.err | tail -1 > gpurun_out/r03g_$tag.json; python - <<PY
import json
j=json.load(open("gpurun_out/r03g_$tag.json"))
print("$tag", round(j["value"]), round(j["ms_per_step"],3), {k:round(v,2) for k,v in j["phase_ms_per_step"].items() if k!="finish_ms"}, round(j["roofline"]["frac"],3))
PY
}
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs"
run base X=1
run relwg AUTOGP_HIP_LIB=$L/libautogp_hip_relwg.so
run nofactor AUTOGP_HIP_LIB=$L/libautogp_hip_nofactor.so
run flow AGP_FLOW=1
run flow_relwg AGP_FLOW=1 AUTOGP_HIP_LIB=$L/libautogp_hip_relwg.so
run flow_chain AGP_FLOW=1 AUTOGP_HIP_LIB=$L/libautogp_hip_chain.so
B="python bench.py --particles 64 --steps 200 --warmup 5 --no-cpu-baseline --no-extra-legs"
run P64 X=1
run P64_relwg AUTOGP_HIP_LIB=$L/libautogp_hip_relwg.so
run P64_chain AUTOGP_HIP_LIB=$L/libautogp_hip_chain.so
run P64_chainrel AUTOGP_HIP_LIB=$L/libautogp_hip_chainrel.so
B="python bench.py --particles 64 --n-obs 1024 --steps 400 --warmup 5 --no-cpu-baseline --no-extra-legs"
run c2 X=1
run c2_relwg AUTOGP_HIP_LIB=$L/libautogp_hip_relwg.so
run c2_chain AUTOGP_HIP_LIB=$L/libautogp_hip_chain.so
run c2_chainrel AUTOGP_HIP_LIB=$L/libautogp_hip_chainrel.so
AUTOGP_HIP_LIB=$L/libautogp_hip_nofactor.so python tools/gpu_launch_times.py 2048 512 2>&1 | grep -v amdgpu | tail -3
