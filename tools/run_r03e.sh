cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs"
run() { tag=$1; shift; env "$@" $B 2>>gpurun_out/r03e.err | tail -1 > gpurun_out/r03e_$tag.json; python - <<PY
import json
j=json.load(open("gpurun_out/r03e_$tag.json"))
print("$tag", round(j["value"]), round(j["ms_per_step"],3), {k:round(v,2) for k,v in j["phase_ms_per_step"].items() if k!="finish_ms"}, round(j["roofline"]["frac"],3))
PY
}
run base X=1
run asym AUTOGP_HIP_LIB=$PWD/autogp.jl_amd/lib/libautogp_hip_asym.so
run fuse4 AGP_FUSE_MAX_US=4
run fuse6 AGP_FUSE_MAX_US=6
run fuse8 AGP_FUSE_MAX_US=8
run fuse10 AGP_FUSE_MAX_US=10
run flow AGP_FLOW=1 AGP_FUSE_MAX_US=10
run streams2 AGP_STREAMS=2 AGP_FUSE_MAX_US=10
run asym_fuse8 AUTOGP_HIP_LIB=$PWD/autogp.jl_amd/lib/libautogp_hip_asym.so AGP_FUSE_MAX_US=8
B="python bench.py --particles 64 --steps 200 --warmup 5 --no-cpu-baseline --no-extra-legs"
run P64 X=1
run P64_asym AUTOGP_HIP_LIB=$PWD/autogp.jl_amd/lib/libautogp_hip_asym.so
run P64_fuse30 AGP_FLOW_FUSE_MAX_US=30
run P64_fuse15 AGP_FLOW_FUSE_MAX_US=15
