#!/bin/bash
# A/B runs of bench.py on the GPU box, one line of output per variant:
#   bash tools/run_ab.sh <tag> "<bench args>" "name1 ENV=.. ENV=.." "name2 ..." ...
# e.g. bash tools/run_ab.sh r03x "--particles 64 --steps 200" "base X=1" "nolag AGP_LAG=0" "lib AUTOGP_HIP_LIB=\$PWD/autogp.jl_amd/lib/libautogp_hip_x.so"
# (variant libraries: hipcc ... -D<SWITCH> -o autogp.jl_amd/lib/libautogp_hip_<x>.so, selected through AUTOGP_HIP_LIB)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
TAG=$1; ARGS=$2; shift 2
for v in "$@"; do
  name=${v%% *}; envs=${v#* }
  env $envs python bench.py $ARGS --warmup 5 --no-cpu-baseline --no-extra-legs 2>>gpurun_out/${TAG}.err | tail -1 > gpurun_out/${TAG}_$name.json
  python - <<PY
import json
j = json.load(open("gpurun_out/${TAG}_$name.json"))
print("$name", round(j["value"]), round(j["ms_per_step"], 3), {k: round(v, 2) for k, v in j["phase_ms_per_step"].items() if k != "finish_ms"},
      round(j["roofline"]["frac"], 3), j["config"]["not_positive_definite"])
PY
done
