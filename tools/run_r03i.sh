cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(time python -m pytest tests/test_gpu_parity.py tests/test_gpu_lag.py -m gpu -x -q) > gpurun_out/r03i_pytest.log 2>&1; tail -6 gpurun_out/r03i_pytest.log
run() { tag=$1; shift; env "$@" $B 2>>gpurun_out/r03i.err | tail -1 > gpurun_out/r03i_$tag.json; python - <<PY
import json
j=json.load(open("gpurun_out/r03i_$tag.json"))
print("$tag", round(j["value"]), round(j["ms_per_step"],3), {k:round(v,2) for k,v in j["phase_ms_per_step"].items() if k!="finish_ms"}, round(j["roofline"]["frac"],3), j["config"]["not_positive_definite"])
PY
}
B="python bench.py --particles 64 --steps 200 --warmup 5 --no-cpu-baseline --no-extra-legs"
run P64 X=1
run P64_off AGP_FLOW_PART_TB=0
run P64_tb3 AGP_FLOW_PART_TB=3
run P64_tb6 AGP_FLOW_PART_TB=6 AGP_FLOW_PART_CH=5
run P64_tb8 AGP_FLOW_PART_TB=8
run P64_ch2 AGP_FLOW_PART_CH=2
run P64_ch6 AGP_FLOW_PART_CH=6
run P64_ch12 AGP_FLOW_PART_CH=12
B="python bench.py --particles 64 --n-obs 1024 --steps 400 --warmup 5 --no-cpu-baseline --no-extra-legs"
run c2 X=1
run c2_off AGP_FLOW_PART_TB=0
run c2_tb2 AGP_FLOW_PART_TB=2 AGP_FLOW_PART_CH=3
run c2_tb4ch2 AGP_FLOW_PART_TB=4 AGP_FLOW_PART_CH=2
B="python bench.py --particles 128 --n-obs 4096 --steps 20 --warmup 2 --no-cpu-baseline --no-extra-legs"
run c4like X=1
run c4like_off AGP_FLOW_PART_TB=0
run c4like_tb6ch8 AGP_FLOW_PART_TB=6 AGP_FLOW_PART_CH=8
python tools/gpu_flow_trace.py 2048 64 2>&1 | grep -v amdgpu > gpurun_out/r03i_flow_trace_P64.txt; head -24 gpurun_out/r03i_flow_trace_P64.txt | tail -8; tail -5 gpurun_out/r03i_flow_trace_P64.txt
