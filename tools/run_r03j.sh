cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AGP_FLOW_PART_TB=0 python tools/gpu_flow_trace.py 2048 64 2>&1 | grep -v amdgpu > gpurun_out/r03j_flow_trace_P64.txt; tail -12 gpurun_out/r03j_flow_trace_P64.txt
AGP_FLOW_PART_TB=0 python tools/gpu_flow_trace.py 1024 64 2>&1 | grep -v amdgpu > gpurun_out/r03j_flow_trace_c2.txt; tail -9 gpurun_out/r03j_flow_trace_c2.txt
AGP_FLOW_PART_TB=0 python tools/gpu_flow_trace.py 2048 512 2>&1 | grep -v amdgpu > gpurun_out/r03j_flow_trace_P512.txt; tail -9 gpurun_out/r03j_flow_trace_P512.txt
