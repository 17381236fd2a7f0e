cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
AUTOGP_HIP_LIB=$PWD/autogp.jl_amd/lib/libautogp_hip_probe.so python tools/gpu_flow_trace.py 2048 64 2>&1 | grep -v amdgpu > gpurun_out/r03s_trace_probe.txt; tail -8 gpurun_out/r03s_trace_probe.txt
