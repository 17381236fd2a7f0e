cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/gpu_launch_times.py 2048 512 2>&1 | grep -v amdgpu > gpurun_out/r03f_launch_times.txt; cat gpurun_out/r03f_launch_times.txt
python tools/gpu_flow_trace.py 2048 64 2>&1 | grep -v amdgpu > gpurun_out/r03f_flow_trace_P64.txt; tail -22 gpurun_out/r03f_flow_trace_P64.txt
python tools/gpu_flow_trace.py 2048 512 2>&1 | grep -v amdgpu > gpurun_out/r03f_flow_trace_P512.txt; tail -22 gpurun_out/r03f_flow_trace_P512.txt
python tools/gpu_flow_trace.py 1024 64 2>&1 | grep -v amdgpu > gpurun_out/r03f_flow_trace_c2.txt; tail -14 gpurun_out/r03f_flow_trace_c2.txt
