cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FLOW_MODES=cols,flow_fused_pm python tools/gpu_flow_perf.py 2048x256 2048x320 2048x384 2048x448 2048x512 2048x640 4096x128 4096x192 1024x256 1024x512 2>&1 | grep -v amdgpu | tee gpurun_out/r03r_flow_perf.txt | tail -22
