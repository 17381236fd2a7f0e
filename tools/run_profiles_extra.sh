#!/bin/bash
# rocprofv3 kernel statistics of the paths beside the headline sweep: gradient sweep, predictive pass, extension sweeps,
# the streamed config 5.   usage (GPU box, repo root): bash tools/run_profiles_extra.sh <tag>   -> gpurun_out/<tag>_*_kernel_stats.txt
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
prof() { name=$1; shift; cd /tmp && export TMPDIR=/tmp; rm -rf $R/gpurun_out/px_$name
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/px_$name -o x -- "$@" > $R/gpurun_out/px_$name.log 2>&1
  cd $R; python tools/rocprof_summary.py $(find gpurun_out/px_$name -name "*.db" | head -1) gpurun_out/${TAG}_${name}_kernel_stats.txt --cmd "$*" | head -14; }
prof grad python $R/tools/gpu_grad_perf.py 2048x512
prof predict python $R/tools/gpu_predict_perf.py 2048:2048:128
prof extend python $R/tools/gpu_extend_profile.py 512
prof stream python $R/tools/run_stream.py --rejuvenate --predict
prof toeplitz python $R/tools/gpu_toeplitz_sweep.py
