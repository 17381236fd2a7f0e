#!/bin/bash
# evidence pass: gpu tests, smoke, bench (with cpu baseline), rocprofv3 kernel stats, PMC passes, configs, schedules
# usage (on the GPU box, from the repo root): bash tools/run_evidence.sh <tag>      -> gpurun_out/<tag>_*
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
(time python -m pytest tests -m gpu -q) > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/${TAG}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py 2>gpurun_out/bench.err | tail -1 > gpurun_out/${TAG}_bench_n1.json; cut -c1-300 gpurun_out/${TAG}_bench_n1.json
python bench.py --particles 64 --no-cpu-baseline --no-extra-legs 2>>gpurun_out/bench.err | tail -1 > gpurun_out/${TAG}_bench_n1_P64_rank_share.json
python tools/run_configs.py ${TAG} 2>&1 | grep -v amdgpu | tail -8
FLOW_MODES=cols,flow python tools/gpu_flow_perf.py 2048x8 2048x16 2048x32 2048x64 2048x128 2048x256 2048x384 2048x512 1024x64 1024x128 512x256 4096x32 4096x128 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_flow_perf.txt; tail -3 gpurun_out/${TAG}_flow_perf.txt
python tools/gpu_extend_perf.py ${TAG} 2>&1 | grep -v amdgpu | tail -4
python tools/gpu_grad_perf.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_grad_perf.txt; cat gpurun_out/${TAG}_grad_perf.txt
# opt-in structured value sweep (Schur algorithm) against the dense sweep: bench population, config 4's shape, kernel statistics
(python tools/gpu_toeplitz_sweep.py; python tools/gpu_toeplitz_sweep.py 4096 128) 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_toeplitz_sweep.txt; cat gpurun_out/${TAG}_toeplitz_sweep.txt
# gradient sweep: lag-domain contraction on / off (spans of the sweep's kernels), then with the histogram instead of the spectra
(python tools/gpu_grad_lagdom_ab.py; AGP_GRAD_FFT=0 python tools/gpu_grad_lagdom_ab.py | head -1) 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_grad_lagdom_ab.txt; head -2 gpurun_out/${TAG}_grad_lagdom_ab.txt
python tools/gpu_grad_lagdom_check.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_grad_lagdom_check.txt; tail -1 gpurun_out/${TAG}_grad_lagdom_check.txt
# sources of the lag sums: Toeplitz solves / spectra of L^-T / element-wise, accuracy against the element-wise contraction; phases of the element-wise sweep
(python tools/gpu_grad_toeplitz_check.py; python tools/gpu_grad_phases.py) 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_grad_toeplitz_check.txt; cat gpurun_out/${TAG}_grad_toeplitz_check.txt
(python tools/run_stream.py --rejuvenate; python tools/run_stream.py --rejuvenate --no-extend; python tools/run_stream.py --rejuvenate --predict; AGP_PREDICT_REUSE=0 python tools/run_stream.py --rejuvenate --predict; python tools/run_stream.py --rejuvenate --predict --time-order) 2>&1 | grep -v amdgpu | grep "^{" > gpurun_out/${TAG}_stream.jsonl; cut -c1-200 gpurun_out/${TAG}_stream.jsonl
python tools/gpu_scratch_via_store.py 2>&1 | grep "^n=" > gpurun_out/${TAG}_store_scratch.txt; cat gpurun_out/${TAG}_store_scratch.txt
python tools/gpu_calendar_bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_calendar.json; cut -c1-300 gpurun_out/${TAG}_calendar.json
(for T in 64 512; do tools/native/hmc_replay 2048 $T 2; tools/native/hmc_replay 2048 $T 2 10 0.02 grid; AGP_LAG=2 tools/native/hmc_replay 2048 $T 2 10 0.02 grid; tools/native/hmc_replay 2048 $T 2 10 0.02 monthly; AGP_FACTOR_CACHE=0 tools/native/hmc_replay 2048 $T 2; done; tools/native/hmc_replay 2048 192 2 10 0.02 grid; AGP_LAG=2 tools/native/hmc_replay 2048 192 2 10 0.02 grid; tools/native/hmc_replay 512 256 4; tools/native/threads_bench 2048 512 8; tools/native/threads_bench 2048 64 8; tools/native/threads_bench 2048 512 4 grad) 2>&1 | grep "^{" > gpurun_out/${TAG}_native.jsonl; cut -c1-220 gpurun_out/${TAG}_native.jsonl
# round 6: the factor store sizing itself (ragged arrivals, no reservation; AGP_STORE_SELF_SIZE=0 = round 5's rule), dependent-issue latencies
(for T in 512 1024; do for S in 1 0; do AGP_STORE_SELF_SIZE=$S HMC_JITTER_US=20000 HMC_WINDOW_US=5 tools/native/hmc_replay 512 $T 2; done; done; for R in 0 1; do HMC_JITTER_US=3000 HMC_WINDOW_US=20 HMC_RESERVE=$R tools/native/hmc_replay 2048 512 2; done) 2>&1 | grep "^{" > gpurun_out/${TAG}_store_selfsize.jsonl; cut -c1-200 gpurun_out/${TAG}_store_selfsize.jsonl
tools/native/lat_bench > gpurun_out/${TAG}_lat_bench.json 2>&1; cut -c1-300 gpurun_out/${TAG}_lat_bench.json
# round 6: the reference's tutorial sizes (135-442 points, 8-64 caller threads): launch-bound sweeps
(for N in 144 443; do for T in 8 16 64; do tools/native/hmc_replay $N $T 20; tools/native/hmc_replay $N $T 20 10 0.02 grid; tools/native/hmc_replay $N $T 20 10 0.02 monthly; done; done; tools/native/threads_bench 144 8 400; tools/native/threads_bench 144 8 400 grad; tools/native/threads_bench 144 64 100; tools/native/threads_bench 144 64 100 grad; tools/native/threads_bench 144 1 400; tools/native/threads_bench 144 1 400 grad) 2>&1 | grep "^{" > gpurun_out/${TAG}_small_native.jsonl; cut -c1-200 gpurun_out/${TAG}_small_native.jsonl | head -3
# stand-alone diagonal-tile harness (per-phase clocks), predictive passes, extension sweeps alone, dataflow traces (measurement library)
(tools/native/diag_bench 8 0 20; tools/native/diag_bench 512 0 20; tools/native/diag_bench 512 8 20; tools/native/diag_bench 64 8 20) > gpurun_out/${TAG}_diag_bench.txt 2>&1; tail -1 gpurun_out/${TAG}_diag_bench.txt | cut -c1-160
(python tools/gpu_predict_perf.py; python tools/gpu_predict_perf.py 2048:2048:128 --off-lattice) 2>&1 | grep "^predict" > gpurun_out/${TAG}_predict_perf.txt; tail -2 gpurun_out/${TAG}_predict_perf.txt
python tools/gpu_launch_times.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_launch_times_P512.txt; tail -2 gpurun_out/${TAG}_launch_times_P512.txt
if [ -f autogp.jl_amd/lib/libautogp_hip_exp.so ]; then
  python tools/gpu_flow_trace.py 1024 64 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_flow_trace_config2.txt; python tools/gpu_flow_trace.py 2048 64 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_flow_trace_P64.txt; grep "^diag items\|^sub items" gpurun_out/${TAG}_flow_trace_config2.txt
fi
# randomised soak runs (short versions): sizes x schedules x store; structured sweeps forced against dense / element-wise engines; the factor store's life cycle; 32 host threads
(python tools/gpu_fuzz.py 100 1 big; python tools/gpu_fuzz_structured.py 300 1; python tools/gpu_fuzz_stream.py 150 1; python tools/gpu_fuzz_pairs.py 100 1; python tools/gpu_stress_threads.py 10) 2>&1 | grep -E "fuzz ok|^stress|Error|assert" | cut -c1-900 > gpurun_out/${TAG}_soak.txt; cut -c1-200 gpurun_out/${TAG}_soak.txt
bash tools/run_profiles_extra.sh ${TAG} 2>&1 | grep -E "k_chol|k_trtri|k_zspec|k_grad" | head -12
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/prof.log 2>&1
cd $R
python tools/rocprof_summary.py $(find gpurun_out/prof -name "*.db" | head -1) gpurun_out/${TAG}_bench_kernel_stats.txt --cmd "python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs" | head -12
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof64 -o bench -- python $R/bench.py --particles 64 --steps 20 --warmup 2 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/prof64.log 2>&1
cd $R
python tools/rocprof_summary.py $(find gpurun_out/prof64 -name "*.db" | head -1) gpurun_out/${TAG}_bench_P64_kernel_stats.txt --particles 64 --cmd "python bench.py --particles 64 --steps 20 --warmup 2 --no-cpu-baseline --no-extra-legs" | head -8
bash tools/run_pmc.sh 2>&1 | tail -5
# the element-wise gradient sweep (what an irregular series pays): kernel statistics and the vector / matrix pipes' activity
bash tools/prof_cmd.sh ${TAG}_grad_elementwise python $R/tools/gpu_grad_phases.py | head -12
(bash tools/run_pmc_cmd.sh gradsq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" -- python $R/tools/gpu_grad_phases.py) > gpurun_out/${TAG}_grad_pmc.txt 2>&1; cd $R; grep -E "k_trtri|k_kinv|k_grad_contract|Counter_Name|kernel" gpurun_out/${TAG}_grad_pmc.txt | cut -c1-260 | head -8
# the 64-particle share (dataflow kernel): MFMA busy, HBM bytes
(bash tools/run_pmc_cmd.sh sq64 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" -- python $R/bench.py --particles 64 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs; bash tools/run_pmc_cmd.sh fetch64 "FETCH_SIZE" -- python $R/bench.py --particles 64 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs; bash tools/run_pmc_cmd.sh write64 "WRITE_SIZE" -- python $R/bench.py --particles 64 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs) 2>&1 | grep -v "at::native\|rocclr" > gpurun_out/${TAG}_pmc_P64.txt; cd $R; head -12 gpurun_out/${TAG}_pmc_P64.txt | cut -c1-250
python tools/pmc_summary.py ${TAG} > /dev/null 2>&1; cp profiles/${TAG}_pmc_summary.txt gpurun_out/ 2>/dev/null; cp profiles/hbm_traffic.json gpurun_out/${TAG}_hbm_traffic.json 2>/dev/null
ls gpurun_out | grep ${TAG}
