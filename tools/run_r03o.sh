cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/r03o_pytest.log 2>&1; tail -4 gpurun_out/r03o_pytest.log
python tools/gpu_grad_perf.py 2>&1 | grep -v amdgpu > gpurun_out/r03o_grad_perf.txt; cat gpurun_out/r03o_grad_perf.txt
python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>gpurun_out/r03o.err | tail -1 > gpurun_out/r03o_bench.json; python - <<PY
import json
j=json.load(open("gpurun_out/r03o_bench.json"))
print(round(j["value"]), j["ms_per_step"], j["roofline"]["frac"], j["roofline_diag_kernel"]["frac"])
for k in ("grad","predict","roofline_cov_kernel","general_path"): print(k, json.dumps(j[k])[:600])
PY
(for T in 64 512; do tools/native/hmc_replay 2048 $T 2; done) 2>&1 | grep "^{" | cut -c1-330
