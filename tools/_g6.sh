cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r06e_pytest.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r06e_pytest.log; grep -B40 "short test summary" gpurun_out/r06e_pytest.log | head -80
