#!/bin/bash
# rocprofv3 PMC passes for the bench workload (each counter group in its own run, kernel-trace only).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs"
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$name -o $name -- $CMD > $R/gpurun_out/pmc_$name.log 2>&1; echo "$name rc=$?"; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE
ls -R $R/gpurun_out/pmc_* | head -40
