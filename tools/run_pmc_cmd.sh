#!/bin/bash
# rocprofv3 PMC pass for an arbitrary command: tools/run_pmc_cmd.sh <name> "<counters...>" -- <cmd...>   (kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; counters=$2; shift 3
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $R/gpurun_out/pmc_$name -o $name -- "$@" > $R/gpurun_out/pmc_$name.log 2>&1
echo "$name rc=$?"
python - <<PY
import pandas as pd
cc = pd.read_csv("$R/gpurun_out/pmc_$name/${name}_counter_collection.csv")
cc["kernel"] = cc["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "")
cc["dur_us"] = (cc["End_Timestamp"] - cc["Start_Timestamp"]) / 1e3
piv = cc.pivot_table(index=["Dispatch_Id", "kernel", "dur_us"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
agg = piv.groupby("kernel").mean(numeric_only=True).drop(columns=["Dispatch_Id"])
agg.insert(0, "calls", piv.groupby("kernel").size())
pd.set_option("display.width", 250); pd.set_option("display.max_columns", 30)
print(agg[agg.index.str.contains("agp::")].to_string(float_format=lambda v: f"{v:,.0f}"))
PY
