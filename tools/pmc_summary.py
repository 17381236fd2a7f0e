"""Summarise the rocprofv3 PMC passes (tools/run_pmc.sh) per kernel: mean counter value per dispatch,
derived clock / MFMA utilisation / HBM bytes.  Writes profiles/<tag>_pmc_summary.txt and
profiles/hbm_traffic.json (bytes per launch for the dominant kernel, used by bench.py)."""
import json
import sys
from pathlib import Path

import pandas as pd

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = []
tabs = {}
for name in ("sq", "lds", "fetch", "write"):
    f = ROOT / f"gpurun_out/pmc_{name}/{name}_counter_collection.csv"
    if not f.exists():
        continue
    cc = pd.read_csv(f)
    cc["dur_us"] = (cc["End_Timestamp"] - cc["Start_Timestamp"]) / 1e3
    cc["kernel"] = cc["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "")
    piv = cc.pivot_table(index=["Dispatch_Id", "kernel", "dur_us"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
    tabs[name] = piv
    agg = piv.groupby("kernel").mean(numeric_only=True).drop(columns=["Dispatch_Id"])
    agg.insert(0, "calls", piv.groupby("kernel").size())
    out.append(f"## pass '{name}': mean per dispatch\n{agg.to_string(float_format=lambda v: f'{v:,.1f}')}\n")

res = {}
step_bytes = {}
if "sq" in tabs:
    t = tabs["sq"]; t = t[t.kernel.str.contains("agp::")]
    g = t.groupby("kernel").sum(numeric_only=True)
    d = pd.DataFrame(index=g.index)
    # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs
    d["clock_GHz"] = g["GRBM_GUI_ACTIVE"] / 8 / (g["dur_us"] * 1e3)
    d["mfma_busy_frac"] = g["SQ_VALU_MFMA_BUSY_CYCLES"] / (g["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
    d["wait_any/wave_cycles"] = g["SQ_WAIT_ANY"] / g["SQ_WAVE_CYCLES"]
    d["wait_inst/wave_cycles"] = g["SQ_WAIT_INST_ANY"] / g["SQ_WAVE_CYCLES"]
    d["active_inst/wave_cycles"] = g["SQ_ACTIVE_INST_ANY"] / g["SQ_WAVE_CYCLES"]
    out.append("## derived (sq pass, totals over all dispatches of the kernel)\n" + d.to_string(float_format=lambda v: f"{v:.3f}") + "\n")
for name, col in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    if name in tabs:
        t = tabs[name]; t = t[t.kernel.str.contains("agp::")]
        m = t.groupby("kernel")[col].mean()
        # FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1 KB (rocprofv3 derived metric); MI355X_MICROARCH.md §HBM:
        # on gfx950 FETCH_SIZE under-reports wide coalesced reads by exactly 2x -> corrected below.
        fac = 2.0 if name == "fetch" else 1.0
        out.append(f"## {col}: mean per dispatch [KB as reported], corrected bytes = value*1024*{fac}\n" +
                   "\n".join(f"{k:40s} {v:14.1f} KB  -> {v * 1024 * fac / 1e6:10.1f} MB" for k, v in m.items()) + "\n")
        for k, v in m.items():
            res.setdefault(k, {})[name] = v * 1024 * fac
        # whole-sweep traffic: every agp:: dispatch of the pass, per step (one k_finish_logpdf dispatch per step)
        n_steps = int(t.kernel.str.contains("k_finish_logpdf").sum())
        if n_steps > 0:
            step_bytes[name] = float(t[col].sum()) * 1024 * fac / n_steps
txt = "\n".join(out)
(ROOT / f"profiles/{tag}_pmc_summary.txt").write_text(txt)
print(txt)
# dominant kernel: the sub-diagonal-tile instantiation (last template argument 2) of the default build
import re
upd = [k for k in res if re.search(r"k_chol_update<true, \d, true, 2\b", k)] or [k for k in res if "k_chol_update<true" in k]
if upd and "fetch" in res[upd[0]] and "write" in res[upd[0]]:
    tot = res[upd[0]]["fetch"] + res[upd[0]]["write"]
    import datetime, hashlib, os
    lib = ROOT / "autogp.jl_amd" / "lib" / "libautogp_hip.so"
    (ROOT / "profiles/hbm_traffic.json").write_text(json.dumps({
        "tag": tag, "date": datetime.date.today().isoformat(),
        # which build the counters were taken on: the library's own hash (bench.py compares it with the library it runs) and, when the
        # evidence script was handed one (AGP_GIT_HEAD: the GPU box holds no .git), the commit
        "library_sha256_16": hashlib.sha256(lib.read_bytes()).hexdigest()[:16] if lib.exists() else None,
        "git_head": os.environ.get("AGP_GIT_HEAD"),
        "k_chol_update_bytes_per_launch": tot, "bytes_per_step": (step_bytes["fetch"] + step_bytes["write"]) if len(step_bytes) == 2 else None,
        "kernel": upd[0], "fetch_bytes_corrected_x2": res[upd[0]]["fetch"], "write_bytes": res[upd[0]]["write"],
        "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python bench.py --steps 2 --warmup 1`, {tag}; "
                  "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM"}))
    print("hbm bytes/launch", tot / 1e6, "MB")
