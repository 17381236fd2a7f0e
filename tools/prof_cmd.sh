#!/bin/bash
# rocprofv3 kernel statistics of one command.  usage (GPU box, repo root): bash tools/prof_cmd.sh <name> <cmd...>  -> gpurun_out/<name>_kernel_stats.txt
# PROF_TRACE=<substring>: also print the durations (us) of the LAST sweep's dispatches of the kernels whose name contains it, in launch order
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
cd /tmp && export TMPDIR=/tmp; rm -rf $R/gpurun_out/px_$name
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/px_$name -o x -- "$@" > $R/gpurun_out/px_$name.log 2>&1
cd $R; DB=$(find gpurun_out/px_$name -name "*.db" | head -1)
python tools/rocprof_summary.py $DB gpurun_out/${name}_kernel_stats.txt --cmd "$*" | head -${PROF_LINES:-22}
if [ -n "$PROF_TRACE" ]; then python - "$DB" "$PROF_TRACE" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name,start,duration,grid_x from kernels order by start"))
sel = [r for r in rows if sys.argv[2] in r[0]]
per = {}
for r in sel: per.setdefault(r[0], []).append(r)
for k, v in per.items():
    n = len(v) // 4 if len(v) >= 4 else len(v)        # (the tools run 1 warm-up + 3 timed sweeps)
    last = v[-n:]
    print(k[:60], "last sweep:", [round(r[2] / 1e3, 1) for r in last], "grid", [r[3] // 256 for r in last][:3], "...")
PY
fi
rm -rf $R/gpurun_out/px_$name
