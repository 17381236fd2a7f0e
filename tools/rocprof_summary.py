"""Turn a rocprofv3 rocpd sqlite database (gpurun_out/...) into the text summary committed under
profiles/.  Usage: python tools/rocprof_summary.py <results.db> <out.txt> [--particles P --n N]"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db"); ap.add_argument("out")
    ap.add_argument("--particles", type=int, default=512); ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--cmd", default="")
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    lines = [f"# rocprofv3 --kernel-trace --stats summary ({a.db})", f"# command: {a.cmd}", ""]
    lines.append(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for name, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append(f"{name[:70]:70s} {calls:6d} {tot:12.1f} {avg:10.1f} {pct:6.2f}")
    rows = list(cur.execute("select name,start,duration,grid_x,vgpr_count,accum_vgpr_count,sgpr_count,lds_size from kernels order by start"))
    res = {}
    for r in rows:
        res.setdefault(r[0], r[4:])
    lines += ["", "# per-kernel resources (vgpr, agpr, sgpr, lds bytes)"]
    for k, v in res.items():
        lines.append(f"{k[:70]:70s} {v}")
    NB = 128; nt = (a.n + NB - 1) // NB; P = a.particles
    import re
    dia = [r for r in rows if re.search(r"k_chol_update<true, \d, true, 1\b", r[0])]
    upd = [r for r in rows if "k_chol_update<true" in r[0] and r not in dia]
    trs = [r for r in rows if "k_chol_trsm" in r[0]]
    split = len(dia) > 0
    per = (nt - 1) if split else nt
    if per > 0 and len(upd) >= per and len(upd) % per == 0:
        lines += ["", f"# last sweep, per block column k (P={P}, n={a.n}): "
                      + ("diagonal-tile launch, sub-diagonal-tile launch with its TF/s (2*128^2*(k*128) + 128^3 flop per tile, nt-k-1 tiles)"
                         if split else "update-kernel duration and GEMM TF/s (2*128^2*(k*128) flop per tile, (nt-k) tiles), trsm duration")]
        last = upd[-per:]; lt = trs[-(nt - 1):] if len(trs) >= nt - 1 else []
        ld = dia[-nt:] if split and len(dia) >= nt else []
        for k, r in enumerate(last):
            if split:
                fl = P * (nt - k - 1) * (2 * NB * NB * (k * NB) + NB ** 3)
                s = (f"k={k:2d} diag {ld[k][2] / 1e3:8.1f} us   sub-diag grid={r[3] // 256:6d} WGs {r[2] / 1e3:9.1f} us  "
                     f"{fl / (r[2] * 1e-9) / 1e12:6.1f} TF/s") if ld else f"k={k:2d} sub-diag {r[2] / 1e3:9.1f} us"
            else:
                fl = P * (nt - k) * 2 * NB * NB * (k * NB) + P * NB ** 3 / 3
                s = f"k={k:2d} grid={r[3] // 256:6d} WGs  update {r[2] / 1e3:9.1f} us  {fl / (r[2] * 1e-9) / 1e12:6.1f} TF/s"
                if k < len(lt):
                    s += f"   trsm {lt[k][2] / 1e3:8.1f} us"
            lines.append(s)
        if split and len(ld) == nt:
            lines.append(f"k={nt - 1:2d} diag {ld[nt - 1][2] / 1e3:8.1f} us")
    open(a.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
