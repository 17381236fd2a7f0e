"""Do fp64 VALU FMAs and fp64 MFMAs share execution resources on gfx950?  k_mfma_peak with MFMAs only, FMAs only, both in the
same wave (independent instruction streams), and split across the waves of a workgroup.  If the times of the first two ADD
UP in the third, the covariance evaluation (fp64 VALU) can never hide under the factorisation's MFMAs."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
eng = pkg.GPEngine(0)
iters = 20000
mf = 16 * 2048.0; vf = 128 * 64 * 2.0          # flops per wave and iteration: MFMAs / FMAs
for wg in (1, 2):
    res = {}
    for mode, name in ((0, "MFMA only"), (1, "VALU fp64 FMA only"), (2, "both, same wave"), (3, "waves 0-1 MFMA, waves 2-3 FMA"),
                       (4, "workgroups alternate roles (b & 1)"), (5, "workgroups by halves (b >> 8) & 1")):
        tf, ghz = eng.debug_mfma_peak(iters, wg | (mode << 8))
        # the entry reports TF/s assuming 16 MFMAs per wave and iteration: turn it back into time per iteration
        t_iter = (4 * 16 * 2048.0) / (tf * 1e12 / (256 * wg)) * 1e9      # ns per iteration per CU-resident workgroup
        res[mode] = t_iter
        print(f"{wg} WG/CU  {name:32s} {t_iter:8.1f} ns / iteration   ({ghz:.2f} GHz)", flush=True)
    print(f"   sum of the first two {res[0] + res[1]:.1f} ns, max {max(res[0], res[1]):.1f} ns, both in one wave {res[2]:.1f} ns")
