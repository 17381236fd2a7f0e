"""GEMM-loop harness: the production strip loop (128 x 128 items, two workgroups per CU) against 256 x 128 macro-items (two row
tiles per workgroup sharing the column operand's LDS slab, one workgroup per CU).  Needs the experiments build."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import os
import __graft_entry__ as g
g.build(experiments=True)
os.environ["AUTOGP_HIP_LIB"] = str(g.LIB_EXP)
pkg = g.load_package()
eng = pkg.GPEngine(0)
nt = 16
for P in (512, 64):
    fl = sum(P * (nt - k - 1) * 2 * 128 * 128 * k * 128 for k in range(1, nt - 1))
    for v, name in ((2000, "128x128 items, one launch"), (4000, "256x128 macro-items, 16-col slabs"), (4032, "256x128 macro-items, 32-col slabs"),
                    (4100, "256x128 macro-items, 8 waves")):
        ms = min(eng.debug_gemm_variant(P, nt, 1, v, 5) for _ in range(3))
        print(f"P={P} {name:36s} {ms:8.3f} ms  {fl/ms/1e9:6.1f} TF/s", flush=True)
