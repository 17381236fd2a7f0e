"""GEMM-loop harness: the production strip loop (4 waves, 16 accumulator blocks per wave) against the 8-wave variant
(column halves, 8 blocks per wave, 4 waves per SIMD).  Needs the experiments build (python __graft_entry__.py --experiments)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import os
import __graft_entry__ as g
g.build(experiments=True)                                  # measurement build: its own library file
os.environ["AUTOGP_HIP_LIB"] = str(g.LIB_EXP)
pkg = g.load_package()
eng = pkg.GPEngine(0)
nt = 16
for P in (512, 64):
    fl = sum(P * (nt - k - 1) * 2 * 128 * 128 * k * 128 for k in range(1, nt - 1))
    for v, name in ((2001, "4 waves, launch per column"), (3001, "8 waves, launch per column"), (3032, "8 waves, 32-column slabs"),
                    (2000, "4 waves, one launch"), (3000, "8 waves, one launch")):
        ms = min(eng.debug_gemm_variant(P, nt, 1, v, 5) for _ in range(2))
        print(f"P={P} {name:32s} {ms:8.3f} ms  {fl/ms/1e9:6.1f} TF/s", flush=True)
