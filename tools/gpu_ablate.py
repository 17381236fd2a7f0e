import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import os
import __graft_entry__ as g
g.build(experiments=True)                                  # measurement build: its own library file
os.environ["AUTOGP_HIP_LIB"] = str(g.LIB_EXP)
pkg = g.load_package()
eng = pkg.GPEngine(0)
P, nt = 512, 16
names = {0: "full", 1: "no global loads", 3: "no gload, no LDS store/barrier", 7: "MFMA only (no ds_read either)", 8: "full + setprio",
         16: "full, no epilogue", 19: "no gload/lds-store, no epilogue", 23: "MFMA only, no epilogue"}
variants = [int(v) for v in sys.argv[1:]] or [0, 1, 3, 7, 8, 16, 19, 23]
for k in (8, 4):
    T = nt - k - 1
    fl = P * T * 2 * 128 * 128 * k * 128
    for v in variants:
        ms = eng.debug_gemm_variant(P, nt, k, v, 5)
        print(f"k={k} T={T} variant {v:2d} {names.get(v, ''):40s} {ms:8.3f} ms  {fl/ms/1e9:6.1f} TF/s")
