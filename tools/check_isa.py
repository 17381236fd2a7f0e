"""Static check of the compiled kernels (gfx950 ISA): no buffer load / store may sit in a "waterfall" loop — the
readfirstlane / compare / s_and_saveexec loop the compiler wraps around a buffer access whose resource descriptor it takes for
lane-dependent.  The operand streams of every K-loop are raw buffer loads with the descriptor in SGPRs; when an index the
descriptor is built from stops being provably wave-uniform the kernels stay correct and silently lose ~10 % (seen once: the
sub-diagonal kernel's slab loop, 31.1 -> 34.1 us per block column).   python tools/check_isa.py  -> exit status 0 / 1"""
import re, shutil, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent


def waterfalled_accesses(asm_text):
    bad = {}
    cur = None
    lines = asm_text.split("\n")
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
        if cur and ("buffer_load" in l or "buffer_store" in l) and any("s_and_saveexec" in x for x in lines[max(0, i - 3):i]):
            bad[cur] = bad.get(cur, 0) + 1
        if "s_endpgm" in l:
            cur = None
    return bad


def main():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "engine.s"
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", str(out),
                        str(ROOT / "autogp.jl_amd" / "csrc" / "agp_engine.hip")], check=True, stderr=subprocess.DEVNULL)
        bad = waterfalled_accesses(out.read_text())
    n_kernels = "all"
    if bad:
        for k, v in bad.items():
            print(f"{v} buffer accesses inside waterfall loops: {k}")
        return 1
    print(f"no buffer access inside a waterfall loop ({n_kernels} kernels of agp_engine.hip)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
