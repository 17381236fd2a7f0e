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


KERNEL_UNITS = ("agp_kernels.hip", "agp_kernels_flow.hip", "agp_kernels_grad.hip")
# scratch (spilled registers) budget in bytes per kernel: the K-loops of the large-population kernels must not spill at all; the
# dataflow kernel carries both tile bodies and a few hoisted constants (88 - 96 B); a rewrite of the diagonal K-loop once took it
# to 376 B — and the 64-particle share from 3.8 to 4.1 ms — without a single test noticing
SCRATCH_BUDGET = {"k_chol_updateILb1ELi4ELb1ELi2ELi2E": 0, "k_chol_updateILb1ELi4ELb1ELi2ELi1E": 0, "k_chol_updateILb1ELi4ELb1ELi2ELi0E": 0,
                  "k_chol_diagILi4ELi2E": 64, "k_chol_flowILi4ELi2E": 128, "k_chol_flowILi4ELi1E": 160, "k_chol_flowILi4ELi0E": 128}


def scratch_bytes(asm_text):
    out = {}
    cur = None
    for l in asm_text.split("\n"):
        m = re.match(r"^\s+\.amdhsa_kernel (_Z\w+)", l)
        if m:
            cur = m.group(1)
        m = re.match(r"^\s+\.amdhsa_private_segment_fixed_size (\d+)", l)
        if m and cur:
            out[cur] = int(m.group(1)); cur = None
    return out


def main():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    bad, over, n_kernels = {}, [], 0
    with tempfile.TemporaryDirectory() as td:
        jobs = []
        for u in KERNEL_UNITS:
            out = Path(td) / (u + ".s")
            jobs.append((out, subprocess.Popen([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", str(out),
                                                str(ROOT / "autogp.jl_amd" / "csrc" / u)], stderr=subprocess.DEVNULL)))
        for out, pr in jobs:
            if pr.wait() != 0:
                print("compilation failed:", out.name); return 1
            txt = out.read_text()
            bad.update(waterfalled_accesses(txt))
            sb = scratch_bytes(txt)
            n_kernels += len(sb)
            for k, v in sb.items():
                for frag, lim in SCRATCH_BUDGET.items():
                    if frag in k and v > lim:
                        over.append(f"{k}: {v} B of scratch (budget {lim})")
    for k, v in bad.items():
        print(f"{v} buffer accesses inside waterfall loops: {k}")
    for o in over:
        print("register spills over budget:", o)
    if bad or over:
        return 1
    print(f"no buffer access inside a waterfall loop, spills within budget ({n_kernels} kernels of {', '.join(KERNEL_UNITS)})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
