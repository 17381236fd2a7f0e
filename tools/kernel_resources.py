"""Per-kernel register / spill / LDS report of the engine (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py [filter-substring] [extra hipcc flags...]"""
import re, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
csrc = ROOT / "autogp.jl_amd" / "csrc"
out = ""
for src in sorted(csrc.glob("agp_kernels*.hip")):          # the kernel translation units (the host units hold no __global__)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
           "-c", str(src), "-o", "/tmp/agp_res.o"] + [a for a in sys.argv[2:]]
    out += subprocess.run(cmd, capture_output=True, text=True, cwd=str(csrc)).stderr
cur = None; rows = []
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}; rows.append(cur); continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("spill_v", r"VGPRs Spill: (\d+)"),
                     ("spill_s", r"SGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, ln)
        if m and cur is not None:
            cur[key] = int(m.group(1))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
print(f"{'kernel':100s} vgpr agpr sgpr spillV scratch occ   lds")
for r in rows:
    if flt in r["name"]:
        print(f"{r['name'][:100]:100s} {r.get('vgpr',0):4d} {r.get('agpr',0):4d} {r.get('sgpr',0):4d} {r.get('spill_v',0):6d} {r.get('scratch',0):7d} {r.get('occ',0):3d} {r.get('lds',0):6d}")
