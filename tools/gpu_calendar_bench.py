"""bench.py's calendar_monthly leg on its own (the benchmark population on n month starts): value / gradient / predictive sweeps of a
lattice with gaps beside the general evaluator.   python tools/gpu_calendar_bench.py [n=2048] [P=512]"""
import json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
P = int(sys.argv[2]) if len(sys.argv) > 2 else 512
pkg = g.load_package()
nodes, noises = pkg.prior.sample_particles(np.random.default_rng(2048), P, max_depth=-1, max_size=63)
out = bench.calendar_leg(pkg, pkg.encode_batch(nodes), nodes, noises, n, 0)
print(json.dumps(out))
