"""Phases of the element-wise gradient sweep (lag domain off: what an irregular series costs), 512 prior-sampled particles, n = 2048."""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
import __graft_entry__ as g
pkg=g.load_package()
n=2048; P=512
ts,xs=pkg.prior.synthetic_series(n,seed=2048,shuffle=True)
nodes,noises=pkg.prior.sample_particles(np.random.default_rng(2048),P,max_depth=-1,max_size=63)
e=pkg.GPEngine(0); e.set_data(ts,xs); e.set_grad_lag_domain(False)
e.logpdf_grad_batch(nodes,noises,check=False)
e.set_profiling(True)
acc={}
for _ in range(3):
    e.logpdf_grad_batch(nodes,noises,check=False)
    for k,v in e.timing().items(): acc[k]=acc.get(k,0)+v/3
print({k:round(v,2) for k,v in acc.items() if v>0.01})
