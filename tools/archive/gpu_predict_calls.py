"""Wall time of eight consecutive predictive passes of a fresh process (128 particles, n = 2048, m = 4096 lattice queries): the first
calls of a process must not cost more than the later ones (host arrays travel through pinned staging: PinnedUploads).
python tools/gpu_predict_calls.py [plain|after_grad|after_value]"""
import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import __graft_entry__ as g
pkg=g.load_package()
n=2048; P=512
ts,xs=pkg.prior.synthetic_series(n,seed=2048,shuffle=True)
nodes,noises=pkg.prior.sample_particles(np.random.default_rng(2048),P,max_depth=-1,max_size=63)
e=pkg.GPEngine(0); e.set_data(ts,xs)
mode=sys.argv[1] if len(sys.argv)>1 else "plain"
if mode=="after_grad":
    e.logpdf_grad_batch(nodes,noises,check=False); e.logpdf_grad_batch(nodes,noises,check=False)
if mode=="after_value":
    for _ in range(30): e.logpdf_batch(nodes,noises,check=False)
tsort=np.sort(ts); hq=(tsort[-1]-tsort[0])/(n-1)
tq=np.concatenate([ts,tsort[0]+hq*np.arange(n,2*n)])
out=[]
for r in range(8):
    t0=time.perf_counter(); e.predict_batch(nodes[:128],noises[:128],tq,n=n,check=False); out.append((time.perf_counter()-t0)*1e3)
print(mode,[round(x,1) for x in out])
