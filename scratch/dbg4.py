import sys, numpy as np
sys.path.insert(0,'.')
from __graft_entry__ import load_package
sipx = load_package()
from sipx import sharded
import bench
TF=np.float32
cfg = sys.argv[1] if len(sys.argv)>1 else "c3"
n,h,kinds = bench.CONFIGS[cfg]
m = bench.synthetic_model(n,TF,20240604)
gs = sipx.compgrid(h,n)
def radius_of(op):
    s = sipx.get_TD_operator(gs, op, TF)[0] @ m
    return float(0.5*np.abs(s.astype(np.float64)).sum())
g,c = bench.build_problem(sipx,n,h,kinds,m,TF,radius_of)
P,A,prop = sipx.setup_constraints(c,g,TF)
opt = bench.bench_options(sipx,TF,40)
A,AtA,l,y = sipx.PARSDMM_precompute_distribute(A,prop,g,opt)
ctx = sipx.host.build_context(m,AtA,A,prop,P,g,opt)
drv = sharded.PhaseDriver(ctx,opt)
for i in range(26):
    drv.step()
    d=ctx.debug_proj(1)
    print(i+1, "need %d th %.5g prev_rel %.3g hw %.3g gathered %d ovf %d spec_ok %d its %d refine %d lo/hi rel %.3g"%(d["need"],d["theta"],(d["theta"]/d["theta_prev"]-1) if d["theta_prev"] else 0,d["hw"],d["gathered"],d["overflow"],d["spec_ok"],d["michelot_its"],d["refine"],(d["hi"]-d["lo"])/max(d["hi"],1e-300)), flush=True)
ctx.close()
