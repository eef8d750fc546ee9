import sys, numpy as np
sys.path.insert(0,'.')
from __graft_entry__ import load_package
sipx = load_package()
from sipx import sharded
import bench
TF=np.float32
cfg = sys.argv[1] if len(sys.argv)>1 else "c3-small"
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
for i in range(30):
    drv.step()
    L=drv.log
    print(cfg,i+1,"cg",L.cg_it[i],"relres %.2e"%L.cg_relres[i],"obj %.4e"%L.obj[i],"rpri",np.array2string(L.r_pri[i],precision=3),"rho",np.array2string(L.rho[i],precision=3),"gam",np.array2string(L.gamma[i],precision=3), flush=True)
    if i%10==9: print("  feas", L.set_feasibility[drv.counter-2])
    if (not np.isfinite(L.obj[i]) and i>0) or L.cg_it[i]>50: break
ctx.close()
