import sys, numpy as np
sys.path.insert(0,'.')
from __graft_entry__ import load_package
sipx = load_package()
from sipx import sharded
import bench
TF=np.float32
for cfg in ["c3-small","c3"]:
    n,h,kinds = bench.CONFIGS[cfg]
    m = bench.synthetic_model(n,TF,20240604)
    gs = sipx.compgrid(h,n)
    def radius_of(op):
        s = sipx.get_TD_operator(gs, op, TF)[0] @ m
        return float(0.5*np.abs(s.astype(np.float64)).sum())
    g,c = bench.build_problem(sipx,n,h,kinds,m,TF,radius_of)
    P,A,prop = sipx.setup_constraints(c,g,TF)
    opt = bench.bench_options(sipx,TF,12)
    A,AtA,l,y = sipx.PARSDMM_precompute_distribute(A,prop,g,opt)
    ctx = sipx.host.build_context(m,AtA,A,prop,P,g,opt)
    drv = sharded.PhaseDriver(ctx,opt)
    for i in range(10):
        drv.step()
        L=drv.log
        print(cfg,i+1,"cg",L.cg_it[i],"obj %.4e"%L.obj[i],"rpri",np.array2string(L.r_pri[i],precision=3),"rho",np.array2string(L.rho[i],precision=3), flush=True)
        if not np.isfinite(L.obj[i]) and i>0: break
    ctx.close()
# large-vector projector, cold
rng=np.random.default_rng(0)
for n in (10**6, 2*10**7):
    v=(rng.standard_normal(n)*np.exp(rng.standard_normal(n))).astype(TF)
    b=float(0.3*np.abs(v.astype(np.float64)).sum())
    Pj=sipx.Projector(sipx.set_definitions("l1","identity",0.0,b,("matrix","")),sipx.compgrid((1.,1.),(4,4)),TF)
    w=Pj(v.copy())
    a=np.abs(v.astype(np.float64)); 
    u=np.sort(a)[::-1]; cs=np.cumsum(u); k=np.arange(1,n+1); rho=np.nonzero(u>(cs-b)/k)[0][-1]; th=(cs[rho]-b)/(rho+1)
    ref=np.sign(v)*np.maximum(a-th,0)
    print(n,"theta",th,"err",np.linalg.norm(w-ref)/np.linalg.norm(ref),"l1",np.abs(w.astype(np.float64)).sum()/b, flush=True)
