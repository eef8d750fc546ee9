import sys, numpy as np
sys.path.insert(0,'.')
from __graft_entry__ import load_package
sipx = load_package()
import bench
TF=np.float32
n,h,kinds = bench.CONFIGS["c3"]
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
ctx.parsdmm_begin(opt)
ctx.parsdmm_steps(8)
for r in range(3):
    ms = ctx.time_spmv(50)
    print("MODE0 back-to-back: %.1f us  %.0f GB/s" % (ms*1e3, 9*256**3*4/ms/1e6))
ctx.kernel_stats(True)
ctx.parsdmm_steps(10)
nl, ms = ctx.kernel_stats(False)
print("MODE1 in CG: %d launches avg %.1f us  %.0f GB/s" % (nl, ms/nl*1e3, 9*256**3*4/(ms/nl)/1e6))
ctx.close()
