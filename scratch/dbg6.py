import sys, numpy as np
sys.path.insert(0,'.')
from __graft_entry__ import load_package
sipx = load_package()
from oracle import parsdmm_oracle as O
sys.path.insert(0,'tests')
import tests.test_gpu_parity as tp
TF=np.float32
n,h=(16,12,8),(25.0,25.0,25.0)
kinds=["bounds","l1dft"]
m=tp.model(n,TF,seed=2)
go, oo, Po, Ao, propo, AtAo = tp._problem(O, n, h, TF, kinds, m, dict(maxit=60))
gs, os_, Ps, As, props, AtAs = tp._problem(sipx, n, h, TF, kinds, m, dict(maxit=60))
ctx = sipx.host.build_context(m, AtAs, As, props, Ps, gs, os_)
print("engine feas0", ctx.feasibility_initial)
for i,P in enumerate(Po):
    Am = O.csc_mul(Ao[i], m)
    print("oracle feas0", i, O.nrm2(P(Am.copy())-Am,TF)/O.nrm2(Am,TF))
print("proj via API:", np.linalg.norm(Ps[1](m.copy())-m)/np.linalg.norm(m))
ctx.close()
