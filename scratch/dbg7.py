import sys, numpy as np
sys.path.insert(0,'.')
from __graft_entry__ import load_package
sipx = load_package()
from sipx import sharded
from oracle import parsdmm_oracle as O
import tests.test_gpu_parity as tp
TF=np.float64
n,h=(16,12,8),(25.0,25.0,25.0)
kinds=["bounds","l1dft"]
m=tp.model(n,TF,seed=2)
go, oo, Po, Ao, propo, AtAo = tp._problem(O, n, h, TF, kinds, m, dict(maxit=8))
gs, os_, Ps, As, props, AtAs = tp._problem(sipx, n, h, TF, kinds, m, dict(maxit=8))
tr=[]
xo, lo, l_o, y_o = O.PARSDMM(m.copy(), AtAo, Ao, propo, Po, go, oo, trace=tr)
ctx = sipx.host.build_context(m, AtAs, As, props, Ps, gs, os_)
drv = sharded.PhaseDriver(ctx, os_)
for i in range(6):
    drv.step()
    x,l,y = ctx.download()
    t=tr[i]
    print(i+1, "x", np.linalg.norm(x-t["x"])/max(np.linalg.norm(t["x"]),1e-300), "y", [np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-300) for a,b in zip(y,t["y"])], "l", [np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-300) for a,b in zip(l,t["l"])], "rho", drv.log.rho[i], lo.rho[i], "gam", drv.log.gamma[i], lo.gamma[i])
    print("    ", ctx.debug_proj(1))
ctx.close()
print("----")
ctx = sipx.host.build_context(m, AtAs, As, props, Ps, gs, os_)
drv = sharded.PhaseDriver(ctx, os_)
drv.step(); x1,l1,y1 = ctx.download()
drv.step(); x2,l2,y2 = ctx.download()
lo_=tr[1]["l"][1]; yo_=tr[1]["y"][1]
print("l_engine[:5]", l2[1][:5]); print("l_oracle[:5]", lo_[:5])
print("10*(y-s) engine[:5]", (10*(y2[1]-x2))[:5], " l1 prev", l1[1][:5], "oracle l prev", tr[0]["l"][1][:5])
print("x1 norm", np.linalg.norm(x1), "y1[1] norm", np.linalg.norm(y1[1]), "oracle y it1", np.linalg.norm(tr[0]["y"][1]))
