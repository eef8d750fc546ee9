import sys, numpy as np, importlib
sys.path.insert(0, '.')
from __graft_entry__ import load_package
sipx = load_package()
from oracle import parsdmm_oracle as O
spec = importlib.util.spec_from_file_location("tg", "tests/test_gpu_parity.py"); tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
np.set_printoptions(linewidth=200, precision=10)
n, h, TF = (32, 24), (25.0, 6.0), np.float64
m = tg.model(n, TF, seed=4)
go, oo, Po, Ao, propo, AtAo = tg._minkowski_problem(O, n, h, TF, m, maxit=4)
gs, os_, Ps, As, props, AtAs = tg._minkowski_problem(sipx, n, h, TF, m, maxit=4)
xo, lo, l_o, y_o = O.PARSDMM(m.copy(), AtAo, Ao, propo, Po, go, oo)
xs, ls, l_s, y_s = sipx.PARSDMM(m.copy(), AtAs, As, props, Ps, gs, os_)
for f in ("cg_it", "cg_relres", "obj", "evol_x", "r_pri", "r_dual", "rho", "gamma"):
    print(f); print(" o", np.asarray(getattr(lo, f))); print(" s", np.asarray(getattr(ls, f)))
for i in range(len(y_o)):
    print(i, "y err", np.abs(y_o[i] - y_s[i]).max(), "l err", np.abs(l_o[i] - l_s[i]).max())
print("x err u", np.abs(xo[:m.size] - xs[:m.size]).max(), "v", np.abs(xo[m.size:] - xs[m.size:]).max())
