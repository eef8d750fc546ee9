import sys, numpy as np
sys.path.insert(0, '.')
from __graft_entry__ import load_package
sipx = load_package()
from sipx import multilevel as ML
TF = np.float64
for n1 in (32, 64, 128):
    n, h = (n1, n1, n1), (25.0, 25.0, 25.0)
    rng = np.random.default_rng(5)
    m = (1500 + 2500 * np.linspace(0, 1, n[2])[None, None, :] + 150 * rng.standard_normal(n)).astype(TF).reshape(-1, order="F")
    g = sipx.compgrid(h, n)
    s = sipx.get_TD_operator(g, "TV", TF)[0] @ m
    c = [sipx.set_definitions("bounds", "identity", 1600.0, 3900.0, ("tensor", "")),
         sipx.set_definitions("l1", "TV", 0.0, float(0.5 * np.abs(s).sum()), ("tensor", ""))]
    opt = sipx.PARSDMM_options(FL=TF, maxit=8, evol_rel_tol=10 * np.finfo(TF).eps)
    # single level, cold
    P, A, prop = sipx.setup_constraints(c, g, TF)
    A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
    x, log, l, y = sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
    print(n1, "cold  cg", log.cg_it, "obj", log.obj[-1], flush=True)
    # warm restart from own output
    opt.zero_ini_guess = False
    x2, log2, l2, y2 = sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, opt, x.copy(), l, y)
    print(n1, "warm  cg", log2.cg_it, "obj", log2.obj[-1], flush=True)
    opt.zero_ini_guess = True
    L = ML.setup_multi_level_PARSDMM(m, 2, 2, g, c, opt)
    x3, log3, _, _ = ML.PARSDMM_multi_level(m, *L[:5], opt)
    print(n1, "ml    cg", log3.cg_it, "obj", log3.obj[-1], "rho", log3.rho[0], flush=True)
