import sys, numpy as np, importlib
sys.path.insert(0, '.')
from __graft_entry__ import load_package
sipx = load_package()
from oracle import parsdmm_oracle as O
spec = importlib.util.spec_from_file_location("tg", "tests/test_gpu_parity.py"); tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
np.set_printoptions(linewidth=200, precision=10)
n, h, TF = (32, 24), (25.0, 6.0), np.float64
m = tg.model(n, TF, seed=4)
for sig in (0.02, 0.5):
    res = {}
    for name, mod in (("o", O), ("s", sipx)):
        g = mod.compgrid(h, n)
        c = [mod.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", "")),
             mod.set_definitions("l1", "identity", 0.0, float(sig * np.abs(m).sum()), ("matrix", ""))]
        P, A, prop = mod.setup_constraints(c, g, TF)
        opt = mod.PARSDMM_options(FL=TF, maxit=4)
        A, AtA, l, y = mod.PARSDMM_precompute_distribute(A, prop, g, opt)
        res[name] = mod.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
    print("sigma frac", sig)
    print(" r_pri o", res["o"][1].r_pri[:, 1]); print(" r_pri s", res["s"][1].r_pri[:, 1])
# standalone projector on vectors with many repeated values
rng = np.random.default_rng(0)
v = np.repeat(rng.standard_normal(50), 16) * 100
for b in (0.02 * np.abs(v).sum(), 0.5 * np.abs(v).sum()):
    a = O.project_l1_Duchi(v.copy(), b)
    c = sipx.set_definitions("l1", "identity", 0.0, float(b), ("matrix", ""))
    P = sipx.host.Projector(c, sipx.compgrid((1.0, 1.0), (len(v), 1)), TF)
    bb = P(v.copy())
    print("standalone", b, np.abs(a - bb).max(), np.abs(bb).sum(), np.abs(a).sum())
