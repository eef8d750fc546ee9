"""One-off check (not a test): a 200-iteration C3 solve at 256^3 with and without the sampled prediction of theta must end at the
same point (the sample only places the speculative range of the first pass)."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
sipx = load_package()
import importlib.util
spec = importlib.util.spec_from_file_location("tg", "tests/test_gpu_parity.py"); tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
TF = np.float32
n = (256, 256, 256)
res = {}
for tag in ("0", "1"):
    os.environ["SIPX_L1_SAMPLE"] = tag
    m, g, opt, P, A, prop, AtA = tg._c3_problem(sipx, n, TF, maxit=200)
    opt.evol_rel_tol = opt.feas_tol = opt.obj_tol = 0.0          # as bench.py: the stop rules that depend on tolerances never fire
    x, log, l, y = sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
    res[tag] = (x.astype(np.float64), log)
    print(tag, "iterations", len(log.obj), "obj", log.obj[-1], "feas", log.set_feasibility[-1], "finite", np.isfinite(log.obj).all(), flush=True)
a, b = res["0"], res["1"]
print("rel diff x", np.linalg.norm(a[0] - b[0]) / np.linalg.norm(a[0]), "iters", len(a[1].obj), len(b[1].obj))
k = min(len(a[1].obj), len(b[1].obj))
print("max rel diff obj", np.max(np.abs(a[1].obj[:k] - b[1].obj[:k]) / np.abs(a[1].obj[:k])))
print("rho equal rows", int((np.abs(a[1].rho[:k] - b[1].rho[:k]) <= 1e-6 * np.abs(a[1].rho[:k])).all(axis=1).sum()), "of", k)
