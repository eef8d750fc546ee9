"""BASELINE config 5 pattern: Float64, {bounds, l1 on TV}, 3 levels, coarsening factor 2 (test_scaling_3D.jl:144-145)."""
import sys, time, json
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from __graft_entry__ import load_package
sipx = load_package()
from sipx import multilevel as ML
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
maxit = int(sys.argv[2]) if len(sys.argv) > 2 else 30
TF = np.float64
n, h = (n1, n1, n1), (25.0, 25.0, 25.0)
rng = np.random.default_rng(20240601 + 5)
m = (1500 + 2500 * np.linspace(0, 1, n[2])[None, None, :] + 150 * rng.standard_normal(n)).astype(TF).reshape(-1, order="F")
g = sipx.compgrid(h, n)
t0 = time.perf_counter()
s = sipx.get_TD_operator(g, "TV", TF)[0] @ m
c = [sipx.set_definitions("bounds", "identity", 1600.0, 3900.0, ("tensor", "")),
     sipx.set_definitions("l1", "TV", 0.0, float(0.5 * np.abs(s).sum()), ("tensor", ""))]
del s
opt = sipx.PARSDMM_options(FL=TF, maxit=maxit, evol_rel_tol=10 * np.finfo(TF).eps)
L = ML.setup_multi_level_PARSDMM(m, 3, 2, g, c, opt)
t1 = time.perf_counter()
x, log, l, y = ML.PARSDMM_multi_level(m, *L[:5], opt)
t2 = time.perf_counter()
print(json.dumps({"grid": n, "levels": [list(gg.n) for gg in L[4]], "setup_s": t1 - t0, "solve_s": t2 - t1,
                  "finest_iterations": int(len(log.obj)), "finest_cg": int(np.sum(log.cg_it)),
                  "obj_last": float(log.obj[-1]), "feas_last": [float(v) for v in log.set_feasibility[-1]],
                  "finite": bool(np.isfinite(x).all())}))
