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
host_path = len(sys.argv) > 3 and sys.argv[3] == "host"
T = {}
x, log, l, y = ML.PARSDMM_multi_level(m, *L[:5], opt, timings=T, host_transfers=host_path)
t2 = time.perf_counter()
fin = T["levels"][-1]
print(json.dumps({"grid": n, "levels": [list(gg.n) for gg in L[4]], "transfers": "host (round-1 path)" if host_path else "device (sipx_warm_start_from)",
                  "setup_s": t1 - t0, "whole_solve_s": t2 - t1,
                  "solve_only_s": sum(v["solve_s"] for v in T["levels"]), "warm_start_total_s": sum(v["warm_start_s"] for v in T["levels"]),
                  "context_total_s": sum(v["context_s"] for v in T["levels"]), "download_s": T.get("download_s"),
                  "per_level": T["levels"],
                  "finest_iterations": fin["iterations"], "finest_cg": fin["cg_iterations"],
                  "finest_level_it_per_s": fin["iterations"] / fin["solve_s"],
                  "obj_last": float(log.obj[-1]), "feas_last": [float(v) for v in log.set_feasibility[-1]],
                  "finite": bool(np.isfinite(x).all())}))
