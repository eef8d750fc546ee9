"""Time the slice-rank projector inside a solve on slices with a separated spectrum (rank-6 structure + 0.1 % noise),
with and without the warm-started subspace route (SIPX_RANK_SUBSPACE).  usage: python tools/rank_route_bench.py [n=256] [r=8]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

sipx = load_package()
TF = np.float32
nn = int(sys.argv[1]) if len(sys.argv) > 1 else 256
r = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n, h = (nn, nn, nn), (10.0, 10.0, 10.0)
rng = np.random.default_rng(5)
U, V = rng.standard_normal((nn, 6)), rng.standard_normal((6, nn))
m3 = 2500.0 + 300.0 * (U @ V)[:, :, None] / 3.0 * np.linspace(0.5, 1.5, nn)[None, None, :] + 2.0 * rng.standard_normal(n)
m = m3.reshape(-1, order="F").astype(TF)
out = {}
for route in ("1", "0"):
    os.environ["SIPX_RANK_SUBSPACE"] = route
    g = sipx.compgrid(h, n)
    c = [sipx.set_definitions("bounds", "identity", 1000.0, 4500.0, ("matrix", "")),
         sipx.set_definitions("rank", "identity", 0, r, ("slice", "z"))]
    opt = sipx.PARSDMM_options(FL=TF, maxit=30)
    opt.evol_rel_tol = opt.feas_tol = opt.obj_tol = 0.0
    P, A, prop = sipx.setup_constraints(c, g, TF)
    A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
    ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt)
    ctx.parsdmm_begin(opt)
    ctx.parsdmm_steps(5)
    t0 = time.perf_counter()
    ctx.parsdmm_steps(20)
    dt = time.perf_counter() - t0
    out["subspace" if route == "1" else "full"] = {"it_per_s": 20 / dt, "ms_per_it": dt / 20 * 1e3}
    ctx.close()
print(json.dumps({"grid": n, "rank": r, **out}))
