"""Time the slice-rank projector inside a solve on slices WITHOUT a spectral gap (the synthetic model of BASELINE config 4:
a constant per z slice plus white noise), with the Chebyshev-filtered subspace route (default) and with the full
decomposition of every call (SIPX_RANK_CHEB=0), and compare the iterates of the two.
usage: python tools/rank_flat_bench.py [n=256] [r=32] [iterations=12]"""
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
r = int(sys.argv[2]) if len(sys.argv) > 2 else 32
its = int(sys.argv[3]) if len(sys.argv) > 3 else 12
n, h = (nn, nn, nn), (25.0, 25.0, 25.0)
rng = np.random.default_rng(20240605)
zz = np.linspace(0.0, 1.0, nn)[None, None, :]
m = (1500.0 + 2500.0 * zz + 150.0 * rng.standard_normal(n)).reshape(-1, order="F").astype(TF)
out, xs = {}, {}
for route in ("1", "0"):
    os.environ["SIPX_RANK_CHEB"] = route
    g = sipx.compgrid(h, n)
    c = [sipx.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", "")),
         sipx.set_definitions("rank", "identity", 0, r, ("slice", "z"))]
    opt = sipx.PARSDMM_options(FL=TF, maxit=its + 3)
    opt.evol_rel_tol = opt.feas_tol = opt.obj_tol = 0.0
    P, A, prop = sipx.setup_constraints(c, g, TF)
    A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
    ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt)
    ctx.parsdmm_begin(opt)
    ctx.parsdmm_steps(3)
    t0 = time.perf_counter()
    ctx.parsdmm_steps(its)
    dt = time.perf_counter() - t0
    xs[route] = np.asarray(ctx.download(want_ly=False)[0], dtype=np.float64)
    out["filtered" if route == "1" else "full"] = {"it_per_s": its / dt, "ms_per_it": dt / its * 1e3}
    ctx.close()
if xs["1"] is not None:
    out["rel_diff_x"] = float(np.linalg.norm(xs["1"] - xs["0"]) / np.linalg.norm(xs["0"]))
print(json.dumps({"grid": n, "rank": r, **out}))
