"""The slice-rank projector inside a solve, iteration by iteration: wall time of every PARSDMM iteration (one native step each,
synchronised), the projector's route counters, and -- with SIPX_EXT_DEBUG=1/2 in the environment -- its own trace on stderr.
usage: python tools/rank_probe.py [config=c4|rank] [n3=512] [iterations=12] [n12=512]
  c4:   the eight sets of BASELINE config 4;  rank: {bounds, slice rank 32} alone (what a rank's share of the set costs)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

sipx = load_package()
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
n3 = int(sys.argv[2]) if len(sys.argv) > 2 else 512
its = int(sys.argv[3]) if len(sys.argv) > 3 else 12
n12 = int(sys.argv[4]) if len(sys.argv) > 4 else 512
TF = np.float32
shape = (n12, n12, n3)
names = bench.CONFIGS["c4"][2] if cfg == "c4" else ["bounds", "rank:32"]
m = bench.synthetic_model(shape, TF, 20240601 + 3)
gs = sipx.compgrid((25.0, 25.0, 25.0), shape)


def radius_of(opname):                      # sigma = 0.5 ||A m||_1 (bench.measure's rule)
    s = sipx.get_TD_operator(gs, opname, TF)[0] @ m
    return float(0.5 * np.abs(s.astype(np.float64)).sum())


g, c = bench.build_problem(sipx, shape, (25.0, 25.0, 25.0), names, m, TF, radius_of)
P, A, prop = sipx.setup_constraints(c, g, TF)
opt = sipx.PARSDMM_options(FL=TF, maxit=its + 1, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0)
A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
t0 = time.perf_counter()
ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt)
t_build = time.perf_counter() - t0
ctx.parsdmm_begin(opt)
ms = []
for i in range(its):
    t0 = time.perf_counter()
    ctx.parsdmm_steps(1)
    ctx.debug_proj(0, 0)
    ms.append((time.perf_counter() - t0) * 1e3)
    print("iteration %2d: %7.2f ms" % (i + 1, ms[-1]), file=sys.stderr, flush=True)
st = ctx.kernel_stats_all(-1)
x = np.asarray(ctx.download(want_ly=False)[0], dtype=np.float64)
out = {"config": cfg, "grid": shape, "build_context_s": t_build, "ms_per_iteration": [round(v, 2) for v in ms],
       "mean_ms_from_3": float(np.mean(ms[2:])) if len(ms) > 2 else None, "rank_route": st.get("rank_route"),
       "x_norm": float(np.linalg.norm(x)), "x_finite": bool(np.isfinite(x).all())}
if len(sys.argv) > 5:
    np.save(sys.argv[5], x.astype(np.float32))
print(json.dumps(out))
ctx.close()
