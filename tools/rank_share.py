import sys, json, time
import numpy as np
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
sipx = load_package()
import bench
TF = np.float32
for n3 in (512, 128, 64):
    shape = (512, 512, n3)
    m = bench.synthetic_model(shape, TF, 20240601 + 3)
    g, c = bench.build_problem(sipx, shape, (25.0, 25.0, 25.0), ["bounds", "rank:32"], m, TF, None)
    P, A, prop = sipx.setup_constraints(c, g, TF)
    opt = sipx.PARSDMM_options(FL=TF, maxit=30, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0)
    A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
    ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt)
    ctx.parsdmm_begin(opt)
    ctx.parsdmm_steps(4)
    ctx.debug_proj(0, 0)
    t0 = time.perf_counter()
    ctx.parsdmm_steps(20)
    ctx.debug_proj(0, 0)
    dt = (time.perf_counter() - t0) / 20
    st = ctx.kernel_stats_all(-1)
    print(shape, "%.2f ms per iteration (its 5-24)" % (dt * 1e3), st.get("rank_route"), flush=True)
    ctx.close()
