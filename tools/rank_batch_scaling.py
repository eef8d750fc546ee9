"""How does the slice-rank projector scale with the number of slices (what a rank would own if the slices of that set were
dealt over the GPUs)?  {bounds, slice rank 32} on n x n x n3 grids, y/l section time per iteration from log.timing."""
import sys, json
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from __graft_entry__ import load_package
sipx = load_package()
import bench
TF = np.float32
out = {}
for n, n3s in ((256, (256, 128, 64, 32)), (512, (512, 128, 64))):
    for n3 in n3s:
        shape = (n, n, n3)
        m = bench.synthetic_model(shape, TF, 20240601 + 3)
        g, c = bench.build_problem(sipx, shape, (25.0, 25.0, 25.0), ["bounds", "rank:32"], m, TF, None)
        P, A, prop = sipx.setup_constraints(c, g, TF)
        opt = sipx.PARSDMM_options(FL=TF, maxit=7, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0)
        A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
        x, log, l, y = sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
        out[f"{n}x{n}x{n3}"] = round(log.timing["argmin y and l update"] * 1e3 / len(log.obj), 2)
        print(shape, out[f"{n}x{n}x{n3}"], "ms per iteration in the y/l section", flush=True)
print(json.dumps(out))
