import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from __graft_entry__ import load_package
sipx = load_package()
import importlib
spec = importlib.util.spec_from_file_location("tg", "tests/test_gpu_parity.py"); tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
TF = np.float32
n = (256, 256, 256) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1].split("x"))
NIT = int(sys.argv[2]) if len(sys.argv) > 2 else 36
m, g, opt, P, A, prop, AtA = tg._c3_problem(sipx, n, TF, maxit=NIT + 1)
opt.evol_rel_tol = opt.feas_tol = opt.obj_tol = 0.0
ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt)
ctx.parsdmm_begin(opt)
prev = [0, 0, 0]
for it in range(1, NIT):
    ctx.parsdmm_steps(1)
    th = [ctx.debug_proj(s, 0) for s in (1, 2, 3)]
    rho = ctx._run[2]["rho"][it - 1]
    line = []
    for k, d in enumerate(th):
        t = d["theta"]
        rel = (t / prev[k] - 1) if prev[k] > 0 else float("nan")
        line.append("th %.3e (%+.4f) ok %d ov %d n %d hw %.0e its %d rf %d ln %d sm %d" % (t, rel, d["spec_ok"], d["overflow"], d["gathered"], d["hw"], d["michelot_its"], d["refine"], d["lean"], d["sampled"]))
        if d["sampled"]:
            d2 = ctx.debug_proj((1, 2, 3)[k], 2)
            line[-1] += " est %.4e [%.4e %.4e] c %d" % (d2["spec_lo"], d2["spec_hi"], d2["lo"], d2["hi"])
        prev[k] = t if t > 0 else prev[k]
    print(it, "rho", np.round(rho, 3), " | ".join(line), flush=True)
ctx.close()
