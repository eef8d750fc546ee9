import sys, numpy as np
sys.path.insert(0, '.')
from __graft_entry__ import load_package
sipx = load_package()
TF = np.float32
for n in [(64, 64, 64), (128, 128, 128), (256, 256, 64), (256, 256, 256)]:
    h = (25.0, 25.0, 25.0)
    N = int(np.prod(n))
    rng = np.random.default_rng(0)
    m = (2500 + 150 * rng.standard_normal(N)).astype(TF)
    g = sipx.compgrid(h, n)
    c = [sipx.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", ""))] + \
        [sipx.set_definitions("l1", k, 0.0, 1e9, ("matrix", "")) for k in ("D_x", "D_y", "D_z")]
    x = rng.standard_normal(N).astype(TF)
    out = {}
    for mode in ("cds", "stencil"):
        P, A, prop = sipx.setup_constraints(c, g, TF)
        opt = sipx.PARSDMM_options(FL=TF)
        opt.Q_mode = mode
        A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
        ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt)
        out[mode] = ctx.apply_Q(x)
        ctx.close()
    d = np.abs(out["cds"].astype(np.float64) - out["stencil"])
    bad = np.flatnonzero(d > 1e-4 * np.abs(out["cds"]).max())
    print(n, "max diff", d.max(), "scale", np.abs(out["cds"]).max(), "nbad", bad.size, flush=True)
    if bad.size:
        b = bad[:10]
        print("  first bad idx", b, [np.unravel_index(int(i), n, order="F") for i in b[:5]])
        print("  cds", out["cds"][b[:5]], "st", out["stencil"][b[:5]])
