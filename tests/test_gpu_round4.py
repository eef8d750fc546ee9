"""GPU tests added in round 4: the logged evol_x of a context without a distance term (the residual product queued ahead must
not overwrite x_old before the log reads it), the device-memory figure of a context, the read-only counters call."""
import numpy as np
import pytest

from oracle import parsdmm_oracle as O      # checker only
from tests.test_gpu_parity import _problem, model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [dict(adjust_rho=False, adjust_gamma=False, adjust_feasibility_rho=False), dict()])
def test_feasibility_only_logs_evol_x_and_stops_like_the_oracle(sipx, kw):
    """feasibility_only = true: no distance term, so obj / evol_x are formed by log_scalars from x_old AFTER the y/l update
    (PARSDMM.jl:139-147).  With rho fixed every right-hand side is known ahead and the residual product of the next x-step
    (which stores x_old <- x) used to be queued before that read: evol_x = 0 on every such iteration, and stop rule 2
    (stop_PARSDMM.jl:31-36) ended the solve six iterations later whether or not it had converged."""
    TF, n, h = np.float64, (32, 24), (25.0, 6.0)
    m = model(n, TF, seed=3)
    # (started from x = m, y_i = A_i m, l = 0: from the zero start the first x-step returns zero and the solve is over in three iterations)
    opts = dict(maxit=300, feasibility_only=True, evol_rel_tol=1e-7, zero_ini_guess=False, **kw)
    go, oo, Po, Ao, propo, AtAo = _problem(O, n, h, TF, ["bounds", "l1:TV"], m, opts)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, ["bounds", "l1:TV"], m, opts)
    y0 = [np.asarray(A @ m) for A in Ao]
    l0 = [np.zeros_like(v) for v in y0]
    xo, lo, _, _ = O.PARSDMM(m.copy(), AtAo, Ao, propo, Po, go, oo, m.copy(), [v.copy() for v in l0], [v.copy() for v in y0])
    xs, ls, _, _ = sipx.PARSDMM(m.copy(), AtAs, As, props, Ps, gs, os_, m.copy(), [v.copy() for v in l0], [v.copy() for v in y0])
    assert len(lo.obj) > 12, len(lo.obj)                       # a solve long enough for rule 2's window to matter
    assert abs(len(ls.obj) - len(lo.obj)) <= max(2, len(lo.obj) // 25), (len(ls.obj), len(lo.obj))
    K = min(len(ls.obj), len(lo.obj), 60)
    assert (ls.evol_x[1:K - 2] > 0).all()
    assert np.allclose(ls.evol_x[:K], lo.evol_x[:K], rtol=1e-5, atol=1e-14), np.abs(ls.evol_x[:K] / lo.evol_x[:K] - 1).max()
    assert np.allclose(ls.obj[:K], lo.obj[:K], rtol=1e-6)
    assert np.linalg.norm(xs - xo) / np.linalg.norm(xo) < 1e-5


def test_device_bytes_and_read_only_counters(sipx):
    """sipx_device_bytes: what a context allocated (at least its x, m, rhs, CG vectors, Q and the y / l pairs of every set);
    sipx_kernel_stats_json(-1): the engine's counters without touching a running statistics collection."""
    TF, n, h = np.float32, (48, 40, 32), (25.0, 25.0, 25.0)
    m = model(n, TF, seed=3)
    g, opt, P, A, prop, AtA = _problem(sipx, n, h, TF, ["bounds", "l1:D_x", "l1:D_z"], m, dict(maxit=12, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0))
    ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt)
    try:
        N, w = int(np.prod(n)), 4
        b = ctx.device_bytes()
        floor = (7 + 5 + 4 * 4) * N * w             # x, x_old, rhs, m, r, p, Ap + 5 bands of Q + (y, l) x 2 pairs for four terms
        assert floor < b["context"] < 4 * floor and b["context"] <= b["device_used"] <= b["device_total"]
        ctx.parsdmm_begin(opt)
        ctx.parsdmm_steps(3)
        ctx.kernel_stats(2)
        ctx.parsdmm_steps(3)
        peek = ctx.kernel_stats_all(-1)
        assert peek["kernels"] == [] and "slab_searches" in peek and "rank_route" in peek
        st = ctx.kernel_stats_all(0)                 # the collection survived the peek
        assert any(k["name"] == "k_cds<MODE=1>" and k["launches"] > 0 for k in st["kernels"])
    finally:
        ctx.close()
