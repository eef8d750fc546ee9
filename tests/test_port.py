"""Cross-check of the two CPU restatements: the C/OpenMP port used as the timed CPU baseline
(oracle/parsdmm_port.c) against the numpy oracle.  Both are test infrastructure."""
import numpy as np
import pytest

from oracle import parsdmm_oracle as O
from oracle import port


def _model(n, TF, seed=0):
    rng = np.random.default_rng(20240601 + seed)
    z = np.linspace(0, 1, n[-1]).reshape((1,) * (len(n) - 1) + (-1,))
    return (1500 + 2500 * z + 150 * rng.standard_normal(n)).astype(TF).reshape(-1, order="F")


@pytest.mark.parametrize("TF", [np.float32, np.float64])
@pytest.mark.parametrize("n,h,ops", [((32, 24), (25.0, 6.0), ["TV"]), ((12, 10, 8), (25.0, 25.0, 25.0), ["D_x", "D_y", "D_z"])])
def test_port_matches_numpy_oracle(TF, n, h, ops):
    m = _model(n, TF)
    g = O.compgrid(h, n)
    sets, cons = [("bounds", "identity", 1600.0, 3900.0)], [O.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", ""))]
    for op in ops:
        A = O.get_TD_operator(g, op, TF)[0]
        r = float(0.5 * np.abs(A @ m).sum())
        sets.append(("l1", op, 0.0, r))
        cons.append(O.set_definitions("l1", op, 0.0, r, ("matrix", "")))
    opt = O.PARSDMM_options(FL=TF, maxit=40)
    P, A, prop = O.setup_constraints(cons, g, TF)
    A, AtA, l, y = O.PARSDMM_precompute_distribute(A, prop, g, opt)
    xo, lo, _, _ = O.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
    r = port.run(n, h, sets, m, 40, nthreads=2)
    K = min(6, len(lo.obj), r["n_iter"])
    rt = 5e-4 if TF == np.float32 else 1e-8
    assert np.array_equal(r["cg_it"][:K], lo.cg_it[:K])
    assert np.allclose(r["obj"][:K], lo.obj[:K], rtol=rt)
    assert np.allclose(r["rho"][:K], lo.rho[:K], rtol=rt) and np.allclose(r["gamma"][:K], lo.gamma[:K], rtol=rt)
    assert np.allclose(r["set_feasibility"][0], lo.set_feasibility[0], rtol=rt)
    err = np.linalg.norm(r["x"].astype(np.float64) - xo) / np.linalg.norm(xo)
    assert err < (5e-4 if TF == np.float32 else 1e-6), err
