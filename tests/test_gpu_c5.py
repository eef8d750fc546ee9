"""BASELINE configs[4] ("C5"): PARSDMM_multi_level, 3-D, Float64, THREE levels, coarsening factor 2, {bounds, l1 on TV}
(reference src/PARSDMM_multi_level.jl:8-89, src/interpolate_y_l.jl:7-97, src/setup_multi_level_PARSDMM.jl:7-137,
src/constraint2coarse.jl:8-104; timed in examples/test_scaling_3D.jl:144-148).  The workload (model, radius, options) is
bench.run_c5 -- the `c5` leg of the default bench line."""
import numpy as np
import pytest

import bench
from oracle import parsdmm_oracle as O      # checker only

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model", ["survey", "layered"])
def test_c5_three_levels_match_oracle(sipx, model):
    """C5's own shape at a size the oracle finishes in seconds: x to 1e-6, the same iteration count on EVERY level (which
    also says that every level iterates: the round-2 workload returned from both coarse levels through the feasible-input exit)."""
    from sipx import multilevel as ML
    TF, n, h = np.float64, (24, 20, 16), (25.0, 25.0, 25.0)
    its_o = []

    def solver(*a, **k):
        r = O.PARSDMM(*a, **k)
        its_o.append((tuple(int(v) for v in a[5].n), len(r[1].obj), int(np.sum(r[1].cg_it))))
        return r
    mo, go, co = bench.c5_problem(O, n, h, TF, model)
    oo = O.PARSDMM_options(FL=TF, maxit=100, evol_rel_tol=10 * float(np.finfo(TF).eps))
    Lo = O.setup_multi_level_PARSDMM(mo, 3, 2, go, co, oo)
    xo, logo, lo, yo = O.PARSDMM_multi_level(mo.copy(), *Lo[:5], oo, solver=solver)

    ms, gs, cs = bench.c5_problem(sipx, n, h, TF, model)
    assert np.array_equal(ms, mo) and cs[1].max == pytest.approx(co[1].max, rel=1e-12)
    os_ = sipx.PARSDMM_options(FL=TF, maxit=100, evol_rel_tol=10 * float(np.finfo(TF).eps))
    Ls = ML.setup_multi_level_PARSDMM(ms, 3, 2, gs, cs, os_)
    assert [tuple(g.n) for g in Ls[4]] == [tuple(g.n) for g in Lo[4]] == [(24, 20, 16), (12, 10, 8), (6, 5, 4)]
    T = {}
    xs, logs, ls, ys = ML.PARSDMM_multi_level(ms.copy(), *Ls[:5], os_, timings=T)
    its_s = [(tuple(v["grid"]), v["iterations"], v["cg_iterations"]) for v in T["levels"]]
    assert its_s == its_o, (its_s, its_o)
    assert all(it > 1 for _, it, _ in its_s), its_s
    err = np.linalg.norm(xs - xo) / np.linalg.norm(xo)
    assert err < 1e-6, err
    assert np.allclose(logs.obj, logo.obj, rtol=1e-8) and np.array_equal(logs.cg_it, logo.cg_it)
    assert np.allclose(logs.rho, logo.rho, rtol=1e-8)
    for a, b in zip(ys, yo):
        assert len(a) == len(b) and np.linalg.norm(a - b) <= 1e-6 * max(np.linalg.norm(b), 1e-300)


def test_c5_workload_keeps_its_promises_at_moderate_size(sipx):
    """run_c5 at 96^3 (seconds): every level iterates, the finest level starts from the coarse solution (its first primal
    residual is below the cold start's) and needs no more iterations than the cold start."""
    r = bench.run_c5(sipx, (96, 96, 96), maxit=100)
    assert r["levels"] == [[96, 96, 96], [48, 48, 48], [24, 24, 24]]
    assert r["finite"] and r["every_level_iterates"], r["iterations_per_level"]
    assert r["finest_first_r_pri_total"] < r["single_level"]["first_r_pri_total"]
    assert r["finest_iterations"] <= r["single_level"]["iterations"]


@pytest.mark.timeout(900)
def test_c5_full_size_properties(sipx):
    """BASELINE configs[4] itself, 512^3 Float64, three levels: finite logs, the three grids, every level iterates, the
    finest level warm-started (first-iteration r_pri below the cold start's), feasible end point."""
    r = bench.run_c5(sipx, (512, 512, 512), maxit=100)
    assert r["levels"] == [[512, 512, 512], [256, 256, 256], [128, 128, 128]]
    assert [v["grid"] for v in r["per_level"]] == [[128, 128, 128], [256, 256, 256], [512, 512, 512]]
    assert r["finite"] and r["every_level_iterates"], r["iterations_per_level"]
    assert r["finest_first_r_pri_total"] < r["single_level"]["first_r_pri_total"]
    assert r["finest_iterations"] <= r["single_level"]["iterations"]
    assert max(r["feas_last"]) < 5e-2                      # the default feas_tol: the stop rule that ended the solve
