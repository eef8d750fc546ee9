"""Pins the CPU oracle against every known-answer / property test the reference holds for
the PARSDMM path.  The reference ships no golden vectors; these are the
deterministic facts its own test-suite asserts, re-asserted on our restatement."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import parsdmm_oracle as O


def _grid2(TF, n=(30, 20), h=(25.0, 25.0)):
    return O.compgrid((TF(h[0]), TF(h[1])), n)


# ---- test/test_prox_l2s!.jl:4-8,15-19 -------------------------------------------------
@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_prox_l2s_known_answers(TF):
    m = np.random.default_rng(1).standard_normal(10).astype(TF)
    x = np.random.default_rng(2).standard_normal(10).astype(TF)
    O.prox_l2s(x, TF(0), m)
    assert np.array_equal(x, m)
    x = np.array([2.0], TF)
    O.prox_l2s(x, TF(3), np.array([1.0], TF))
    assert x[0] == TF(7.0 / 4.0)


# ---- test/test_projectors.jl:49-56 ----------------------------------------------------
def test_cardinality_closed_forms():
    x = np.array([0, 0, 1, 2, 3], np.float64)
    assert np.array_equal(O.project_cardinality(x, 2), [0, 0, 0, 2, 3])
    x = np.array([0, 0, -1, 2, -3], np.float64)
    assert np.array_equal(O.project_cardinality(x, 2), [0, 0, 0, 2, -3])


# ---- test/test_projectors.jl:22-29,94-104 (properties) --------------------------------
@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_projector_properties(TF):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(1000).astype(TF)
    # l1: result has ||.||_1 == radius; feasible input untouched
    b = TF(0.3) * O.asum(x, TF)
    y = O.project_l1_Duchi(x.copy(), b)
    assert abs(float(O.asum(y, TF)) - float(b)) <= 1e-4 * float(b)
    z = O.project_l1_Duchi(y.copy(), TF(2) * b)
    assert np.array_equal(z, y)
    # bounds
    y = O.project_bounds(x.copy(), TF(-0.5), TF(0.25))
    assert y.min() >= TF(-0.5) and y.max() <= TF(0.25)
    # l2 / annulus: norm equals the violated radius
    y = O.project_l2(x.copy(), TF(1.0))
    assert abs(float(O.nrm2(y, TF)) - 1.0) < 1e-5
    n = float(O.nrm2(x, TF))
    y = O.project_annulus(x.copy(), TF(2 * n), TF(3 * n))
    assert abs(float(O.nrm2(y, TF)) - 2 * n) < 1e-4 * n
    y = O.project_annulus(np.zeros(16, TF), TF(2), TF(3))
    assert np.allclose(y, 0.5)


# ---- l1 threshold equals the exact fixed point of sum(max(|v|-t,0)) = b ----------------
def test_l1_theta_is_fixed_point():
    rng = np.random.default_rng(4)
    v = rng.standard_normal(5000)
    b = 0.25 * np.abs(v).sum()
    t = float(O.l1ball_theta_duchi(np.abs(v), b))
    assert abs(np.maximum(np.abs(v) - t, 0).sum() - b) < 1e-9 * b


# ---- test/test_TD_OPs.jl:13-40,54-81 --------------------------------------------------
def test_TD_ops_2d_are_diff_over_h():
    TF = np.float64
    n1, n2, h1, h2 = 7, 5, 2.0, 3.0
    g = O.compgrid((h1, h2), (n1, n2))
    img = np.zeros((n1, n2)); img[3, :] = 1; img[:, 2] = 1          # the "cross" image
    v = img.reshape(-1, order="F")
    Dx = O.get_TD_operator(g, "D_x", TF)[0]
    Dz = O.get_TD_operator(g, "D_z", TF)[0]
    TV = O.get_TD_operator(g, "TV", TF)[0]
    dx = (np.diff(img, axis=0) / h1).reshape(-1, order="F")
    dz = (np.diff(img, axis=1) / h2).reshape(-1, order="F")
    assert np.allclose(Dx @ v, dx) and np.allclose(Dz @ v, dz)
    assert np.allclose(TV @ v, np.concatenate([dz, dx]))            # TV = [D_z; D_x]


def test_TD_ops_3d_are_diff_over_h():
    TF = np.float64
    n, h = (5, 4, 3), (1.0, 2.0, 4.0)
    g = O.compgrid(h, n)
    img = np.random.default_rng(5).standard_normal(n)
    v = img.reshape(-1, order="F")
    for ax, name in enumerate(["D_x", "D_y", "D_z"]):
        D = O.get_TD_operator(g, name, TF)[0]
        assert np.allclose(D @ v, (np.diff(img, axis=ax) / h[ax]).reshape(-1, order="F"))
    TV = O.get_TD_operator(g, "TV", TF)[0]
    parts = [(np.diff(img, axis=ax) / h[ax]).reshape(-1, order="F") for ax in (2, 1, 0)]
    assert np.allclose(TV @ v, np.concatenate(parts))               # z, y, x


# ---- test/test_CDS_Mvp.jl:9-35 ---------------------------------------------------------
def test_CDS_MVp_matches_sparse():
    TF = np.float32
    g = _grid2(TF)
    TV = O.get_TD_operator(g, "TV", TF)[0]
    A = O.ata_ordered(TV, TF)
    x = np.random.default_rng(6).standard_normal(600).astype(TF)
    R, off = O.mat2CDS(A, TF)
    assert np.allclose(O.CDS_MVp(R, off, x, np.zeros(600, TF)), A @ x, rtol=10 * np.finfo(TF).eps, atol=1e-6)
    A = sp.random(300, 300, 0.1, random_state=7, format="csc")
    x = np.random.default_rng(8).standard_normal(300)
    R, off = O.mat2CDS(A, np.float64)
    assert np.allclose(O.CDS_MVp(R, off, x, np.zeros(300)), A @ x, rtol=1e-12, atol=1e-12)


# ---- test/test_CDS_scaled_add.jl:22-33 (exact) and missing-diagonal error --------------
def test_CDS_scaled_add_exact():
    TF = np.float64
    g = _grid2(TF, n=(12, 9))
    A = O.ata_ordered(O.get_TD_operator(g, "TV", TF)[0], TF)
    B = O.ata_ordered(O.get_TD_operator(g, "D_z", TF)[0], TF)
    RA, oA = O.mat2CDS(A, TF)
    RB, oB = O.mat2CDS(B, TF)
    RS, oS = O.mat2CDS(sp.csc_matrix(A + B), TF)
    O.CDS_scaled_add(RA, RB, oA, oB, 1.0)
    assert np.array_equal(oA, oS) and np.array_equal(RA, RS)
    with pytest.raises(ValueError):
        O.CDS_scaled_add(RB, RA, oB, oA, 1.0)


# ---- test/test_Q_update.jl: CDS branch reproduces A + sum d_rho B_i under SpMV ---------
def test_Q_update_cds():
    TF = np.float64
    g = _grid2(TF, n=(10, 8))
    ops = [O.get_TD_operator(g, k, TF)[0] for k in ("TV", "D_x", "identity")]
    mats = [O.ata_ordered(A, TF) for A in ops]
    cds = [O.mat2CDS(M, TF) for M in mats]
    AtA = [c[0] for c in cds]
    prop = O.set_properties(AtA_offsets=[c[1] for c in cds])
    rho_old = np.array([1.0, 2.0, 3.0]); rho_new = np.array([1.5, 2.0, 0.5])
    Q, Qo = O.assemble_Q(AtA, prop.AtA_offsets, rho_old, TF)

    class L: pass
    log = L(); log.rho = rho_old[None, :]
    O.Q_update(Q, AtA, prop, rho_new, [0, 2], log, 0, Qo)
    dense = sum(r * M for r, M in zip(rho_new, mats))
    x = np.random.default_rng(9).standard_normal(80)
    assert np.allclose(O.Ax_CDS(x, Q, Qo), dense @ x, rtol=1e-12)


# ---- test/test_cg.jl:1-29 --------------------------------------------------------------
def test_cg_properties():
    rng = np.random.default_rng(10)
    A = rng.standard_normal((200, 100)); A = A.T @ A
    xt = rng.standard_normal(100); b = A @ xt
    Af = lambda v: A @ v
    x, flag, relres, it1 = O.cg(Af, b, 1e-5, 1000, np.zeros(100))
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) <= 1e-5
    x, flag, relres, it2 = O.cg(Af, b, 1e-5, 1000, xt + np.finfo(float).eps)
    assert it2 < it1
    x0 = xt.copy()
    x, flag, relres, it = O.cg(Af, b, 1e-14, 1000, x0)       # exact guess: iter==1, x untouched
    assert it == 1 and np.array_equal(x, xt) and flag == 0


# ---- test/test_update_y_l.jl:61-75: equals the 4-line formula --------------------------
@pytest.mark.parametrize("gamma", [1.0, 1.3])
def test_update_y_l_formula(gamma):
    TF = np.float64
    g = _grid2(TF, n=(9, 7))
    A = O.get_TD_operator(g, "TV", TF)[0]
    rng = np.random.default_rng(11)
    N, M = A.shape[1], A.shape[0]
    x = rng.standard_normal(N)
    y = [rng.standard_normal(M)]; l = [rng.standard_normal(M)]
    y0, l0 = y[0].copy(), l[0].copy()
    rho = np.array([2.5]); gam = np.array([gamma])
    P = [lambda v: O.project_bounds(v, -0.1, 0.1)]
    z = lambda: [np.zeros(M)]
    y_old, l_old, x_hat, r_pri, s = z(), z(), z(), z(), z()

    class L: pass
    log = L(); log.r_pri = np.zeros((3, 1)); log.r_dual = np.zeros((3, 1)); log.set_feasibility = np.zeros((3, 1))
    O.update_y_l(x, 1, 1, y, y_old, l, l_old, rho, gam, P, [A], log, P, 2, x_hat, r_pri, s, True)
    s_ref = A @ x
    xh = gamma * s_ref + (1 - gamma) * y0
    y_ref = np.clip(xh - l0 / 2.5, -0.1, 0.1)
    l_ref = l0 + 2.5 * (y_ref - xh)
    assert np.allclose(y[0], y_ref, atol=1e-14) and np.allclose(l[0], l_ref, atol=1e-13)
    assert np.array_equal(y_old[0], y0) and np.array_equal(l_old[0], l0)
    assert np.isclose(log.r_pri[0, 0], np.linalg.norm(y_ref - s_ref))


# ---- solver-level pins (test/test_PARSDMM.jl:17-36, 77-89, 192-242) -------------------
def _setup(constraints, g, opt):
    P_sub, TD_OP, prop = O.setup_constraints(constraints, g, opt.FL)
    TD_OP, AtA, l, y = O.PARSDMM_precompute_distribute(TD_OP, prop, g, opt)
    return P_sub, TD_OP, prop, AtA


def test_feasible_input_returned_untouched():
    opt = O.PARSDMM_options(FL=np.float64)
    g = O.compgrid((1.0, 1.0), (20, 31))
    x = np.random.default_rng(12).standard_normal(20 * 31)
    c = [O.set_definitions("bounds", "identity", float(x.min()), float(x.max()), ("matrix", ""))]
    P_sub, TD_OP, prop, AtA = _setup(c, g, opt)
    xo, log, l, y = O.PARSDMM(x.copy(), AtA, TD_OP, prop, P_sub, g, opt)
    assert np.array_equal(xo, x) and len(log.obj) == 1


@pytest.mark.parametrize("kind", ["bounds", "l1", "annulus"])
def test_single_identity_set_equals_projector(kind):
    TF = np.float64
    opt = O.PARSDMM_options(FL=TF, maxit=400, feas_tol=1e-10, obj_tol=1e-10, evol_rel_tol=1e-12)
    g = O.compgrid((1.0, 1.0), (16, 11))
    m = np.random.default_rng(13).standard_normal(16 * 11)
    if kind == "bounds":
        c = O.set_definitions("bounds", "identity", -0.3, 0.4, ("matrix", ""))
        ref = np.clip(m, -0.3, 0.4)
    elif kind == "l1":
        b = 0.4 * np.abs(m).sum()
        c = O.set_definitions("l1", "identity", 0.0, b, ("matrix", ""))
        ref = O.project_l1_Duchi(m.copy(), b)
    else:
        nm = np.linalg.norm(m)
        c = O.set_definitions("annulus", "identity", 0.3 * nm, 0.5 * nm, ("matrix", ""))
        ref = m * 0.5
    P_sub, TD_OP, prop, AtA = _setup([c], g, opt)
    x, log, l, y = O.PARSDMM(m.copy(), AtA, TD_OP, prop, P_sub, g, opt)
    assert np.linalg.norm(x - ref) / np.linalg.norm(ref) < 1e-7


def test_converged_result_is_feasible():
    TF = np.float64
    opt = O.PARSDMM_options(FL=TF, maxit=600, evol_rel_tol=10 * np.finfo(TF).eps)
    g = O.compgrid((1.0, 1.0), (24, 19))
    x = np.random.default_rng(14).standard_normal(24 * 19)
    Dz = O.get_TD_operator(g, "D_z", TF)[0]; TV = O.get_TD_operator(g, "TV", TF)[0]
    c = [O.set_definitions("bounds", "identity", 0.5 * x.min(), 0.5 * x.max(), ("matrix", "")),
         O.set_definitions("bounds", "D_z", 0.5 * (Dz @ x).min(), 0.5 * (Dz @ x).max(), ("matrix", "")),
         O.set_definitions("l1", "TV", 0.0, 0.5 * np.abs(TV @ x).sum(), ("matrix", ""))]
    P_sub, TD_OP, prop, AtA = _setup(c, g, opt)
    xo, log, l, y = O.PARSDMM(x.copy(), AtA, TD_OP, prop, P_sub, g, opt)
    for i in range(len(TD_OP) - 1):
        s = TD_OP[i] @ xo
        assert np.linalg.norm(P_sub[i](s.copy()) - s) / np.linalg.norm(s) <= 1.5 * float(opt.feas_tol)
    # log bookkeeping (src/PARSDMM.jl:261-278)
    it = len(log.obj)
    assert log.r_pri.shape == (it, 4) and log.set_feasibility.shape[1] == 3
    assert log.cg_it[0] == 0 and np.isnan(log.evol_x[0])      # zero start: rhs==0 => cg flag -9


def test_nearest_neighbour_index_rounds_half_up_on_the_julia_range():
    # exact halves of 1 + k (nc-1)/(nf-1) are representable in float64, so floor(x + 0.5) on the float expression
    # is the reference for the integer routine
    for nc in range(1, 20):
        for nf in range(1, 40):
            k = np.arange(nf)
            want = (np.floor(1 + k * (nc - 1) / (nf - 1) + 0.5) - 1).astype(np.int64) if nf > 1 and nc > 1 else np.zeros(nf, np.int64)
            got = O._nn_index(nc, nf)
            assert np.array_equal(got, want), (nc, nf)
            assert got.min() >= 0 and got.max() <= nc - 1


def test_resample_identity_and_coarsen_refine_shapes():
    rng = np.random.default_rng(5)
    a = rng.standard_normal(6 * 5 * 4)
    assert np.array_equal(O.resample_nn(a, (6, 5, 4), (6, 5, 4)), a)
    c = O.resample_nn(a, (6, 5, 4), (3, 3, 2))
    assert c.shape == (18,)
    A = a.reshape((6, 5, 4), order="F")
    assert c.reshape((3, 3, 2), order="F")[2, 2, 1] == A[5, 4, 3] and c[0] == a[0]      # end points map to end points


def test_constraint2coarse_scalings():
    cs = [O.set_definitions("l1", "TV", 0.0, 8.0, ("matrix", "")), O.set_definitions("l2", "identity", 0.0, 4.0, ("matrix", "")),
          O.set_definitions("cardinality", "identity", 0, 10 ** 9, ("matrix", "")), O.set_definitions("rank", "identity", 0, 99, ("matrix", "")),
          O.set_definitions("nuclear", "identity", 0.0, 2.7, ("matrix", ""))]
    out = O.constraint2coarse(cs, O.compgrid((1.0, 1.0), (10, 8)), 2)
    assert [c.max for c in out[:4]] == [2.0, 2.0, 80, 8] and abs(out[4].max - 1.0) < 1e-15


# ---- the remaining get_projector branches, pinned by the properties test/test_projectors.jl:58-330 checks -------------
def test_cardinality_per_fiber_and_slice_counts():
    rng = np.random.default_rng(11)
    for mode, k, view in ((("fiber", "x"), 7, lambda X, i, j: X[:, i, j]), (("fiber", "y"), 6, lambda X, i, j: X[i, :, j]),
                          (("fiber", "z"), 4, lambda X, i, j: X[i, j, :])):
        n = (10, 12, 9)
        x = rng.standard_normal(int(np.prod(n)))
        x0 = x.copy()
        O.project_cardinality_mode(x, k, n, mode)
        X, X0 = x.reshape(n, order="F"), x0.reshape(n, order="F")
        ax = {"x": 0, "y": 1, "z": 2}[mode[1]]
        dims = [d for a, d in enumerate(n) if a != ax]
        for i in range(dims[0]):
            for j in range(dims[1]):
                f, f0 = view(X, i, j), view(X0, i, j)
                assert np.count_nonzero(f) == k                                       # test_projectors.jl:70-80
                keep = np.argsort(-np.abs(f0), kind="stable")[:k]
                assert np.array_equal(f[keep], f0[keep])
    for d, k in (("x", 7), ("y", 6), ("z", 5)):
        n = (10, 12, 9)
        x = rng.standard_normal(int(np.prod(n)))
        O.project_cardinality_mode(x, k, n, ("slice", d))
        X = x.reshape(n, order="F")
        ax = {"x": 0, "y": 1, "z": 2}[d]
        assert all(np.count_nonzero(np.take(X, i, axis=ax)) == k for i in range(n[ax]))   # :83-93
    X = rng.standard_normal((50, 100))
    x = X.reshape(-1, order="F").copy()
    O.project_cardinality_mode(x, 11, (50, 100), ("fiber", "z"))
    assert all(np.count_nonzero(x.reshape((50, 100), order="F")[i, :]) == 11 for i in range(50))   # :64-67


def test_cardinality_tie_goes_to_the_earlier_entry():
    x = np.array([1.0, -2.0, 2.0, 0.5, 2.0, -2.0])
    O.project_cardinality_mode(x, 2, (6, 1), ("fiber", "x"))
    assert np.array_equal(x, [0, -2.0, 2.0, 0, 0, 0])


def test_rank_and_nuclear_per_slice_properties():
    rng = np.random.default_rng(12)
    n = (14, 12, 9)
    for d, r in (("x", 7), ("y", 6), ("z", 5)):
        x = rng.standard_normal(int(np.prod(n)))
        O.project_rank(x, r, n, ("slice", d))
        X = x.reshape(n, order="F")
        ax = {"x": 0, "y": 1, "z": 2}[d]
        assert all(np.linalg.matrix_rank(np.take(X, i, axis=ax)) == r for i in range(n[ax]))   # test_projectors.jl:145-156
        x = rng.standard_normal(int(np.prod(n)))
        O.project_nuclear(x, 1.234, n, ("slice", d))
        X = x.reshape(n, order="F")
        for i in range(n[ax]):
            nn = np.linalg.svd(np.take(X, i, axis=ax), compute_uv=False).sum()
            assert abs(nn - 1.234) < 1e-12                                              # :206-217
    for shp in ((30, 12), (12, 30), (20, 20)):
        X = rng.standard_normal(shp)
        nn = np.linalg.svd(X, compute_uv=False).sum()
        x = X.reshape(-1, order="F").copy()
        x0 = x.copy()
        O.project_nuclear(x, 1.1 * nn, shp)
        assert np.allclose(x, x0, rtol=1e-13, atol=1e-13)                               # :173-181 (untouched up to SVD round trip)
        O.project_nuclear(x, 0.5 * nn, shp)
        assert abs(np.linalg.svd(x.reshape(shp, order="F"), compute_uv=False).sum() - 0.5 * nn) < 1e-11   # :183-204


def test_subspace_closed_forms():
    rng = np.random.default_rng(13)
    M = rng.standard_normal((40, 12))
    U = np.linalg.svd(M, full_matrices=False)[0]
    x = rng.standard_normal(40); y = x.copy()
    assert np.allclose(O.project_subspace(x, U, True), U @ (U.T @ y), rtol=1e-13)        # test_projectors.jl:221-227
    x = y.copy()
    ref = M @ np.linalg.solve(M.T @ M, M.T @ y)
    assert np.allclose(O.project_subspace(x, M, False), ref, rtol=1e-12)                 # :229-234
    X = rng.standard_normal((40, 5)); x = X.reshape(-1, order="F").copy()
    O.project_subspace(x, M, False, (40, 5), ("fiber", "x"))
    assert np.allclose(x.reshape((40, 5), order="F"), M @ np.linalg.solve(M.T @ M, M.T @ X), rtol=1e-12)   # :236-241
    M2 = rng.standard_normal((23, 8))
    X = rng.standard_normal((5, 23)); x = X.reshape(-1, order="F").copy()
    O.project_subspace(x, M2, False, (5, 23), ("fiber", "z"))
    assert np.allclose(x.reshape((5, 23), order="F"), (M2 @ np.linalg.solve(M2.T @ M2, M2.T @ X.T)).T, rtol=1e-12)   # :243-248
    for d, shp in (("z", (6, 7, 5)), ("y", (6, 5, 7)), ("x", (5, 6, 7))):
        Mb = rng.standard_normal((42, 4))
        X = rng.standard_normal(shp); x = X.reshape(-1, order="F").copy()
        O.project_subspace(x, Mb, False, shp, ("slice", d))
        Xp = x.reshape(shp, order="F")
        ax = {"x": 0, "y": 1, "z": 2}[d]
        for i in range(5):
            v = np.take(X, i, axis=ax).reshape(-1, order="F")
            assert np.allclose(np.take(Xp, i, axis=ax).reshape(-1, order="F"), Mb @ np.linalg.solve(Mb.T @ Mb, Mb.T @ v), rtol=1e-11)  # :250-272


def test_histogram_properties():
    rng = np.random.default_rng(14)
    ref = np.sort(rng.standard_normal(100))
    x = rng.standard_normal(100)
    O.project_histogram_relaxed(x, ref, ref)
    assert np.array_equal(np.sort(x), ref)                                             # test_projectors.jl:276-280
    LB = np.sort(rng.standard_normal(100)); UB = LB + 0.7
    x = rng.standard_normal(100); x0 = x.copy()
    O.project_histogram_relaxed(x, LB, UB)
    xs = np.sort(x)
    assert (xs <= UB).all() and (xs >= LB).all()                                       # :282-289


def test_bounds_per_fiber():
    rng = np.random.default_rng(15)
    n = (6, 5, 4)
    for d in ("x", "y", "z"):
        ax = {"x": 0, "y": 1, "z": 2}[d]
        LB = -rng.random(n[ax]); UB = rng.random(n[ax])
        x = 3 * rng.standard_normal(int(np.prod(n)))
        O.project_bounds_mode(x, LB, UB, n, ("fiber", d))
        X = np.moveaxis(x.reshape(n, order="F"), ax, 0)
        assert (X <= UB[:, None, None]).all() and (X >= LB[:, None, None]).all()
    with pytest.raises(ValueError):
        O.project_bounds_mode(np.zeros(120), np.zeros(4), np.ones(4), n, ("slice", "z"))
