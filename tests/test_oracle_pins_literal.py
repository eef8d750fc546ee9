"""The reference's own known-answer tests, LITERALLY: same sizes, same constants, and `==` where the reference asserts
`==` (isapprox(a, b, rtol=r) is Julia's norm-wise |a - b| <= r * max(|a|, |b|)).  Inputs the reference draws with `randn`
are drawn here from numpy (Julia's RNG stream is not reproducible outside Julia); every assertion below holds for ANY draw,
which is what makes it a pin.  tests/test_oracle_pins.py keeps the same facts at other sizes / precisions.

What stays UNPINNED by the reference (no fixture exists, and none can be generated: no Julia toolchain here or on the GPU
box): the VALUES of the Barzilai-Borwein rule of adapt_rho_gamma (the reference only tests that its serial and per-worker forms agree:
restated at the end of this file), the stop rules, and any iteration trace of PARSDMM.  For those the HIP
engine is compared with this restatement only ("parity unpinned" above the leaf functions)."""
import numpy as np
import scipy.sparse as sp

from oracle import parsdmm_oracle as O

EPS = np.finfo(np.float64).eps


def isapprox(a, b, rtol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) <= rtol * max(np.linalg.norm(a), np.linalg.norm(b))


# ---- test/test_TD_OPs.jl:4-40 -------------------------------------------------------------------------------------------
def test_TD_OPs_2d_literal():
    n1, n2, h1, h2 = 9, 6, 0.99, 1.123
    TF = np.float64
    D2D = O.get_discrete_Grad((n1, n2), (h1, h2), "TV", TF)
    D2x = O.get_discrete_Grad((n1, n2), (h1, h2), "D_x", TF)
    D2z = O.get_discrete_Grad((n1, n2), (h1, h2), "D_z", TF)
    x = np.zeros((n1, n2))                      # the 'cross' image: x[:,3] .= 1; x[4,:] .= 1 (1-based)
    x[:, 2] = 1.0
    x[3, :] = 1.0
    v = x.reshape(-1, order="F")
    a1 = (D2x @ v).reshape((n1 - 1, n2), order="F")
    a2 = (D2z @ v).reshape((n1, n2 - 1), order="F")
    a3 = D2D @ v
    a3a = a3[:(n2 - 1) * n1].reshape((n1, n2 - 1), order="F")
    a3b = a3[(n2 - 1) * n1:].reshape((n1 - 1, n2), order="F")
    assert np.array_equal(a1, np.diff(x, axis=0) / h1)                      # @test a1==diff(x, dims=1)./h1
    assert np.array_equal(a2, np.diff(x, axis=1) / h2)                      # @test a2==diff(x, dims=2)./h2
    assert np.count_nonzero(a1[:, 2]) == 0
    for i in (1, 2, 4, 5, 6):
        assert np.array_equal(a1[:, 0], a1[:, i - 1])
    assert np.count_nonzero(a2[3, :]) == 0
    for i in (1, 2, 3, 5, 6, 7, 8, 9):
        assert np.array_equal(a2[0, :], a2[i - 1, :])
    assert np.array_equal(a3a, a2) and np.array_equal(a3b, a1)              # TV = [D_z; D_x]


# ---- test/test_TD_OPs.jl:42-81 -----------------------------------------------------------------------------------------
def test_TD_OPs_3d_literal():
    n1, n2, n3, h1, h2, h3 = 4, 6, 5, 0.99, 1.123, 1.0
    TF = np.float64
    n, h = (n1, n2, n3), (h1, h2, h3)
    D3x, D3y, D3z = (O.get_discrete_Grad(n, h, k, TF) for k in ("D_x", "D_y", "D_z"))
    x = np.zeros(n)                             # x[2,:,:] .= 1; x[:,4,:] .= 1; x[:,:,3] .= 1 (1-based)
    x[1, :, :] = 1.0
    x[:, 3, :] = 1.0
    x[:, :, 2] = 1.0
    v = x.reshape(-1, order="F")
    a1 = (D3x @ v).reshape((n1 - 1, n2, n3), order="F")
    a2 = (D3y @ v).reshape((n1, n2 - 1, n3), order="F")
    a3 = (D3z @ v).reshape((n1, n2, n3 - 1), order="F")
    for i in range(n2):
        assert np.array_equal(a1[:, i, :], np.diff(x[:, i, :], axis=0) / h1)
    for i in range(n3):
        assert np.array_equal(a1[:, :, i], np.diff(x[:, :, i], axis=0) / h1)
    for i in range(n1):
        assert np.array_equal(a2[i, :, :], np.diff(x[i, :, :], axis=0) / h2)
    for i in range(n3):
        assert np.array_equal(a2[:, :, i], np.diff(x[:, :, i], axis=1) / h2)
    for i in range(n1):
        assert np.array_equal(a3[i, :, :], np.diff(x[i, :, :], axis=1) / h3)
    for i in range(n2):
        assert np.array_equal(a3[:, i, :], np.diff(x[:, i, :], axis=1) / h3)
    # D3D = [D_z; D_y; D_x] (get_discrete_Grad.jl:69-72)
    D3D = O.get_discrete_Grad(n, h, "TV", TF)
    assert np.array_equal(D3D @ v, np.concatenate([D3z @ v, D3y @ v, D3x @ v]))


# ---- test/test_CDS_scaled_add.jl:3-33 ------------------------------------------------------------------------------------
def test_CDS_scaled_add_literal_structured():
    TF = np.float64
    n1, n2 = 30, 20
    N = n1 * n2
    x = np.random.default_rng(101).standard_normal(N)
    g = O.compgrid((TF(25), TF(25)), (n1, n2))
    A = O.ata_ordered(O.get_TD_operator(g, "TV", TF)[0], TF)
    B = O.ata_ordered(O.get_TD_operator(g, "D_z", TF)[0], TF)
    C = sp.csc_matrix(A + B)
    R_A, offset_A = O.mat2CDS(A, TF)
    R_B, offset_B = O.mat2CDS(B, TF)
    R_C, offset_C = O.mat2CDS(C, TF)
    O.CDS_scaled_add(R_A, R_B, offset_A, offset_B, 1.0)
    assert np.array_equal(R_C, R_A)                                          # @test R_C==R_A
    assert np.array_equal(offset_C, offset_A)                                # @test offset_C==offset_A
    Cx_CDS = O.CDS_MVp(R_A, offset_A, x, np.zeros(N))
    assert isapprox(C @ x, Cx_CDS, 10 * EPS)                                 # @test isapprox(Cx_native,Cx_CDS,rtol=10*eps())


# ---- test/test_CDS_scaled_add.jl:36-62 ----------------------------------------------------------------------------------
def test_CDS_scaled_add_literal_random():
    rs = np.random.RandomState(102)
    A = sp.random(100, 100, 0.01, random_state=rs, data_rvs=rs.standard_normal, format="csc")
    B = sp.random(100, 100, 0.01, random_state=rs, data_rvs=rs.standard_normal, format="csc")
    x = rs.standard_normal(100)
    A = sp.csc_matrix(A + B)                                                 # make sure A has got all offsets of A and B
    R_A, offset_A = O.mat2CDS(A, np.float64)
    R_B, offset_B = O.mat2CDS(B, np.float64)
    C = sp.csc_matrix(A + B)
    O.CDS_scaled_add(R_A, R_B, offset_A, offset_B, 1.0)
    assert isapprox(C @ x, O.CDS_MVp(R_A, offset_A, x, np.zeros(100)), 10 * EPS)
    R_C, offset_C = O.mat2CDS(C, np.float64)
    assert np.array_equal(R_C, R_A) and np.array_equal(offset_C, offset_A)


# ---- test/test_cg.jl:5-29 ---------------------------------------------------------------------------------------------------
def test_cg_literal():
    rng = np.random.default_rng(103)
    A = rng.standard_normal((200, 100))
    A = A.T @ A
    xt = rng.standard_normal(100)
    b = A @ xt
    Af = lambda v: A @ v
    nrm = np.linalg.norm
    x, flag, relres, iter1 = O.cg(Af, b, 1e-5, 1000, np.zeros(100))
    assert nrm(A @ x - b) / nrm(b) <= 1e-5                                   # inexact
    x, flag, relres, it = O.cg(Af, b, 1e-14, 1000, np.zeros(100))
    assert nrm(A @ x - b) / nrm(b) <= 1.01e-14                               # very accurate solution can be achieved
    x, flag, relres, iter2 = O.cg(Af, b, 1e-5, 1000, xt.copy() + EPS)
    assert nrm(A @ x - b) / nrm(b) <= 1e-5 and iter2 < iter1                 # very good initial guess
    x, flag, relres, iter2 = O.cg(Af, b, 1e-14, 1000, xt.copy())
    assert nrm(A @ x - b) / nrm(b) <= 1e-14                                  # :27
    assert iter2 == 1                                                        # :28
    assert np.array_equal(x, xt)                                             # :29  @test x==xt


# ---- test/test_update_y_l.jl:8-90 (the Blas_active = false block; the oracle restates those formulas) --------------------------
def test_update_y_l_literal():
    rng = np.random.default_rng(123)
    TF = np.float64
    x = rng.standard_normal(100)
    p, i = 2, 10
    M = (51, 100)
    y = [rng.standard_normal(k) for k in M]
    y_old = [rng.standard_normal(k) for k in M]
    l_old = [rng.standard_normal(k) for k in M]
    l = [rng.standard_normal(k) for k in M]
    rho = np.array([1.234, 10.23432])
    gamma = np.array([1.0, 1.345])
    m = rng.standard_normal(100)
    prox = [lambda v: 1.0 * v, lambda v: O.prox_l2s(v, rho[1], m)]
    TD_OP = [sp.csc_matrix(sp.eye(51, 100, format="csc") * 2.0), sp.identity(100, format="csc")]
    maxit = 39

    class L: pass
    log = L()
    log.r_pri, log.r_dual, log.set_feasibility = np.zeros((maxit, p)), np.zeros((maxit, p)), np.zeros((maxit, p - 1))
    P_sub = [prox[0]]
    counter = 12
    x_hat = [rng.standard_normal(k) for k in M]
    r_pri = [rng.standard_normal(k) for k in M]
    s = [rng.standard_normal(k) for k in M]
    # reference solution (:61-70)
    y2, l2 = [v.copy() for v in y], [v.copy() for v in l]
    x_hat2 = [None, None]
    for k in range(p):
        x_hat2[k] = gamma[k] * (TD_OP[k] @ x) + (1 - gamma[k]) * y2[k]
        y2[k] = prox[k](x_hat2[k] - l2[k] / rho[k])
        l2[k] = l2[k] + rho[k] * (y2[k] - x_hat2[k])
    y_in, l_in = [v.copy() for v in y], [v.copy() for v in l]
    counter = O.update_y_l(x, p, i, y, y_old, l, l_old, rho, gamma, prox, TD_OP, log, P_sub, counter, x_hat, r_pri, s)
    nrm = np.linalg.norm
    for k in range(p):                                                       # :75-80, absolute 1e-14
        assert nrm(y[k] - y2[k]) <= 1e-14
        assert nrm(l[k] - l2[k]) <= 1e-14
        assert nrm(s[k] - TD_OP[k] @ x) <= 1e-14
        assert nrm(r_pri[k] - (-(TD_OP[k] @ x) + y[k])) <= 1e-14
    assert gamma[1] == 1.345 and gamma[0] == 1.0 and rho[0] == 1.234 and rho[1] == 10.23432     # :83-86
    for k in range(p):                                                       # :89-92
        assert nrm(y[k] - y_old[k]) > 10 * EPS and nrm(l[k] - l_old[k]) > 10 * EPS
        assert np.array_equal(y_old[k], y_in[k]) and np.array_equal(l_old[k], l_in[k])
    assert counter == 13                                                     # mod(i,10)==0: one feasibility row was written


# ---- test/test_Q_update.jl:3-63 -------------------------------------------------------------------------------------------
def test_Q_update_literal():
    rs = np.random.RandomState(104)
    sprandn = lambda: sp.random(100, 100, 0.1, random_state=rs, data_rvs=rs.standard_normal, format="csc")
    A = sprandn()
    B = [sprandn(), sprandn()]
    A = sp.csc_matrix(A + B[0] + B[1])          # all nonzero diagonals of B are also in A
    A3 = A.copy()
    rho = np.array([1.0, 1.0])

    class L: pass
    log = L()
    log.rho = np.zeros((100, 2))
    log.rho[0] = [2.0, 3.0]
    i = 0
    ind_updated = [k for k in range(2) if rho[k] != log.rho[i, k]]
    assert ind_updated == [0, 1]
    for k in ind_updated:                       # explicit solution (:26-30)
        A = A + B[k] * (rho[k] - log.rho[i, k])
    x = rs.standard_normal(100)
    # sparse matrix in CDS format (:46-59)
    R_A, offset_A = O.mat2CDS(A3, np.float64)
    cds = [O.mat2CDS(b, np.float64) for b in B]
    prop = O.set_properties(AtA_offsets=[c[1] for c in cds])
    O.Q_update(R_A, [c[0] for c in cds], prop, rho, ind_updated, log, i, offset_A)
    y = O.CDS_MVp(R_A, offset_A, x, np.zeros(100))
    assert isapprox(y, A @ x, 10 * EPS)                                      # @test isapprox(y,A*x,rtol=10*eps())


# ---- test/test_projectors.jl:5-56 -----------------------------------------------------------------------------------------
def test_projectors_literal_bounds_l1_cardinality():
    rng = np.random.default_rng(123)
    x = rng.standard_normal(100)
    lo, u = -0.11, 0.01
    O.project_bounds(x, lo, u)
    assert x.max() <= u and x.min() >= lo                                    # :9-10
    x = 100.0 * rng.standard_normal(100)
    lv, uv = rng.standard_normal(100) - 10.0, rng.standard_normal(100) + 10.0
    O.project_bounds(x, lv, uv)
    assert (x <= uv).all() and (x >= lv).all()                               # :17-18
    x = rng.standard_normal(100)
    tau = np.abs(x).sum() * 2
    y = x.copy()
    O.project_l1_Duchi(x, tau)
    assert np.array_equal(x, y)                                              # :24  untouched inside the ball
    x = rng.standard_normal(100)
    tau = np.abs(x).sum() * 0.234
    O.project_l1_Duchi(x, tau)
    assert abs(np.abs(x).sum() - tau) <= 10 * EPS * max(np.abs(x).sum(), tau)       # :28  isapprox(norm(x,1),tau,rtol=10*eps())
    x = rng.standard_normal(100)
    O.project_cardinality(x, 5)
    assert np.count_nonzero(x) == 5                                          # :45
    x = np.array([0, 0, 1, 2, 3])
    assert np.array_equal(O.project_cardinality(x.astype(np.float64), 2), [0, 0, 0, 2, 3])        # :48-50
    x = np.array([0, 0, -1, 2, -3])
    assert np.array_equal(O.project_cardinality(x.astype(np.float64), 2), [0, 0, 0, 2, -3])       # :53-55


# ---- test/test_projectors.jl:94-104 ---------------------------------------------------------------------------------------
def test_projectors_literal_l2():
    rng = np.random.default_rng(124)
    x = rng.standard_normal(100)
    O.project_l2(x, 0.123)
    assert abs(np.linalg.norm(x) - 0.123) <= 10 * EPS * max(np.linalg.norm(x), 0.123)           # :97
    x = rng.standard_normal(100)
    y = x.copy()
    O.project_l2(x, 1.234 * np.linalg.norm(x))
    assert np.array_equal(x, y)                                              # :103  @test x==y


# ---- test/test_prox_l2s!.jl:4-19 -----------------------------------------------------------------------------------------
def test_prox_l2s_literal():
    rng = np.random.default_rng(125)
    m, x = rng.standard_normal(10), rng.standard_normal(10)
    O.prox_l2s(x, 0.0, m)
    assert np.array_equal(x, m)                                              # :5-8   rho = 0: x == m
    y = x.copy()
    O.prox_l2s(x, 1e10, m)
    assert isapprox(x, y, 1e-14)                                             # :10-13 rho = 1e10: x stays
    x, m = np.array([2.0]), np.array([1.0])
    O.prox_l2s(x, 3.0, m)
    assert x[0] == 7 / 4                                                     # :15-19 (2*3 + 1) / (3 + 1)


# ---- test/test_rhs_compose.jl:1-38 ---------------------------------------------------------------------------------------
def test_rhs_compose_literal():
    """Same sizes and constants: y_1, l_1 in R^51000 behind TD_OP[1] = 2 * speye(51000, 100000), y_2, l_2 in R^100000 behind
    speye(100000), rho = [1.234, 10.23432].  The loop-fusion form (what the oracle restates, rhs_compose.jl:31-35) against
    the explicit-BLAS form (:24-30: temp = A'(rho y + l); axpy!(1, temp, rhs)) and against the parallel form (:17-20: the
    (+) reduction of the per-set terms), each to 10 eps as the reference asserts (:36, :49)."""
    TF = np.float64
    rng = np.random.default_rng(20240611)
    p, N = 2, 100000
    y = [rng.standard_normal(51000), rng.standard_normal(100000)]
    l = [rng.standard_normal(51000), rng.standard_normal(100000)]
    rho = np.array([1.234, 10.23432])
    TD_OP = [sp.eye(51000, 100000, format="csc", dtype=TF) * 2.0, sp.eye(100000, format="csc", dtype=TF)]
    rhs = O.rhs_compose(l, y, rho, TD_OP, p, N)                                   # Blas_active = false
    rhs_2 = np.zeros(N, TF)                                                       # Blas_active = true, restated literally
    for ii in range(p):
        temp = O.csc_mul_adj(TD_OP[ii], TF(rho[ii]) * y[ii] + l[ii])              # mul!(temp_array, TD_OP[ii]', rho[ii] .* y[ii] .+ l[ii])
        rhs_2 = TF(1.0) * temp + rhs_2                                            # BLAS.axpy!(TF(1.0), temp_array, rhs)
    assert isapprox(rhs, rhs_2, 10 * EPS)                                         # :36
    rhs_3 = np.zeros(N, TF)                                                       # parallel = true: @distributed (+) over ii
    for ii in range(p):
        rhs_3 = rhs_3 + O.rhs_compose(l, y, rho, TD_OP, p, N, only=[ii])
    assert isapprox(rhs, rhs_3, 10 * EPS)                                         # :49
    # and the closed form the two scaled identities admit, entry by entry
    want = rho[1] * y[1] + l[1]
    want[:51000] += 2.0 * (rho[0] * y[0] + l[0])
    assert isapprox(rhs, want, 10 * EPS)


# ---- test/test_argmin_x.jl:21-59 -----------------------------------------------------------------------------------------
def _spd_100(rng):
    """A = sprandn(100, 100, 0.01) + I; A = A'A; while rank(A) < 100: A += I   (test_argmin_x.jl:4-8, 37-41)."""
    A = sp.random(100, 100, density=0.01, random_state=rng, data_rvs=rng.standard_normal, format="csc") + sp.eye(100, format="csc")
    A = (A.T @ A).tocsc()
    while np.linalg.matrix_rank(A.toarray()) < 100:
        A = A + sp.eye(100, format="csc")
    return A


def test_argmin_x_literal():
    """Same sizes, tolerances and iteration numbers.  The reference test calls argmin_x with the CSC matrix (and, in its second
    half, passes CDS offsets along); the path this repository restates and replaces is the CDS branch (argmin_x.jl:23-39), so
    both halves run it on mat2CDS(A) -- the tolerance rule and cg are the same code for every storage format."""
    nrm = np.linalg.norm
    rng = np.random.default_rng(20240612)
    for half in range(2):
        A = _spd_100(rng)
        xt = rng.standard_normal(100)
        b = A @ xt
        R, off = O.mat2CDS(A)
        res = lambda x: nrm(A @ x - b) / nrm(b)
        if half == 0:
            x, it, relres, _ = O.argmin_x(R, b, np.zeros(100), 1e-5, 5, off)               # zero initial guess  :22-24
            assert res(x) <= 1e-5
            assert abs(res(x) - relres) <= 1e-5 * max(res(x), relres)                      # isapprox(..., relres, rtol = x_solve_tol_ref)
            x2, it2, relres2, _ = O.argmin_x(R, b, xt + rng.standard_normal(100) * 1e-4, 1e-5, 5, off)   # good initial guess  :27-30
            assert it2 < it
            assert res(x2) <= 1e-5
            assert abs(res(x2) - relres2) <= 1e-5 * max(res(x2), relres2)
            x, it, relres, _ = O.argmin_x(R, b, np.zeros(100), 10 * EPS, 15, off)          # 10 eps reachable  :33-34
            assert res(x) <= 20 * EPS
        else:
            x, it, relres, _ = O.argmin_x(R, b, np.zeros(100), 1e-5, 5, off)               # CDS  :46-50
            assert res(x) <= 1e-5 and res(x) <= 2.0 * relres
            x, it, relres, _ = O.argmin_x(R, b, np.zeros(100), 1e-10, 5, off)              # more accurate  :53-57
            assert res(x) <= 1e-10 and res(x) <= 2.0 * relres


# ---- test/test_update_y_l_parallel.jl:1-118 and test/test_adapt_rho_gamma_parallel.jl:1-141: one worker per set == the serial loop ------
def _parallel_leaf_inputs(rng):
    M = (51, 100)
    vec = lambda: [rng.standard_normal(k) for k in M]
    d = dict(x=rng.standard_normal(100), y=vec(), y_0=vec(), y_old=vec(), l_old=vec(), l=vec(), l_0=vec(), l_hat=vec(), l_hat_0=vec(),
             x_hat=vec(), r_pri=vec(), s=vec(), s_0=vec(), m=rng.standard_normal(100))
    d["rho"] = np.array([1.234, 10.23432])
    d["gamma"] = np.array([1.0, 1.345])
    d["TD_OP"] = [sp.csc_matrix(sp.eye(51, 100, format="csc") * 2.0), sp.identity(100, format="csc")]
    return d


def _bb_one_worker(TF, gamma, rho, adjust_gamma, adjust_rho, y, y_old, s, s_0, l, l_hat_0, l_0, l_old, y_0, l_hat):
    """src/adapt_rho_gamma_parallel.jl:30-127 restated on its own (one set: index [1] of the worker's local part), in that file's
    nesting -- the correlation test inside the reliability test -- which differs from the serial file the oracle follows."""
    eps_correlation = TF(0.3)
    safeguard = TF(1e-10) if TF == np.float64 else TF(1e-6)
    l_hat[:] = l_old + rho * (-s + y_old)
    d_l_hat, d_H_hat, d_l, d_G_hat = l_hat - l_hat_0, s - s_0, l - l_0, y_0 - y
    d_dHh_dlh, d_dGh_dl = TF(np.dot(d_H_hat, d_l_hat)), TF(np.dot(d_G_hat, d_l))
    n_d_H_hat, n_d_l_hat, n_d_l, n_d_G_hat = (TF(np.linalg.norm(v)) for v in (d_H_hat, d_l_hat, d_l, d_G_hat))
    alpha_comp = beta_comp = False
    if (n_d_H_hat * n_d_l_hat) > safeguard and (n_d_H_hat ** 2) > safeguard and d_dHh_dlh > safeguard:
        if d_dHh_dlh / (n_d_H_hat * n_d_l_hat) > eps_correlation:
            alpha_comp = True
            mg, sd = d_dHh_dlh / (n_d_H_hat ** 2), (n_d_l_hat ** 2) / d_dHh_dlh
            alpha_hat = mg if (TF(2.0) * mg) > sd else sd - mg / TF(2.0)
    if (n_d_G_hat * n_d_l) > safeguard and (n_d_G_hat ** 2) > safeguard and d_dGh_dl > safeguard:
        if d_dGh_dl / (n_d_G_hat * n_d_l) > eps_correlation:
            beta_comp = True
            mg, sd = d_dGh_dl / (n_d_G_hat ** 2), (n_d_l ** 2) / d_dGh_dl
            beta_hat = mg if (TF(2.0) * mg) > sd else sd - mg / TF(2.0)
    if adjust_rho:
        if alpha_comp and beta_comp:
            rho = TF(np.sqrt(alpha_hat * beta_hat))
        elif alpha_comp:
            rho = alpha_hat
        elif beta_comp:
            rho = beta_hat
    if adjust_gamma:
        if alpha_comp and beta_comp:
            gamma = TF(1.0) + ((TF(2.0) * TF(np.sqrt(alpha_hat * beta_hat))) / (alpha_hat + beta_hat))
        elif alpha_comp:
            gamma = TF(1.9)
        elif beta_comp:
            gamma = TF(1.1)
        else:
            gamma = TF(1.5)
    return rho, gamma, (alpha_comp, beta_comp)


def _bb_compare(d, adjust_gamma=True, adjust_rho=True):
    p, TF = 2, np.float64
    cp = lambda v: [a.copy() for a in v]
    # "distributed computation": every worker runs the rule on its own set
    rho_d, gamma_d, l_hat_d, branches = d["rho"].copy(), d["gamma"].copy(), cp(d["l_hat"]), []
    for k in range(p):
        rho_d[k], gamma_d[k], b = _bb_one_worker(TF, TF(gamma_d[k]), TF(rho_d[k]), adjust_gamma, adjust_rho, d["y"][k], d["y_old"][k], d["s"][k],
                                                 d["s_0"][k], d["l"][k], d["l_hat_0"][k], d["l_0"][k], d["l_old"][k], d["y_0"][k], l_hat_d[k])
        branches.append(b)
    # the same through the oracle's one-set entry (what the sharded oracle comparisons use) ...
    rho_w, gamma_w, l_hat_w = d["rho"].copy(), d["gamma"].copy(), cp(d["l_hat"])
    for k in range(p):
        O.adapt_rho_gamma(gamma_w, rho_w, adjust_gamma, adjust_rho, d["y"], d["y_old"], d["s"], d["s_0"], d["l"], d["l_hat_0"], d["l_0"],
                          d["l_old"], d["y_0"], p, l_hat_w, only=[k])
    # ... and "serial computation"
    rho, gamma, l_hat = d["rho"].copy(), d["gamma"].copy(), cp(d["l_hat"])
    O.adapt_rho_gamma(gamma, rho, adjust_gamma, adjust_rho, d["y"], d["y_old"], d["s"], d["s_0"], d["l"], d["l_hat_0"], d["l_0"], d["l_old"],
                      d["y_0"], p, l_hat)
    assert np.array_equal(rho_d, rho) and np.array_equal(gamma, gamma_d)                 # `@test rho_d==rho`, `@test gamma==gamma_d`
    assert np.array_equal(rho_w, rho) and np.array_equal(gamma_w, gamma)
    for k in range(p):
        assert isapprox(l_hat[k], l_hat_d[k], 10 * EPS) and np.array_equal(l_hat[k], l_hat_w[k])
        # "another related piece of code" (:121-139): l_hat = l_old + rho (-s + y_old), here with the rho the rule was entered with
        assert isapprox(d["l_old"][k] + d["rho"][k] * (-d["s"][k] + d["y_old"][k]), l_hat[k], 10 * EPS)
    return branches, rho, gamma


def test_adapt_rho_gamma_parallel_literal():
    d = _parallel_leaf_inputs(np.random.default_rng(123))
    branches, rho, gamma = _bb_compare(d)
    # independent draws are uncorrelated: neither step length is computed, rho stays, gamma falls back to 1.5
    assert branches == [(False, False)] * 2 and np.array_equal(rho, d["rho"]) and np.array_equal(gamma, [1.5, 1.5])


def test_adapt_rho_gamma_parallel_every_branch():
    """The same comparison on inputs built so that each combination of (alpha, beta) is computed: the differences of the
    snapshots are made to correlate (or not) by construction."""
    seen = set()
    for seed in range(40):
        rng = np.random.default_rng(1000 + seed)
        d = _parallel_leaf_inputs(rng)
        for k in range(2):
            ca, cb = (seed >> (2 * k)) & 1, (seed >> (2 * k + 1)) & 1
            if ca:       # l_hat - l_hat_0 along s - s_0:  l_hat = l_old + rho (y_old - s)
                d["l_hat_0"][k] = d["l_old"][k] + d["rho"][k] * (-d["s"][k] + d["y_old"][k]) - (0.5 + rng.random()) * (d["s"][k] - d["s_0"][k]) \
                    + 0.3 * rng.standard_normal(d["s"][k].size)
            if cb:       # l - l_0 along y_0 - y
                d["l_0"][k] = d["l"][k] - (0.5 + 2 * rng.random()) * (d["y_0"][k] - d["y"][k]) + 0.3 * rng.standard_normal(d["s"][k].size)
        for ag, ar in ((True, True), (False, True), (True, False)):
            branches, rho, gamma = _bb_compare(d, adjust_gamma=ag, adjust_rho=ar)
            seen.update(branches)
    assert seen == {(False, False), (True, False), (False, True), (True, True)}


def test_update_y_l_parallel_literal():
    rng = np.random.default_rng(123)
    d = _parallel_leaf_inputs(rng)
    p, i, maxit = 2, 10, 39
    rho, gamma, x, TD_OP = d["rho"], d["gamma"], d["x"], d["TD_OP"]
    prox = [lambda v: 1.0 * v, lambda v: O.prox_l2s(v, rho[1], d["m"])]
    P_sub = [prox[0]]

    class L: pass

    def fresh_log():
        g = L()
        g.r_pri, g.r_dual, g.set_feasibility = np.zeros((maxit, p)), np.zeros((maxit, p)), np.zeros((maxit, p - 1))
        return g
    cp = lambda v: [a.copy() for a in v]
    # reference solution (:86-98)
    y2, l2, x_hat2 = cp(d["y"]), cp(d["l"]), [None, None]
    for k in range(p):
        x_hat2[k] = gamma[k] * (TD_OP[k] @ x) + (1 - gamma[k]) * y2[k]
        y2[k] = prox[k](x_hat2[k] - l2[k] / rho[k])
        l2[k] = l2[k] + rho[k] * (y2[k] - x_hat2[k])
    rhs_ref = sum(TD_OP[k].T @ (l2[k] + y2[k]) for k in range(p))
    # one worker per set (:68-76) -- the oracle's one-set entry -- and the serial call (:101-107)
    outs = {}
    for name, groups in (("workers", [[0], [1]]), ("serial", [None])):
        y, l, y_old, l_old, x_hat, r_pri, s = (cp(d[k]) for k in ("y", "l", "y_old", "l_old", "x_hat", "r_pri", "s"))
        log, counter = fresh_log(), 12
        for only in groups:
            counter = O.update_y_l(x, p, i, y, y_old, l, l_old, rho, gamma, prox, TD_OP, log, P_sub, counter, x_hat, r_pri, s, only=only)
        outs[name] = dict(y=y, l=l, s=s, r_pri=r_pri, rhs=sum(TD_OP[k].T @ (l[k] + y[k]) for k in range(p)), log=log)
    a, b = outs["serial"], outs["workers"]
    assert isapprox(a["rhs"], rhs_ref, 10 * EPS) and isapprox(b["rhs"], rhs_ref, 10 * EPS)            # :109-110
    for k in range(p):                                                                                 # :111-113
        assert isapprox(a["s"][k], b["s"][k], 10 * EPS) and isapprox(a["l"][k], b["l"][k], 10 * EPS) and isapprox(a["r_pri"][k], b["r_pri"][k], 10 * EPS)
    assert np.array_equal(a["log"].r_pri, b["log"].r_pri) and np.array_equal(a["log"].set_feasibility, b["log"].set_feasibility)
