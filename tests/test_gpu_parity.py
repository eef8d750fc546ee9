"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bit-exact where the arithmetic is element-wise (stencils, CDS SpMV, Q assembly,
bounds / distance prox); stated floating-point tolerances where a reduction or the l1 threshold
is involved (the reference itself accepts rtol 5e-4 Float32 between its serial and parallel
paths, test/test_PARSDMM_parallel.jl:72, and 1e-12 Float64 between BLAS and loop code,
test/test_PARSDMM.jl:314)."""
import re

import numpy as np
import pytest

from oracle import parsdmm_oracle as O

pytestmark = pytest.mark.gpu

GRIDS = [((32, 24), (25.0, 6.0)), ((30, 21), (2.0, 3.0)), ((16, 12, 8), (25.0, 20.0, 10.0)),
         ((9, 7, 5), (1.0, 2.0, 4.0))]


def model(n, TF, seed=0):
    rng = np.random.default_rng(20240601 + seed)
    z = np.linspace(0, 1, n[-1]).reshape((1,) * (len(n) - 1) + (-1,))
    return (1500 + 2500 * z + 150 * rng.standard_normal(n)).astype(TF).reshape(-1, order="F")


def ops_for(n):
    return ["identity", "D_x", "D_z", "TV"] + (["D_y"] if len(n) == 3 else [])


def julia_max(v):
    v = np.asarray(v, np.float64)
    return float("nan") if np.isnan(v).any() else float(v.max())


# ---- K1: CDS SpMV (test/test_CDS_Mvp.jl) -----------------------------------------------------
@pytest.mark.parametrize("TF", [np.float32, np.float64])
@pytest.mark.parametrize("n", [(30, 20), (32, 24), (7, 5, 3)])
def test_cds_spmv_bitexact(sipx, TF, n):
    g = O.compgrid(tuple(TF(25) for _ in n), n)
    A = O.ata_ordered(O.get_TD_operator(g, "TV", TF)[0], TF)
    R, off = O.mat2CDS(A, TF)
    x = np.random.default_rng(1).standard_normal(A.shape[0]).astype(TF)
    ref = O.Ax_CDS(x, R, off)
    assert np.array_equal(sipx.cds_spmv(R, off, x), ref)
    perm = np.random.default_rng(2).permutation(len(off))         # band order = summation order
    assert np.array_equal(sipx.cds_spmv(R[:, perm], off[perm], x), O.Ax_CDS(x, np.asfortranarray(R[:, perm]), off[perm]))


def test_cds_spmv_random_bands(sipx):
    import scipy.sparse as sp
    A = sp.random(1000, 1000, 0.01, random_state=3, format="csc")   # 1000 % 4 == 0 but arbitrary offsets
    R, off = O.mat2CDS(A, np.float64)
    R, off = R[:, :32], off[:32]
    x = np.random.default_rng(4).standard_normal(1000)
    assert np.array_equal(sipx.cds_spmv(R, off, x), O.Ax_CDS(x, np.asfortranarray(R), off))


# ---- operators: forward / adjoint stencils vs the CSC products (test/test_TD_OPs.jl) ----------
@pytest.mark.parametrize("TF", [np.float32, np.float64])
@pytest.mark.parametrize("n,h", GRIDS)
def test_operators_bitexact(sipx, TF, n, h):
    go, gs = O.compgrid(h, n), sipx.compgrid(h, n)
    x = model(n, TF)
    for name in ops_for(n):
        Ao = O.get_TD_operator(go, name, TF)[0]
        As = sipx.get_TD_operator(gs, name, TF)[0]
        assert As.shape == Ao.shape
        s = O.csc_mul(Ao, x)
        assert np.array_equal(As @ x, s), name
        v = np.random.default_rng(5).standard_normal(Ao.shape[0]).astype(TF)
        assert np.array_equal(As.T @ v, O.csc_mul_adj(Ao, v)), name


# ---- Q assembly from device-generated AtA bands == oracle (PARSDMM_initialize.jl:216-230) ------
def _problem(mod, n, h, TF, kinds, m, opt_kw=None):
    g = mod.compgrid(h, n)
    opt = mod.PARSDMM_options(FL=TF, **(opt_kw or {}))
    c = []
    for k in kinds:
        if k == "bounds":
            c.append(mod.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", "")))
        elif k.startswith("l1:"):
            A = O.get_TD_operator(O.compgrid(h, n), k[3:], TF)[0]
            c.append(mod.set_definitions("l1", k[3:], 0.0, float(0.5 * np.abs(A @ m).sum()), ("matrix", "")))
        elif k.startswith("bnd:"):
            A = O.get_TD_operator(O.compgrid(h, n), k[4:], TF)[0]
            s = A @ m
            c.append(mod.set_definitions("bounds", k[4:], float(0.5 * s.min()), float(0.5 * s.max()), ("matrix", "")))
        elif k == "annulus":
            nm = float(np.linalg.norm(m.astype(np.float64)))
            c.append(mod.set_definitions("annulus", "identity", 0.9 * nm, 0.98 * nm, ("matrix", "")))
        elif k.startswith("card:"):
            A = O.get_TD_operator(O.compgrid(h, n), k[5:], TF)[0]
            c.append(mod.set_definitions("cardinality", k[5:], 0, int(0.3 * A.shape[0]), ("matrix", "")))
        elif k == "l1dft":
            Z = np.abs(np.fft.fftn(m.reshape(n, order="F").astype(np.float64), norm="ortho"))
            c.append(mod.set_definitions("l1", "DFT", 0.0, float(0.25 * Z.sum()), ("matrix", "")))
        elif k == "l1dct":
            import scipy.fft as sfft
            Z = np.abs(sfft.dctn(m.reshape(n, order="F").astype(np.float64), norm="ortho"))
            c.append(mod.set_definitions("l1", "DCT", 0.0, float(0.5 * Z.sum()), ("matrix", "")))
        elif k == "dftmask":                  # low-pass mask in the Fourier domain (symmetric, so the result stays real)
            f = np.meshgrid(*[np.minimum(np.arange(d), d - np.arange(d)) / (d / 2) for d in n], indexing="ij")
            keep = (sum(v ** 2 for v in f) <= 0.6 ** 2).astype(TF).reshape(-1, order="F")
            c.append(mod.set_definitions("bounds", "DFT", np.zeros(m.size, TF), keep, ("matrix", "")))
        elif k.startswith("rank:"):
            mode = ("matrix", "") if len(n) == 2 else ("slice", "z")
            c.append(mod.set_definitions("rank", "identity", 0, int(k[5:]), mode))
        elif k.startswith("l1id:"):           # l1 ball on the model itself: every entry stays active (lv-1 cap of the scan)
            c.append(mod.set_definitions("l1", "identity", 0.0, float(k[5:]) * float(np.abs(m.astype(np.float64)).sum()), ("matrix", "")))
        elif k == "l2":
            nm = float(np.linalg.norm(m.astype(np.float64)))
            c.append(mod.set_definitions("l2", "identity", 0.0, 0.9 * nm, ("matrix", "")))
        elif k.startswith("cardf:"):          # cardf:<op>:<fiber|slice>:<dir>: keep 30% of every fiber / slice
            _, opn, md, d = k.split(":")
            tdn = O.get_TD_operator(O.compgrid(h, n), opn, TF)[3]
            ax = {"x": 0, "y": 1, "z": len(n) - 1}[d]
            L = tdn[ax] if md == "fiber" else int(np.prod(tdn)) // tdn[ax]
            c.append(mod.set_definitions("cardinality", opn, 0, max(1, int(0.3 * L)), (md, d)))
        elif k.startswith("nuc:"):            # nuclear norm of every slice (or of the 2-D model): half the mean value
            d = k[4:]
            X = m.astype(np.float64).reshape(n, order="F")
            if len(n) == 2:
                sig, mode = 0.5 * np.linalg.svd(X, compute_uv=False).sum(), ("matrix", "")
            else:
                ax = {"x": 0, "y": 1, "z": 2}[d]
                sig = 0.5 * np.mean([np.linalg.svd(np.take(X, i, axis=ax), compute_uv=False).sum() for i in range(n[ax])])
                mode = ("slice", d)
            c.append(mod.set_definitions("nuclear", "identity", 0.0, float(sig), mode))
        elif k == "hist":                     # relaxed histogram of the model: sorted values of a smoother model +- 50
            ref = np.sort(0.5 * (m.astype(np.float64) + np.mean(m)))
            c.append(mod.set_definitions("histogram", "identity", (ref - 50).astype(TF), (ref + 50).astype(TF), ("matrix", "")))
        elif k.startswith("bndf:"):           # bounds per depth level (fiber along the last dimension)
            d = k[5:]
            ax = {"x": 0, "y": 1, "z": len(n) - 1}[d]
            lo = np.linspace(1600.0, 2200.0, n[ax]).astype(TF)
            c.append(mod.set_definitions("bounds", "identity", lo, (lo + 1500).astype(TF), ("fiber", d)))
        elif k.startswith("sub:"):            # subspace spanned by 6 smooth vectors, per fiber (2-D) / slice (3-D)
            d = k[4:]
            ax = {"x": 0, "y": 1, "z": len(n) - 1}[d]
            L = n[ax] if len(n) == 2 else int(np.prod(n)) // n[ax]
            t = np.linspace(0, 1, L)
            A = np.stack([np.cos(np.pi * q * t) for q in range(6)], axis=1).astype(TF)
            sd = mod.set_definitions("subspace", "identity", 0, 0, ("fiber" if len(n) == 2 else "slice", d))
            sd.custom_TD_OP = (A, False)
            c.append(sd)
    P, A, prop = mod.setup_constraints(c, g, TF)
    A, AtA, l, y = mod.PARSDMM_precompute_distribute(A, prop, g, opt)
    return g, opt, P, A, prop, AtA


@pytest.mark.parametrize("TF", [np.float32, np.float64])
@pytest.mark.parametrize("n,h", GRIDS)
def test_Q_assembly_bitexact(sipx, TF, n, h):
    m = model(n, TF)
    kinds = ["bounds", "l1:D_x", "l1:D_z", "l1:TV"]
    go, oo, Po, Ao, propo, AtAo = _problem(O, n, h, TF, kinds, m)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m)
    rho = [3.0, 0.5, 7.0, 11.0, 2.0]
    Qo, offo = O.assemble_Q(AtAo, propo.AtA_offsets, np.array(rho, TF), TF)
    os_.rho_ini = rho
    for explicit in (False, True):
        ctx = sipx.host.build_context(m, AtAo if explicit else AtAs, As, propo if explicit else props, Ps, gs, os_)
        Q, off = ctx.get_Q()
        assert np.array_equal(off, offo)
        assert np.array_equal(Q, Qo)
        rho_new = [3.0, 0.25, 7.0, 12.5, 1.0]
        ctx.q_update(rho_new, rho)
        Q2, _ = ctx.get_Q()
        ctx.close()

        class L: pass
        log = L(); log.rho = np.array([rho])
        Qr = O.Q_update(Qo.copy(order="F"), AtAo, propo, np.array(rho_new, TF), [1, 3, 4], log, 0, offo)
        assert np.array_equal(Q2, Qr)


def test_missing_diagonal_is_an_error(sipx):
    TF = np.float32
    n, h = (16, 12), (1.0, 1.0)
    m = model(n, TF)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, ["bounds"], m)
    ctx = sipx.host.build_context(m, AtAs, As, props, Ps, gs, os_)
    ctx.close()
    with pytest.raises(sipx.SipxError):       # l1 radius must be positive (project_l1_Duchi!.jl:22)
        c = [sipx.set_definitions("l1", "identity", 0.0, -1.0, ("matrix", ""))]
        P, A, prop = sipx.setup_constraints(c, gs, TF)
        A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, gs, os_)
        sipx.host.build_context(m, AtA, A, prop, P, gs, os_)


# ---- projectors (test/test_projectors.jl) ------------------------------------------------------
@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_projectors(sipx, TF):
    rng = np.random.default_rng(6)
    g = sipx.compgrid((1.0, 1.0), (10, 10))
    for n in (1000, 4099, 200000):
        v = (rng.standard_normal(n) * np.exp(rng.standard_normal(n))).astype(TF)
        # bounds: exact
        P = sipx.Projector(sipx.set_definitions("bounds", "identity", -0.5, 0.25, ("matrix", "")), g, TF)
        assert np.array_equal(P(v.copy()), O.project_bounds(v.copy(), TF(-0.5), TF(0.25)))
        lb = (rng.standard_normal(n) - 1).astype(TF); ub = lb + TF(0.5)
        P = sipx.Projector(sipx.set_definitions("bounds", "identity", lb, ub, ("matrix", "")), g, TF)
        assert np.array_equal(P(v.copy()), O.project_bounds(v.copy(), lb, ub))
        # l1 ball: same threshold up to the working precision; result on the sphere
        b = float(0.3 * np.abs(v.astype(np.float64)).sum())
        P = sipx.Projector(sipx.set_definitions("l1", "identity", 0.0, b, ("matrix", "")), g, TF)
        w = P(v.copy())
        ref = O.project_l1_Duchi(v.copy(), TF(b))
        tol = 2e-5 if TF == np.float32 else 1e-12
        assert np.linalg.norm(w - ref) <= tol * np.linalg.norm(ref)
        assert abs(np.abs(w.astype(np.float64)).sum() - b) <= tol * b
        assert np.array_equal(P(w.copy() * TF(0.5)), w * TF(0.5))           # feasible input untouched
        # exact fixed point in float64
        if TF == np.float64:
            a = np.abs(v); nz = w != 0
            theta = (a[nz] - np.abs(w[nz])).mean()
            assert abs(np.maximum(a - theta, 0).sum() - b) < 1e-9 * b
        # l2 / annulus
        nv = float(np.linalg.norm(v.astype(np.float64)))
        P = sipx.Projector(sipx.set_definitions("l2", "identity", 0.0, 0.5 * nv, ("matrix", "")), g, TF)
        assert np.allclose(P(v.copy()), O.project_l2(v.copy(), TF(0.5 * nv)), rtol=4 * np.finfo(TF).eps, atol=0)
        P = sipx.Projector(sipx.set_definitions("annulus", "identity", 2 * nv, 3 * nv, ("matrix", "")), g, TF)
        assert np.allclose(P(v.copy()), O.project_annulus(v.copy(), TF(2 * nv), TF(3 * nv)), rtol=4 * np.finfo(TF).eps, atol=0)
        z = P(np.zeros(n, TF))
        assert np.array_equal(z, O.project_annulus(np.zeros(n, TF), TF(2 * nv), TF(3 * nv)))
    # heavy ties at the threshold and a tiny vector
    v = np.array([3, 3, 3, 3, -3, 1, 0, 0], TF)
    P = sipx.Projector(sipx.set_definitions("l1", "identity", 0.0, 5.0, ("matrix", "")), g, TF)
    assert np.allclose(P(v.copy()), O.project_l1_Duchi(v.copy(), TF(5.0)), rtol=1e-6)


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_prox_l1(sipx, TF):
    """SIPX_PROJ_PROX_L1 == prox_l1!(x, rho): soft threshold at 1/rho (src/prox_l1!.jl:8-10, get_projector.jl:21-27)."""
    g = sipx.compgrid((1.0, 1.0), (10, 10))
    rng = np.random.default_rng(61)
    for n, rho in ((5, 2.0), (1000, 0.7), (4099, 3.0), (200000, 11.0)):
        v = (rng.standard_normal(n) * np.exp(rng.standard_normal(n))).astype(TF)
        v[::7] = 0
        P = sipx.Projector(sipx.set_definitions("prox_l1", "identity", 0.0, rho, ("matrix", "")), g, TF)
        assert np.array_equal(P(v.copy()), O.prox_l1(v.copy(), TF(rho))), (n, rho)
    # closed forms: threshold 1/rho = 1/2
    P = sipx.Projector(sipx.set_definitions("prox_l1", "identity", 0.0, 2.0, ("matrix", "")), g, TF)
    assert np.array_equal(P(np.array([3, -3, 0.25, -0.25, 0.5, 0], TF)), np.array([2.5, -2.5, 0, 0, 0, 0], TF))


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_prox_l2s_known_answers(sipx, TF):
    """test/test_prox_l2s!.jl:4-19 on the HIP path: rho = 0 returns m; (x, m, rho) = (2, 1, 3) gives 7/4; random inputs
    bit for bit against the oracle (numerator in TF, division in Float64)."""
    rng = np.random.default_rng(62)
    x, m = rng.standard_normal(100).astype(TF), rng.standard_normal(100).astype(TF)
    assert np.array_equal(sipx.prox_l2s(x.copy(), 0.0, m), m)                                  # :4-8
    assert np.array_equal(sipx.prox_l2s(np.full(10, 2, TF), 3.0, np.ones(10, TF)), np.full(10, 7 / 4, TF))   # :15-19
    for n, rho in ((1000, 0.3), (4099, 10.0), (200001, 1234.5)):
        x = (1500 + 900 * rng.standard_normal(n)).astype(TF)
        m = (1500 + 900 * rng.standard_normal(n)).astype(TF)
        assert np.array_equal(sipx.prox_l2s(x.copy(), rho, m), O.prox_l2s(x.copy(), TF(rho), m)), (n, rho)


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_cardinality_projector(sipx, TF):
    """test/test_projectors.jl:49-56 closed forms + stable tie breaking of sortperm(by=abs, rev=true)."""
    g = sipx.compgrid((1.0, 1.0), (10, 10))
    P = lambda k: sipx.Projector(sipx.set_definitions("cardinality", "identity", 0, k, ("matrix", "")), g, TF)
    assert np.array_equal(P(2)(np.array([0, 0, 1, 2, 3], TF)), [0, 0, 0, 2, 3])
    assert np.array_equal(P(2)(np.array([0, 0, -1, 2, -3], TF)), [0, 0, 0, 2, -3])
    assert np.array_equal(P(1)(np.array([1, -1, 1], TF)), [1, 0, 0])                 # ties: lowest index survives
    assert np.array_equal(P(3)(np.array([2, -2, 2, 2, -2, 1], TF)), [2, -2, 2, 0, 0, 0])
    assert np.array_equal(P(7)(np.array([1, 0, 2], TF)), [1, 0, 2])                  # k >= length
    assert np.array_equal(P(0)(np.array([1, 0, 2], TF)), [0, 0, 0])
    rng = np.random.default_rng(21)
    for n, k in ((1000, 100), (4099, 1), (200000, 60000), (300000, 299999)):
        v = (rng.standard_normal(n) * np.exp(rng.standard_normal(n))).astype(TF)
        v[rng.integers(0, n, n // 10)] = 0                                            # some exact zeros
        v[rng.integers(0, n, n // 20)] = TF(0.5)                                      # and a big tie group
        assert np.array_equal(P(k)(v.copy()), O.project_cardinality(v.copy(), k)), (n, k)


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_library_backed_projectors(sipx, TF):
    """DFT-folded l1 ball (hipFFT) and matrix / slice rank (rocSOLVER) against numpy."""
    rng = np.random.default_rng(31)
    tol = 2e-5 if TF == np.float32 else 1e-10
    for n in ((16, 12, 8), (32, 24)):
        g = sipx.compgrid(tuple(1.0 for _ in n), n)
        go = O.compgrid(tuple(1.0 for _ in n), n)
        v = rng.standard_normal(int(np.prod(n))).astype(TF)
        b = float(0.4 * np.abs(np.fft.fftn(v.reshape(n, order="F").astype(np.float64), norm="ortho")).sum())
        P = sipx.Projector(sipx.set_definitions("l1", "DFT", 0.0, b, ("matrix", "")), g, TF)
        Po = O.get_projector(O.set_definitions("l1", "DFT", 0.0, b, ("matrix", "")), TF, go)
        w, ref = P(v.copy()), Po(v.copy())
        assert np.linalg.norm(w - ref) <= tol * np.linalg.norm(ref)
        assert np.array_equal(P(w.copy() * TF(0.5)), w * TF(0.5)) or np.allclose(P(w.copy() * TF(0.5)), w * TF(0.5), rtol=0, atol=tol * np.abs(w).max())
        mode = ("matrix", "") if len(n) == 2 else ("slice", "z")
        P = sipx.Projector(sipx.set_definitions("rank", "identity", 0, 3, mode), g, TF)
        Po = O.get_projector(O.set_definitions("rank", "identity", 0, 3, mode), TF, go)
        w, ref = P(v.copy()), Po(v.copy())
        assert np.linalg.norm(w - ref) <= 20 * tol * np.linalg.norm(ref)
        W = w.reshape(n, order="F")
        for S in ([W] if W.ndim == 2 else [W[:, :, i] for i in range(n[2])]):
            assert np.linalg.matrix_rank(S.astype(np.float64), tol=1e-3 * np.linalg.norm(S)) <= 3


# ---- one phase-level iteration in lock-step with the oracle -------------------------------------
@pytest.mark.parametrize("TF", [np.float32, np.float64])
@pytest.mark.parametrize("n,h", GRIDS[:3])
def test_phases_lockstep(sipx, TF, n, h):
    m = model(n, TF)
    kinds = ["bounds", "bnd:D_z", "l1:TV"]
    go, oo, Po, Ao, propo, AtAo = _problem(O, n, h, TF, kinds, m)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m)
    rng = np.random.default_rng(7)
    p = len(Ao)
    N = len(m)
    # a random warm state
    x0 = (m + 50 * rng.standard_normal(N)).astype(TF)
    y0 = [O.csc_mul(Ao[i], x0) + (5 * rng.standard_normal(Ao[i].shape[0])).astype(TF) * TF(0.01) for i in range(p)]
    l0 = [(rng.standard_normal(Ao[i].shape[0])).astype(TF) for i in range(p)]
    rho = np.array([10.0, 3.0, 7.0, 2.0], TF); gamma = np.array([1.0, 1.3, 1.7, 1.0], TF)
    os_.zero_ini_guess = False
    os_.rho_ini = [float(r) for r in rho]
    ctx = sipx.host.build_context(m, AtAs, As, props, Ps, gs, os_, x=x0, l=l0, y=y0)
    # rhs_compose: bit exact (src/rhs_compose.jl:24-36; every entry is a short sum of products added in set order)
    ctx.rhs_compose(rho)
    rhs = O.rhs_compose(l0, y0, rho, Ao, p, N)
    assert np.array_equal(ctx.get_rhs(), rhs)
    # x-minimisation
    Qo, offo = O.assemble_Q(AtAo, propo.AtA_offsets, rho, TF)
    xo, it_o, relres_o, tol_o = O.argmin_x(Qo, rhs, x0.copy(), TF(1.0), 1, offo)
    tol, cg_it, relres, flag = ctx.argmin_x(1, 1.0)
    xs, _, _ = ctx.download(False)
    eps = np.finfo(TF).eps
    assert cg_it == it_o and flag == 0
    assert abs(tol - float(tol_o)) <= 8 * eps * abs(float(tol_o))
    assert np.linalg.norm(xs - xo) <= 50 * eps * np.linalg.norm(xo)
    assert abs(relres - float(relres_o)) <= 1e-3 * float(relres_o)
    # y/l update with BB first-iteration snapshots and feasibility
    prox = list(Po) + [lambda v: O.prox_l2s(v, rho[p - 1], m)]
    z = lambda: [np.zeros(Ao[i].shape[0], TF) for i in range(p)]
    y, l = [v.copy() for v in y0], [v.copy() for v in l0]
    y_old, l_old, x_hat, r_pri, s = z(), z(), z(), z(), z()

    class L: pass
    log = L(); log.r_pri = np.zeros((10, p)); log.r_dual = np.zeros((10, p)); log.set_feasibility = np.zeros((10, p - 1))
    O.update_y_l(xs.copy(), p, 10, y, y_old, l, l_old, rho, gamma, prox, Ao, log, Po, 2, x_hat, r_pri, s)
    rp, rd, fe = ctx.update_y_l(10, sipx.host.YL_FEAS | sipx.host.YL_FIRST, rho, gamma)
    _, ls, ys = ctx.download()
    obj, evol = ctx.log_scalars()
    rt = 2e-5 if TF == np.float32 else 1e-11
    for i in range(p):
        exact = i != 2          # set 2 is the l1 ball: threshold agrees to working precision only
        if exact:
            assert np.array_equal(ys[i], y[i]) and np.array_equal(ls[i], l[i]), i
        else:
            assert np.linalg.norm(ys[i] - y[i]) <= rt * np.linalg.norm(y[i])
            assert np.linalg.norm(ls[i] - l[i]) <= rt * max(np.linalg.norm(l[i]), 1)
    assert np.allclose(rp, log.r_pri[9], rtol=rt) and np.allclose(rd, log.r_dual[9], rtol=10 * rt, atol=1e-30)
    assert np.allclose(fe, log.set_feasibility[1], rtol=10 * rt, atol=1e-12)
    nd = O.nrm2(xs - m, TF)
    assert np.isclose(obj, float(TF(0.5) * TF(nd * nd)), rtol=4 * eps)
    assert np.isclose(evol, float(O.nrm2(x0 - xs, TF) / O.nrm2(xs, TF)), rtol=1e-4)
    # second iteration with BB sums: compare the adapted rho/gamma with the oracle rule
    l_hat, l_hat_0, y_0, s_0, l_0 = z(), z(), z(), z(), z()
    for ii in range(p):
        l_hat[ii][:] = l_old[ii] + TF(rho[ii]) * (-s[ii] + y_old[ii])
        l_hat_0[ii][:] = l_hat[ii]; y_0[ii][:] = y[ii]; s_0[ii][:] = s[ii]; l_0[ii][:] = l[ii]
    ctx.rhs_compose(rho)
    rhs2 = O.rhs_compose(ls, ys, rho, Ao, p, N)          # from the engine's own y, l (the l1 set agrees to rounding only)
    assert np.array_equal(ctx.get_rhs(), rhs2)
    rhs2 = O.rhs_compose(l, y, rho, Ao, p, N)
    x2, *_ = O.argmin_x(Qo, rhs2, xs.copy(), TF(tol), 2, offo)
    ctx.argmin_x(2, tol)
    x2s, _, _ = ctx.download(False)
    assert np.linalg.norm(x2s - x2) <= 1e-4 * np.linalg.norm(x2)
    O.update_y_l(x2s.copy(), p, 2, y, y_old, l, l_old, rho, gamma, prox, Ao, log, Po, 3, x_hat, r_pri, s)
    ctx.update_y_l(2, sipx.host.YL_BB, rho, gamma)
    rho_o, gam_o = rho.copy(), gamma.copy()
    O.adapt_rho_gamma(gam_o, rho_o, True, True, y, y_old, s, s_0, l, l_hat_0, l_0, l_old, y_0, p, l_hat)
    rho_s, gam_s = ctx.adapt_rho_gamma(True, True, rho, gamma)
    ctx.close()
    # the rule divides sums that agree to the order of summation (float64 accumulation on both sides)
    rt_bb = 5e-5 if TF == np.float32 else 1e-10
    assert np.allclose(rho_s, rho_o, rtol=rt_bb) and np.allclose(gam_s, gam_o, rtol=rt_bb), (rho_s, rho_o, gam_s, gam_o)


# ---- whole solve --------------------------------------------------------------------------------
C4_KINDS = ["bounds", "l1:D_x", "l1:D_y", "l1:D_z", "annulus", "l1dft", "rank:3", "card:D_z"]
CASES = [
    ("c1-2d-bounds-tv", (32, 24), (25.0, 6.0), ["bounds", "l1:TV"]),
    ("2d-bounds-dz-tv", (40, 28), (1.0, 1.0), ["bounds", "bnd:D_z", "l1:TV"]),
    ("c3-3d-bounds-l1xyz", (16, 12, 8), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"]),
    ("3d-odd-tv-annulus", (9, 7, 5), (25.0, 25.0, 25.0), ["bounds", "l1:TV", "annulus"]),
    ("2d-nonconvex-cardinality", (32, 24), (1.0, 1.0), ["bounds", "card:D_z"]),
    ("3d-dft-l1", (16, 12, 8), (25.0, 25.0, 25.0), ["bounds", "l1dft"]),
    ("3d-slice-rank", (16, 12, 8), (25.0, 25.0, 25.0), ["bounds", "rank:3"]),
    ("3d-dft-lowpass-mask", (16, 12, 8), (25.0, 25.0, 25.0), ["bounds", "dftmask", "l1:D_z"]),
    ("3d-dct-l1", (16, 12, 8), (25.0, 25.0, 25.0), ["bounds", "l1dct"]),
    ("2d-rank", (32, 24), (25.0, 6.0), ["bounds", "rank:4", "l1:TV"]),
    ("2d-l1-identity-all-active", (32, 24), (25.0, 6.0), ["bounds", "l1id:0.9"]),
    ("3d-card-fiber-Dz", (16, 12, 8), (25.0, 25.0, 25.0), ["bounds", "cardf:D_z:fiber:x"]),
    ("3d-card-slice-Dy", (16, 12, 8), (25.0, 25.0, 25.0), ["bounds", "cardf:D_y:slice:z"]),
    ("3d-nuclear-slice-x", (16, 12, 8), (25.0, 25.0, 25.0), ["bounds", "nuc:x", "l1:D_z"]),
    ("2d-nuclear", (32, 24), (25.0, 6.0), ["bounds", "nuc:"]),
    ("3d-histogram-bounds-fiber", (16, 12, 8), (25.0, 25.0, 25.0), ["bndf:z", "hist"]),
    ("2d-subspace", (32, 24), (25.0, 6.0), ["bounds", "sub:z"]),
    ("3d-subspace-slice", (16, 12, 8), (25.0, 25.0, 25.0), ["bounds", "sub:x", "l1:D_z"]),
    # BASELINE configs[3] (C4): the full 8-set list, two of the sets non-convex
    ("c4-3d-eight-sets", (16, 12, 8), (25.0, 25.0, 25.0), C4_KINDS),
]


# (case, dtype) -> (bound on the relative end-point difference, why).  Only cases whose rho traces separate are looked
# up here; every other case is held to 5e-4 (Float32) / 1e-6 (Float64).
DOCUMENTED_EXCEPTIONS = {
    # Two non-convex sets (slice rank, cardinality): the Float32 rho traces separate at iteration 16 (a BB threshold
    # flip), after which the cardinality projector keeps a different support in the two runs -- the problem has no
    # unique solution and 60 iterations do not converge it.  Measured 3.9e-3.  In Float64 the same case stays in
    # lock-step with the oracle for all 60 iterations and ends within 1e-6 (held to that by this very test).
    ("c4-3d-eight-sets", "f32"): (1e-2, "non-convex sets, support of the cardinality projector differs after the separation"),
}


def _record_separation(key, sep, n_engine, n_oracle, err):
    """Which parametrisations separate from the oracle's trace, and where: printed (pytest -rP / -s) and appended to
    gpurun_out/parity_separations.jsonl so the list in DESIGN.md can be checked against a run."""
    import json
    import os
    rec = {"case": key[0], "dtype": key[1], "first_iteration_with_different_rho_or_cg_it": None if sep is None else sep + 1,
           "iterations_engine": n_engine, "iterations_oracle": n_oracle, "rel_diff_x": float(err)}
    print("trace separation:", json.dumps(rec))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_separations.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("TF", [np.float32, np.float64])
@pytest.mark.parametrize("name,n,h,kinds", CASES)
def test_parsdmm_matches_oracle(sipx, TF, name, n, h, kinds):
    m = model(n, TF, seed=len(kinds))
    kw = dict(maxit=60)
    if any(k.startswith("sub:") for k in kinds):
        kw["feas_tol"] = 1e-3          # the smooth model is within the default 5e-2 of the subspace already
    go, oo, Po, Ao, propo, AtAo = _problem(O, n, h, TF, kinds, m, kw)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m, kw)
    xo, lo, l_o, y_o = O.PARSDMM(m.copy(), AtAo, Ao, propo, Po, go, oo)
    xs, ls, l_s, y_s = sipx.PARSDMM(m.copy(), AtAs, As, props, Ps, gs, os_)
    # the first iterations run in lock-step (before threshold flips of the BB rule can separate the traces)
    K = min(6, len(lo.obj), len(ls.obj))
    rt = 5e-4 if TF == np.float32 else 1e-8
    assert np.array_equal(ls.cg_it[:K], lo.cg_it[:K])
    for f in ("obj", "r_pri_total", "r_dual_total", "rho", "gamma"):
        a, b = np.asarray(getattr(ls, f))[:K], np.asarray(getattr(lo, f))[:K]
        assert np.allclose(a, b, rtol=rt, atol=1e-12), (f, a, b)
    assert len(ls.obj) == len(lo.obj) or min(len(ls.obj), len(lo.obj)) > 6
    if len(ls.obj) > 1:
        assert np.isnan(ls.evol_x[0]) and ls.cg_it[0] == 0         # zero start: rhs == 0 -> cg flag -9
    assert np.array_equal(ls.set_feasibility[0], lo.set_feasibility[0]) or \
        np.allclose(ls.set_feasibility[0], lo.set_feasibility[0], rtol=rt)
    # solution level: the reference's own serial/parallel tolerance for Float32, tighter for Float64
    err = np.linalg.norm(xs.astype(np.float64) - xo) / np.linalg.norm(xo)
    tol64 = 1e-6
    if any(k.startswith("sub:") for k in kinds):
        # once y lies in the subspace, P(s) - s is pure GEMM rounding noise and the BB ratios built from it differ between
        # two correct implementations; the traces agree for the first iterations (above) and the solutions to 1e-3
        tol64 = 1e-3
    # Traces are compared up to the first iteration at which the two rho histories differ (a threshold flip of the BB
    # rule: once a set is exactly feasible its multiplier is a rounding residue and sum(dG*dl) / (|dG| |dl|) is the
    # correlation of noise, so which side of eps_correlation it falls on depends on the summation order of the
    # reductions -- in the reference as much as here).  After a separation the end points must still agree to the
    # reference's own serial-vs-parallel tolerance (test/test_PARSDMM_parallel.jl:72), unless the case is listed in
    # DOCUMENTED_EXCEPTIONS with the reason and its own measured bound.
    Kc = min(len(ls.obj), len(lo.obj))
    # (Float64: a rho history counts as separated at the tolerance the lock-step comparison below asserts -- the eight-set list with
    #  the l1-DFT set through the REAL transform, round 4, has one rho 1.4e-6 off at iteration 60: the BB ratio of a set whose
    #  multiplier is FFT rounding noise; with 1e-5 here that fell between "separated" and "in lock step")
    sep_rt = 1e-5 if TF == np.float32 else 1e-6
    sep = next((k for k in range(Kc) if ls.cg_it[k] != lo.cg_it[k] or not np.allclose(ls.rho[k], lo.rho[k], rtol=sep_rt)), None)
    upto = Kc if sep is None else sep
    for f in ("obj", "r_pri_total", "rho", "gamma"):      # lock-step up to the first separation, at the reference's own tolerance
        a, b = np.asarray(getattr(ls, f))[:upto], np.asarray(getattr(lo, f))[:upto]
        assert np.allclose(a, b, rtol=(5e-4 if TF == np.float32 else 1e-6), atol=1e-12), (f, upto)
    key = (name, "f32" if TF == np.float32 else "f64")
    tol = 5e-4 if TF == np.float32 else tol64
    if sep is not None or len(ls.obj) != len(lo.obj):
        # PRIMARY check after a separation: the oracle again with the ENGINE's rho / gamma history forced on it (replay), so
        # that a flipped threshold of the BB rule cannot separate the two -- the end points then agree to the reference's
        # serial-vs-parallel tolerance for EVERY case, the eight-set C4 list in Float32 included
        gr, orr, Pr, Ar, propr, AtAr = _problem(O, n, h, TF, kinds, m, dict(kw, maxit=len(ls.obj)))
        xr, lr, _, _ = O.PARSDMM(m.copy(), AtAr, Ar, propr, Pr, gr, orr, replay=(ls.rho, ls.gamma))
        err_replay = np.linalg.norm(xs.astype(np.float64) - xr) / np.linalg.norm(xr)
        assert err_replay < tol, (key, "replayed", sep, err_replay)
        # SECONDARY: the free-running oracle (its own rho history from the separation on)
        tol = max(tol, 5e-4)
        _record_separation(key, sep, len(ls.obj), len(lo.obj), err)
        if key in DOCUMENTED_EXCEPTIONS:
            tol = DOCUMENTED_EXCEPTIONS[key][0]
        assert julia_max(ls.set_feasibility[-1]) <= max(julia_max(lo.set_feasibility[-1]), float(os_.feas_tol))
    assert err < tol, (key, sep, err)
    # log bookkeeping (PARSDMM.jl:261-278)
    it = len(ls.obj)
    p = len(kinds) + 1
    assert ls.r_pri.shape == (it, p) and ls.set_feasibility.shape[1] == p - 1
    assert len(y_s) == p and [len(v) for v in y_s] == [len(v) for v in y_o]


def test_feasible_input_returned_untouched(sipx):
    TF = np.float64
    n, h = (20, 31), (1.0, 1.0)
    x = np.random.default_rng(12).standard_normal(20 * 31)
    g = sipx.compgrid(h, n)
    opt = sipx.PARSDMM_options(FL=TF)
    c = [sipx.set_definitions("bounds", "identity", float(x.min()), float(x.max()), ("matrix", ""))]
    P, A, prop = sipx.setup_constraints(c, g, TF)
    A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
    xo, log, l, y = sipx.PARSDMM(x.copy(), AtA, A, prop, P, g, opt)
    assert np.array_equal(xo, x) and len(log.obj) == 1 and log.set_feasibility.shape[0] == 1


@pytest.mark.parametrize("kind", ["bounds", "l1", "annulus"])
def test_single_identity_set_equals_projector(sipx, kind):
    """test/test_PARSDMM.jl:192-242 pattern: one set, identity operator => PARSDMM(m) == P_C(m)."""
    TF = np.float64
    g = sipx.compgrid((1.0, 1.0), (16, 12))
    opt = sipx.PARSDMM_options(FL=TF, maxit=400, feas_tol=1e-10, obj_tol=1e-10, evol_rel_tol=1e-12)
    m = np.random.default_rng(13).standard_normal(16 * 12)
    if kind == "bounds":
        c = sipx.set_definitions("bounds", "identity", -0.3, 0.4, ("matrix", "")); ref = np.clip(m, -0.3, 0.4)
    elif kind == "l1":
        b = 0.4 * np.abs(m).sum()
        c = sipx.set_definitions("l1", "identity", 0.0, b, ("matrix", "")); ref = O.project_l1_Duchi(m.copy(), b)
    else:
        nm = np.linalg.norm(m)
        c = sipx.set_definitions("annulus", "identity", 0.3 * nm, 0.5 * nm, ("matrix", "")); ref = m * 0.5
    P, A, prop = sipx.setup_constraints([c], g, TF)
    A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
    x, log, l, y = sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
    assert np.linalg.norm(x - ref) / np.linalg.norm(ref) < 1e-7


def test_converged_result_is_feasible(sipx):
    """test/test_PARSDMM.jl:77-89 at a size the oracle would need minutes for: property only."""
    TF = np.float32
    n, h = (256, 192), (25.0, 6.0)
    m = model(n, TF)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, ["bounds", "bnd:D_z", "l1:TV"], m,
                                            dict(maxit=500, evol_rel_tol=10 * np.finfo(TF).eps))
    x, log, l, y = sipx.PARSDMM(m.copy(), AtAs, As, props, Ps, gs, os_)
    for i in range(3):
        s = As[i] @ x
        f = np.linalg.norm(Ps[i](s.copy()).astype(np.float64) - s) / np.linalg.norm(s.astype(np.float64))
        assert f <= 1.5 * float(os_.feas_tol), (i, f)


# ---- sharded solve on the real engine: several ranks sharing the one GPU, the engine's collectives over gloo ----------
def _sharded_worker(rank, world, port, out, kinds, n, backend, mode, phase, decomp="sets", tf="f32"):
    import os
    import sys
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")          # the container hostname may not resolve
    import datetime
    kw = dict(device_id=torch.device("cuda", 0)) if backend == "nccl" else {}
    if backend == "nccl":
        torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180), **kw)
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        from __graft_entry__ import load_package
        sipx = load_package()
        from sipx import sharded
        TF = np.float32 if tf == "f32" else np.float64
        h = (25.0, 25.0, 25.0)[:len(n)]
        m = model(n, TF, seed=5)
        gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m, dict(maxit=40))
        x, log, l, y = sharded.PARSDMM_sharded(m.copy(), AtAs, As, props, Ps, gs, os_, dist=dist, device=0, comm_mode=mode,
                                               phase_driver=phase, decomp=decomp)
        owned = sharded.shard_sets(len(As), world, rank) if decomp == "sets" else [1] * len(As)
        np.savez(os.path.join(out, f"r{rank}.npz"), x=x, obj=log.obj, evol_x=log.evol_x, cg_it=log.cg_it, rho=log.rho, gamma=log.gamma,
                 r_pri=log.r_pri, r_dual=log.r_dual, feas=log.set_feasibility, cg_relres=log.cg_relres)
        np.savez(os.path.join(out, f"yl{rank}.npz"), **{f"y{i}": y[i] for i in range(len(y)) if owned[i]},
                 **{f"l{i}": l[i] for i in range(len(l)) if owned[i]})
    finally:
        dist.destroy_process_group()


# The pytest process holds the GPU too and a box allows 6 processes on it: 4 ranks at most.  The x-step runs on z-slabs:
# 16 planes over 2 / 4 ranks (even), over 3 ranks (6, 6, 4: ragged), 5 planes over 4 ranks (2, 2, 1 and an EMPTY slab), a
# 2-D grid (slabs of rows); 4 ranks on 3 terms: one rank owns no set (the 8-GPU / 5-term case in small).
SHARDED = [
    (2, ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], (32, 24, 16), False),
    (4, ["bounds", "l1:D_z"], (32, 24, 16), False),
    (3, ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], (32, 24, 16), False),
    (4, ["bounds", "l1:D_z", "l1:D_x"], (12, 10, 5), False),
    (2, ["l1:TV"], (32, 24, 16), False),              # test/test_PARSDMM_parallel.jl:13-66
    (2, ["l1dft"], (32, 24, 16), False),              # test/test_PARSDMM_parallel.jl:69-121
    (2, ["bounds", "l1:TV"], (64, 48), False),        # 2-D
    (2, ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], (32, 24, 16), True),      # the loop kept on the host, phase entry points
    # slice-wise rank / nuclear norm: the owner broadcasts v, EVERY rank factorises the slices of its slab, all-gather
    (2, ["bounds", "rank:3", "l1:D_z"], (32, 24, 16), False),
    (3, ["bounds", "rank:3", "l1:D_z"], (32, 24, 16), False),               # slabs of 6, 6 and 4 slices
    (4, ["bounds", "nuc:z"], (12, 10, 5), False),                           # 2, 2, 1 slices and a rank with none
    (3, ["bounds", "rank:8", "l1:D_z"], (128, 128, 6), False),             # slices large enough for the warm-started (filtered) subspace route, per rank
    (4, C4_KINDS, (16, 12, 8), False),                                      # BASELINE config 4's set list over 4 ranks
]


# The WHOLE iteration on z-slabs (sipx_set_decomp(SIPX_DECOMP_SLAB)): every rank holds every set; even, ragged (6, 6, 4) and
# empty slabs, a 2-D grid, TV (a z-block whose adjoint reads the recomputed plane below), the phase-level loop.
SLAB = [
    (2, ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], (32, 24, 16), False),
    (3, ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], (32, 24, 16), False),
    (4, ["bounds", "l1:D_z", "l1:D_x"], (12, 10, 5), False),
    (2, ["l1:TV"], (32, 24, 16), False),
    (2, ["bounds", "l1:TV"], (64, 48), False),
    (4, ["bounds", "annulus", "l1:D_z"], (32, 24, 16), False),
    (2, ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], (32, 24, 16), True),
]


# Round 5: the long lists inside the slab iteration.  The slice-wise rank / nuclear-norm set is projected by every rank on the
# z-slices of its own slab (no exchange); cardinality through a search of its own over the slab collectives (all-reduced probe
# counts, the pairs inside the final bracket all-gathered); a projector that needs the whole array (l1 behind the DFT) by an owner
# rank on the gathered v (two fan exchanges for that set); y, l of every set stay on the slabs.  Even, ragged and empty slabs,
# BASELINE config 4's list.
SLAB_LOOSE = [
    (2, ["bounds", "rank:3", "l1:D_z"], (32, 24, 16)),
    (3, ["bounds", "rank:3", "l1:D_z"], (32, 24, 16)),
    (4, ["bounds", "nuc:z"], (12, 10, 5)),
    (2, ["bounds", "l1dft"], (32, 24, 16)),                  # the slab-decomposed transform: 8 + 8 planes, 12 + 12 rows
    (3, ["bounds", "l1dft"], (20, 18, 10)),                  # ragged: 4 + 4 + 2 planes, 6 + 6 + 6 rows
    (4, ["l1dft", "bounds"], (12, 10, 5)),                   # 2 + 2 + 1 planes and an empty slab; rows 3 + 3 + 3 + 1
    (3, ["bounds", "card:D_z", "l1:D_x"], (32, 24, 16)),
    (4, ["bounds", "card:D_z"], (12, 10, 5)),
    (2, ["l1dft", "card:D_z", "bounds"], (32, 24, 16)),
    (2, ["bounds", "card:identity"], (64, 48)),             # 2-D, the identity
    (4, C4_KINDS, (16, 12, 8)),
    (3, ["bounds", "rank:8", "l1:D_z"], (128, 128, 6)),      # the warm-started (filtered) subspace route per rank
]


@pytest.mark.timeout(400)
@pytest.mark.parametrize("world,kinds,n", SLAB_LOOSE)
def test_slab_decomposed_long_lists(sipx, tmp_path, world, kinds, n):
    test_sharded_ranks_on_one_gpu(sipx, tmp_path, world, kinds, n, False, decomp="slab")


@pytest.mark.timeout(400)
@pytest.mark.parametrize("world,kinds,n", [(3, ["bounds", "card:D_z", "l1:D_x"], (32, 24, 16)), (4, ["bounds", "card:D_z"], (12, 10, 5))])
def test_slab_cardinality_through_an_owner_rank(sipx, tmp_path, monkeypatch, world, kinds, n):
    """The gathered form of a set on D_z (SIPX_SLAB_CARD_GATHER=1: cardinality projected by an owner rank on the whole array instead
    of the search through the slab collectives): one more plane of P(v) travels for the adjoint stencil; ragged and empty slabs."""
    monkeypatch.setenv("SIPX_SLAB_CARD_GATHER", "1")
    test_sharded_ranks_on_one_gpu(sipx, tmp_path, world, kinds, n, False, decomp="slab")


@pytest.mark.timeout(400)
def test_slab_dft_falls_back_when_the_all_to_all_fails_its_self_test(sipx, tmp_path, monkeypatch):
    """sipx_finalize tests the communicator's all-to-all on known data when a set needs it (the slab-decomposed DFT).  A failure on
    ONE rank (test hook) does not fail the context: the ranks agree and the set goes through an owner rank on all of them -- the
    same end point as the serial solve."""
    monkeypatch.setenv("SIPX_COMM_SELFTEST_FAIL", "alltoall:1")
    test_sharded_ranks_on_one_gpu(sipx, tmp_path, 2, ["bounds", "l1dft"], (32, 24, 16), False, decomp="slab")


@pytest.mark.timeout(400)
def test_slab_decomposed_runs_repeat_bit_for_bit(sipx, tmp_path):
    """Two runs of BASELINE config 4's list on four ranks: the same bits in x and in every log.  (Every collective protocol of the
    slab decomposition rests on all ranks taking the same decisions from the same state; round 5 found a set-up race -- a zero-fill
    on the null stream against a kernel on the non-blocking engine stream -- through runs that differed.)"""
    import os
    import torch.multiprocessing as mp
    res = []
    for rep in range(2):
        out = tmp_path / f"run{rep}"
        out.mkdir()
        mp.spawn(_sharded_worker, args=(4, 30900 + (os.getpid() % 1000) + rep, str(out), C4_KINDS, (16, 12, 8), "gloo", "torch", False, "slab", "f32"),
                 nprocs=4, join=True)
        res.append(np.load(out / "r0.npz"))
    for k in res[0].files:
        assert np.array_equal(res[0][k], res[1][k], equal_nan=True), k


@pytest.mark.timeout(400)
def test_slab_decomposed_rank_set_on_its_lane_is_bit_identical(sipx, tmp_path, monkeypatch):
    """Slab-decomposed, BASELINE config 4's list on three ranks (a ragged last slab): the slice-rank set -- the z-slices of a
    rank's own planes, no collective in its update -- runs on the lane (engine.cpp, lane_start / lane_join: a stream and a host
    thread of its own) beside the lock-step searches, the sweep and the decomposed transform.  Same kernels on the same
    operands: x, y, l and every log equal the in-turn run (SIPX_RANK_LANE=0) bit for bit.
    Reference: src/update_y_l_parallel.jl:6-90 (the sets are independent given x)."""
    import os
    import torch.multiprocessing as mp
    res = []
    for rep, lane in enumerate(("1", "0")):
        monkeypatch.setenv("SIPX_RANK_LANE", lane)
        out = tmp_path / f"lane{lane}"
        out.mkdir()
        mp.spawn(_sharded_worker, args=(3, 31900 + (os.getpid() % 1000) + rep, str(out), C4_KINDS, (16, 12, 8), "gloo", "torch", False, "slab", "f32"),
                 nprocs=3, join=True)
        res.append(np.load(out / "r0.npz"))
    assert len(res[0].files) > 3
    for k in res[0].files:
        assert np.array_equal(res[0][k], res[1][k], equal_nan=True), k


@pytest.mark.timeout(400)
@pytest.mark.parametrize("world,kinds,n", [(2, ["bounds", "l1dft"], (32, 24, 16)), (3, ["l1dft", "card:D_z", "bounds"], (20, 18, 10))])
def test_slab_dft_through_an_owner_rank(sipx, tmp_path, monkeypatch, world, kinds, n):
    """The gathered form of the l1-DFT set (SIPX_SLAB_DFT_GATHER=1: an owner rank projects the whole array on a stream of its own)
    instead of the slab-decomposed transform."""
    monkeypatch.setenv("SIPX_SLAB_DFT_GATHER", "1")
    test_sharded_ranks_on_one_gpu(sipx, tmp_path, world, kinds, n, False, decomp="slab")


@pytest.mark.timeout(400)
def test_slab_decomposed_in_float64(sipx, tmp_path):
    test_sharded_ranks_on_one_gpu(sipx, tmp_path, 3, ["bounds", "l1:D_x", "l1:D_z", "annulus"], (32, 24, 16), False, decomp="slab", tf="f64")


@pytest.mark.timeout(400)
@pytest.mark.parametrize("kinds,n", [(["bounds", "l1dft"], (15, 12, 10)), (["bounds", "l1dft", "l1:D_z"], (16, 12, 10))])
def test_slab_decomposed_dft_in_float64(sipx, tmp_path, kinds, n):
    """The slab-decomposed transform in Float64 (D2Z / Z2Z / Z2D plans), three ranks with a ragged last slab; odd n0 (the half
    spectrum then has no Nyquist column) and even."""
    test_sharded_ranks_on_one_gpu(sipx, tmp_path, 3, kinds, n, False, decomp="slab", tf="f64")


@pytest.mark.timeout(400)
def test_slab_decomposed_with_the_sampled_prediction(sipx, tmp_path, monkeypatch):
    """The sampled prediction of theta inside the slab-decomposed iteration (every rank samples its planes, one all-reduce of
    the histograms), forced on for a grid this small: same end point as the serial solve, identical on every rank."""
    monkeypatch.setenv("SIPX_L1_SAMPLE_RUNS", "96")
    test_sharded_ranks_on_one_gpu(sipx, tmp_path, 2, ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], (64, 48, 40), False, decomp="slab")


@pytest.mark.timeout(400)
@pytest.mark.parametrize("world,kinds,n,phase", SLAB)
def test_slab_decomposed_ranks_on_one_gpu(sipx, tmp_path, world, kinds, n, phase):
    """Serial == slab-decomposed to the reference's own serial-vs-parallel tolerance; every rank ends with identical x, y, l
    and logs (the sums are all-reduced: identical bits, identical decisions)."""
    test_sharded_ranks_on_one_gpu(sipx, tmp_path, world, kinds, n, phase, decomp="slab")


@pytest.mark.timeout(400)
@pytest.mark.parametrize("world,kinds,n,phase", SHARDED)
def test_sharded_ranks_on_one_gpu(sipx, tmp_path, world, kinds, n, phase, decomp="sets", tf="f32"):
    """Serial == sharded to the reference's own tolerance (test/test_PARSDMM_parallel.jl:72,121: 5e-4 on x); every rank ends
    with identical x and logs; r_dual is filled (the reference's parallel mode leaves it zero)."""
    import os
    import torch.multiprocessing as mp
    port = 29600 + (os.getpid() % 2000) + 11 * world
    mp.spawn(_sharded_worker, args=(world, port, str(tmp_path), kinds, n, "gloo", "torch", phase, decomp, tf), nprocs=world, join=True)
    r0 = np.load(tmp_path / "r0.npz")
    for r in range(1, world):
        r1 = np.load(tmp_path / f"r{r}.npz")
        for k in r0.files:
            assert np.array_equal(r0[k], r1[k], equal_nan=True), k
        if decomp == "slab":                      # y, l are complete (gathered) on every rank
            a, b = np.load(tmp_path / "yl0.npz"), np.load(tmp_path / f"yl{r}.npz")
            for k in a.files:
                assert np.array_equal(a[k], b[k], equal_nan=True), k
    TF = np.float32 if tf == "f32" else np.float64
    h = (25.0, 25.0, 25.0)[:len(n)]
    m = model(n, TF, seed=5)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m, dict(maxit=40))
    xs, ls, l_s, y_s = sipx.PARSDMM(m.copy(), AtAs, As, props, Ps, gs, os_)
    K = min(8, len(ls.obj), len(r0["obj"]))
    assert np.array_equal(r0["cg_it"][:K], ls.cg_it[:K])
    dft_slab = (decomp == "slab" and "l1dft" in kinds and len(n) == 3 and os.environ.get("SIPX_SLAB_DFT_GATHER") != "1" and
                not os.environ.get("SIPX_COMM_SELFTEST_FAIL", "").startswith("alltoall"))
    assert np.allclose(r0["obj"][:K], ls.obj[:K], rtol=5e-4) and np.allclose(r0["r_pri"][:K], ls.r_pri[:K], rtol=5e-4, atol=1e-12)
    assert np.allclose(r0["r_dual"][:K], ls.r_dual[:K], rtol=2e-3, atol=1e-10) and (r0["r_dual"][1:K] > 0).any()
    assert np.allclose(r0["evol_x"][1:K], ls.evol_x[1:K], rtol=5e-4) and np.allclose(r0["rho"][:K], ls.rho[:K], rtol=5e-4)
    assert np.allclose(r0["feas"][0], ls.set_feasibility[0], rtol=1e-5)
    # end point: the reference's serial-vs-parallel tolerance (test/test_PARSDMM_parallel.jl:72).  With a non-convex set (rank,
    # cardinality) the problem has no unique solution: the first iterations above agree, after a Barzilai-Borwein threshold
    # flip the two runs settle on neighbouring points (same documented bound as DOCUMENTED_EXCEPTIONS for the C4 list)
    ncvx = any(k.startswith(("rank:", "card")) for k in kinds)
    # (measured: 9e-4 for {bounds, rank, l1}, 2e-2 for the eight-set C4 list, whose annulus leaves the scale of x loose)
    assert np.linalg.norm(r0["x"] - xs) / np.linalg.norm(xs) < (5e-2 if ncvx else 5e-4)
    if ncvx:
        # PRIMARY check for the non-convex lists: the oracle with the sharded run's own rho / gamma history forced on it
        # (replay) -- no threshold flip of the BB rule can separate them, and the end points agree to the reference's 5e-4
        go, oo, Po, Ao, propo, AtAo = _problem(O, n, h, TF, kinds, m, dict(maxit=len(r0["obj"])))
        xr, _, _, _ = O.PARSDMM(m.copy(), AtAo, Ao, propo, Po, go, oo, replay=(r0["rho"], r0["gamma"]))
        assert np.linalg.norm(r0["x"] - xr) / np.linalg.norm(xr) < 5e-4
    if not ncvx and len(r0["obj"]) == len(ls.obj) and np.array_equal(r0["cg_it"], ls.cg_it):      # same trajectory: the owners' y, l too
        for r in range(world):
            yl = np.load(tmp_path / f"yl{r}.npz")
            for k in yl.files:
                i = int(k[1:])
                ref = (y_s if k[0] == "y" else l_s)[i]
                # a multiplier of a set that is not active is a rounding residue of size rho * eps * |y|: that is its scale
                scale = max(np.linalg.norm(ref), float(ls.rho.max()) * np.finfo(TF).eps * np.linalg.norm(y_s[i]))
                # (the slab-decomposed DFT rounds differently from hipFFT's 3-D plan of the serial run: x agrees to 1e-7, and a
                #  multiplier that is nothing but rho * (rounding of y - s) then differs by about its own size -- 1e-3 (Float32) to
                #  2e-2 (Float64) of rho * eps * |y| for the inactive bounds set beside the l1-DFT set, measured: allowed on top)
                noise = float(ls.rho.max()) * np.finfo(TF).eps * np.linalg.norm(y_s[i])
                assert np.linalg.norm(yl[k] - ref) <= 5e-4 * scale + (5e-2 * noise if dft_slab else 0.0), (r, k)


def test_sharded_one_rank_through_rccl(sipx, tmp_path):
    """The engine's native RCCL communicator (librccl looked up at run time, ncclUniqueId through torch.distributed) with
    a world of one: every collective of the sharded loop is issued; the result equals the plain solve bit for bit except
    for the obj / evol_x sums, which are taken over the slab in a separate pass."""
    import os
    import torch.multiprocessing as mp
    kinds, n = ["bounds", "l1:D_x", "l1:D_z"], (32, 24, 16)
    port = 31100 + (os.getpid() % 2000)
    mp.spawn(_sharded_worker, args=(1, port, str(tmp_path), kinds, n, "nccl", "rccl", False), nprocs=1, join=True)
    r0 = np.load(tmp_path / "r0.npz")
    TF = np.float32
    m = model(n, TF, seed=5)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, (25.0, 25.0, 25.0), TF, kinds, m, dict(maxit=40))
    xs, ls, _, _ = sipx.PARSDMM(m.copy(), AtAs, As, props, Ps, gs, os_)
    assert len(r0["obj"]) == len(ls.obj) and np.array_equal(r0["cg_it"], ls.cg_it)
    assert np.array_equal(r0["x"], xs) and np.array_equal(r0["r_pri"], ls.r_pri) and np.array_equal(r0["rho"], ls.rho)
    assert np.allclose(r0["obj"], ls.obj, rtol=1e-6)


# ---- multilevel (BASELINE config 5 pattern; parity unpinned in the reference, oracle == engine here) ----------
def test_resample_nn_matches_oracle(sipx):
    rng = np.random.default_rng(41)
    for nc, nf in (((7, 5), (13, 9)), ((16, 12, 8), (8, 6, 4)), ((9, 7, 5), (17, 13, 9)), ((4, 4, 4), (9, 9, 9)), ((5,), (5,))):
        a = rng.standard_normal(int(np.prod(nc)))
        assert np.array_equal(sipx.host.resample_nn(a, nc, nf), O.resample_nn(a, nc, nf))


@pytest.mark.parametrize("TF,n,h", [(np.float64, (32, 24), (25.0, 6.0)), (np.float32, (16, 16, 16), (25.0, 25.0, 25.0))])
def test_multilevel_matches_oracle(sipx, TF, n, h):
    from sipx import multilevel as ML
    m = model(n, TF, seed=9)

    def cons(mod):
        TV = O.get_TD_operator(O.compgrid(h, n), "TV", TF)[0]
        return [mod.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", "")),
                mod.set_definitions("l1", "TV", 0.0, float(0.5 * np.abs(TV @ m).sum()), ("matrix", ""))]
    oo = O.PARSDMM_options(FL=TF, maxit=40)
    Lo = O.setup_multi_level_PARSDMM(m, 2, 2, O.compgrid(h, n), cons(O), oo)
    xo, logo, lo, yo = O.PARSDMM_multi_level(m.copy(), *Lo[:5], oo)
    os_ = sipx.PARSDMM_options(FL=TF, maxit=40)
    Ls = ML.setup_multi_level_PARSDMM(m, 2, 2, sipx.compgrid(h, n), cons(sipx), os_)
    assert [tuple(g.n) for g in Ls[4]] == [tuple(g.n) for g in Lo[4]]
    xs, logs, ls, ys = ML.PARSDMM_multi_level(m.copy(), *Ls[:5], os_)
    err = np.linalg.norm(xs.astype(np.float64) - xo) / np.linalg.norm(xo)
    assert err < (5e-4 if TF == np.float32 else 1e-6), err
    assert [len(v) for v in ys] == [len(v) for v in yo]
    assert list(os_.rho_ini) == [10.0] or np.allclose(os_.rho_ini, 10.0)          # restored (PARSDMM_multi_level.jl:87)


@pytest.mark.parametrize("TF,n,h,levels", [(np.float64, (32, 24), (25.0, 6.0), 2), (np.float32, (16, 16, 16), (25.0, 25.0, 25.0), 2),
                                           (np.float64, (22, 18, 14), (25.0, 20.0, 10.0), 3)])
def test_multilevel_device_transfers_equal_host_transfers(sipx, TF, n, h, levels):
    """sipx_warm_start_from (x, l, y resampled from context to context on the device) == the round-1 path (download,
    resample through sipx_resample_nn and the host interpolate_y_l, upload at sipx_finalize): the same integer index
    arithmetic either way, so the warm starts and hence the solves agree bit for bit."""
    from sipx import multilevel as ML
    m = model(n, TF, seed=10)
    TV = O.get_TD_operator(O.compgrid(h, n), "TV", TF)[0]
    Dz = O.get_TD_operator(O.compgrid(h, n), "D_z", TF)[0]
    s = Dz @ m
    cons = lambda: [sipx.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", "")),
                    sipx.set_definitions("bounds", "D_z", float(0.5 * s.min()), float(0.5 * s.max()), ("matrix", "")),
                    sipx.set_definitions("l1", "TV", 0.0, float(0.5 * np.abs(TV @ m).sum()), ("matrix", ""))]
    out = []
    for host_path in (True, False):
        opt = sipx.PARSDMM_options(FL=TF, maxit=25)
        L = ML.setup_multi_level_PARSDMM(m, levels, 2, sipx.compgrid(h, n), cons(), opt)
        T = {}
        x, log, l, y = ML.PARSDMM_multi_level(m.copy(), *L[:5], opt, timings=T, host_transfers=host_path)
        assert len(T["levels"]) == levels and [v["grid"] for v in T["levels"]][-1] == list(n)
        out.append((x, log, l, y))
    (xa, la, l_a, y_a), (xb, lb, l_b, y_b) = out
    assert np.array_equal(xa, xb) and np.array_equal(la.obj, lb.obj) and np.array_equal(la.cg_it, lb.cg_it)
    assert all(np.array_equal(a, b) for a, b in zip(y_a, y_b)) and all(np.array_equal(a, b) for a, b in zip(l_a, l_b))


# ---- stencil form of Q (sipx_set_q_mode(SIPX_Q_STENCIL)): rounding-level agreement with CDS --------
@pytest.mark.parametrize("TF", [np.float32, np.float64])
@pytest.mark.parametrize("n,h", GRIDS)
def test_stencil_Q_matches_cds_Q(sipx, TF, n, h):
    m = model(n, TF)
    kinds = ["bounds", "l1:D_x", "l1:D_z", "l1:TV"]
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m)
    rho = [3.0, 0.5, 7.0, 11.0, 2.0]
    os_.rho_ini = rho
    x = np.random.default_rng(3).standard_normal(m.size).astype(TF)
    out = {}
    for mode in ("cds", "stencil"):
        os_.Q_mode = mode
        ctx = sipx.host.build_context(m, AtAs, As, props, Ps, gs, os_)
        y0 = ctx.apply_Q(x)
        ctx.q_update([3.0, 0.25, 7.0, 12.5, 1.0], rho)
        out[mode] = (y0, ctx.apply_Q(x))
        if mode == "stencil":
            with pytest.raises(sipx.SipxError, match="stores no bands"):
                ctx.get_Q()
        ctx.close()
    eps = np.finfo(TF).eps
    for a, b in zip(out["cds"], out["stencil"]):
        scale = np.abs(a).max()
        assert np.abs(a.astype(np.float64) - b).max() <= 16 * eps * scale


def test_stencil_Q_refuses_explicit_bands(sipx):
    TF = np.float32
    n, h = (16, 12), (1.0, 1.0)
    m = model(n, TF)
    go, oo, Po, Ao, propo, AtAo = _problem(O, n, h, TF, ["bounds", "l1:TV"], m)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, ["bounds", "l1:TV"], m)
    os_.Q_mode = "stencil"
    with pytest.raises(sipx.SipxError, match="descriptor-generated"):
        sipx.host.build_context(m, AtAo, As, propo, Ps, gs, os_)


@pytest.mark.parametrize("TF", [np.float32, np.float64])
@pytest.mark.parametrize("name,n,h,kinds", CASES[:4])
def test_parsdmm_stencil_mode_matches_oracle(sipx, TF, name, n, h, kinds):
    m = model(n, TF, seed=len(kinds))
    kw = dict(maxit=60)
    go, oo, Po, Ao, propo, AtAo = _problem(O, n, h, TF, kinds, m, kw)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m, kw)
    os_.Q_mode = "stencil"
    xo, lo, _, _ = O.PARSDMM(m.copy(), AtAo, Ao, propo, Po, go, oo)
    xs, ls, _, _ = sipx.PARSDMM(m.copy(), AtAs, As, props, Ps, gs, os_)
    K = min(4, len(lo.obj), len(ls.obj))
    rt = 1e-3 if TF == np.float32 else 1e-7
    for f in ("obj", "r_pri_total", "rho"):
        a, b = np.asarray(getattr(ls, f))[:K], np.asarray(getattr(lo, f))[:K]
        assert np.allclose(a, b, rtol=rt, atol=1e-12), (f, a, b)
    err = np.linalg.norm(xs.astype(np.float64) - xo) / np.linalg.norm(xo)
    assert err < (5e-4 if TF == np.float32 else 1e-6), err


def test_stencil_mode_agrees_with_cds_mode_when_every_block_has_work(sipx):
    # 128^3 / 4 = 2048 * 256 vectors: every block of the widest launch is active, so the CG partial sums of the two
    # modes must cover the same block range (regression: a 2048-block producer next to 1792-block producers)
    TF, n, h = np.float32, (128, 128, 128), (25.0, 25.0, 25.0)
    m = model(n, TF, seed=2)
    rows = {}
    for mode in ("cds", "stencil"):
        g = sipx.compgrid(h, n)
        c = [sipx.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", ""))]
        for k in ("D_x", "D_z"):
            s = sipx.get_TD_operator(g, k, TF)[0] @ m
            c.append(sipx.set_definitions("l1", k, 0.0, float(0.5 * np.abs(s.astype(np.float64)).sum()), ("matrix", "")))
        P, A, prop = sipx.setup_constraints(c, g, TF)
        opt = sipx.PARSDMM_options(FL=TF, maxit=12)
        opt.Q_mode = mode
        A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
        x, log, _, _ = sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
        rows[mode] = (x, log)
    (xc, lc), (xs, ls) = rows["cds"], rows["stencil"]
    assert np.isfinite(ls.obj).all() and ls.cg_it.max() < 50
    assert np.array_equal(lc.cg_it, ls.cg_it)
    assert np.linalg.norm(xc.astype(np.float64) - xs) / np.linalg.norm(xc.astype(np.float64)) < 1e-4


def test_multilevel_survives_a_feasible_coarse_level(sipx):
    # a coarse level that is already feasible returns through the early exit with an all-zero rho log row
    # (PARSDMM.jl:63-82); the next level must not start from rho = 0 (documented deviation in multilevel._carry_rho)
    from sipx import multilevel as ML
    TF, n, h = np.float64, (32, 32, 32), (25.0, 25.0, 25.0)
    rng = np.random.default_rng(5)
    m = (1500 + 2500 * np.linspace(0, 1, n[2])[None, None, :] + 150 * rng.standard_normal(n)).astype(TF).reshape(-1, order="F")
    res = {}
    for name, mod, ml in (("oracle", O, O), ("sipx", sipx, ML)):
        g = mod.compgrid(h, n)
        s = O.get_TD_operator(O.compgrid(h, n), "TV", TF)[0] @ m
        c = [mod.set_definitions("bounds", "identity", 1600.0, 3900.0, ("tensor", "")),
             mod.set_definitions("l1", "TV", 0.0, float(0.5 * np.abs(s).sum()), ("tensor", ""))]
        opt = mod.PARSDMM_options(FL=TF, maxit=8)
        L = ml.setup_multi_level_PARSDMM(m, 2, 2, g, c, opt)
        x, log, _, _ = ml.PARSDMM_multi_level(m.copy(), *L[:5], opt)
        res[name] = (x, log)
    (xo, lo), (xs, ls) = res["oracle"], res["sipx"]
    assert np.isfinite(xs).all() and (ls.rho[0] > 0).all() and ls.cg_it.max() < 100
    assert np.linalg.norm(xs - xo) / np.linalg.norm(xo) < 1e-6


# ---- remaining get_projector branches: fiber / slice modes, nuclear, histogram, subspace ----------
def _proj(sipx, st, n, TF, mn, mx, mode, custom=((), False)):
    c = sipx.set_definitions(st, "identity", mn, mx, mode)
    c.custom_TD_OP = custom
    return sipx.host.Projector(c, sipx.compgrid(tuple(1.0 for _ in n), n), TF)


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_cardinality_fiber_and_slice_modes(sipx, TF):
    rng = np.random.default_rng(21)
    cases = [((10, 12, 9), ("fiber", "x"), 7), ((10, 12, 9), ("fiber", "y"), 6), ((10, 12, 9), ("fiber", "z"), 4),
             ((10, 12, 9), ("slice", "x"), 7), ((10, 12, 9), ("slice", "y"), 6), ((10, 12, 9), ("slice", "z"), 5),
             ((50, 100), ("fiber", "x"), 7), ((50, 100), ("fiber", "z"), 11), ((300, 4), ("fiber", "x"), 290),
             ((8, 6, 5), ("fiber", "x"), 0), ((8, 6, 5), ("fiber", "z"), 5), ((40, 40, 3), ("slice", "z"), 1000)]
    for n, mode, k in cases:
        v = rng.standard_normal(int(np.prod(n))).astype(TF)
        v[rng.integers(0, v.size, v.size // 5)] = TF(0.5)          # plenty of exact ties
        v[rng.integers(0, v.size, v.size // 5)] = TF(-0.5)
        want = O.project_cardinality_mode(v.copy(), k, n, mode)
        got = _proj(sipx, "cardinality", n, TF, 0, k, mode)(v.copy())
        assert np.array_equal(got, want), (n, mode, k)


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_bounds_fiber_modes(sipx, TF):
    rng = np.random.default_rng(22)
    for n, d in (((6, 5, 4), "x"), ((6, 5, 4), "y"), ((6, 5, 4), "z"), ((9, 7), "x"), ((9, 7), "z")):
        ax = {"x": 0, "y": 1, "z": len(n) - 1}[d]
        LB = (-rng.random(n[ax])).astype(TF); UB = rng.random(n[ax]).astype(TF)
        v = (3 * rng.standard_normal(int(np.prod(n)))).astype(TF)
        want = O.project_bounds_mode(v.copy(), LB, UB, n, ("fiber", d))
        got = _proj(sipx, "bounds", n, TF, LB, UB, ("fiber", d))(v.copy())
        assert np.array_equal(got, want), (n, d)
    with pytest.raises(sipx.SipxError, match="per slice"):
        _proj(sipx, "bounds", (6, 5, 4), TF, np.zeros(4, TF), np.ones(4, TF), ("slice", "z"))(np.zeros(120, TF))
    with pytest.raises(sipx.SipxError, match="entries"):
        _proj(sipx, "bounds", (6, 5, 4), TF, np.zeros(3, TF), np.ones(3, TF), ("fiber", "z"))(np.zeros(120, TF))


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_rank_and_nuclear_slice_modes(sipx, TF):
    rng = np.random.default_rng(23)
    tol = 2e-5 if TF == np.float32 else 1e-11
    n = (14, 12, 9)
    for d, r in (("x", 7), ("y", 6), ("z", 5)):
        v = rng.standard_normal(int(np.prod(n))).astype(TF)
        want = O.project_rank(v.copy(), r, n, ("slice", d))
        got = _proj(sipx, "rank", n, TF, 0, r, ("slice", d))(v.copy())
        assert np.linalg.norm(got.astype(np.float64) - want) <= tol * np.linalg.norm(want), ("rank", d)
        want = O.project_nuclear(v.copy(), TF(3.5), n, ("slice", d))
        got = _proj(sipx, "nuclear", n, TF, 0.0, 3.5, ("slice", d))(v.copy())
        assert np.linalg.norm(got.astype(np.float64) - want) <= tol * np.linalg.norm(want), ("nuclear", d)
        ax = {"x": 0, "y": 1, "z": 2}[d]
        G = got.astype(np.float64).reshape(n, order="F")
        for i in range(n[ax]):
            assert abs(np.linalg.svd(np.take(G, i, axis=ax), compute_uv=False).sum() - 3.5) < 3.5 * 50 * tol
    for shp in ((30, 12), (12, 30), (20, 20)):
        X = rng.standard_normal(shp).astype(TF)
        nn = float(np.linalg.svd(X.astype(np.float64), compute_uv=False).sum())
        v = X.reshape(-1, order="F").copy()
        assert np.array_equal(_proj(sipx, "nuclear", shp, TF, 0.0, 1.1 * nn, ("matrix", ""))(v.copy()), v)   # inside: untouched
        got = _proj(sipx, "nuclear", shp, TF, 0.0, 0.5 * nn, ("matrix", ""))(v.copy())
        want = O.project_nuclear(v.copy(), TF(0.5 * nn), shp)
        assert np.linalg.norm(got.astype(np.float64) - want) <= tol * np.linalg.norm(want)
    # r = min(n1, n2): the projection is the identity (constraint2coarse clips r to min(n))
    v = rng.standard_normal(20 * 12).astype(TF)
    assert np.array_equal(_proj(sipx, "rank", (20, 12), TF, 0, 12, ("matrix", ""))(v.copy()), v)


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_histogram_projector(sipx, TF):
    rng = np.random.default_rng(24)
    for M in (100, 4097):
        LB = np.sort(rng.standard_normal(M)).astype(TF); UB = (LB + TF(0.7)).astype(TF)
        v = rng.standard_normal(M).astype(TF)
        v[:10] = v[10:20]                                  # ties: the stable order decides who gets which bound
        want = O.project_histogram_relaxed(v.copy(), LB, UB)
        got = _proj(sipx, "histogram", (M, 1), TF, LB, UB, ("matrix", ""))(v.copy())
        assert np.array_equal(got, want)
        ref = np.sort(rng.standard_normal(M)).astype(TF)
        got = _proj(sipx, "histogram", (M, 1), TF, ref, ref, ("matrix", ""))(rng.standard_normal(M).astype(TF))
        assert np.array_equal(np.sort(got), ref)           # test_projectors.jl:276-280


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_subspace_projector(sipx, TF):
    rng = np.random.default_rng(25)
    tol = 5e-5 if TF == np.float32 else 1e-11

    def close(a, b):
        return np.linalg.norm(a.astype(np.float64) - b) <= tol * max(np.linalg.norm(b), 1e-30)
    M = rng.standard_normal((40, 12)).astype(TF)
    U = np.linalg.svd(M.astype(np.float64), full_matrices=False)[0].astype(TF)
    v = rng.standard_normal(40).astype(TF)
    for A, orth in ((U, True), (M, False)):
        want = O.project_subspace(v.astype(np.float64), A.astype(np.float64), orth)
        assert close(_proj(sipx, "subspace", (40, 1), TF, 0, 0, ("matrix", ""), (A, orth))(v.copy()), want)
    X = rng.standard_normal((40, 5)).astype(TF)
    want = O.project_subspace(X.reshape(-1, order="F").astype(np.float64), M.astype(np.float64), False, (40, 5), ("fiber", "x"))
    assert close(_proj(sipx, "subspace", (40, 5), TF, 0, 0, ("fiber", "x"), (M, False))(X.reshape(-1, order="F").copy()), want)
    M2 = rng.standard_normal((23, 8)).astype(TF)
    X = rng.standard_normal((5, 23)).astype(TF)
    want = O.project_subspace(X.reshape(-1, order="F").astype(np.float64), M2.astype(np.float64), False, (5, 23), ("fiber", "z"))
    assert close(_proj(sipx, "subspace", (5, 23), TF, 0, 0, ("fiber", "z"), (M2, False))(X.reshape(-1, order="F").copy()), want)
    for d, shp in (("z", (6, 7, 5)), ("y", (6, 5, 7)), ("x", (5, 6, 7))):
        Mb = rng.standard_normal((42, 4)).astype(TF)
        X = rng.standard_normal(shp).astype(TF)
        want = O.project_subspace(X.reshape(-1, order="F").astype(np.float64), Mb.astype(np.float64), False, shp, ("slice", d))
        assert close(_proj(sipx, "subspace", shp, TF, 0, 0, ("slice", d), (Mb, False))(X.reshape(-1, order="F").copy()), want), d
    with pytest.raises(sipx.SipxError, match="rows of A"):
        _proj(sipx, "subspace", (40, 5), TF, 0, 0, ("fiber", "z"), (M, False))(np.zeros(200, TF))


# ---- Minkowski sets: x = [u; v], operators [A 0] / [0 A] / [A A]; parity unpinned in the reference
# (no test exercises PARSDMM_precompute_distribute_Minkowski.jl), oracle == engine here --------------------------------
def _minkowski_problem(mod, n, h, TF, m, maxit=40, feasibility_only=False):
    g = mod.compgrid(h, n)
    opt = mod.PARSDMM_options(FL=TF, maxit=maxit, feasibility_only=feasibility_only)
    opt.Minkowski = True
    Og = O.compgrid(h, n)
    TV = O.get_TD_operator(Og, "TV", TF)[0]
    Dz = O.get_TD_operator(Og, "D_z", TF)[0]
    smooth = np.sort(m)      # not used as data, only to scale the radii
    c1 = [mod.set_definitions("bounds", "identity", 1400.0, 4100.0, ("matrix", "")),
          mod.set_definitions("l1", "TV", 0.0, float(0.25 * np.abs(TV @ m).sum()), ("matrix", ""))]
    c2 = [mod.set_definitions("bounds", "identity", -300.0, 300.0, ("matrix", "")),
          mod.set_definitions("l1", "identity", 0.0, float(0.02 * np.abs(m).sum()), ("matrix", ""))]
    cs = [mod.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", "")),
          mod.set_definitions("bounds", "D_z", float(0.5 * (Dz @ m).min()), float(0.5 * (Dz @ m).max()), ("matrix", ""))]
    del smooth
    P1, A1, p1 = mod.setup_constraints(c1, g, TF)
    P2, A2, p2 = mod.setup_constraints(c2, g, TF)
    P3, A3, p3 = mod.setup_constraints(cs, g, TF)
    TD_OP, prop, AtA, l, y = mod.PARSDMM_precompute_distribute_Minkowski(A1, A2, A3, p1, p2, p3, g, opt)
    return g, opt, P1 + P2 + P3, TD_OP, prop, AtA


@pytest.mark.parametrize("TF", [np.float32, np.float64])
@pytest.mark.parametrize("n,h", [((32, 24), (25.0, 6.0)), ((12, 10, 8), (25.0, 25.0, 25.0)), ((9, 7, 2), (1.0, 1.0, 1.0))])
def test_minkowski_Q_bitexact(sipx, TF, n, h):
    m = model(n, TF, seed=4)
    go, oo, Po, Ao, propo, AtAo = _minkowski_problem(O, n, h, TF, m)
    gs, os_, Ps, As, props, AtAs = _minkowski_problem(sipx, n, h, TF, m)
    rho = [3.0, 0.5, 7.0, 11.0, 2.0, 1.5, 4.0]
    os_.rho_ini = rho
    assert all(np.array_equal(a, b) for a, b in zip(props.AtA_offsets, propo.AtA_offsets))
    Qo, offo = O.assemble_Q(AtAo, propo.AtA_offsets, np.array(rho, TF), TF)
    ctx = sipx.host.build_context(m, AtAs, As, props, Ps, gs, os_)
    Q, off = ctx.get_Q()
    assert np.array_equal(off, offo) and Q.shape == Qo.shape == (2 * m.size, len(offo))
    assert np.array_equal(Q, Qo)
    x = np.random.default_rng(1).standard_normal(2 * m.size).astype(TF)
    assert np.array_equal(ctx.apply_Q(x), O.Ax_CDS(x, Qo, offo))
    rho_new = [3.0, 0.25, 7.0, 12.5, 2.0, 1.0, 8.0]
    ctx.q_update(rho_new, rho)
    Q2, _ = ctx.get_Q()
    ctx.close()

    class L: pass
    log = L(); log.rho = np.array([rho])
    Qr = O.Q_update(Qo.copy(order="F"), AtAo, propo, np.array(rho_new, TF), [1, 3, 5, 6], log, 0, offo)
    assert np.array_equal(Q2, Qr)


@pytest.mark.parametrize("TF", [np.float32, np.float64])
@pytest.mark.parametrize("n,h", [((32, 24), (25.0, 6.0)), ((12, 10, 8), (25.0, 25.0, 25.0))])
def test_minkowski_parsdmm_matches_oracle(sipx, TF, n, h):
    m = model(n, TF, seed=4)
    go, oo, Po, Ao, propo, AtAo = _minkowski_problem(O, n, h, TF, m)
    gs, os_, Ps, As, props, AtAs = _minkowski_problem(sipx, n, h, TF, m)
    xo, lo, l_o, y_o = O.PARSDMM(m.copy(), AtAo, Ao, propo, Po, go, oo)
    xs, ls, l_s, y_s = sipx.PARSDMM(m.copy(), AtAs, As, props, Ps, gs, os_)
    assert xs.shape == xo.shape == (2 * m.size,)
    K = min(6, len(lo.obj), len(ls.obj))
    rt = 5e-4 if TF == np.float32 else 1e-8
    assert np.array_equal(ls.cg_it[:K], lo.cg_it[:K])
    for f in ("obj", "evol_x", "r_pri_total", "r_dual_total", "rho", "gamma"):
        a, b = np.asarray(getattr(ls, f))[1:K], np.asarray(getattr(lo, f))[1:K]
        assert np.allclose(a, b, rtol=rt, atol=1e-12), (f, a, b)
    assert np.allclose(ls.set_feasibility[0], lo.set_feasibility[0], rtol=rt)
    scale = np.linalg.norm(xo[:m.size] + xo[m.size:])
    err = np.linalg.norm(xs.astype(np.float64) - xo) / scale
    assert err < (5e-4 if TF == np.float32 else 1e-6), err
    assert [len(v) for v in y_s] == [len(v) for v in y_o]
    # warm restart from the result: x0 carries both components
    os_.zero_ini_guess = False; oo.zero_ini_guess = False
    os_.maxit = oo.maxit = 5
    x2o, l2o, _, _ = O.PARSDMM(m.copy(), AtAo, Ao, propo, Po, go, oo, xo.copy(), l_o, y_o)
    x2s, l2s, _, _ = sipx.PARSDMM(m.copy(), AtAs, As, props, Ps, gs, os_, xs.copy(), l_s, y_s)
    assert np.linalg.norm(x2s.astype(np.float64) - x2o) / scale < (1e-3 if TF == np.float32 else 1e-5)


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_l1_projector_when_every_entry_stays_active(sipx, TF):
    # project_l1_Duchi!.jl:42: the scan stops at lv-1, so theta = (||v||_1 - min|v| - b)/(lv-1) when nothing is zeroed
    rng = np.random.default_rng(31)
    for M in (5, 64, 1000):
        v = (rng.uniform(2.0, 3.0, M) * rng.choice([-1.0, 1.0], M)).astype(TF)
        b = 0.8 * float(np.abs(v.astype(np.float64)).sum())
        want = O.project_l1_Duchi(v.copy(), TF(b))
        c = sipx.set_definitions("l1", "identity", 0.0, b, ("matrix", ""))
        got = sipx.host.Projector(c, sipx.compgrid((1.0, 1.0), (M, 1)), TF)(v.copy())
        assert np.count_nonzero(want) >= M - 1
        assert np.allclose(got, want, rtol=(2e-5 if TF == np.float32 else 1e-12), atol=0), M


# ---- BASELINE.json full sizes: size-independent properties (no oracle run at these sizes) -------------------------------
def _c3_problem(sipx, n, TF, q_mode="cds", maxit=12, rho=None):
    h = (25.0,) * len(n)
    rng = np.random.default_rng(20240601 + 3)
    z = np.linspace(0.0, 1.0, n[-1]).reshape((1,) * (len(n) - 1) + (-1,))
    m = (1500.0 + 2500.0 * z + 150.0 * rng.standard_normal(n)).astype(TF).reshape(-1, order="F")
    g = sipx.compgrid(h, n)
    ops = ("D_x", "D_y", "D_z") if len(n) == 3 else ("TV",)
    c = [sipx.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", ""))]
    for k in ops:
        s = sipx.get_TD_operator(g, k, TF)[0] @ m
        c.append(sipx.set_definitions("l1", k, 0.0, float(0.5 * np.abs(s.astype(np.float64)).sum()), ("matrix", "")))
    P, A, prop = sipx.setup_constraints(c, g, TF)
    opt = sipx.PARSDMM_options(FL=TF, maxit=maxit)
    opt.Q_mode = q_mode
    if rho is not None:
        opt.rho_ini = rho
    A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
    return m, g, opt, P, A, prop, AtA


@pytest.mark.parametrize("n", [(256, 256, 256), (2048, 2048)])
def test_full_size_Q_properties(sipx, n):
    # BASELINE configs[2] / configs[1]: Q = sum rho_i A_i'A_i is symmetric, annihilates constants up to the identity
    # sets' share, and its CDS and stencil forms agree
    TF = np.float32
    p = len(n) + 2
    rho = [3.0, 0.5, 7.0, 11.0, 2.0][:p] if len(n) == 3 else [3.0, 0.5, 2.0]
    out = {}
    rng = np.random.default_rng(7)
    N = int(np.prod(n))
    x = rng.standard_normal(N).astype(TF); y = rng.standard_normal(N).astype(TF)
    for mode in ("cds", "stencil"):
        m, g, opt, P, A, prop, AtA = _c3_problem(sipx, n, TF, mode, rho=rho)
        ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt)
        Qx, Qy, Q1 = ctx.apply_Q(x), ctx.apply_Q(y), ctx.apply_Q(np.ones(N, TF))
        ctx.close()
        out[mode] = Qx
        sym = abs(np.dot(Qx.astype(np.float64), y) - np.dot(x.astype(np.float64), Qy))
        assert sym <= 1e-5 * np.linalg.norm(Qx.astype(np.float64)) * np.linalg.norm(y.astype(np.float64))
        w0 = rho[0] + rho[-1]                                   # identity sets: bounds and the distance term
        assert np.abs(Q1 - w0).max() <= 1e-4 * max(rho)         # difference operators annihilate constants
    d = np.abs(out["cds"].astype(np.float64) - out["stencil"]).max()
    assert d <= 32 * np.finfo(TF).eps * np.abs(out["cds"]).max()


def test_full_size_c3_solver_properties(sipx):
    # BASELINE configs[2], 256^3 Float32: finite logs, zero-start conventions, feasibility improves, x stays within the
    # bounds' reach, warm restart continues the solve, a feasible model is returned untouched
    TF, n = np.float32, (256, 256, 256)
    m, g, opt, P, A, prop, AtA = _c3_problem(sipx, n, TF, maxit=30)
    x, log, l, y = sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
    it = len(log.obj)
    assert it == 30 or it > 6
    assert np.isfinite(log.obj).all() and np.isfinite(log.r_pri).all() and np.isfinite(log.r_dual).all()
    assert np.isnan(log.evol_x[0]) and log.cg_it[0] == 0 and (log.cg_it[1:] >= 1).all() and log.cg_it.max() < 20
    assert log.r_pri.shape == (it, 5) and log.set_feasibility.shape[1] == 4
    f0, f1 = log.set_feasibility[0], log.set_feasibility[min(2, len(log.set_feasibility) - 2)]
    assert (f1[1:] < f0[1:]).all()                                  # the three l1 sets get closer to feasible
    assert [len(v) for v in y] == [op.shape[0] for op in A]
    assert np.isfinite(x).all() and x.min() > 1000 and x.max() < 4500
    # projector idempotence at full size: the l1 projection of y_2 (already inside its ball) changes nothing
    assert np.array_equal(P[1](y[1].copy()), y[1]) or np.abs(y[1]).sum() > float(P[1].pmax)
    # a model that is feasible for every set comes back untouched (PARSDMM.jl:63-82)
    flat = np.full(m.size, 2500.0, TF)
    xf, logf, _, _ = sipx.PARSDMM(flat.copy(), AtA, A, prop, P, g, opt)
    assert np.array_equal(xf, flat) and len(logf.obj) == 1


@pytest.mark.parametrize("TF", [np.float32])
def test_full_size_projector_properties(sipx, TF):
    # 256^3 entries: l1 norm equals the radius, cardinality keeps exactly k entries, both are idempotent
    rng = np.random.default_rng(11)
    M = 256 ** 3
    v = rng.standard_normal(M).astype(TF)
    b = 0.234 * float(np.abs(v.astype(np.float64)).sum())
    g1 = sipx.compgrid((1.0, 1.0), (M, 1))
    Pl1 = sipx.host.Projector(sipx.set_definitions("l1", "identity", 0.0, b, ("matrix", "")), g1, TF)
    w = Pl1(v.copy())
    assert abs(float(np.abs(w.astype(np.float64)).sum()) - b) <= 2e-6 * b          # test_projectors.jl:27-29
    assert np.array_equal(np.sign(w[w != 0]), np.sign(v[w != 0]))
    w2 = Pl1(w.copy())
    assert np.abs(w2 - w).max() <= 1e-6 * np.abs(w).max()
    k = M // 10
    Pc = sipx.host.Projector(sipx.set_definitions("cardinality", "identity", 0, k, ("matrix", "")), g1, TF)
    u = Pc(v.copy())
    assert np.count_nonzero(u) == k                                               # test_projectors.jl:44-46
    kept = u != 0
    assert np.abs(v[~kept]).max() <= np.abs(v[kept]).min() and np.array_equal(u[kept], v[kept])
    assert np.array_equal(Pc(u.copy()), u)


# ---- long solves: the stop rules (stop_PARSDMM.jl:23-52), incl. the rho/gamma freeze of rule 3 and the final stop of rule 4 ----
@pytest.mark.parametrize("name,n,h,kinds,kw", [
    ("tight-tolerances", (32, 24), (25.0, 6.0), ["bounds", "l1:TV"], dict(maxit=400, evol_rel_tol=1e-9, feas_tol=1e-9, obj_tol=1e-9)),
    ("default-tolerances", (16, 12, 8), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_z"], dict(maxit=400)),
    ("feasibility-only", (32, 24), (25.0, 6.0), ["bounds", "l1:TV"], dict(maxit=200, feasibility_only=True)),
])
def test_long_solves_stop_like_the_oracle(sipx, name, n, h, kinds, kw):
    TF = np.float64
    m = model(n, TF, seed=3)
    go, oo, Po, Ao, propo, AtAo = _problem(O, n, h, TF, kinds, m, kw)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m, kw)
    xo, lo, l_o, y_o = O.PARSDMM(m.copy(), AtAo, Ao, propo, Po, go, oo)
    xs, ls, l_s, y_s = sipx.PARSDMM(m.copy(), AtAs, As, props, Ps, gs, os_)
    p = len(Ao)
    assert ls.r_pri.shape[1] == p == len(y_s) and ls.set_feasibility.shape[1] == len(kinds)
    # same stopping iteration (the BB rule can flip a threshold late in a long run: allow a few iterations of slack)
    assert abs(len(ls.obj) - len(lo.obj)) <= max(3, len(lo.obj) // 20), (len(ls.obj), len(lo.obj))
    K = min(len(ls.obj), len(lo.obj), 40)
    assert np.array_equal(ls.cg_it[:K], lo.cg_it[:K])
    assert np.allclose(ls.rho[:K], lo.rho[:K], rtol=1e-6) and np.allclose(ls.gamma[:K], lo.gamma[:K], rtol=1e-6)
    assert np.allclose(ls.obj[:K], lo.obj[:K], rtol=1e-6)
    err = np.linalg.norm(xs - xo) / np.linalg.norm(xo)
    assert err < 1e-5, err
    if kw.get("feasibility_only"):
        assert len(Ps) == p                      # no distance term appended (PARSDMM_precompute_distribute.jl:17-26)


# ---- symmetric band read of the SpMV (negative bands = shifted positive partners) ---------------------------------------
@pytest.mark.parametrize("TF", [np.float32, np.float64])
@pytest.mark.parametrize("n,h", GRIDS)
def test_symmetric_band_read_is_bit_identical(sipx, TF, n, h, monkeypatch):
    m = model(n, TF)
    kinds = ["bounds", "l1:D_x", "l1:D_z", "l1:TV"]
    go, oo, Po, Ao, propo, AtAo = _problem(O, n, h, TF, kinds, m)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m)
    rho = [3.0, 0.5, 7.0, 11.0, 2.0]
    os_.rho_ini = rho
    x = np.random.default_rng(9).standard_normal(m.size).astype(TF)
    Qo, offo = O.assemble_Q(AtAo, propo.AtA_offsets, np.array(rho, TF), TF)
    want = O.Ax_CDS(x, Qo, offo)
    out = {}
    for full in ("0", "1"):
        monkeypatch.setenv("SIPX_CDS_FULL", full)
        ctx = sipx.host.build_context(m, AtAs, As, props, Ps, gs, os_)
        y0 = ctx.apply_Q(x)
        ctx.q_update([3.0, 0.25, 7.0, 12.5, 1.0], rho)
        out[full] = (y0, ctx.apply_Q(x))
        Q, off = ctx.get_Q()
        ctx.close()
        # the stored matrix itself is symmetric bit for bit, before and after the incremental update
        for b, o in enumerate(off):
            if o < 0:
                pb = list(off).index(-o)
                assert np.array_equal(Q[-o:, b], Q[:o, pb])
    assert np.array_equal(out["0"][0], want) and np.array_equal(out["1"][0], want)
    assert np.array_equal(out["0"][1], out["1"][1])


def test_asymmetric_explicit_bands_fall_back_to_the_full_read(sipx):
    TF, n, h = np.float64, (16, 12), (1.0, 1.0)
    m = model(n, TF)
    go, oo, Po, Ao, propo, AtAo = _problem(O, n, h, TF, ["bounds", "l1:TV"], m)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, ["bounds", "l1:TV"], m)
    bad = [np.array(a, order="F", copy=True) for a in AtAo]
    bad[1][5, 0] += 0.125                      # one entry of the most negative band: A'A no longer symmetric
    Qo, offo = O.assemble_Q(bad, propo.AtA_offsets, np.full(3, 10.0, TF), TF)
    x = np.random.default_rng(2).standard_normal(m.size).astype(TF)
    ctx = sipx.host.build_context(m, bad, As, propo, Ps, gs, os_)
    y = ctx.apply_Q(x)
    ctx.close()
    assert np.array_equal(y, O.Ax_CDS(x, Qo, offo))


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_dft_domain_mask_projector(sipx, TF):
    rng = np.random.default_rng(17)
    for n in ((16, 12, 8), (32, 24)):
        N = int(np.prod(n))
        v = rng.standard_normal(N).astype(TF)
        mask = (rng.random(N) < 0.4).astype(TF)
        want = O.project_bounds_dft(v.copy(), mask, n)
        c = sipx.set_definitions("bounds", "DFT", np.zeros(N, TF), mask, ("matrix", ""))
        got = sipx.host.Projector(c, sipx.compgrid(tuple(1.0 for _ in n), n), TF)(v.copy())
        tol = 5e-6 if TF == np.float32 else 1e-13
        assert np.abs(got.astype(np.float64) - want).max() <= tol * max(1.0, np.abs(want).max())
    with pytest.raises(sipx.SipxError, match="two-valued mask"):
        sipx.host.Projector(sipx.set_definitions("bounds", "DFT", np.zeros(8, TF), np.arange(8).astype(TF), ("matrix", "")),
                            sipx.compgrid((1.0, 1.0), (4, 2)), TF)


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_l2_and_annulus_behind_the_dft_equal_the_plain_projectors(sipx, TF):
    rng = np.random.default_rng(18)
    n = (16, 12, 8)
    N = int(np.prod(n))
    v = rng.standard_normal(N).astype(TF)
    nv = float(np.linalg.norm(v.astype(np.float64)))
    g_o, g_s = O.compgrid((1.0,) * 3, n), sipx.compgrid((1.0,) * 3, n)
    for st_, lo, hi in (("l2", 0.0, 0.5 * nv), ("l2", 0.0, 2.0 * nv), ("annulus", 1.5 * nv, 2.0 * nv), ("annulus", 0.2 * nv, 0.7 * nv)):
        want = O.get_projector(O.set_definitions(st_, "DFT", lo, hi, ("tensor", "")), TF, g_o)(v.copy())
        got = sipx.host.Projector(sipx.set_definitions(st_, "DFT", lo, hi, ("tensor", "")), g_s, TF)(v.copy())
        tol = 5e-6 if TF == np.float32 else 1e-12
        assert np.abs(got.astype(np.float64) - want).max() <= tol * np.abs(want).max(), (st_, lo, hi)


# ---- sets behind the DCT (orthogonal transform folded into the projector; joDCT normalisation unpinned) ---------------
@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_dct_domain_projectors(sipx, TF):
    import scipy.fft as sfft
    rng = np.random.default_rng(19)
    tol = 2e-5 if TF == np.float32 else 1e-11
    for n in ((16, 12, 8), (32, 24), (9, 7, 5)):
        N = int(np.prod(n))
        g_o, g_s = O.compgrid((1.0,) * len(n), n), sipx.compgrid((1.0,) * len(n), n)
        v = rng.standard_normal(N).astype(TF)
        c = sfft.dctn(v.astype(np.float64).reshape(n, order="F"), norm="ortho").reshape(-1, order="F")
        lbv = (-0.3 - rng.random(N)).astype(TF); ubv = (0.2 + rng.random(N)).astype(TF)
        cases = [("l1", 0.0, 0.3 * float(np.abs(c).sum())), ("l1", 0.0, 2.0 * float(np.abs(c).sum())), ("bounds", -0.4, 0.6),
                 ("bounds", lbv, ubv), ("cardinality", 0, N // 5), ("l2", 0.0, 0.5 * float(np.linalg.norm(c))),
                 ("annulus", 1.2 * float(np.linalg.norm(c)), 2.0 * float(np.linalg.norm(c)))]
        for st_, lo, hi in cases:
            want = O.get_projector(O.set_definitions(st_, "DCT", lo, hi, ("matrix", "")), TF, g_o)(v.copy())
            got = sipx.host.Projector(sipx.set_definitions(st_, "DCT", lo, hi, ("matrix", "")), g_s, TF)(v.copy())
            if st_ == "cardinality" and TF == np.float32:
                # the k-th largest coefficient is decided on TF-rounded transforms: allow a swap of near-equal entries
                assert np.linalg.norm(got.astype(np.float64) - want) <= 1e-3 * np.linalg.norm(want), (n, st_)
            else:
                assert np.abs(got.astype(np.float64) - want).max() <= tol * max(1.0, np.abs(want).max()), (n, st_, np.ndim(lo))
        if True:      # inside the l1 ball: returned bit for bit
            big = sipx.set_definitions("l1", "DCT", 0.0, 2.0 * float(np.abs(c).sum()), ("matrix", ""))
            assert np.array_equal(sipx.host.Projector(big, g_s, TF)(v.copy()), v)


# ---- caller-supplied sparse TD_OP (constraint.custom_TD_OP, setup_constraints.jl:70-72), e.g. the reference's own D_xz ------
def _custom_problem(mod, n, h, TF, m, which, maxit=40):
    import scipy.sparse as sp
    g = mod.compgrid(h, n)
    Og = O.compgrid(h, n)
    Dx = O.get_TD_operator(Og, "D_x", TF)[0]
    Dz1 = O.get_TD_operator(O.compgrid(h, (n[0] - 1, n[1])), "D_z", TF)[0]      # get_TD_operator.jl:66-70 (D_xz = D_z * D_x)
    Dxz = sp.csc_matrix(Dz1 @ Dx, dtype=TF)
    Dxz.sort_indices()
    s = Dxz @ m
    c = [mod.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", ""))]
    if which == "l1":
        sd = mod.set_definitions("l1", "identity", 0.0, float(0.4 * np.abs(s).sum()), ("matrix", ""))
    else:
        sd = mod.set_definitions("bounds", "identity", float(0.3 * s.min()), float(0.3 * s.max()), ("matrix", ""))
    sd.custom_TD_OP = (Dxz, False)
    c.append(sd)
    opt = mod.PARSDMM_options(FL=TF, maxit=maxit)
    P, A, prop = mod.setup_constraints(c, g, TF)
    A, AtA, l, y = mod.PARSDMM_precompute_distribute(A, prop, g, opt)
    return g, opt, P, A, prop, AtA, Dxz


@pytest.mark.parametrize("TF", [np.float32, np.float64])
@pytest.mark.parametrize("which", ["l1", "bounds"])
def test_custom_sparse_operator_matches_oracle(sipx, TF, which):
    n, h = (30, 22), (25.0, 6.0)
    m = model(n, TF, seed=6)
    go, oo, Po, Ao, propo, AtAo, Dxz = _custom_problem(O, n, h, TF, m, which)
    gs, os_, Ps, As, props, AtAs, _ = _custom_problem(sipx, n, h, TF, m, which)
    assert As[1].kind == "custom" and As[1].shape == Dxz.shape and len(props.AtA_offsets[1]) == 9
    assert np.array_equal(props.AtA_offsets[1], propo.AtA_offsets[1])
    xo, lo, l_o, y_o = O.PARSDMM(m.copy(), AtAo, Ao, propo, Po, go, oo)
    # (a) A'A handed over explicitly, as a binding that reuses the reference's own setup would: lock-step
    AtA_explicit = [None, AtAo[1], None]
    xs, ls, l_s, y_s = sipx.PARSDMM(m.copy(), AtA_explicit, As, props, Ps, gs, os_)
    K = min(6, len(lo.obj), len(ls.obj))
    rt = 5e-4 if TF == np.float32 else 1e-8
    assert np.array_equal(ls.cg_it[:K], lo.cg_it[:K])
    for f in ("obj", "r_pri_total", "r_dual_total", "rho", "gamma"):
        a, b = np.asarray(getattr(ls, f))[:K], np.asarray(getattr(lo, f))[:K]
        assert np.allclose(a, b, rtol=rt, atol=1e-12), (f, a, b)
    assert np.allclose(ls.set_feasibility[0], lo.set_feasibility[0], rtol=rt)
    err = np.linalg.norm(xs.astype(np.float64) - xo) / np.linalg.norm(xo)
    assert err < (5e-4 if TF == np.float32 else 1e-6), err
    assert [len(v) for v in y_s] == [len(v) for v in y_o] == [m.size, Dxz.shape[0], m.size]
    # (b) A'A computed by the host mirror (scipy product: summation order unpinned): same solution to tolerance
    xs2, ls2, _, _ = sipx.PARSDMM(m.copy(), AtAs, As, props, Ps, gs, os_)
    assert np.linalg.norm(xs2.astype(np.float64) - xo) / np.linalg.norm(xo) < (1e-3 if TF == np.float32 else 1e-5)


def test_stepwise_driver_equals_whole_solve(sipx):
    """sipx_parsdmm_begin / sipx_parsdmm_steps (what bench.py drives) == sipx_parsdmm, bit for bit."""
    TF, n, h = np.float32, (24, 20, 12), (25.0, 25.0, 25.0)
    kinds = ["bounds", "l1:D_x", "l1:D_z", "annulus"]
    m = model(n, TF, seed=8)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m, dict(maxit=35))
    x1, log1, l1, y1 = sipx.PARSDMM(m.copy(), AtAs, As, props, Ps, gs, os_)
    ctx = sipx.host.build_context(m, AtAs, As, props, Ps, gs, os_)
    ctx.parsdmm_begin(os_)
    done, steps = False, 0
    while not done:
        done = ctx.parsdmm_steps(3 if steps % 2 else 1)      # uneven chunks
        steps += 1
        assert steps < 100
    log2 = ctx.parsdmm_log()
    x2, l2, y2 = ctx.download()
    ctx.close()
    assert len(log2.obj) == len(log1.obj)
    assert np.array_equal(x1, x2) and all(np.array_equal(a, b) for a, b in zip(y1, y2)) and all(np.array_equal(a, b) for a, b in zip(l1, l2))
    for f in ("obj", "evol_x", "r_pri", "r_dual", "rho", "gamma", "cg_it", "cg_relres", "set_feasibility"):
        assert np.array_equal(np.asarray(getattr(log1, f)), np.asarray(getattr(log2, f)), equal_nan=True), f


# ---- test/test_setup_constraints.jl: the same properties, through setup_constraints -> P_sub[i] -----------------------
@pytest.mark.parametrize("which", ["oracle", "sipx"])
def test_setup_constraints_projectors_have_the_reference_properties(sipx, which):
    mod = O if which == "oracle" else sipx
    TF = np.float64
    rng = np.random.default_rng(33)
    g2, g3 = mod.compgrid((1.0, 1.0), (20, 31)), mod.compgrid((1.0, 1.0, 1.0), (10, 12, 6))
    N2, N3 = 20 * 31, 10 * 12 * 6

    def P(st, lo, hi, mode, grid, custom=((), False)):
        c = mod.set_definitions(st, "identity", lo, hi, mode)
        c.custom_TD_OP = custom
        return mod.setup_constraints([c], grid, TF)[0][0]

    x = rng.standard_normal(N2)
    y = P("bounds", -0.11, 0.01, ("matrix", ""), g2)(x.copy())
    assert y.max() <= 0.01 and y.min() >= -0.11                                            # :13-26
    lo, hi = rng.standard_normal(N2) - 10, rng.standard_normal(N2) + 10
    y = P("bounds", lo, hi, ("matrix", ""), g2)(100 * rng.standard_normal(N2))
    assert (y <= hi).all() and (y >= lo).all()                                             # :28-42
    assert np.array_equal(P("l1", 0, 2 * np.abs(x).sum(), ("matrix", ""), g2)(x.copy()), x)   # :45-59
    y = P("l1", 0, 0.234 * np.abs(x).sum(), ("matrix", ""), g2)(x.copy())
    assert abs(np.abs(y).sum() - 0.234 * np.abs(x).sum()) < 1e-10 * np.abs(x).sum()        # :61-74
    assert np.count_nonzero(P("cardinality", 0, 5, ("matrix", ""), g2)(x.copy())) == 5      # :77-91
    X = P("cardinality", 0, 7, ("fiber", "x"), g2)(x.copy()).reshape((20, 31), order="F")
    assert all(np.count_nonzero(X[:, i]) == 7 for i in range(31))                           # :108-120
    X = P("cardinality", 0, 11, ("fiber", "z"), g2)(x.copy()).reshape((20, 31), order="F")
    assert all(np.count_nonzero(X[i, :]) == 11 for i in range(20))                          # :123-135
    x3 = rng.standard_normal(N3)
    for d, ax, k in (("x", 0, 7), ("y", 1, 6), ("z", 2, 5)):
        X = P("cardinality", 0, k, ("slice", d), g3)(x3.copy()).reshape((10, 12, 6), order="F")
        assert all(np.count_nonzero(np.take(X, i, axis=ax)) == k for i in range(X.shape[ax]))   # :181-221
        X = P("rank", 0, 3, ("slice", d), g3)(x3.copy()).reshape((10, 12, 6), order="F")
        assert all(np.linalg.matrix_rank(np.take(X, i, axis=ax)) == 3 for i in range(X.shape[ax]))   # :289-329
        X = P("nuclear", 0.0, 1.234, ("slice", d), g3)(x3.copy()).reshape((10, 12, 6), order="F")
        assert all(abs(np.linalg.svd(np.take(X, i, axis=ax), compute_uv=False).sum() - 1.234) < 1e-9
                   for i in range(X.shape[ax]))                                                  # :369-409
    y = P("l2", 0, 0.123, ("matrix", ""), g2)(x.copy())
    assert abs(np.linalg.norm(y) - 0.123) < 1e-12                                           # :224-237
    assert np.linalg.matrix_rank(P("rank", 0, 12, ("matrix", ""), g2)(x.copy()).reshape((20, 31), order="F")) == 12   # :273-286
    M = rng.standard_normal((20, 7))
    y = P("subspace", 0, 0, ("fiber", "x"), g2, (M, False))(x.copy())
    Xr = x.reshape((20, 31), order="F")
    assert np.allclose(y.reshape((20, 31), order="F"), M @ np.linalg.solve(M.T @ M, M.T @ Xr), rtol=1e-9, atol=1e-11)   # :448-462
    ref = np.sort(rng.standard_normal(N2))
    assert np.array_equal(np.sort(P("histogram", ref, ref, ("matrix", ""), g2)(x.copy())), ref)   # :482-495
    with pytest.raises(Exception):
        mod.setup_constraints([mod.set_definitions("rank", "identity", 0, 3, ("tensor", ""))], g3, TF)   # setup_constraints.jl:60-62
    with pytest.raises(Exception):
        mod.setup_constraints([mod.set_definitions("l1", "identity", 0, 1.0, ("fiber", "x"))], g2, TF)   # :65-67


@pytest.mark.parametrize("which", ["oracle", "sipx"])
def test_single_nuclear_norm_set_reaches_the_closed_form(sipx, which):
    """test/test_PARSDMM.jl:192-242: one nuclear-norm set on the identity, tolerances 10 eps, maxit 2500:
    PARSDMM(m) agrees with project_nuclear!(m) to 1e-9 and is feasible to 2 feas_tol."""
    mod = O if which == "oracle" else sipx
    TF = np.float64
    n = (100, 201)
    rng = np.random.default_rng(44)
    m = rng.standard_normal(n[0] * n[1])
    tau = 0.54321
    closed = O.project_nuclear(m.copy(), tau, n)
    g = mod.compgrid((1.0, 1.0), n)
    c = [mod.set_definitions("nuclear", "identity", 0.0, tau, ("matrix", ""))]
    eps = float(np.finfo(TF).eps)
    opt = mod.PARSDMM_options(FL=TF, maxit=2500, obj_tol=10 * eps, feas_tol=10 * eps, evol_rel_tol=10 * eps, Blas_active=False)
    P, A, prop = mod.setup_constraints(c, g, TF)
    A, AtA, l, y = mod.PARSDMM_precompute_distribute(A, prop, g, opt)
    x, log, _, _ = mod.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
    assert np.linalg.norm(x - closed) / np.linalg.norm(closed) <= 1e-9
    px = P[0](x.copy())
    assert np.linalg.norm(px - x) / np.linalg.norm(x) <= 2.0 * 10 * eps + 1e-13


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_named_D_xz_operator(sipx, TF):
    """TD_OP = "D_xz" (get_TD_operator.jl:66-70, 2-D only) goes through the sparse-operator path."""
    n, h = (30, 22), (25.0, 6.0)
    m = model(n, TF, seed=7)
    Ao = O.get_TD_operator(O.compgrid(h, n), "D_xz", TF)[0]
    As = sipx.get_TD_operator(sipx.compgrid(h, n), "D_xz", TF)[0]
    assert As.A.shape == Ao.shape == ((n[0] - 1) * (n[1] - 1), n[0] * n[1])
    assert np.array_equal(As.A.toarray(), Ao.toarray())                      # same rounded entries +-fl(1/h1) fl(1/h2)
    res = {}
    for name, mod in (("o", O), ("s", sipx)):
        g = mod.compgrid(h, n)
        c = [mod.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", "")),
             mod.set_definitions("l1", "D_xz", 0.0, float(0.4 * np.abs(Ao @ m).sum()), ("matrix", ""))]
        opt = mod.PARSDMM_options(FL=TF, maxit=40)
        P, A, prop = mod.setup_constraints(c, g, TF)
        A, AtA, l, y = mod.PARSDMM_precompute_distribute(A, prop, g, opt)
        res[name] = mod.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
    (xo, lo, _, _), (xs, ls, _, _) = res["o"], res["s"]
    K = min(5, len(lo.obj), len(ls.obj))
    assert np.array_equal(ls.cg_it[:K], lo.cg_it[:K])
    assert np.allclose(ls.obj[:K], lo.obj[:K], rtol=1e-3 if TF == np.float32 else 1e-7)
    assert np.linalg.norm(xs.astype(np.float64) - xo) / np.linalg.norm(xo) < (1e-3 if TF == np.float32 else 1e-5)
    with pytest.raises(sipx.SipxError):
        sipx.get_TD_operator(sipx.compgrid((1.0,) * 3, (4, 4, 4)), "D_xz", TF)


@pytest.mark.parametrize("strict", ["1", "0"])
def test_rank_projection_subspace_route(sipx, capfd, monkeypatch, strict):
    """Inside a solve the slice-rank projector (Float32, Gram route) restarts a block subspace iteration from the previous
    call's Ritz vectors and accepts it when the top-r residuals are below its level (ext_proj.hip): strict = 1, the level of
    rounds 3-4 (1e-12 theta_max on the Gram matrix: the two routes then agree to 2e-6); strict = 0, the default since round 5 -- the
    backward error of the reference's own Float32 svd on the slice (project_rank!.jl:26-45), where two correct routes differ like
    two Float32 SVDs do (1e-5 over 16 iterations).  Either way the iterates must agree with the full decomposition of every call
    and with the oracle's LAPACK SVD."""
    monkeypatch.setenv("SIPX_RANK_STRICT", strict)
    TF = np.float32
    n, h = (96, 96, 5), (10.0, 10.0, 10.0)
    rng = np.random.default_rng(77)
    m3 = np.zeros(n)
    for k in range(n[2]):                                   # rank-3 slices + a little noise, velocity-like range
        U, V = rng.standard_normal((n[0], 3)), rng.standard_normal((3, n[1]))
        m3[:, :, k] = 2500.0 + 300.0 * (U @ V) / 3.0 + 2.0 * rng.standard_normal(n[:2])
    m = m3.reshape(-1, order="F").astype(TF)

    def solve(mod):
        g = mod.compgrid(h, n)
        c = [mod.set_definitions("bounds", "identity", 1000.0, 4500.0, ("matrix", "")),
             mod.set_definitions("rank", "identity", 0, 4, ("slice", "z"))]
        opt = mod.PARSDMM_options(FL=TF, maxit=16)
        opt.evol_rel_tol = opt.feas_tol = opt.obj_tol = 0.0          # run all iterations
        P, A, prop = mod.setup_constraints(c, g, TF)
        A, AtA, l, y = mod.PARSDMM_precompute_distribute(A, prop, g, opt)
        return mod.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)

    monkeypatch.setenv("SIPX_EXT_DEBUG", "1")
    capfd.readouterr()
    xs, ls, _, _ = solve(sipx)
    err = capfd.readouterr().err
    assert err.count("subspace accepted") >= 5, err[-2000:]           # the warm-started route carried most iterations
    monkeypatch.setenv("SIPX_EXT_DEBUG", "0")
    monkeypatch.setenv("SIPX_RANK_SUBSPACE", "0")
    xf, lf, _, _ = solve(sipx)
    monkeypatch.delenv("SIPX_RANK_SUBSPACE")
    assert "subspace" not in capfd.readouterr().err
    xo, lo, _, _ = solve(O)
    nrm = np.linalg.norm(xo)
    assert np.linalg.norm(xs.astype(np.float64) - xf.astype(np.float64)) / nrm < (2e-6 if strict == "1" else 1e-5)
    assert np.linalg.norm(xs.astype(np.float64) - xo) / nrm < 1e-4
    K = min(len(ls.obj), len(lo.obj), 8)
    assert np.allclose(ls.obj[:K], lo.obj[:K], rtol=2e-3)


@pytest.mark.parametrize("strict", ["1", "0"])
@pytest.mark.parametrize("r,strong", [(8, 0), (7, 0), (8, 5)])
def test_rank_projection_filtered_route_on_flat_spectra(sipx, capfd, monkeypatch, r, strong, strict):
    """Slices that are a constant plus white noise (the synthetic model of BASELINE config 4) have no gap behind any block of
    singular values: plain subspace iteration never gets there, the Chebyshev-filtered one (ext_proj.hip, rank_cheb_route)
    does, and is accepted on the inertia certificate (one batched Cholesky factorisation of mu I - G + X_r Theta_r X_r').
    r = 7: an odd block (the Jacobi ordering pads it); strong = 5: five more directions far above the noise, of decaying
    weight, so that the projections of the filter take their GEMM form (more than two vectors far above a column).
    The iterates must agree with the full decomposition of every call (SIPX_RANK_CHEB=0) and with the oracle's LAPACK SVD
    (reference: src/projectors/project_rank!.jl:26-45)."""
    monkeypatch.setenv("SIPX_RANK_STRICT", strict)       # 1: the acceptance level of rounds 3-4; 0: the Float32-svd class (round 5 default)
    TF = np.float32
    n, h = (128, 128, 6), (25.0, 25.0, 25.0)
    rng = np.random.default_rng(20240604)
    zz = np.linspace(0.0, 1.0, n[2])[None, None, :]
    m3 = 1500.0 + 2500.0 * zz + 150.0 * rng.standard_normal(n)
    for i in range(strong):
        u, v = rng.standard_normal(n[0]), rng.standard_normal(n[1])
        m3 += (60.0 / 2.0 ** i) * (u[:, None] * v[None, :])[:, :, None]
    m = m3.reshape(-1, order="F").astype(TF)

    def solve(mod):
        g = mod.compgrid(h, n)
        c = [mod.set_definitions("bounds", "identity", 1000.0 if strong else 1600.0, 4500.0 if strong else 3900.0, ("matrix", "")),
             mod.set_definitions("rank", "identity", 0, r, ("slice", "z"))]
        opt = mod.PARSDMM_options(FL=TF, maxit=14)
        opt.evol_rel_tol = opt.feas_tol = opt.obj_tol = 0.0          # run all iterations
        P, A, prop = mod.setup_constraints(c, g, TF)
        A, AtA, l, y = mod.PARSDMM_precompute_distribute(A, prop, g, opt)
        return mod.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)

    monkeypatch.setenv("SIPX_EXT_DEBUG", "1")
    capfd.readouterr()
    xs, ls, _, _ = solve(sipx)
    err = capfd.readouterr().err
    assert err.count("inertia certificate holds") >= 4, err[-3000:]   # flat spectra: the energy bound cannot certify them
    assert err.count("subspace accepted") >= 4, err[-3000:]
    far = [int(v) for v in re.findall(r"(\d+) vectors far above", err)]
    assert far and (not strong or max(far) > 2), far                # strong directions: the GEMM form of the projections ran
    monkeypatch.setenv("SIPX_EXT_DEBUG", "0")
    monkeypatch.setenv("SIPX_RANK_CHEB", "0")
    xf, lf, _, _ = solve(sipx)
    monkeypatch.delenv("SIPX_RANK_CHEB")
    xo, lo, _, _ = solve(O)
    nrm = np.linalg.norm(xo)
    d_so = np.linalg.norm(xs.astype(np.float64) - xo) / nrm
    d_fo = np.linalg.norm(xf.astype(np.float64) - xo) / nrm
    d_sf = np.linalg.norm(xs.astype(np.float64) - xf.astype(np.float64)) / nrm
    # (r = 7 is the case with a near-tie on the way -- see below: the full decomposition of every call ends 2e-4 from the oracle
    #  there, and with the block of round 4, 24 guard columns, the filtered route 1.4e-4: neither is "the" answer to 1e-4)
    print(f"rank route r={r} strong={strong} strict={strict}: filtered-oracle {d_so:.2e}, full-oracle {d_fo:.2e}, filtered-full {d_sf:.2e}")
    # fixed bounds (not "the other route's error"): r = 8 both routes end within 2e-5 of the oracle in either mode; r = 7
    # is the near-tie case -- the full decomposition of every call itself ends 2.0e-4 from the oracle's Float32 LAPACK SVD, the filtered
    # route 1.4e-4 (strict) / 1.8e-4 (Float32-svd class), and the two routes 3.7e-4 from each other
    lim_so, lim_sf = ((4e-4, 6e-4) if r == 7 else (1e-4, 1e-4))
    assert d_so < lim_so, (d_so, d_fo, d_sf)
    # the two routes agree to Float32 rounding -- unless the full decomposition itself has left the oracle (r = 7: sigma_7 and
    # sigma_8 of one slice come within 1e-2 of each other on the way and the truncation there is all but discontinuous; the
    # filtered route stays at 1e-5 of the oracle, the full one ends at 2e-4): then they cannot both be near it
    # (the bound is the spread of trajectories, not of accuracy: with the acceptance tolerance of the filtered route at 1e-14,
    #  1e-12 -- the product's -- and 1e-10 the r = 8 case ends 2.0e-6, 6.7e-6 and 2.2e-6 from the oracle, scratch run of round 4:
    #  white noise truncated inside its own flat spectrum amplifies which Float32 roundings a call happened to make)
    assert d_sf < lim_sf, (d_so, d_fo, d_sf)
    K = min(len(ls.obj), len(lo.obj), 8)
    assert np.allclose(ls.obj[:K], lo.obj[:K], rtol=2e-3)


# ---- BASELINE configs[3] (C4): 8 constraint sets, two of them non-convex ------------------------------------------------
@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_c4_nonconvex_switches_and_lockstep(sipx, TF):
    """The full C4 set list {bounds, l1 D_x / D_y / D_z, annulus, l1-DFT, slice rank, cardinality on D_z} against the
    oracle: a non-convex set switches the solve to rho_update_frequency = 3, gamma = 0.75 and adjust_gamma = false
    (src/PARSDMM_initialize.jl:107-114) whatever the options say."""
    n, h = (16, 12, 8), (25.0, 25.0, 25.0)
    m = model(n, TF, seed=4)
    kw = dict(maxit=45, rho_update_frequency=2, gamma_ini=1.0, adjust_gamma=True)
    go, oo, Po, Ao, propo, AtAo = _problem(O, n, h, TF, C4_KINDS, m, kw)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, C4_KINDS, m, kw)
    assert list(props.ncvx) == list(propo.ncvx) and sum(bool(v) for v in props.ncvx) == 2       # rank, cardinality
    xo, lo, _, _ = O.PARSDMM(m.copy(), AtAo, Ao, propo, Po, go, oo)
    xs, ls, l_s, y_s = sipx.PARSDMM(m.copy(), AtAs, As, props, Ps, gs, os_)
    it = len(ls.obj)
    assert it > 12 and ls.r_pri.shape == (it, 9) and ls.set_feasibility.shape[1] == 8
    assert np.all(ls.gamma == 0.75) and np.all(lo.gamma == 0.75)                  # gamma fixed, never adapted
    # rho may only change after iterations that are multiples of 3 (BB rule) or of 10 beyond the 10th (feasibility
    # doubling, src/PARSDMM.jl:213-223): row i+1 differs from row i only then
    for log in (ls, lo):
        for i in range(1, len(log.obj)):                       # i = 1-based number of the iteration that made the change
            if not np.array_equal(log.rho[i], log.rho[i - 1]):
                assert i % 3 == 0 or (i % 10 == 0 and i > 10), i
    assert any(not np.array_equal(ls.rho[i], ls.rho[i - 1]) for i in range(1, it))      # and it does adapt
    K = min(9, it, len(lo.obj))
    rt = 5e-4 if TF == np.float32 else 1e-8
    assert np.array_equal(ls.cg_it[:K], lo.cg_it[:K])
    for f in ("obj", "r_pri", "r_dual", "rho", "gamma", "evol_x"):
        a, b = np.asarray(getattr(ls, f))[:K], np.asarray(getattr(lo, f))[:K]
        assert np.allclose(a, b, rtol=rt, atol=1e-10, equal_nan=True), (f, a, b)
    assert np.allclose(ls.set_feasibility[0], lo.set_feasibility[0], rtol=rt, atol=1e-12)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n,maxit", [((256, 256, 256), 24), ((512, 512, 512), 12)])
def test_full_size_c4_solver_properties(sipx, n, maxit):
    """C4 at BASELINE's sizes (bench.py's c4-256 / c4 set lists, 512^3 = configs[3] itself): finite logs, the non-convex
    switches, set feasibilities that come down, per-set vector lengths."""
    import bench
    TF = np.float32
    kinds = bench.CONFIGS["c4"][2]
    m = bench.synthetic_model(n, TF, 20240601 + 3)
    gsx = sipx.compgrid((25.0, 25.0, 25.0), n)

    def radius_of(opname):
        sv = sipx.get_TD_operator(gsx, opname, TF)[0] @ m
        return float(0.5 * np.abs(sv.astype(np.float64)).sum())
    g, c = bench.build_problem(sipx, n, (25.0, 25.0, 25.0), kinds, m, TF, radius_of)
    P, A, prop = sipx.setup_constraints(c, g, TF)
    opt = sipx.PARSDMM_options(FL=TF, maxit=maxit, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0)
    A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
    x, log, l, y = sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
    assert len(log.obj) == maxit and log.r_pri.shape == (maxit, 9) and log.set_feasibility.shape[1] == 8
    for f in ("obj", "r_pri", "r_dual", "rho", "cg_relres"):
        assert np.isfinite(np.asarray(getattr(log, f))).all(), f
    assert np.isfinite(log.evol_x[1:]).all() and np.isfinite(x).all()
    assert np.all(log.gamma == 0.75)
    for i in range(1, maxit):
        if not np.array_equal(log.rho[i], log.rho[i - 1]):
            assert i % 3 == 0 or (i % 10 == 0 and i > 10), i
    f0, f1 = log.set_feasibility[0], log.set_feasibility[1]          # initial, and after 10 iterations
    assert np.isfinite(f1).all()
    # decreasing feasibility: the worst set improves, and so do at least six of the eight (the annulus is entered from
    # inside its outer radius and moves outward first: 0.020 -> 0.032 at both sizes)
    assert julia_max(f1) < julia_max(f0) and (f1 < f0).sum() >= 6, (f0, f1)
    assert [len(v) for v in y] == [op.shape[0] for op in A]


# ---- a second solve on one context after a stop-rule exit (sequential solves per handle, include/sipx.h) ----------------
def test_second_solve_after_stop_rule_exit_continues_with_the_current_rho(sipx):
    """After a stop-rule exit Q holds sum_i rho_last_i AtA_i; the next solve on the same context has to start from that
    rho (as after the maxit exit), i.e. equal a fresh context warm-started from the downloaded x, l, y with rho_last."""
    TF, n, h = np.float64, (24, 20, 12), (25.0, 25.0, 25.0)
    kinds = ["bounds", "bnd:D_z", "bnd:D_x"]                       # element-wise sets: no order-dependent reductions in y, l
    m = model(n, TF, seed=11)
    for stop_kw in (dict(maxit=200, evol_rel_tol=2e-3), dict(maxit=9)):        # stop-rule exit, maxit exit
        gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m, dict(adjust_gamma=False, **stop_kw))
        ctx = sipx.host.build_context(m, AtAs, As, props, Ps, gs, os_)
        log1, _ = ctx.parsdmm(os_)
        if "evol_rel_tol" in stop_kw:
            assert 6 < len(log1.obj) < 200                                      # it was the stop rule
        x1, l1, y1 = ctx.download()
        rho_last = np.array(log1.rho[-1])
        assert not np.allclose(rho_last, 10.0)                                   # rho did move away from rho_ini
        o2 = sipx.PARSDMM_options(FL=TF, maxit=8, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0, adjust_gamma=False)
        log2, _ = ctx.parsdmm(o2)
        x2, _, _ = ctx.download()
        ctx.close()
        if "evol_rel_tol" in stop_kw:
            assert np.array_equal(log2.rho[0], rho_last)
        o3 = sipx.PARSDMM_options(FL=TF, maxit=8, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0, adjust_gamma=False,
                                  zero_ini_guess=False, rho_ini=[float(r) for r in np.array(log2.rho[0])])
        ctx3 = sipx.host.build_context(m, AtAs, As, props, Ps, gs, o3, x=x1, l=l1, y=y1)
        log3, _ = ctx3.parsdmm(o3)
        x3, _, _ = ctx3.download()
        ctx3.close()
        assert np.array_equal(log2.cg_it, log3.cg_it)
        assert np.allclose(log2.obj, log3.obj, rtol=1e-9) and np.allclose(log2.r_pri, log3.r_pri, rtol=1e-6, atol=1e-12)
        assert np.linalg.norm(x2 - x3) <= 1e-9 * np.linalg.norm(x3)


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_l1_projector_with_a_large_gather(sipx, TF):
    """Cold start on 3 M entries: the bracket of the threshold search holds far more than 2^17 magnitudes, so the sweeps of
    k_l1_solve are shared by its 32 workgroups (counter barrier, double-double shares).  Against the oracle's sort-based
    threshold; twice, bit for bit (the shares are added in slot order, whatever order the workgroups arrive in)."""
    rng = np.random.default_rng(71)
    n = 3_000_000
    v = (rng.standard_normal(n) * np.exp(0.5 * rng.standard_normal(n))).astype(TF)
    g = sipx.compgrid((1.0, 1.0), (10, 10))
    for frac in (0.6, 0.05):
        b = float(frac * np.abs(v.astype(np.float64)).sum())
        P = sipx.Projector(sipx.set_definitions("l1", "identity", 0.0, b, ("matrix", "")), g, TF)
        w1, w2 = P(v.copy()), P(v.copy())
        assert np.array_equal(w1, w2)
        ref = O.project_l1_Duchi(v.astype(np.float64), b)
        tol = 2e-5 if TF == np.float32 else 1e-11
        assert np.linalg.norm(w1 - ref) <= tol * np.linalg.norm(ref)
        assert abs(np.abs(w1.astype(np.float64)).sum() - b) <= tol * b


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_sampled_prediction_of_the_l1_threshold_changes_nothing(sipx, TF, monkeypatch):
    """The sampled estimate of theta (k_sample: histogram of every 16th run of 64 entries, Newton / secant bounds of the
    sample's root) only places the speculative range of the first pass; theta itself stays the exact fixed point.  On a grid
    small enough for the oracle, with the sample forced on (SIPX_L1_SAMPLE_RUNS: large grids sample by themselves): the solve
    with and without it must agree to rounding of the one float64 sum whose split between 'above the range' and 'gathered'
    differs, every search after the cold ones must have used the estimate at least once, and both must agree with the oracle
    as the unsampled path does."""
    n = (64, 48, 40)
    m, g, opt, P, A, prop, AtA = _c3_problem(sipx, n, TF, maxit=24)
    res = {}
    for tag, env in (("off", {"SIPX_L1_SAMPLE": "0"}), ("on", {"SIPX_L1_SAMPLE": "1", "SIPX_L1_SAMPLE_RUNS": "96"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt)
        ctx.parsdmm_begin(opt)
        used = 0
        for it in range(24):
            ctx.parsdmm_steps(1)
            used += sum(int(ctx.debug_proj(s, 0)["sampled"]) for s in (1, 2, 3))
        x, _, _ = ctx.download()
        log = ctx._run[2]
        res[tag] = (x.astype(np.float64), np.array(log["rho"]), np.array(log["r_pri"]), used)
        ctx.close()
    assert res["off"][3] == 0 and res["on"][3] >= 12
    tol = 2e-6 if TF == np.float32 else 1e-12
    # the same rho history (no threshold of the Barzilai-Borwein rule flipped), to the rounding theta carries
    assert np.allclose(res["on"][1], res["off"][1], rtol=50 * tol, atol=0)
    assert np.allclose(res["on"][2], res["off"][2], rtol=50 * tol, atol=0)
    assert np.linalg.norm(res["on"][0] - res["off"][0]) <= tol * np.linalg.norm(res["off"][0])


@pytest.mark.parametrize("kind", ["blocky", "spiky", "wide"])
def test_sampled_prediction_on_hostile_models(sipx, kind, monkeypatch):
    """Models that make a sampled estimate useless -- piecewise constant (almost every difference is zero, a few are huge),
    a few spikes on a flat background, magnitudes spread over twenty octaves: whatever the estimate says, the end point must
    be the one of the unsampled search (a miss only costs the fallback sweeps) and every log finite."""
    TF, n = np.float32, (64, 48, 40)
    rng = np.random.default_rng(3)
    if kind == "blocky":
        m = np.repeat(np.repeat(np.repeat(rng.uniform(1500, 4500, (8, 6, 5)), 8, 0), 8, 1), 8, 2)
    elif kind == "spiky":
        m = np.full(n, 2500.0)
        m.reshape(-1)[rng.choice(m.size, 200, replace=False)] += rng.uniform(500, 2000, 200)
    else:
        m = 2500.0 + np.exp(rng.uniform(-20, 7, n)) * rng.choice([-1.0, 1.0], n)
    m = m.astype(TF).reshape(-1, order="F")
    h = (25.0, 25.0, 25.0)
    g = sipx.compgrid(h, n)
    c = [sipx.set_definitions("bounds", "identity", 1600.0, 4400.0, ("matrix", ""))]
    for k in ("D_x", "D_z", "TV"):
        s_ = sipx.get_TD_operator(g, k, TF)[0] @ m
        c.append(sipx.set_definitions("l1", k, 0.0, float(0.3 * np.abs(s_.astype(np.float64)).sum()), ("matrix", "")))
    P, A, prop = sipx.setup_constraints(c, g, TF)
    opt = sipx.PARSDMM_options(FL=TF, maxit=30, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0)
    A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
    out = {}
    for tag, env in (("off", {"SIPX_L1_SAMPLE": "0"}), ("on", {"SIPX_L1_SAMPLE": "1", "SIPX_L1_SAMPLE_RUNS": "64"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        x, log, _, _ = sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
        assert np.isfinite(log.obj).all() and np.isfinite(log.r_pri).all() and np.isfinite(x).all()
        out[tag] = (x.astype(np.float64), log)
    assert len(out["on"][1].obj) == len(out["off"][1].obj)
    assert np.linalg.norm(out["on"][0] - out["off"][0]) <= 2e-6 * np.linalg.norm(out["off"][0])


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_fused_cg_iterations_are_bit_identical(sipx, TF, monkeypatch):
    """CG iterations from the second on as one kernel (k_cds_fused: scalar step + product on p = r + beta p_old formed on the
    fly; the default on grids up to 2^23 points) against the three-kernel form (SIPX_CG_FUSED=0): same arithmetic, so the same
    bits -- x, the CG iteration counts and residuals, every log."""
    n = (48, 40, 24)
    m, g, opt, P, A, prop, AtA = _c3_problem(sipx, n, TF, maxit=30)
    out = {}
    for tag in ("0", "1"):
        monkeypatch.setenv("SIPX_CG_FUSED", tag)
        x, log, l, y = sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
        out[tag] = (x, log)
    (x0, l0), (x1, l1) = out["0"], out["1"]
    assert l0.cg_it.sum() > len(l0.cg_it)                       # iterations beyond the first did run
    assert np.array_equal(l0.cg_it, l1.cg_it) and np.array_equal(l0.cg_relres, l1.cg_relres)
    assert np.array_equal(x0, x1) and np.array_equal(l0.obj, l1.obj) and np.array_equal(l0.r_pri, l1.r_pri)
    assert np.array_equal(l0.rho, l1.rho)


# ---- BASELINE config 5 on more than one GPU: the multilevel wrapper over slab-decomposed levels ---------------------------
def _ml_problem(sipx, n, h, TF, levels):
    from sipx import multilevel as ML
    m = model(n, TF, seed=9)
    TV = O.get_TD_operator(O.compgrid(h, n), "TV", TF)[0]
    cons = [sipx.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", "")),
            sipx.set_definitions("l1", "TV", 0.0, float(0.5 * np.abs(TV @ m).sum()), ("matrix", ""))]
    opt = sipx.PARSDMM_options(FL=TF, maxit=40)
    L = ML.setup_multi_level_PARSDMM(m, levels, 2, sipx.compgrid(h, n), cons, opt)
    return ML, m, opt, L


def _ml_worker(rank, world, port, out, n, h, levels):
    import os
    import sys
    import datetime
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from __graft_entry__ import load_package
        sipx = load_package()
        ML, m, opt, L = _ml_problem(sipx, n, h, np.float32, levels)
        rec = {}
        x, log, l, y = ML.PARSDMM_multi_level(m.copy(), *L[:5], opt, device=0, dist=dist, comm_mode="torch", timings=rec)
        np.savez(os.path.join(out, f"ml{rank}.npz"), x=x, obj=log.obj, cg_it=log.cg_it, rho=log.rho, **{f"y{i}": v for i, v in enumerate(y)},
                 **{f"l{i}": v for i, v in enumerate(l)})
        np.savez(os.path.join(out, f"mlmem{rank}.npz"), device_bytes=np.array([lv.get("device_bytes", -1) for lv in rec["levels"]], dtype=np.int64),
                 sparse=np.array([int(lv.get("sparse_arrays", False)) for lv in rec["levels"]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(400)
@pytest.mark.parametrize("world,n,h,levels", [(2, (32, 24, 16), (25.0, 25.0, 25.0), 2), (3, (24, 20, 20), (25.0, 20.0, 10.0), 3)])
def test_multilevel_over_slab_decomposed_levels(sipx, tmp_path, world, n, h, levels):
    """BASELINE config 5's pattern on more than one rank: every level solved slab-decomposed (sparse arrays), the coarse slabs
    all-gathered on the device one block at a time, every rank resampling the grid points it stores (sipx_warm_start_from).
    Identical results on every rank; the single-rank multilevel solve to the reference's serial-vs-parallel tolerance."""
    _run_ml_sharded(sipx, tmp_path, world, n, h, levels)


def _run_ml_sharded(sipx, tmp_path, world, n, h, levels):
    import os
    import torch.multiprocessing as mp
    mp.spawn(_ml_worker, args=(world, 31700 + (os.getpid() % 2000) + world, str(tmp_path), n, h, levels), nprocs=world, join=True)
    r0 = np.load(tmp_path / "ml0.npz")
    for r in range(1, world):
        r1 = np.load(tmp_path / f"ml{r}.npz")
        for k in r0.files:
            assert np.array_equal(r0[k], r1[k], equal_nan=True), k
    ML, m, opt, L = _ml_problem(sipx, n, h, np.float32, levels)
    xs, logs, ls, ys = ML.PARSDMM_multi_level(m.copy(), *L[:5], opt)
    assert np.linalg.norm(r0["x"] - xs) <= 5e-4 * np.linalg.norm(xs)
    K = min(6, len(logs.obj), len(r0["obj"]))
    assert np.array_equal(r0["cg_it"][:K], logs.cg_it[:K]) and np.allclose(r0["obj"][:K], logs.obj[:K], rtol=5e-4)
    return r0, np.load(tmp_path / "mlmem0.npz")


@pytest.mark.timeout(600)
def test_multilevel_levels_hold_sparse_arrays(sipx, tmp_path, monkeypatch):
    """Round 5: the levels of a slab-decomposed multilevel solve hold the rank's planes only (sipx_warm_start_from completes one
    coarse block at a time in a coarse-sized temporary).  Same iterates, bit for bit, as with whole arrays on every level
    (SIPX_MULTILEVEL_SLAB_FULL=1)."""
    n, h, levels, world = (48, 40, 36), (25.0, 20.0, 10.0), 3, 3
    (tmp_path / "sparse").mkdir()
    (tmp_path / "full").mkdir()
    a, mem_a = _run_ml_sharded(sipx, tmp_path / "sparse", world, n, h, levels)
    monkeypatch.setenv("SIPX_MULTILEVEL_SLAB_FULL", "1")
    b, mem_b = _run_ml_sharded(sipx, tmp_path / "full", world, n, h, levels)
    for k in a.files:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert mem_a["sparse"].all() and not mem_b["sparse"].any()
    # (what a rank then holds is measured where arrays are larger than the 2 MiB granules of the mapping: the `c5` leg of
    #  bench.py at N > 1 reports `device_bytes_per_level`; on a grid this small the granules outweigh the arrays)
    assert (mem_a["device_bytes"] > 0).all()


@pytest.mark.timeout(300)
def test_bench_line_reports_the_rank_projector_route():
    """A set list with a slice-rank set (BASELINE config 4 at 256^3): the bench line carries the engine's own counters of the
    rank projector's route -- calls, calls served by the warm-started filtered subspace iteration, full decompositions, products
    with the Gram matrices -- so that a C4 number can be read together with what produced it."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(prefix="sipx_bench_"), "detail.json")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "c4-256", "--steps", "5", "--warmup", "2", "--detail", detail,
                        "--no-cpu-baseline", "--no-512", "--no-c4", "--no-c5"], capture_output=True, text=True, timeout=280, env=dict(os.environ))
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(r.stdout.strip()) < 4096
    d = json.load(open(detail))
    rr = d["rank_route"]
    assert rr["calls"] >= 7 and rr["calls"] == rr["warm_started_subspace"] + rr["full_decomposition"]
    assert rr["warm_started_subspace"] >= 3 and rr["products_with_gram"] >= 10 * rr["warm_started_subspace"]
    assert d["config"]["all_logs_finite"]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("dist_env", [None, "1"])
def test_bench_contract_line(dist_env):
    """bench.py prints ONE JSON line with the contract's fields (metric, value, unit, n_gpus, steps, warmup, ms_per_step,
    higher_is_better, scaling, vs_baseline, dtype, data, config.workload, roofline, iteration_roofline) -- on one GPU, and
    through the sharded code path (RCCL communicator with a world of one: both decompositions timed and reported under fixed
    keys, the headline always the slab decomposition; what RCCL itself reports about the communicator in every leg)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if dist_env:
        env.update(SIPX_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(32100 + os.getpid() % 2000), RANK="0", WORLD_SIZE="1")
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(prefix="sipx_bench_"), "detail.json")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "c3-small", "--steps", "4", "--warmup", "2", "--detail", detail,
                        "--no-cpu-baseline", "--no-512", "--no-c4", "--no-c5"], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    assert len(lines[0]) < 4096, len(lines[0])       # the driver keeps a few KB of stdout tail: round 3's line (24 KB) lost its head
    h = json.loads(lines[0])                         # the headline: the contract's keys and the figures a review reads first
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "iteration_roofline", "dominant_kernel", "comm"):
        assert k in h, k
    assert h["n_gpus"] == 1 and h["steps"] == 4 and h["warmup"] == 2 and h["value"] > 0 and h["vs_baseline"] is None
    assert "workload" in h["config"] and h["config"]["all_logs_finite"]
    assert {"bound", "achieved", "peak", "unit", "frac", "frac_survey", "traffic", "kernel", "launches", "avg_launch_ms"} <= set(h["roofline"])
    assert h["roofline"]["frac"] <= h["roofline"]["frac_survey"]                 # bytes that move <= SURVEY's count
    assert h["dominant_kernel"]["kernel"].startswith("k_") and len(h["libsipx_sha16"]) == 16
    d = json.load(open(detail))                      # everything else: the side file
    assert abs(d["value"] - h["value"]) <= 1e-4 * h["value"]
    dom = d["dominant_kernel"]
    assert dom["kernel"].startswith("k_") and dom["launches"] > 0 and dom["avg_launch_ms"] > 0 and 0 <= dom["frac"] < 1.2      # (64^3: a one-workgroup step may lead)
    assert any(r["kernel"] == dom["kernel"] for r in d["kernels"]) and len(d["libsipx_sha16"]) == 16
    if dist_env:
        assert set(h["decompositions"]) == {"slab", "sets"} and h["decomposition"] == "slab" and h["faster_decomposition"] in ("slab", "sets")
        assert h["value"] == h["decompositions"]["slab"]["value"]
        assert all(v["ranks_agree_on_x"] is True for v in h["decompositions"].values())
        assert h["comm"]["rccl_nranks"] == 1 and h["comm"]["rccl_version"].startswith("rccl ") and h["comm"]["device_bytes_per_rank"] > 0
        for v in d["decompositions"].values():
            assert v["value"] > 0 and v["comm"]["rccl_nranks"] == 1 and v["comm"]["rccl_version"].startswith("rccl ")
            assert v["comm"]["ranks_agree_on_x"] is True
        ss = d["decompositions"]["slab"]["comm"]["slab_searches"]               # the engine's counters of the speculative exchange
        assert ss["speculative_exchange"] > 0 and 0 <= ss["fallbacks"] <= ss["speculative_exchange"]
    else:
        assert h["comm"]["rccl_version"] == "none" and "decompositions" not in h
