"""The reference's own SOLVER-level tests, restated literally (same grid, same sets, same option blocks, same inequalities)
and run on BOTH implementations: the CPU oracle (`-m "not gpu"`; these assertions pin the oracle above its leaf functions)
and the HIP engine through the C ABI (`-m gpu`).

  test/test_PARSDMM.jl:1-36      a feasible input is returned untouched (x == m)
  test/test_PARSDMM.jl:38-190    {bounds, D_z bounds, TV l1} on 100 x 201 Float64 under seven option blocks: every set is
                                 feasible to 1.5 feas_tol
  test/test_PARSDMM.jl:192-242   one nuclear-norm set: tests/test_gpu_parity.py::test_single_nuclear_norm_set_reaches_the_closed_form
  test/test_PARSDMM.jl:244-316   {nuclear, bounds, D_z bounds, TV l1}: Blas_active = false / true agree to rtol 1e-12
  test/test_PARSDMM_parallel.jl:4-38, 57-72     {TV l1} on 201 x 100 Float32, 1e-5 tolerances: feasible to 1.5 feas_tol
  test/test_PARSDMM_parallel.jl:75-100          {DFT l1} likewise

These hold for any draw of the model (the reference seeds Julia's generator, whose stream numpy cannot reproduce: the draw
here is numpy's with the same seed number).  Nothing under /root/reference is read at run time."""
import numpy as np
import pytest

from oracle import parsdmm_oracle as O

N2 = (100, 201)
EPS = float(np.finfo(np.float64).eps)

# the option blocks of test/test_PARSDMM.jl, in file order (line of `options = PARSDMM_options()`)
BLOCKS = {
    "default_79": dict(evol_rel_tol=10 * EPS, maxit=5000),
    "accurate_92": dict(obj_tol=1e-12, feas_tol=1e-12, evol_rel_tol=10 * EPS, maxit=5000),
    "accurate_noblas_108": dict(Blas_active=False, obj_tol=1e-12, feas_tol=1e-12, evol_rel_tol=10 * EPS, maxit=5000),
    "no_gamma_124": dict(adjust_gamma=False, obj_tol=1e-12, feas_tol=1e-12, evol_rel_tol=10 * EPS, maxit=5000),
    "no_rho_140": dict(adjust_gamma=True, adjust_rho=False, obj_tol=1e-6, feas_tol=1e-6, evol_rel_tol=10 * EPS, maxit=10000),
    "no_rho_no_gamma_157": dict(adjust_gamma=False, adjust_rho=False, obj_tol=1e-6, feas_tol=1e-6, evol_rel_tol=10 * EPS, maxit=25000),
    "no_feasibility_rho_174": dict(adjust_feasibility_rho=False, adjust_gamma=True, adjust_rho=True, obj_tol=1e-12, feas_tol=1e-12,
                                   evol_rel_tol=10 * EPS, maxit=5000),
}
# what the numpy oracle finishes in seconds on the CPU suite's budget; the engine runs every block
ORACLE_BLOCKS = ["default_79", "accurate_92", "no_gamma_124", "no_feasibility_rho_174"]


def _draw(TF, n, seed=123):
    return np.random.default_rng(seed).standard_normal(n).astype(TF).reshape(-1, order="F")


def _three_sets(mod, x, g, TF):
    """test/test_PARSDMM.jl:41-73 -- the radii come from the model through the reference's own operators."""
    go = O.compgrid((1.0, 1.0), N2)
    Dz = O.get_TD_operator(go, "D_z", TF)[0]
    TV = O.get_TD_operator(go, "TV", TF)[0]
    return [mod.set_definitions("bounds", "identity", float(0.5 * x.min()), float(0.5 * x.max()), ("matrix", "")),
            mod.set_definitions("bounds", "D_z", float(0.5 * (Dz @ x).min()), float(0.5 * (Dz @ x).max()), ("matrix", "")),
            mod.set_definitions("l1", "TV", 0.0, float(0.5 * np.abs(TV @ x).sum()), ("matrix", ""))]


def _solve(mod, c, g, TF, m, kw):
    opt = mod.PARSDMM_options(FL=TF, **kw)
    P, A, prop = mod.setup_constraints(c, g, TF)
    A, AtA, l, y = mod.PARSDMM_precompute_distribute(A, prop, g, opt)
    x, log, l, y = mod.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
    return x, log, P, A, opt


def _assert_feasible(x, P, A, factor, feas_tol):
    """`for i=1:length(TD_OP)-1  @test norm(P_sub[i](TD_OP[i]*x) - TD_OP[i]*x) / norm(TD_OP[i]*x) <= factor*feas_tol`"""
    assert np.all(np.isfinite(x))
    for i in range(len(A) - 1):
        s = np.asarray(A[i] @ x)
        p = np.asarray(P[i](s.copy()))
        f = np.linalg.norm(p.astype(np.complex128) - s) / np.linalg.norm(s)
        assert f <= factor * float(feas_tol), (i, f, factor * float(feas_tol))


def _feasible_input(mod):
    TF = np.float64
    x = _draw(TF, N2)
    g = mod.compgrid((1.0, 1.0), N2)
    c = [mod.set_definitions("bounds", "identity", float(x.min()), float(x.max()), ("matrix", ""))]
    xo, log, P, A, opt = _solve(mod, c, g, TF, x, dict(zero_ini_guess=True))
    assert np.array_equal(xo, x)                                               # `@test x==m`, test_PARSDMM.jl:36


def _three_set_block(mod, block):
    TF = np.float64
    x = _draw(TF, N2)
    g = mod.compgrid((1.0, 1.0), N2)
    xo, log, P, A, opt = _solve(mod, _three_sets(mod, x, g, TF), g, TF, x, BLOCKS[block])
    if BLOCKS[block].get("adjust_rho", True):
        # not a reference assertion: with the adaptive penalty every block stops on its tolerances well before maxit; the two blocks
        # with a FIXED rho (maxit 10000 / 25000 in the reference) may use all their iterations, the reference only asks for feasibility
        assert len(log.obj) < BLOCKS[block]["maxit"], "stopped by maxit, not by its tolerances"
    _assert_feasible(xo, P, A, 1.5, opt.feas_tol)
    return xo


def _blas_pair(mod):
    """test/test_PARSDMM.jl:244-316."""
    TF = np.float64
    x = _draw(TF, N2, seed=124)
    g = mod.compgrid((1.0, 1.0), N2)
    go = O.compgrid((1.0, 1.0), N2)
    Dz = O.get_TD_operator(go, "D_z", TF)[0]
    TV = O.get_TD_operator(go, "TV", TF)[0]
    c = [mod.set_definitions("nuclear", "identity", 0.0, 1.123, ("matrix", "")),
         mod.set_definitions("bounds", "identity", float(1.0 * x.min()), float(0.50 * x.max()), ("matrix", "")),
         mod.set_definitions("bounds", "D_z", float(0.9 * (Dz @ x).min()), float(0.67 * (Dz @ x).max()), ("matrix", "")),
         mod.set_definitions("l1", "TV", 0.0, float(0.2 * np.abs(TV @ x).sum()), ("matrix", ""))]
    kw = dict(adjust_feasibility_rho=True, adjust_gamma=True, adjust_rho=True, obj_tol=1e-6, feas_tol=1e-6, evol_rel_tol=1e-6, maxit=2500)
    x_noblas = _solve(mod, c, g, TF, x, dict(Blas_active=False, **kw))[0]
    x_blas = _solve(mod, c, g, TF, x, dict(Blas_active=True, **kw))[0]
    assert np.all(np.isfinite(x_blas))
    # Julia's isapprox(a, b, rtol=r) on vectors: norm(a - b) <= r * max(norm(a), norm(b))
    assert np.linalg.norm(x_blas - x_noblas) <= 1e-12 * max(np.linalg.norm(x_blas), np.linalg.norm(x_noblas))


def _parallel_suite_serial_leg(mod, which):
    """test/test_PARSDMM_parallel.jl:4-38 (TV) and :75-100 (DFT), the serial solve x3 / x1 of each block."""
    TF = np.float32
    n = (201, 100)
    x = _draw(TF, n)
    g = mod.compgrid((1.0, 1.0), n)
    c = _parallel_suite_sets(mod, which, x, n, TF)
    kw = dict(evol_rel_tol=1e-5, feas_tol=1e-5, obj_tol=1e-5, maxit=10000)
    xo, log, P, A, opt = _solve(mod, c, g, TF, x, kw)
    assert len(log.obj) < 10000
    _assert_feasible(xo, P, A, 1.5, opt.feas_tol)
    return xo


# ---- the oracle (CPU suite): the reference's solver-level assertions pin the restatement ------------------------------------
def test_oracle_feasible_input_untouched():
    _feasible_input(O)


@pytest.mark.parametrize("block", ORACLE_BLOCKS)
def test_oracle_three_sets_feasible(block):
    _three_set_block(O, block)


def test_oracle_parallel_suite_TV_leg():
    _parallel_suite_serial_leg(O, "TV")


# ---- the second, independent restatement (oracle/parsdmm_port.c, C / OpenMP): EVERY block, the fixed-rho ones included ------------
def _port_three_set_block(block):
    from oracle import port
    TF = np.float64
    x = _draw(TF, N2)
    g = O.compgrid((1.0, 1.0), N2)
    c = _three_sets(O, x, g, TF)
    o = O.PARSDMM_options(FL=TF, **BLOCKS[block])
    O.convert_options(o, TF)
    r = port.run(N2, (1.0, 1.0), [(k.set_type, k.TD_OP, k.min, k.max) for k in c], x, int(o.maxit), float(o.evol_rel_tol), float(o.feas_tol),
                 float(o.obj_tol), rho_ini=float(np.atleast_1d(o.rho_ini)[0]), gamma_ini=float(o.gamma_ini), freq=int(o.rho_update_frequency),
                 adjust_rho=bool(o.adjust_rho), adjust_gamma=bool(o.adjust_gamma), adjust_feasibility_rho=bool(o.adjust_feasibility_rho),
                 nthreads=min(8, port.host_threads()))      # (never the OpenMP default: a GPU box shows every hardware thread of the host behind a quota of 16)
    P, A, prop = O.setup_constraints(c, g, TF)
    _assert_feasible(r["x"], P, A, 1.5, o.feas_tol)
    return r


_PORT_ITERATIONS = {}


@pytest.mark.parametrize("block", list(BLOCKS))
def test_port_three_sets_feasible(block):
    r = _port_three_set_block(block)
    _PORT_ITERATIONS[block] = r["n_iter"]
    if not BLOCKS[block].get("adjust_rho", True):
        assert r["n_iter"] == BLOCKS[block]["maxit"]        # a fixed rho uses all its iterations (why the reference gives it 10000 / 25000)


def test_oracle_and_port_take_the_same_number_of_iterations():
    """Two restatements written apart from each other (numpy; C with the reference's loop structure) stop at the same iteration."""
    for block in ("default_79", "no_gamma_124"):
        x = _draw(np.float64, N2)
        g = O.compgrid((1.0, 1.0), N2)
        xo, log, P, A, opt = _solve(O, _three_sets(O, x, g, np.float64), g, np.float64, x, BLOCKS[block])
        n_port = _PORT_ITERATIONS.get(block) or _port_three_set_block(block)["n_iter"]
        assert len(log.obj) == n_port, (block, len(log.obj), n_port)


# ---- the HIP engine (GPU suite): every block, and the engine against the oracle on the blocks the oracle ran ---------------
@pytest.mark.gpu
def test_engine_feasible_input_untouched(sipx):
    _feasible_input(sipx)


@pytest.mark.gpu
@pytest.mark.parametrize("block", list(BLOCKS))
def test_engine_three_sets_feasible(sipx, block):
    xs = _three_set_block(sipx, block)
    # the C restatement on the same block: same end point (both are feasible to the block's tolerance on a problem with one solution)
    # (the two fixed-rho blocks take the port a minute: they are run by the CPU suite only)
    if BLOCKS[block].get("adjust_rho", True):
        rp = _port_three_set_block(block)
        tol = 1e-9 if float(BLOCKS[block].get("feas_tol", 5e-2)) <= 1e-12 else 5e-4
        assert np.linalg.norm(xs - rp["x"]) / np.linalg.norm(rp["x"]) <= tol, (block, np.linalg.norm(xs - rp["x"]) / np.linalg.norm(rp["x"]))


@pytest.mark.gpu
def test_engine_blas_and_loop_paths_agree(sipx):
    _blas_pair(sipx)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["TV", "DFT"])
def test_engine_parallel_suite_serial_legs(sipx, which):
    xs = _parallel_suite_serial_leg(sipx, which)
    if which == "TV":
        xo = _parallel_suite_serial_leg(O, "TV")
        # `@test isapprox(x1, x3, rtol=5*1f-4)`, test_PARSDMM_parallel.jl:72: the reference's tolerance between two of its own paths
        assert np.linalg.norm(xs.astype(np.float64) - xo) <= 5e-4 * max(np.linalg.norm(xs.astype(np.float64)), np.linalg.norm(xo))


# ---- test/test_PARSDMM_parallel.jl:4-72, 75-121, the parallel legs ("test is for 2 workers"): two ranks sharing the one GPU, the
# engine's collectives over gloo callbacks; x_parallel is feasible and isapprox(x_parallel, x_serial, rtol = 5e-4) ---------------
def _parallel_worker(rank, world, port, out, which, decomp):
    import datetime
    import os
    import sys
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        from __graft_entry__ import load_package
        sipx = load_package()
        from sipx import sharded
        TF, n = np.float32, (201, 100)
        x = _draw(TF, n)
        g = sipx.compgrid((1.0, 1.0), n)
        c = _parallel_suite_sets(sipx, which, x, n, TF)
        opt = sipx.PARSDMM_options(FL=TF, evol_rel_tol=1e-5, feas_tol=1e-5, obj_tol=1e-5, maxit=10000)
        P, A, prop = sipx.setup_constraints(c, g, TF)
        A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
        xo, log, l, y = sharded.PARSDMM_sharded(x.copy(), AtA, A, prop, P, g, opt, dist=dist, device=0, comm_mode="torch", decomp=decomp)
        np.savez(os.path.join(out, f"p{rank}.npz"), x=xo, its=len(log.obj))
    finally:
        dist.destroy_process_group()


def _parallel_suite_sets(mod, which, x, n, TF):
    if which == "TV":
        TV = O.get_TD_operator(O.compgrid((1.0, 1.0), n), "TV", TF)[0]
        return [mod.set_definitions("l1", "TV", 0.0, float(TF(0.5) * np.abs(TV @ x).sum(dtype=TF)), ("matrix", ""))]
    Z = np.abs(np.fft.fftn(x.reshape(n, order="F").astype(np.float64), norm="ortho"))
    return [mod.set_definitions("l1", "DFT", 0.0, float(TF(0.5) * TF(Z.sum())), ("matrix", ""))]


@pytest.mark.gpu
@pytest.mark.timeout(400)
@pytest.mark.parametrize("which,decomp", [("TV", "sets"), ("TV", "slab"), ("DFT", "sets")])
def test_engine_parallel_suite_two_ranks(sipx, tmp_path, which, decomp):
    import os
    import torch.multiprocessing as mp
    port = 29400 + (os.getpid() % 2000) + (7 if decomp == "slab" else 0)
    mp.spawn(_parallel_worker, args=(2, port, str(tmp_path), which, decomp), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "p0.npz"), np.load(tmp_path / "p1.npz")
    assert np.array_equal(r0["x"], r1["x"]) and int(r0["its"]) == int(r1["its"]) < 10000      # every rank ends with the same x
    x_par = r0["x"]
    TF, n = np.float32, (201, 100)
    x = _draw(TF, n)
    g = sipx.compgrid((1.0, 1.0), n)
    x_ser, log, P, A, opt = _solve(sipx, _parallel_suite_sets(sipx, which, x, n, TF), g, TF, x,
                                   dict(evol_rel_tol=1e-5, feas_tol=1e-5, obj_tol=1e-5, maxit=10000))
    _assert_feasible(x_par, P, A, 1.5, opt.feas_tol)                                            # test_PARSDMM_parallel.jl:34-38 / :115-119
    a, b = x_par.astype(np.float64), x_ser.astype(np.float64)
    assert np.linalg.norm(a - b) <= 5e-4 * max(np.linalg.norm(a), np.linalg.norm(b))            # :72 / :121
