"""GPU tests added in round 4: the logged evol_x of a context without a distance term (the residual product queued ahead must
not overwrite x_old before the log reads it), the device-memory figure of a context, the read-only counters call."""
import re

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


@pytest.mark.parametrize("TF,n,kinds,runs", [
    (np.float32, (64, 48, 40), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], "0"),
    (np.float32, (64, 48, 40), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], "96"),        # with the sampled prediction forced on this small grid
    (np.float64, (48, 40, 24), ["bounds", "l1:TV", "annulus"], "64"),
    (np.float32, (256, 192), ["bounds", "l1:TV"], "0"),
    (np.float32, (36, 30, 20), ["bounds", "l1:D_y"], "0"),                             # a grid line shorter than a wave, one searching set
])
@pytest.mark.parametrize("multi", ["0", "1"])
def test_batched_searches_are_bit_identical(sipx, monkeypatch, TF, n, kinds, runs, multi):
    """One rank, the sweep does the updates: the threshold / scale searches of all sets as ONE chain of launches on the engine
    stream (rescaling, sampled prediction, lean passes in one sweep, sums, decision + solve; the fallback sweeps launched only
    for the sets whose pinned verdict asks for them) against the per-set chains on the set streams (SIPX_SEARCH_BATCH=0): the same
    decisions on the same sums, the same gathered values, the same double-double solve -- theta, x, y, l, the CG counts and every
    log bit for bit over 60 iterations that include feasibility estimates, rho changes and (where forced) sampled predictions."""
    h = (25.0, 25.0, 25.0)[:len(n)]
    m = model(n, TF, seed=7)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m, dict(maxit=60))
    os_.evol_rel_tol = os_.feas_tol = os_.obj_tol = 0.0
    if runs != "0":
        monkeypatch.setenv("SIPX_L1_SAMPLE_RUNS", runs)
    # multi = 1: the full first passes and the fallback passes of the chain as one sweep per group of sets too (k_pass_multi)
    monkeypatch.setenv("SIPX_PASS_MULTI", multi)
    out = {}
    for tag in ("0", "1"):
        monkeypatch.setenv("SIPX_SEARCH_BATCH", tag)
        ctx = sipx.host.build_context(m, AtAs, As, props, Ps, gs, os_)
        ctx.parsdmm_begin(os_)
        ctx.parsdmm_steps(60)
        cnt = ctx.kernel_stats_all(-1)["batched_searches"]
        log = ctx.parsdmm_log()
        x, l, y = ctx.download()
        ctx.close()
        out[tag] = (x, l, y, log, cnt)
    (x0, l0, y0, g0, c0), (x1, l1, y1, g1, c1) = out["0"], out["1"]
    ntp = sum(1 for k in kinds if k.startswith("l1") or k == "annulus")
    assert c0["searches"] == 0 and c1["searches"] == ntp * (60 + 6) and 0 <= c1["fallbacks"] < c1["searches"] // 2, (c0, c1)
    assert np.array_equal(x0, x1)
    for a, b in zip(y0 + l0, y1 + l1):
        assert np.array_equal(a, b)
    assert np.array_equal(g0.cg_it, g1.cg_it) and np.array_equal(g0.obj, g1.obj) and np.array_equal(g0.set_feasibility, g1.set_feasibility)
    assert np.array_equal(g0.rho, g1.rho) and np.array_equal(g0.gamma, g1.gamma)
    assert np.array_equal(g0.r_pri, g1.r_pri) and np.array_equal(g0.r_dual, g1.r_dual)


@pytest.mark.parametrize("TF,n,h,kinds,full", [
    (np.float32, (16, 12, 8), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], "0"),
    (np.float64, (16, 12, 8), (25.0, 12.5, 6.0), ["bounds", "l1:TV", "l1:D_y"], "0"),
    (np.float32, (32, 24), (25.0, 6.0), ["bounds", "l1:TV", "l1:D_x"], "0"),
    (np.float32, (13, 7, 5), (1.0, 2.0, 3.0), ["bounds", "l1:D_z", "l1:TV"], "0"),            # scalar path (n1 not a multiple of 4)
    (np.float64, (16, 12, 8), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:TV"], "1"),         # every band maintained (SIPX_CDS_FULL)
])
def test_planned_q_update_is_bit_identical(sipx, monkeypatch, TF, n, h, kinds, full):
    """k_q_update_plan (alpha_i * (A_i'A_i)[g, g+o] taken from a table over the boundary classes of g that the host formed with
    ata_value's arithmetic, added in set order) against k_q_update (every value regenerated per element, SIPX_Q_PLAN=0) and
    against the oracle's mat2CDS / CDS_scaled_add! (Q_update!.jl:45-48): the assembled Q and Q after three incremental updates
    (all sets, one set, a subset with a negative step) bit for bit."""
    m = model(n, TF, seed=5)
    monkeypatch.setenv("SIPX_CDS_FULL", full)
    go, oo, Po, Ao, propo, AtAo = _problem(O, n, h, TF, kinds, m)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m)
    p = len(Ao)
    rng = np.random.default_rng(1)
    rhos = [list(np.round(rng.uniform(0.5, 20.0, p), 3))]
    r1 = list(np.round(rng.uniform(0.5, 20.0, p), 3))
    r2 = list(r1); r2[1] = 0.731
    r3 = list(r2); r3[0] = 0.0625; r3[-1] = 17.5; r3[2] = r3[2] * 0.25
    rhos += [r1, r2, r3]
    os_.rho_ini = rhos[0]
    got = {}
    for tag in ("0", "1"):
        monkeypatch.setenv("SIPX_Q_PLAN", tag)
        ctx = sipx.host.build_context(m, AtAs, As, props, Ps, gs, os_)
        Qs = [ctx.get_Q()]
        for a, b in zip(rhos[:-1], rhos[1:]):
            ctx.q_update(b, a)
            Qs.append(ctx.get_Q())
        ctx.close()
        got[tag] = Qs
    Qo, offo = O.assemble_Q(AtAo, propo.AtA_offsets, np.array(rhos[0], TF), TF)
    for step, ((Q0, off0), (Q1, off1)) in enumerate(zip(got["0"], got["1"])):
        assert list(off0) == list(off1) == list(offo)
        assert np.array_equal(Q0, Q1), step
        if step > 0:
            class L: pass
            log = L(); log.rho = np.array([rhos[step - 1]])
            changed = [i for i in range(p) if TF(rhos[step][i]) != TF(rhos[step - 1][i])]
            Qo = O.Q_update(Qo.copy(order="F"), AtAo, propo, np.array(rhos[step], TF), changed, log, 0, offo)
        assert np.array_equal(Q1, Qo), step


def test_full_size_c2_solver_properties(sipx):
    """BASELINE configs[1] at its own size -- 2048 x 2048 Float32, {bounds on I, l1 on TV = [D_z; D_x]} + distance term, h = (25, 6)
    as examples/projection_intersection_2D.jl:45 -- solved through the product's entry point: finite logs, the zero-start
    conventions of the first iteration (cg.jl:51), per-set log shapes (PARSDMM_initialize.jl:233-236), the TV set gets closer to
    feasible and ends inside 1.5 feas_tol like the reference's own solver test asks (test_PARSDMM.jl:86-89), y of the l1 set lies
    in its ball, x within the bounds' reach, a second solve warm-started from (x, l, y) stops at once with the same x, and a model
    that is feasible for every set comes back untouched (PARSDMM.jl:63-82)."""
    TF, n, h = np.float32, (2048, 2048), (25.0, 6.0)
    rng = np.random.default_rng(20240601 + 2)
    z = np.linspace(0.0, 1.0, n[1]).reshape(1, -1)
    m = (1500.0 + 2500.0 * z + 150.0 * rng.standard_normal(n)).astype(TF).reshape(-1, order="F")
    g = sipx.compgrid(h, n)
    TV = sipx.get_TD_operator(g, "TV", TF)[0]
    sigma = float(0.5 * np.abs((TV @ m).astype(np.float64)).sum())
    c = [sipx.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", "")),
         sipx.set_definitions("l1", "TV", 0.0, sigma, ("matrix", ""))]
    P, A, prop = sipx.setup_constraints(c, g, TF)
    opt = sipx.PARSDMM_options(FL=TF, maxit=300)
    A, AtA, _, _ = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
    x, log, l, y = sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
    it = len(log.obj)
    assert 6 < it <= 300
    for v in (log.obj, log.r_pri, log.r_dual, log.rho, log.gamma, log.set_feasibility):
        assert np.isfinite(v).all()
    assert np.isnan(log.evol_x[0]) and log.cg_it[0] == 0 and (log.cg_it[1:] >= 1).all() and log.cg_it.max() < 40
    assert log.r_pri.shape == (it, 3) and log.set_feasibility.shape[1] == 2 and len(y) == 3
    assert [len(v) for v in y] == [op.shape[0] for op in A] == [n[0] * n[1], n[0] * (n[1] - 1) + (n[0] - 1) * n[1], n[0] * n[1]]
    f0, f1 = log.set_feasibility[0], log.set_feasibility[-1]
    assert f0[1] > 0.3 and f1[1] < f0[1] and (it == 300 or f1.max() < 1.5 * float(opt.feas_tol)), (it, f0, f1)
    assert np.abs(y[1].astype(np.float64)).sum() <= sigma * (1 + 1e-5)
    assert x.min() > 1500.0 and x.max() < 4000.0 and np.isfinite(x).all()
    # the l1 norm of TV x itself has come down to the radius (to the feasibility the solve stopped at)
    tvx = float(np.abs((TV @ x).astype(np.float64)).sum())
    assert tvx <= sigma * (1 + 2 * max(float(f1[1]), 1e-3)), (tvx, sigma)
    opt2 = sipx.PARSDMM_options(FL=TF, maxit=50, zero_ini_guess=False, rho_ini=[float(r) for r in log.rho[-1]])
    x2, log2, _, _ = sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, opt2, x.copy(), [v.copy() for v in l], [v.copy() for v in y])
    assert len(log2.obj) <= 50 and np.linalg.norm(x2.astype(np.float64) - x) <= 2e-3 * np.linalg.norm(x.astype(np.float64))
    flat = np.full(m.size, 2500.0, TF)
    xf, logf, _, _ = sipx.PARSDMM(flat.copy(), AtA, A, prop, P, g, opt)
    assert np.array_equal(xf, flat) and len(logf.obj) == 1


def _slab_local_worker(rank, world, port, out, kinds, n):
    import datetime
    import os
    import sys
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        from __graft_entry__ import load_package
        sipx = load_package()
        from sipx import sharded
        import torch
        TF = np.float32
        h = (25.0, 25.0, 25.0)[:len(n)]
        m = model(n, TF, seed=5)
        for tag in ("0", "1"):
            os.environ["SIPX_SLAB_LOCAL"] = tag
            g, opt, P, A, prop, AtA = _problem(sipx, n, h, TF, kinds, m, dict(maxit=30, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0))
            A, AtA2, _, _ = A, AtA, None, None
            keep = []

            def attach(cx):
                keep.append(sharded.attach_comm(cx, dist, torch.device("cuda", 0), "torch"))
                cx.set_decomp("slab")
            ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt, device=0, owned=[1] * len(A), attach=attach)
            nbytes = ctx.device_bytes()["context"]
            sparse = ctx.kernel_stats_all(-1)["sparse_arrays"]
            log, _ = ctx.parsdmm(opt)
            x, l, y = ctx.download()
            ctx.close()
            np.savez(os.path.join(out, f"loc{tag}_r{rank}.npz"), x=x, obj=log.obj, rho=log.rho, cg_it=log.cg_it, r_pri=log.r_pri, feas=log.set_feasibility,
                     nbytes=nbytes, sparse=sparse, **{f"y{i}": v for i, v in enumerate(y)}, **{f"l{i}": v for i, v in enumerate(l)})
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(400)
@pytest.mark.parametrize("world,kinds,n", [
    (4, ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], (32, 24, 32)),       # eight planes per rank
    (3, ["bounds", "l1:TV", "annulus"], (32, 24, 16)),                 # ragged slabs (6, 6, 4); a set of three blocks
    (4, ["bounds", "l1:D_z", "l1:D_x"], (12, 10, 5)),                  # 2, 2, 1 planes and a rank with none
    (2, ["bounds", "l1:TV"], (64, 48)),                                # 2-D: slabs of rows
    (4, ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], (256, 128, 64)),     # arrays of 8 MiB: the mapping is really sparse (2 MiB granules)
])
def test_slab_ranks_hold_their_planes_only(sipx, tmp_path, world, kinds, n):
    """sipx_set_decomp(SIPX_DECOMP_SLAB) with sparse arrays (every N-sized array keeps its global index space but is backed by
    memory for the rank's planes and the halo planes around them only) against full-size arrays on every rank (SIPX_SLAB_LOCAL=0):
    x, every y_i, l_i and the logs are IDENTICAL on every rank in both modes (no kernel changed; an access outside a rank's share
    would fault), and what a context allocates falls with the number of ranks."""
    import os
    import torch.multiprocessing as mp
    port = 30100 + (os.getpid() % 1500) + 7 * world
    mp.spawn(_slab_local_worker, args=(world, port, str(tmp_path), kinds, n), nprocs=world, join=True)
    full = [np.load(tmp_path / f"loc0_r{r}.npz") for r in range(world)]
    loc = [np.load(tmp_path / f"loc1_r{r}.npz") for r in range(world)]
    for r in range(world):
        assert not bool(full[r]["sparse"]) and bool(loc[r]["sparse"])
        for k in full[0].files:
            if k in ("nbytes", "sparse"):
                continue
            assert np.array_equal(full[0][k], loc[r][k], equal_nan=True), (r, k)
            assert np.array_equal(full[0][k], full[r][k], equal_nan=True), (r, k)
    N = int(np.prod(n))
    if N >= (1 << 21):                # (smaller grids: an array is one or two 2 MiB granules of the mapping whatever the rank holds)
        assert max(int(v["nbytes"]) for v in loc) < 0.8 * int(full[0]["nbytes"]), ([int(v["nbytes"]) for v in loc], int(full[0]["nbytes"]))


def test_rank_route_packs_the_slices_that_still_need_a_filter(sipx, capfd, monkeypatch):
    """Filtered subspace route of the slice-rank projector (ext_proj.hip, rank_cheb_route): once three quarters of the batch have
    converged, the remaining matrices are packed and the later filters run on those alone.  Fourteen slices with a clear gap behind
    the block and two that are a constant plus white noise: the packed route must be taken, accept on the inertia certificate, and
    end where the whole-batch route (SIPX_RANK_PACK=0) and the full decomposition of every call (SIPX_RANK_CHEB=0) end.
    Reference: src/projectors/project_rank!.jl:26-45."""
    TF = np.float32
    n, h = (128, 128, 16), (25.0, 25.0, 25.0)
    rng = np.random.default_rng(20241005)
    m3 = np.zeros(n)
    for k in range(n[2]):
        if k in (5, 11):
            m3[:, :, k] = 2000.0 + 100.0 * k + 150.0 * rng.standard_normal(n[:2])
        else:
            U, V = rng.standard_normal((n[0], 6)), rng.standard_normal((6, n[1]))
            m3[:, :, k] = 2500.0 + 200.0 * (U @ V) / 6.0 + 1.0 * rng.standard_normal(n[:2])
    m = m3.reshape(-1, order="F").astype(TF)

    def solve():
        g = sipx.compgrid(h, n)
        c = [sipx.set_definitions("bounds", "identity", 1000.0, 4500.0, ("matrix", "")),
             sipx.set_definitions("rank", "identity", 0, 8, ("slice", "z"))]
        opt = sipx.PARSDMM_options(FL=TF, maxit=12)
        opt.evol_rel_tol = opt.feas_tol = opt.obj_tol = 0.0          # run all iterations
        P, A, prop = sipx.setup_constraints(c, g, TF)
        A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
        return sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)[0].astype(np.float64)

    monkeypatch.setenv("SIPX_EXT_DEBUG", "1")
    capfd.readouterr()
    xp = solve()
    err = capfd.readouterr().err
    assert "(packed)" in err and err.count("subspace accepted") >= 4, err[-3000:]
    packed = [int(a) for a in re.findall(r"(\d+) of 16 matrices \(packed\)", err)]
    assert packed and max(packed) <= 4, packed
    monkeypatch.setenv("SIPX_EXT_DEBUG", "0")
    monkeypatch.setenv("SIPX_RANK_PACK", "0")
    xw = solve()
    monkeypatch.delenv("SIPX_RANK_PACK")
    monkeypatch.setenv("SIPX_RANK_CHEB", "0")
    xf = solve()
    nrm = np.linalg.norm(xf)
    assert np.linalg.norm(xp - xw) / nrm < 5e-6, np.linalg.norm(xp - xw) / nrm
    assert np.linalg.norm(xp - xf) / nrm < 2e-5, np.linalg.norm(xp - xf) / nrm


def test_certificate_factorisation_agrees_with_the_library(sipx, capfd, monkeypatch):
    """The inertia certificate of the filtered rank route factors mu I - G + X_r Theta_r X_r' with a blocked Cholesky of its own
    (ext_proj.hip, rank_cert_factor: k_chol_diag + two batched GEMMs per 64 columns).  SIPX_RANK_CERT_CHECK runs it and rocSOLVER's
    potrf side by side on the certificate's matrices (positive definite when the certificate holds) and on matrices shifted below an
    eigenvalue (never definite): same verdict for every matrix, at 128 x 128 (two blocks) and 160 x 160 (a ragged last block)."""
    TF = np.float32
    for n in ((128, 128, 6), (160, 160, 5)):
        h = (25.0, 25.0, 25.0)
        rng = np.random.default_rng(20241006)
        zz = np.linspace(0.0, 1.0, n[2])[None, None, :]
        m = (1500.0 + 2500.0 * zz + 150.0 * rng.standard_normal(n)).reshape(-1, order="F").astype(TF)
        g = sipx.compgrid(h, n)
        c = [sipx.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", "")),
             sipx.set_definitions("rank", "identity", 0, 8, ("slice", "z"))]
        opt = sipx.PARSDMM_options(FL=TF, maxit=8)
        opt.evol_rel_tol = opt.feas_tol = opt.obj_tol = 0.0
        P, A, prop = sipx.setup_constraints(c, g, TF)
        A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
        monkeypatch.setenv("SIPX_RANK_CERT_CHECK", "1")
        capfd.readouterr()
        sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
        err = capfd.readouterr().err
        monkeypatch.delenv("SIPX_RANK_CERT_CHECK")
        lines = re.findall(r"certificate check \((.*?) shift\): (\d+) of (\d+) matrices not positive definite, the two factorisations differ on (\d+)", err)
        assert len(lines) >= 4, err[-2000:]
        for which, indef, batch, differ in lines:
            assert int(differ) == 0
            if which == "low":
                assert int(indef) == int(batch) == n[2]


@pytest.mark.parametrize("TF,n", [(np.float32, (32, 32, 12)), (np.float64, (16, 12, 8))])
def test_rank_set_on_its_lane_is_bit_identical(sipx, TF, n):
    """One rank, BASELINE config 4's list: the slice-rank set's update runs on a stream of its own, queued by a host thread of its
    own (engine.cpp, lane_start / lane_join), beside the searches, the sweep and the other loose sets.  Same kernels on the same
    operands: x, y, l and every logged scalar must equal the in-turn run (SIPX_RANK_LANE=0) bit for bit, feasibility iterations
    (the lane set's estimate is formed after the join) and Barzilai-Borwein iterations included.
    Reference: src/update_y_l.jl:36-101 (the sets are independent given x)."""
    import os
    h = (25.0, 25.0, 25.0)
    m = model(n, TF, seed=11)
    kinds = ["bounds", "l1:D_x", "l1:D_y", "l1:D_z", "annulus", "l1dft", "rank:3", "card:D_z"]
    kw = dict(maxit=23)
    out = []
    for lane in ("1", "0"):
        os.environ["SIPX_RANK_LANE"] = lane
        try:
            g, o, P, A, prop, AtA = _problem(sipx, n, h, TF, kinds, m, kw)
            o.evol_rel_tol = o.feas_tol = o.obj_tol = 0.0
            out.append(sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, o))
        finally:
            del os.environ["SIPX_RANK_LANE"]
    g, o, P, A, prop, AtA = _problem(sipx, n, h, TF, kinds, m, kw)
    ctx = sipx.host.build_context(m, AtA, A, prop, P, g, o)
    try:
        assert ctx.kernel_stats_all(-1)["lane_set"] == 6          # the rank set
    finally:
        ctx.close()
    (x1, log1, l1, y1), (x0, log0, l0, y0) = out
    assert len(log1.obj) == len(log0.obj) == 23
    assert np.array_equal(x1, x0)
    for a, b in zip(list(l1) + list(y1), list(l0) + list(y0)):
        assert np.array_equal(a, b)
    for f in ("obj", "evol_x", "r_pri", "r_dual", "rho", "gamma", "set_feasibility", "cg_it"):
        assert np.array_equal(np.asarray(getattr(log1, f)), np.asarray(getattr(log0, f)), equal_nan=True), f


def test_bench_headline_survives_a_leg_that_fails_on_one_rank():
    """bench.py at N > 1: a secondary leg that fails on ONE rank (round 4's four-rank rehearsal: the fourth rank ran out of device
    memory in the c5 leg, went on into the next leg's collectives while the others waited in this one's -- gloo aborted on the
    mismatch and no line was printed).  The failing rank now waits for the others on the process group's store and parks when they
    do not arrive; the rank left in the leg's collectives is released by the leg's deadline; rank 0 prints the line it has -- the
    headline and both decompositions -- with the leg's error, exit code 0.  Rehearsed with two ranks on the one GPU, rank 1 failing
    the c3_512 leg by a test hook."""
    import json
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SIPX_BENCH_SHARE_GPU="1", SIPX_BENCH_FAIL_LEG="c3_512:1", SIPX_BENCH_LEG_DEADLINE="15", SIPX_BENCH_AGREE_S="3")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    detail = os.path.join(tempfile.mkdtemp(prefix="sipx_bench_"), "detail.json")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-c4", "--no-c5",
                        "--detail", detail], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096, r.stdout[:2000]
    h = json.loads(lines[0])
    assert h["n_gpus"] == 2 and h["value"] > 0 and set(h["decompositions"]) == {"slab", "sets"}
    assert "error" in h["c3_512"] and "comm_probe_us" not in h             # the leg's error; nothing behind it was attempted
    d = json.load(open(detail))
    assert d["legs_abandoned_at"] == "c3_512" and d["c3_512"]["rank_local_failure"] is True


@pytest.mark.parametrize("TF,n", [(np.float32, (32, 24, 16)), (np.float64, (15, 12, 9)), (np.float64, (30, 21)), (np.float32, (64, 48))])
def test_l1_behind_the_dft_through_the_real_transform(sipx, monkeypatch, TF, n):
    """l1 ball on the Fourier coefficients (`A'*project_l1_Duchi!(A*x)` with joDFT, src/get_projector.jl:25-33): the model is real, so
    the engine transforms it with hipFFT's R2C / C2R pair -- half the spectrum, no packing -- and lets the search see all N magnitudes
    by writing those of the planes whose conjugates are not stored a second time (ext_proj.hip, k_cabs_half).  Against the complex
    transform of the packed model (SIPX_DFT_REAL=0) and the oracle: even and odd leading dimensions, 2-D and 3-D, both precisions;
    a model inside the ball comes back bit for bit."""
    h = (25.0, 25.0, 25.0)[:len(n)]
    m = model(n, TF, seed=5)
    kinds = ["bounds", "l1dft"]
    out = []
    for real in ("1", "0"):
        monkeypatch.setenv("SIPX_DFT_REAL", real)
        g, o, P, A, prop, AtA = _problem(sipx, n, h, TF, kinds, m, dict(maxit=40))
        out.append(sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, o))
    monkeypatch.delenv("SIPX_DFT_REAL")
    go, oo, Po, Ao, propo, AtAo = _problem(O, n, h, TF, kinds, m, dict(maxit=40))
    xo, lo, _, _ = O.PARSDMM(m.copy(), AtAo, Ao, propo, Po, go, oo)
    (xr, lr, _, _), (xc, lc, _, _) = out
    nrm = np.linalg.norm(xo)
    tol = 2e-5 if TF == np.float32 else 1e-9
    assert np.linalg.norm(xr.astype(np.float64) - xc.astype(np.float64)) / nrm < tol
    assert np.linalg.norm(xr.astype(np.float64) - xo) / nrm < (5e-4 if TF == np.float32 else 1e-6)
    assert len(lr.obj) == len(lc.obj) == len(lo.obj)
    # the projector alone on a vector inside the ball: untouched
    P1 = sipx.setup_constraints([sipx.set_definitions("l1", "DFT", 0.0, 1e30, ("matrix", ""))], sipx.compgrid(h, n), TF)[0]
    v = m.copy()
    assert np.array_equal(np.asarray(P1[0](v.copy())), v)


@pytest.mark.parametrize("stride", ["1", "7"])
def test_section_timing_adds_up(sipx, monkeypatch, stride):
    """log_PARSDMM's timing sections (src/PARSDMM.jl:84-257: T_rhs, T_cg, T_y_l_upd, T_Q_upd ...) are intervals between event records on
    the engine stream.  Round 4: a step whose residual product had been queued ahead opened its x-step interval nowhere, and "argmin x"
    went uncounted on every such iteration (0.21 of 0.5 ms at 256^3: the sections summed to 82 % of the wall time) -- the interval
    now opens where the product is queued.  With a record on every iteration the sections add up to the wall time of the loop; on
    small grids the engine records on a sample of the iterations (every seventh) and scales: an estimate, still within a quarter."""
    import time
    TF, n, h = np.float32, (128, 128, 96), (25.0, 25.0, 25.0)
    m = model(n, TF, seed=9)
    monkeypatch.setenv("SIPX_MARK_STRIDE", stride)
    g, o, P, A, prop, AtA = _problem(sipx, n, h, TF, ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], m,
                                     dict(maxit=90, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0))
    ctx = sipx.host.build_context(m, AtA, A, prop, P, g, o)
    try:
        ctx.parsdmm_begin(o)
        ctx.parsdmm_steps(6)                       # (first launches of every kernel)
        ctx.debug_proj(1, 0)                       # (drains the engine's stream)
        t0 = time.perf_counter()
        done = ctx.parsdmm_steps(84)
        ctx.debug_proj(1, 0)
        wall = time.perf_counter() - t0
        log = ctx.parsdmm_log()
        assert len(log.obj) == 90 and done
        secs = {k: v for k, v in log.timing.items() if k != "initialization"}
        total = sum(secs.values())
        # (the six warm-up iterations are in the sections too: a fifteenth of the run)
        lo, hi = (0.9, 1.25) if stride == "1" else (0.7, 1.45)
        assert lo * wall < total < hi * wall, (wall, total, secs)
        assert secs["argmin x"] > 0.2 * total and secs["argmin y and l update"] > 0.3 * total, secs
    finally:
        ctx.close()
