"""GPU tests added in round 5: a context used again (sipx_reset, the context cache of host.PARSDMM), the slice-rank projector's
acceptance level (the class of the reference's Float32 svd, src/projectors/project_rank!.jl:26-45) and its start without a
previous call, the communicator self-test of sipx_finalize."""
import os
import re

import numpy as np
import pytest

from oracle import parsdmm_oracle as O      # checker only
from tests.test_gpu_parity import C4_KINDS, _problem, model

pytestmark = pytest.mark.gpu

LOG_FIELDS = ("set_feasibility", "r_dual", "r_pri", "r_dual_total", "r_pri_total", "obj", "evol_x", "rho", "gamma", "cg_it", "cg_relres")

RESET_CASES = [
    ("c3-3d", (48, 40, 32), (25.0, 25.0, 25.0), np.float32, ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], 33),
    ("c1-2d-tv", (96, 64), (25.0, 6.0), np.float32, ["bounds", "l1:TV"], 40),
    ("c5-3d-tv-f64", (24, 20, 16), (25.0, 25.0, 25.0), np.float64, ["bounds", "l1:TV"], 30),
    ("c4-eight-sets", (32, 24, 16), (25.0, 25.0, 25.0), np.float32, C4_KINDS, 24),
    ("flat-slice-rank", (96, 96, 4), (25.0, 25.0, 25.0), np.float32, ["bounds", "rank:6"], 12),
]


def _same_logs(a, b):
    for f in LOG_FIELDS:
        x, y = np.asarray(getattr(a, f)), np.asarray(getattr(b, f))
        assert x.shape == y.shape and np.array_equal(x, y, equal_nan=True), f


@pytest.mark.parametrize("name,n,h,TF,kinds,maxit", RESET_CASES, ids=[c[0] for c in RESET_CASES])
def test_a_reset_context_gives_the_bits_of_a_new_one(sipx, name, n, h, TF, kinds, maxit):
    """sipx_reset (include/sipx.h): the same sets on the same grid with another model, on a context that has been solved before --
    how the reference's callers use PARSDMM as a projector inside an outer loop (examples/constrained_freq_FWI_simple.jl:468).
    x, l, y and every log of the solve that follows must EQUAL those of a newly built context, bit for bit, whatever the context
    went through before: a full solve on another model, a solve cut short, a warm start."""
    m1, m2 = model(n, TF, seed=11), model(n, TF, seed=12)
    m2 = (m2 * TF(0.97) + TF(40.0)).astype(TF)                       # another model, not a re-seeded copy of the scale
    kw = dict(maxit=maxit, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0)
    g, opt, P, A, prop, AtA = _problem(sipx, n, h, TF, kinds, m1, kw)

    def fresh(m):
        ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt)
        try:
            log, _ = ctx.parsdmm(opt)
            return ctx.download(), log
        finally:
            ctx.close()

    (x1, l1, y1), log1 = fresh(m1)
    (x2, l2, y2), log2 = fresh(m2)
    assert not np.array_equal(x1, x2)
    TFt = np.dtype(TF).type
    rho_ini = [float(TFt(r)) for r in opt.rho_ini]
    ctx = sipx.host.build_context(m1, AtA, A, prop, P, g, opt)
    try:
        ctx.parsdmm(opt)                                             # a whole solve on m1 ...
        ctx.reset(m2, rho_ini, float(TFt(opt.gamma_ini)))
        log, _ = ctx.parsdmm(opt)                                    # ... then m2 on the same context
        xr, lr, yr = ctx.download()
        assert np.array_equal(xr, x2)
        for a, b in zip(lr + yr, l2 + y2):
            assert np.array_equal(a, b)
        _same_logs(log, log2)
        # a solve cut short (state in the middle of an iteration pattern), then back to m1
        ctx.reset(m1, rho_ini, float(TFt(opt.gamma_ini)))
        ctx.parsdmm_begin(opt)
        ctx.parsdmm_steps(maxit // 2 + 1)
        ctx.reset(m1, rho_ini, float(TFt(opt.gamma_ini)))
        log, _ = ctx.parsdmm(opt)
        xr, lr, yr = ctx.download()
        assert np.array_equal(xr, x1)
        for a, b in zip(lr + yr, l1 + y1):
            assert np.array_equal(a, b)
        _same_logs(log, log1)
    finally:
        ctx.close()


def test_reset_takes_a_warm_start_and_another_rho(sipx):
    """sipx_reset with zero_ini_guess = 0 and other initial penalties == sipx_finalize of a new context with the same arguments
    (src/PARSDMM_initialize.jl:120-127,304-313; src/PARSDMM_multi_level.jl:81-83 carries x, l, y and rho from solve to solve)."""
    TF, n, h = np.float32, (40, 32, 24), (25.0, 25.0, 25.0)
    m = model(n, TF, seed=5)
    kw = dict(maxit=25, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0)
    g, opt, P, A, prop, AtA = _problem(sipx, n, h, TF, ["bounds", "l1:D_x", "l1:D_z"], m, kw)
    x0, log0, l0, y0 = sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
    rho2 = [float(TF(v)) for v in log0.rho[-1]]
    g2, opt2, P2, A2, prop2, AtA2 = _problem(sipx, n, h, TF, ["bounds", "l1:D_x", "l1:D_z"], m, dict(kw, zero_ini_guess=False, rho_ini=rho2))
    ref = sipx.host.build_context(m, AtA2, A2, prop2, P2, g2, opt2, x0.copy(), [v.copy() for v in l0], [v.copy() for v in y0])
    try:
        logr, _ = ref.parsdmm(opt2)
        xr, lr, yr = ref.download()
    finally:
        ref.close()
    ctx = sipx.host.build_context(model(n, TF, seed=6), AtA, A, prop, P, g, opt)
    try:
        ctx.parsdmm(opt)
        ctx.reset(m, rho2, float(TF(opt.gamma_ini)), zero_ini_guess=False, x0=x0, l0=l0, y0=y0)
        log, _ = ctx.parsdmm(opt2)
        xs, ls, ys = ctx.download()
    finally:
        ctx.close()
    assert np.array_equal(xs, xr)
    for a, b in zip(ls + ys, lr + yr):
        assert np.array_equal(a, b)
    _same_logs(log, logr)


def test_parsdmm_keeps_its_context_between_calls(sipx, monkeypatch):
    """host.PARSDMM looks its context up by (device, precision, grid, Q mode, set descriptors, SIPX_* switches): the second call
    with the same sets runs on the first call's context (sipx_reset) and returns what a call without the cache returns; a list
    with array arguments (bound vectors) is never cached; outputs="x" leaves l and y on the device."""
    TF, n, h = np.float32, (40, 32, 24), (25.0, 25.0, 25.0)
    m1, m2 = model(n, TF, seed=1), model(n, TF, seed=2)
    kw = dict(maxit=60)
    g, opt, P, A, prop, AtA = _problem(sipx, n, h, TF, ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], m1, kw)
    sipx.clear_context_cache()
    monkeypatch.setenv("SIPX_CONTEXT_CACHE", "0")
    xa, loga, la, ya = sipx.PARSDMM(m1.copy(), AtA, A, prop, P, g, opt)
    xb, logb, lb, yb = sipx.PARSDMM(m2.copy(), AtA, A, prop, P, g, opt)
    assert not loga.context_reused and not logb.context_reused
    monkeypatch.delenv("SIPX_CONTEXT_CACHE")
    x1, log1, l1, y1 = sipx.PARSDMM(m1.copy(), AtA, A, prop, P, g, opt)
    x2, log2, l2, y2 = sipx.PARSDMM(m2.copy(), AtA, A, prop, P, g, opt)
    assert not log1.context_reused and log2.context_reused
    assert np.array_equal(x1, xa) and np.array_equal(x2, xb)
    _same_logs(log1, loga)
    _same_logs(log2, logb)
    for a, b in zip(l2 + y2, lb + yb):
        assert np.array_equal(a, b)
    buf = np.zeros_like(m1)
    x3, log3, l3, y3 = sipx.PARSDMM(m1.copy(), AtA, A, prop, P, g, opt, x=buf, outputs="x")
    assert log3.context_reused and x3 is buf and l3 is None and y3 is None and np.array_equal(buf, xa)
    # a switch the engine reads when it builds a context: another context
    monkeypatch.setenv("SIPX_YL_MULTI", "0")
    x4, log4, _, _ = sipx.PARSDMM(m1.copy(), AtA, A, prop, P, g, opt)
    assert not log4.context_reused and np.array_equal(x4, xa)
    monkeypatch.delenv("SIPX_YL_MULTI")
    # array arguments inside a descriptor: built per call
    lo, hi = np.full(m1.size, 1600.0, TF), np.full(m1.size, 3900.0, TF)
    c = [sipx.set_definitions("bounds", "identity", lo, hi, ("matrix", ""))]
    Pv, Av, propv = sipx.setup_constraints(c, g, TF)
    Av, AtAv, _, _ = sipx.PARSDMM_precompute_distribute(Av, propv, g, opt)
    for _ in range(2):
        _, lv, _, _ = sipx.PARSDMM(m1.copy(), AtAv, Av, propv, Pv, g, opt)
        assert not lv.context_reused
    sipx.clear_context_cache()


def _flat_slices(n, TF, seed):
    """slices that are a constant plus white noise -- BASELINE config 4's synthetic model: no gap behind any singular value"""
    rng = np.random.default_rng(seed)
    zz = np.linspace(0.0, 1.0, n[2])[None, None, :]
    return (1500.0 + 2500.0 * zz + 150.0 * rng.standard_normal(n)).reshape(-1, order="F").astype(TF)


@pytest.mark.parametrize("n,r", [((192, 192, 3), 32), ((256, 160, 2), 12), ((160, 256, 2), 12)])
def test_slice_rank_projection_is_in_the_class_of_the_float32_svd(sipx, n, r, capfd, monkeypatch):
    """The reference projects a Float32 slice through svd() IN Float32 (src/projectors/project_rank!.jl:26-45: LAPACK's sgesdd).
    Round 5 accepts a Ritz pair of the slice-rank projector at the backward error such an SVD leaves on the slice itself,
    ||E||_2 <= 2^-23 ||X||_2 = eps(Float32) ||X||_2 (ext_proj.hip, k_sub_residual), instead of 1e-12 theta_max on the Gram matrix in Float64.  Leaf test of
    that class against the oracle's Float32 LAPACK SVD, on slices without a spectral gap, the projector starting COLD (no previous
    call: the ramp of short filters, no full decomposition):
      * the engine's projection is as close to the exact (Float64 SVD) projection of the same Float32 slices as the reference's
        own arithmetic is -- within twice its error plus Float32 rounding of the result;
      * engine and oracle agree to a Float32 tolerance; every projected slice has rank r; a second projection changes nothing more
        than Float32 rounding (project_rank! is idempotent: test/test_projectors.jl:94-104)."""
    TF = np.float32
    v = _flat_slices(n, TF, 20240611)
    g, go = sipx.compgrid((1.0, 1.0, 1.0), n), O.compgrid((1.0, 1.0, 1.0), n)
    c = ("rank", "identity", 0, r, ("slice", "z"))
    monkeypatch.setenv("SIPX_EXT_DEBUG", "1")
    capfd.readouterr()
    w = sipx.Projector(sipx.set_definitions(*c), g, TF)(v.copy())
    err = capfd.readouterr().err
    monkeypatch.delenv("SIPX_EXT_DEBUG")
    assert "ramp" in err and "subspace accepted" in err and "full decomposition" not in err, err[-2000:]
    ref32 = O.get_projector(O.set_definitions(*c), TF, go)(v.copy())                 # Float32 LAPACK SVD: the reference's arithmetic
    exact = O.get_projector(O.set_definitions(*c), np.float64, go)(v.astype(np.float64))
    nrm = np.linalg.norm(exact)
    e_ref = np.linalg.norm(ref32.astype(np.float64) - exact) / nrm
    e_eng = np.linalg.norm(w.astype(np.float64) - exact) / nrm
    d = np.linalg.norm(w.astype(np.float64) - ref32.astype(np.float64)) / nrm
    print(f"rel. distance to the exact projection: reference's Float32 SVD {e_ref:.2e}, engine {e_eng:.2e}; engine - oracle {d:.2e}")
    assert e_eng <= 2.0 * e_ref + 2e-7, (e_eng, e_ref)
    assert d <= 3.0 * e_ref + 3e-7, (d, e_ref)
    W = w.reshape(n, order="F").astype(np.float64)
    for k in range(n[2]):
        s = np.linalg.svd(W[:, :, k], compute_uv=False)
        assert s[r] <= 1e-5 * s[0], (k, s[r] / s[0])                  # rank r to Float32 rounding of the stored slice
    w2 = sipx.Projector(sipx.set_definitions(*c), g, TF)(w.copy())
    assert np.linalg.norm(w2.astype(np.float64) - w.astype(np.float64)) <= 2e-6 * np.linalg.norm(w)


def test_strict_rank_route_is_still_there(sipx, capfd, monkeypatch):
    """SIPX_RANK_STRICT=1: the acceptance level of rounds 3-4, 1e-12 theta_max on the Gram matrix -- the result then agrees with the
    exact projection as well as Float32 storage allows."""
    TF, n, r = np.float32, (160, 160, 2), 12
    v = _flat_slices(n, TF, 20240612)
    g, go = sipx.compgrid((1.0, 1.0, 1.0), n), O.compgrid((1.0, 1.0, 1.0), n)
    c = ("rank", "identity", 0, r, ("slice", "z"))
    monkeypatch.setenv("SIPX_RANK_STRICT", "1")
    w = sipx.Projector(sipx.set_definitions(*c), g, TF)(v.copy())
    monkeypatch.delenv("SIPX_RANK_STRICT")
    exact = O.get_projector(O.set_definitions(*c), np.float64, go)(v.astype(np.float64))
    assert np.linalg.norm(w.astype(np.float64) - exact) <= 1.5e-7 * np.linalg.norm(exact)


def test_first_iterations_of_a_solve_do_not_decompose_fully(sipx, capfd, monkeypatch):
    """Rounds 3-4 decomposed the slices fully on iterations 1 and 2 of every solve (166 + 233 ms at 512^3): iteration 1 projects
    v = 0 (rhs = 0, x = 0 with y = l = 0: PARSDMM.jl:101-107) -- nothing to do, P(0) = 0 -- and iteration 2 has no usable start.
    Now: no full decomposition at all in a solve of the flat-spectrum model, and the iterates stay within the reference's own
    serial-vs-parallel tolerance (test/test_PARSDMM_parallel.jl:72) of the oracle's (Float32 LAPACK SVD in every call)."""
    TF, n, h, r = np.float32, (128, 128, 6), (25.0, 25.0, 25.0), 8
    m = _flat_slices(n, TF, 20240604)
    kw = dict(maxit=12, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0)

    def solve(mod):
        g, opt, P, A, prop, AtA = _problem(mod, n, h, TF, ["bounds", f"rank:{r}"], m, kw)
        return mod.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)

    monkeypatch.setenv("SIPX_EXT_DEBUG", "1")
    capfd.readouterr()
    xs, ls, _, _ = solve(sipx)
    err = capfd.readouterr().err
    monkeypatch.delenv("SIPX_EXT_DEBUG")
    assert "every slice is zero" in err and "full decomposition" not in err, err[-3000:]
    assert len(re.findall(r"subspace accepted", err)) >= 11
    xo, lo, _, _ = solve(O)
    assert np.linalg.norm(xs.astype(np.float64) - xo) <= 5e-4 * np.linalg.norm(xo)
    assert np.allclose(ls.obj[:8], lo.obj[:8], rtol=2e-3)


def _selftest_worker(rank, world, port, out, backend, mode, fail):
    import json
    import sys
    import datetime
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    if fail:
        os.environ["SIPX_COMM_SELFTEST_FAIL"] = fail
    kw = dict(device_id=torch.device("cuda", 0)) if backend == "nccl" else {}
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180), **kw)
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        from __graft_entry__ import load_package
        sipx = load_package()
        from sipx import sharded
        TF, n, h = np.float32, (32, 24, 16), (25.0, 25.0, 25.0)
        m = model(n, TF, seed=2)
        g, opt, P, A, prop, AtA = _problem(sipx, n, h, TF, ["bounds", "l1:D_x", "l1:D_z"], m, dict(maxit=8, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0))
        keep = []

        def attach(cx):
            keep.append(sharded.attach_comm(cx, dist, torch.device("cuda", 0), mode=mode))
            cx.set_decomp("slab")
        res = {}
        try:
            ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt, attach=attach)
        except sipx.SipxError as e:
            res = {"error": str(e)}
        else:
            try:
                st = ctx.kernel_stats_all(-1)
                log, _ = ctx.parsdmm(opt)
                x, _, _ = ctx.download(want_ly=False)
                res = {"comm_selftest": st["comm_selftest"], "sparse_arrays": st["sparse_arrays"], "finite": bool(np.isfinite(log.obj).all()),
                       "x_sum": float(x.astype(np.float64).sum())}
            finally:
                ctx.close()
        json.dump(res, open(os.path.join(out, f"s{rank}.json"), "w"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,backend,mode,fail", [(1, "nccl", "rccl", ""), (1, "nccl", "rccl", "mapped"), (2, "gloo", "torch", ""),
                                                     (3, "gloo", "torch", "mapped:1"), (2, "gloo", "torch", "base:1")])
def test_communicator_self_test_at_finalize(sipx, tmp_path, world, backend, mode, fail):
    """sipx_finalize runs the attached communicator's operations once on KNOWN data before anything is allocated (engine.cpp,
    comm_self_test): the grouped all-reduce + neighbour exchange, the in-place reduce-scatter / all-gather, the fan scatter /
    gather, then the neighbour exchange out of hipMemMap-backed memory -- through RCCL with a world of one (all a one-GPU box
    allows: the operations are degenerate but issued) and through the callback communicator with two and three ranks on one GPU
    (the checks themselves at N > 1).  A failure of the mapped exchange on ONE rank switches EVERY rank to full-size arrays and the
    solve goes on, the same on all ranks; wrong data from a base operation is an error of sipx_finalize on every rank."""
    import json
    import torch.multiprocessing as mp
    port = 32100 + (os.getpid() % 2000)
    mp.spawn(_selftest_worker, args=(world, port, str(tmp_path), backend, mode, fail), nprocs=world, join=True)
    res = [json.load(open(tmp_path / f"s{r}.json")) for r in range(world)]
    if fail.startswith("base"):
        assert all("communicator self-test failed" in r.get("error", "") for r in res), res
        return
    for r in res:
        assert r["finite"], r
        if fail:
            assert r["comm_selftest"].startswith("passed; exchange out of hipMemMap-backed memory failed") and r["sparse_arrays"] is False, r
        else:
            assert r["comm_selftest"] == "passed" and r["sparse_arrays"] is True, r
    assert len({r["x_sum"] for r in res}) == 1, res


def _bench_two_ranks(extra_env, extra_args=(), timeout=400):
    import json
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SIPX_BENCH_SHARE_GPU="1", SIPX_BENCH_AGREE_S="5", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    detail = os.path.join(tempfile.mkdtemp(prefix="sipx_bench_"), "detail.json")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-512", "--no-c4",
                        "--no-c5", "--detail", detail, *extra_args], capture_output=True, text=True, timeout=timeout, env=env)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    return r, lines, (json.load(open(detail)) if os.path.exists(detail) else None)


def test_bench_headline_survives_a_rank_that_fails_it():
    """bench.py at N > 1 (round 5): the HEADLINE fails on one rank before any collective (SIPX_BENCH_FAIL_LEG=headline:1) -- the ranks
    tell each other through the store, every rank moves on to the next attempt of the chain (whole arrays instead of the rank's
    planes), and the line carries `value`, both decompositions, what was fallen back from and the error; exit code 0.  Round 4 ran the
    headline outside every protection: the run came back empty."""
    import json
    r, lines, d = _bench_two_ranks({"SIPX_BENCH_FAIL_LEG": "headline:1"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1 and len(lines[0]) < 4096, r.stdout[:2000]
    h = json.loads(lines[0])
    assert h["n_gpus"] == 2 and h["value"] > 0 and set(h["decompositions"]) == {"slab", "sets"}
    assert h["comm"]["fell_back_from"] and h["comm"]["sparse_arrays"] is False and h["comm"]["ranks_agree_on_x"] is True
    att = h["headline_attempts"]
    assert att[0]["ok"] is False and "test hook" in att[0]["error"] and att[1]["ok"] is True


def test_bench_headline_falls_back_when_finalize_fails_on_one_rank():
    """A rank that runs out of device memory inside sipx_finalize (test hook SIPX_FINALIZE_FAIL_RANK) used to throw by itself while the
    others waited for it in the collectives of the initial feasibility.  sipx_finalize now lets the ranks agree on the outcome of their
    allocations (one all-reduce on the self-test's buffer): every rank raises, bench.py's agreement sees the attempt fail alike and the
    chain goes on -- here to its end, because the hook fails every context of rank 1: the line then says so, value 0, and still prints."""
    import json
    r, lines, d = _bench_two_ranks({"SIPX_FINALIZE_FAIL_RANK": "1"})
    assert len(lines) == 1, (r.stdout[:2000], r.stderr[-3000:])
    h = json.loads(lines[0])
    assert h["value"] == 0.0 and "every attempt" in h["error"]
    assert len(h["headline_attempts"]) >= 3 and all(not a["ok"] for a in h["headline_attempts"])
    assert all("failed on 1 of 2 ranks" in a["error"] for a in d["headline_attempts"]), d["headline_attempts"]


def test_float32_loop_of_the_rank_projector(sipx, capfd, monkeypatch):
    """SIPX_RANK_F32=1 (measured, not the default: DESIGN 3): the filters of a warm call run in Float32 on the Gram matrices with the
    pairs far above the rest taken out in Float64 (ext_proj.hip, k_defl_*, cheb_loop<T, float>).  The loop must be taken, end in
    accepted calls (inertia certificate in Float64 on the matrices themselves) and leave the iterates where the Float64 loop leaves
    them -- both inside the reference's serial-vs-parallel tolerance of the oracle's Float32 LAPACK SVD (test_PARSDMM_parallel.jl:72)."""
    TF, n, h, r = np.float32, (128, 128, 6), (25.0, 25.0, 25.0), 8
    m = _flat_slices(n, TF, 20240604)
    kw = dict(maxit=14, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0)

    def solve(mod):
        g, opt, P, A, prop, AtA = _problem(mod, n, h, TF, ["bounds", f"rank:{r}"], m, kw)
        return mod.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)

    monkeypatch.setenv("SIPX_RANK_F32", "1")
    monkeypatch.setenv("SIPX_EXT_DEBUG", "1")
    capfd.readouterr()
    x32, l32, _, _ = solve(sipx)
    err = capfd.readouterr().err
    monkeypatch.delenv("SIPX_EXT_DEBUG")
    monkeypatch.delenv("SIPX_RANK_F32")
    assert err.count("Float32 loop on the deflated matrices: yes") >= 6 and "(Float32)" in err, err[-3000:]
    assert err.count("subspace accepted") >= 11 and "full decomposition" not in err
    x64, l64, _, _ = solve(sipx)
    xo, lo, _, _ = solve(O)
    nrm = np.linalg.norm(xo)
    d32, d64 = np.linalg.norm(x32.astype(np.float64) - xo) / nrm, np.linalg.norm(x64.astype(np.float64) - xo) / nrm
    print(f"Float32 loop - oracle {d32:.2e}, Float64 loop - oracle {d64:.2e}")
    assert d32 < 5e-4 and d64 < 5e-4
    assert np.allclose(l32.obj[:8], lo.obj[:8], rtol=2e-3)


@pytest.mark.gpu
def test_multilevel_returns_x_alone_when_asked(sipx):
    """PARSDMM_multi_level(..., outputs="x"): the same x and log, l and y left on the device (None) -- what a caller that uses the
    projection alone asks for (bench.py reports the difference as c5.whole_solve_x_only_s)."""
    from sipx import multilevel as ML
    from tests.test_gpu_parity import _ml_problem
    ML_, m, opt, L = _ml_problem(sipx, (24, 20, 16), (25.0, 20.0, 10.0), np.float32, 2)
    xa, loga, la, ya = ML.PARSDMM_multi_level(m.copy(), *L[:5], opt)
    xb, logb, lb, yb = ML.PARSDMM_multi_level(m.copy(), *L[:5], opt, outputs="x")
    assert lb is None and yb is None and la is not None
    assert np.array_equal(xa, xb) and np.array_equal(loga.obj, logb.obj)
    with pytest.raises(sipx.host.SipxError):
        ML.PARSDMM_multi_level(m.copy(), *L[:5], opt, outputs="y")


@pytest.mark.gpu
@pytest.mark.parametrize("TF,levels", [(np.float32, 3), (np.float64, 2)])
def test_multilevel_keeps_its_level_contexts_for_the_next_call(sipx, monkeypatch, TF, levels):
    """PARSDMM_multi_level called again with the same sets on the same grids (the projector of an outer loop,
    examples/constrained_freq_FWI_simple.jl:468): the level contexts of the last call are reset (sipx_reset) instead of built anew
    -- an allocation right behind the release of the last call's arrays waits for the driver's wipe of them, 2.4 s against 1.0 at
    512^3 Float64 -- and the call returns the bits of a call that builds them (another model, carried rho, warm start between the
    levels included).  Another problem releases the kept contexts; SIPX_MULTILEVEL_CACHE=0 keeps none.
    Reference: src/PARSDMM_multi_level.jl:8-89."""
    from sipx import multilevel as ML
    from tests.test_gpu_parity import _ml_problem
    _, m, opt, L = _ml_problem(sipx, (24, 20, 16), (25.0, 20.0, 10.0), TF, levels)
    m2 = (m + TF(7.0) * np.sin(np.arange(m.size, dtype=TF) * TF(0.01))).astype(TF)
    fields = ("obj", "evol_x", "r_pri", "r_dual", "rho", "gamma", "set_feasibility", "cg_it")

    def same(a, b):
        assert np.array_equal(a[0], b[0])
        for u, v in zip(list(a[2]) + list(a[3]), list(b[2]) + list(b[3])):
            assert np.array_equal(u, v)
        for f in fields:
            assert np.array_equal(np.asarray(getattr(a[1], f)), np.asarray(getattr(b[1], f)), equal_nan=True), f

    sipx.clear_context_cache()
    try:
        T = [{} for _ in range(5)]
        ML.PARSDMM_multi_level(m.copy(), *L[:5], opt, timings=T[0])
        assert T[0]["contexts_reused"] is False and len(ML._level_cache) == 1
        hit = ML.PARSDMM_multi_level(m2.copy(), *L[:5], opt, timings=T[1])            # the kept levels, another model
        assert T[1]["contexts_reused"] is True and len(ML._level_cache) == 1
        hit_x = ML.PARSDMM_multi_level(m2.copy(), *L[:5], opt, timings=T[2], outputs="x")
        assert T[2]["contexts_reused"] is True and np.array_equal(hit_x[0], hit[0]) and hit_x[2] is None
        sipx.clear_context_cache()
        assert len(ML._level_cache) == 0
        new = ML.PARSDMM_multi_level(m2.copy(), *L[:5], opt, timings=T[3])            # built anew
        assert T[3]["contexts_reused"] is False
        same(hit, new)
        # another problem (another grid): the kept contexts go, this call's stay
        key = next(iter(ML._level_cache))
        _, mb, opt_b, Lb = _ml_problem(sipx, (20, 16, 16), (25.0, 20.0, 10.0), TF, levels)
        ML.PARSDMM_multi_level(mb.copy(), *Lb[:5], opt_b, timings=T[4])
        assert T[4]["contexts_reused"] is False and len(ML._level_cache) == 1 and next(iter(ML._level_cache)) != key
        monkeypatch.setenv("SIPX_MULTILEVEL_CACHE", "0")
        sipx.clear_context_cache()
        Tn = {}
        off = ML.PARSDMM_multi_level(m2.copy(), *L[:5], opt, timings=Tn)
        assert Tn["contexts_reused"] is False and len(ML._level_cache) == 0
        same(off, new)
    finally:
        sipx.clear_context_cache()
