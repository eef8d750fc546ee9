"""GPU tests added in round 3: per-kernel statistics, what the communicator reports, the slab-decomposed l1 search on ragged
and empty slabs with the sampled prediction forced, the exchange-segment overflow as an ERROR (never NaN iterates), a rho_ini
outside the clamp of src/PARSDMM.jl:226."""
import os

import numpy as np
import pytest

from oracle import parsdmm_oracle as O      # checker only
from tests.test_gpu_parity import _problem, model

pytestmark = pytest.mark.gpu


def test_per_kernel_statistics(sipx, monkeypatch):
    """sipx_kernel_stats_json in mode 2: every kernel of an iteration shows up with launches, time and algorithmic bytes;
    mode 1 records the CG product only; the byte counts are the ones DESIGN 3 states.  (With the separate per-set kernels:
    SIPX_YL_MULTI=0; the one-sweep update has its own test below.)"""
    monkeypatch.setenv("SIPX_YL_MULTI", "0")
    TF, n, h = np.float32, (64, 48, 40), (25.0, 25.0, 25.0)
    m = model(n, TF, seed=3)
    g, opt, P, A, prop, AtA = _problem(sipx, n, h, TF, ["bounds", "l1:D_x", "l1:D_z"], m, dict(maxit=30, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0))
    ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt)
    try:
        ctx.parsdmm_begin(opt)
        ctx.parsdmm_steps(4)
        ctx.kernel_stats(1)
        ctx.parsdmm_steps(4)
        st1 = ctx.kernel_stats_all(2)
        assert [k["name"] for k in st1["kernels"]] == ["k_cds<MODE=1>"] and st1["mode"] == 1
        cg4 = int(np.sum(ctx._run[2]["cg_it"][4:8]))
        assert 0 < st1["kernels"][0]["launches"] <= cg4          # (an x-step that finds x good enough logs cg_it = 1 without a product, cg.jl:73-76)
        ctx.parsdmm_steps(6)
        st2 = ctx.kernel_stats_all(0)
        names = {k["name"]: k for k in st2["kernels"]}
        for want in ("k_cds<MODE=1>", "k_cds<MODE=2>", "k_cg_update_xr", "k_rhs", "k_yl", "k_adj_norm", "k_fin_sum", "k_l1_solve"):
            assert want in names, (want, sorted(names))
        N, w, d = int(np.prod(n)), 4, 5
        cds = names["k_cds<MODE=1>"]
        assert cds["bytes_survey"] == cds["launches"] * (d + 2) * N * w                 # (d + 2) N w
        assert cds["bytes_moved"] == cds["launches"] * ((d + 1) // 2 + 2) * N * w       # bands with a non-negative offset only
        assert names["k_yl"]["launches"] == 6 * 4 and names["k_rhs"]["launches"] == 6   # four terms, six iterations
        assert names["k_rhs"]["bytes_moved"] == 6 * (2 * 4 + 1) * N * w                 # (sum 2 M_i + N) w on padded blocks
        assert all(k["total_ms"] > 0 for k in st2["kernels"] if k["name"] in ("k_yl", "k_rhs", "k_cds<MODE=1>"))
        assert ctx.kernel_stats_all(0)["kernels"] == []                                  # collection is off now
        info = ctx.comm_info()
        assert info == {"nranks": 1, "rank": 0, "version": "none", "decomposition": "sets"}
    finally:
        ctx.close()


def _rccl_info_worker(rank, world, port, out):
    import json
    import sys
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        from __graft_entry__ import load_package
        sipx = load_package()
        from sipx import sharded
        TF, n, h = np.float32, (32, 24, 16), (25.0, 25.0, 25.0)
        m = model(n, TF, seed=5)
        g, opt, P, A, prop, AtA = _problem(sipx, n, h, TF, ["bounds", "l1:D_z"], m, dict(maxit=5))
        keep = []

        def attach(cx):
            keep.append(sharded.attach_comm(cx, dist, torch.device("cuda", 0), "rccl"))
            cx.set_decomp("slab")
        ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt, device=0, owned=[1] * len(A), attach=attach)
        info = ctx.comm_info()
        ctx.close()
        with open(os.path.join(out, "info.json"), "w") as f:
            json.dump(info, f)
    finally:
        dist.destroy_process_group()


def test_comm_info_asks_rccl_itself(sipx, tmp_path):
    import json
    import torch.multiprocessing as mp
    mp.spawn(_rccl_info_worker, args=(1, 31900 + os.getpid() % 1000, str(tmp_path)), nprocs=1, join=True)
    info = json.load(open(tmp_path / "info.json"))
    assert info["nranks"] == 1 and info["rank"] == 0 and info["decomposition"] == "slab"
    assert info["version"].startswith("rccl ") and info["version"] != "rccl 0.0.0", info


def _slab_worker(rank, world, port, out, kinds, n, env):
    import sys
    import datetime
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **env)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        from __graft_entry__ import load_package
        sipx = load_package()
        from sipx import sharded
        TF = np.float32
        h = (25.0, 25.0, 25.0)[:len(n)]
        m = model(n, TF, seed=5)
        gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m, dict(maxit=40))
        try:
            x, log, l, y = sharded.PARSDMM_sharded(m.copy(), AtAs, As, props, Ps, gs, os_, dist=dist, device=0, comm_mode="torch", decomp="slab")
            ss = getattr(log, "slab_searches", None) or {}
            np.savez(os.path.join(out, f"r{rank}.npz"), x=x, obj=log.obj, cg_it=log.cg_it, rho=log.rho, r_pri=log.r_pri, err="",
                     searches=np.array([ss.get("speculative_exchange", -1), ss.get("fallbacks", -1)]))
        except sipx.SipxError as e:
            np.savez(os.path.join(out, f"r{rank}.npz"), err=str(e))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(400)
def test_slab_gather_overflow_is_an_error_on_every_rank(sipx, tmp_path):
    """An exchange segment too small for the bracket (forced: 4 values per rank, and one refinement round where the engine
    would enqueue up to six): every rank returns the SAME error from the y/l update whose search overflowed -- no NaN
    iterates, no rank left waiting in a collective.  (The initial feasibility estimate runs all its rounds and fits.)"""
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_slab_worker, args=(world, 30200 + os.getpid() % 1000, str(tmp_path), ["bounds", "l1:D_x", "l1:D_z"], (32, 24, 16),
                                 {"SIPX_GATHER_CAP": "4", "SIPX_L1_ROUNDS_MAX": "1"}), nprocs=world, join=True)
    errs = [str(np.load(tmp_path / f"r{r}.npz")["err"]) for r in range(world)]
    assert errs[0] and errs[0] == errs[1], errs
    assert "exchange segment" in errs[0] and ("l1 threshold search of set" in errs[0] or "initial feasibility of set" in errs[0])


@pytest.mark.timeout(400)
@pytest.mark.parametrize("world,n", [(4, (16, 12, 5)), (3, (16, 12, 5))])
def test_sampled_prediction_on_ragged_and_empty_slabs(sipx, tmp_path, world, n):
    """The sampled prediction forced on slabs of 2, 2, 1 planes and an EMPTY one (4 ranks; 2, 2, 1 for 3): whether the sample is
    taken, its stride and its capacity are functions of the whole grid only, a rank with nothing to sample contributes zeros --
    every rank ends with identical iterates, equal to the serial solve to the reference's tolerance."""
    import torch.multiprocessing as mp
    kinds = ["bounds", "l1:D_z", "l1:D_x"]
    mp.spawn(_slab_worker, args=(world, 30700 + os.getpid() % 1000 + world, str(tmp_path), kinds, n, {"SIPX_L1_SAMPLE_RUNS": "3"}),
             nprocs=world, join=True)
    r0 = np.load(tmp_path / "r0.npz")
    assert str(r0["err"]) == ""
    for r in range(1, world):
        r1 = np.load(tmp_path / f"r{r}.npz")
        for k in ("x", "obj", "cg_it", "rho", "r_pri"):
            assert np.array_equal(r0[k], r1[k], equal_nan=True), (r, k)
    assert np.isfinite(r0["x"]).all() and np.isfinite(r0["obj"]).all()
    TF, h = np.float32, (25.0, 25.0, 25.0)
    m = model(n, TF, seed=5)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m, dict(maxit=40))
    xs, ls, _, _ = sipx.PARSDMM(m.copy(), AtAs, As, props, Ps, gs, os_)
    assert np.linalg.norm(r0["x"] - xs) / np.linalg.norm(xs) < 5e-4


@pytest.mark.parametrize("rho_ini", [1e5, 1e-3])
def test_rho_ini_outside_the_clamp(sipx, rho_ini):
    """rho is clamped to [1e-2, 1e4] at the end of EVERY iteration (src/PARSDMM.jl:226): a rho_ini outside it changes after
    iteration 1 whatever the adaptation rules say -- the pipelined loop must not have queued a right-hand side with the old
    value.  Same logs as the oracle."""
    TF, n, h = np.float64, (24, 18), (25.0, 6.0)
    m = model(n, TF, seed=2)
    kinds = ["bounds", "l1:TV"]
    kw = dict(maxit=12, rho_ini=[rho_ini], adjust_rho=False, adjust_gamma=False, adjust_feasibility_rho=False)
    go, oo, Po, Ao, po, AtAo = _problem(O, n, h, TF, kinds, m, kw)
    xo, lo, _, _ = O.PARSDMM(m.copy(), AtAo, Ao, po, Po, go, oo)
    gs, os_, Ps, As, ps_, AtAs = _problem(sipx, n, h, TF, kinds, m, kw)
    xs, ls, _, _ = sipx.PARSDMM(m.copy(), AtAs, As, ps_, Ps, gs, os_)
    assert len(ls.obj) == len(lo.obj)
    assert np.array_equal(ls.rho, lo.rho) and ls.rho[0, 0] == rho_ini and ls.rho[1, 0] in (1e4, 1e-2)
    assert np.allclose(ls.obj, lo.obj, rtol=1e-9) and np.array_equal(ls.cg_it, lo.cg_it)
    assert np.linalg.norm(xs - xo) <= 1e-8 * np.linalg.norm(xo)


def test_rhs_compose_reference_case_on_the_engine(sipx):
    """test/test_rhs_compose.jl:1-38 through the HIP path: TD_OP = [2 speye(51000, 100000), speye(100000)] (the second one is the
    distance term's identity here), rho = [1.234, 10.23432], Float64 -- k_rhs + the CSC adjoint of the caller-supplied operator
    against the closed form the two scaled identities admit, to the reference's 10 eps."""
    import scipy.sparse as sp
    TF = np.float64
    rng = np.random.default_rng(20240611)
    n, h = (400, 250), (25.0, 6.0)
    N = 100000
    y = [rng.standard_normal(51000), rng.standard_normal(N)]
    l = [rng.standard_normal(51000), rng.standard_normal(N)]
    rho = [1.234, 10.23432]
    A1 = sp.eye(51000, N, format="csc", dtype=TF) * 2.0
    g = sipx.compgrid(h, n)
    sd = sipx.set_definitions("bounds", "identity", -1e12, 1e12, ("matrix", ""))
    sd.custom_TD_OP = (sp.csc_matrix(A1), False)
    opt = sipx.PARSDMM_options(FL=TF, maxit=5)
    P, A, prop = sipx.setup_constraints([sd], g, TF)
    A, AtA, _, _ = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
    opt.zero_ini_guess = False
    ctx = sipx.host.build_context(rng.standard_normal(N), AtA, A, prop, P, g, opt, np.zeros(N), l, y)
    try:
        ctx.rhs_compose(rho)
        rhs = ctx.get_rhs()
    finally:
        ctx.close()
    want = rho[1] * y[1] + l[1]
    want[:51000] += 2.0 * (rho[0] * y[0] + l[0])
    assert np.linalg.norm(rhs - want) <= 10 * np.finfo(TF).eps * max(np.linalg.norm(rhs), np.linalg.norm(want))


MULTI_CASES = [
    # (grid, spacings, sets, precision, options, does the one-sweep kernel take this list?)
    ((64, 48, 40), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], np.float32, {}, True),            # the headline list: I X Y Z D
    ((64, 48, 40), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], np.float64, {}, True),
    ((36, 20, 9), (25.0, 20.0, 10.0), ["bounds", "l1:TV", "annulus"], np.float32, {}, True),                         # ragged tiles; TV: three blocks of one set
    ((36, 20, 9), (25.0, 20.0, 10.0), ["bounds", "l1:TV"], np.float64, {}, True),                                    # C5's list
    ((520, 12, 6), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], np.float32, {}, True),              # two tiles along x: the recomputed point
    ((36, 28, 30), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_z"], np.float64, {}, True),                        # several chunks of planes: the recomputed plane
    ((96, 80), (25.0, 6.0), ["bounds", "l1:TV"], np.float32, {}, True),                                              # 2-D (C2's list)
    ((96, 80), (25.0, 6.0), ["bounds", "l1:D_z", "l1:D_x"], np.float64, {}, True),
    ((40, 24, 16), (25.0, 25.0, 25.0), ["bounds", "l1:D_z"], np.float32, {"adjust_rho": False, "adjust_gamma": False}, True),   # every iteration qualifies
    ((36, 28, 30), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_z"], np.float32, {"rho_update_frequency": 3}, True),   # two plain iterations in a row: the third pair
    ((36, 20, 9), (25.0, 20.0, 10.0), ["l2", "l1:D_z", "bnd:D_y"], np.float64, {}, False),                           # a layout that is not instantiated: per-set kernels
    ((96, 80), (25.0, 6.0), ["bounds", "l1:D_z", "card:D_x"], np.float32, {}, True),                                 # round 4, partial sweep: I Z D in the sweep, cardinality per set
    ((96, 80), (25.0, 6.0), ["bounds", "l1:D_z", "card:D_x"], np.float64, {}, True),                                 # ... and in Float64 (the case round 4 had replaced)
    # BASELINE config 4's list: the sweep takes I X Y Z I(annulus) D; l1 behind the DFT, slice rank and cardinality keep their kernels,
    # the fused right-hand side holds the five sets in front of the first of them, k_rhs adds the rest in order
    ((32, 24, 16), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z", "annulus", "l1dft", "rank:4", "card:D_z"], np.float32, {}, True),
    ((32, 24, 16), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z", "annulus", "l1dft"], np.float64, {}, True),
]


@pytest.mark.parametrize("n,h,kinds,TF,okw,taken", MULTI_CASES)
def test_one_sweep_update_is_bit_identical(sipx, monkeypatch, n, h, kinds, TF, okw, taken):
    """k_yl_multi (one sweep: every set's y/l update + r_pri / r_dual / obj sums, the Barzilai-Borwein sums and snapshot refresh
    when due, the feasibility estimate of the element-wise sets every tenth iteration, the next right-hand side when rho
    cannot change) against the separate kernels (SIPX_YL_MULTI=0: k_yl per set, k_adj_norm, k_rhs): the same arithmetic per
    element, so x, every y_i and l_i and the rho / gamma histories are IDENTICAL bit for bit; the float64 sums are taken in
    another order (1e-12)."""
    m = model(n, TF, seed=7)
    out = {}
    for tag in ("0", "1"):
        monkeypatch.setenv("SIPX_YL_MULTI", tag)
        g, opt, P, A, prop, AtA = _problem(sipx, n, h, TF, kinds, m, dict(maxit=27, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0, **okw))
        ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt)
        try:
            ctx.parsdmm_begin(opt)
            ctx.parsdmm_steps(12)
            ctx.kernel_stats(2)
            ctx.parsdmm_steps(15)
            names = {k["name"] for k in ctx.kernel_stats_all(0)["kernels"]}
            log = ctx.parsdmm_log()
            x, l, y = ctx.download()
        finally:
            ctx.close()
        out[tag] = (x, l, y, log, names)
    (x0, l0, y0, g0, k0), (x1, l1, y1, g1, k1) = out["0"], out["1"]
    assert ("k_yl_multi" in k1) == taken and "k_yl_multi" not in k0
    assert len(g0.obj) == len(g1.obj) == 27 and np.array_equal(g0.cg_it, g1.cg_it)
    if TF == np.float32 or not taken:
        assert np.array_equal(x0, x1)
        for a, b in zip(y0 + l0, y1 + l1):
            assert np.array_equal(a, b)
        assert np.array_equal(g0.rho, g1.rho) and np.array_equal(g0.gamma, g1.gamma)
    else:
        # Float64: the six Barzilai-Borwein sums reach the rule unrounded, so their summation order (another partition of the
        # grid) moves rho in its last bits, and every iterate after it at that level
        assert np.linalg.norm(x0 - x1) <= 1e-11 * np.linalg.norm(x0)
        for a, b in zip(y0, y1):
            assert np.linalg.norm(a - b) <= 1e-9 * max(np.linalg.norm(a), 1e-300)
        for a, b, yy in zip(l0, l1, y0):          # (the multiplier of an inactive set is a rounding residue of size rho eps |y|)
            assert np.linalg.norm(a - b) <= 1e-9 * np.linalg.norm(a) + 10 * float(g0.rho.max()) * np.finfo(TF).eps * np.linalg.norm(yy)
        assert np.allclose(g0.rho, g1.rho, rtol=1e-11) and np.allclose(g0.gamma, g1.gamma, rtol=1e-11)
    for f in ("obj", "evol_x", "r_pri", "r_dual", "set_feasibility"):
        assert np.allclose(getattr(g0, f), getattr(g1, f), rtol=1e-6 if TF == np.float32 else 1e-9, atol=0, equal_nan=True), f


@pytest.mark.parametrize("TF", [np.float32, np.float64])
@pytest.mark.parametrize("n,kinds", [((40, 24, 20), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"]),      # band order 0, -1, +1, -n1, +n1, -n1n2, +n1n2
                                     ((36, 20, 9), ["bounds", "l1:TV"]),                              # band order 0, -n1n2, -n1, -1, +1, +n1, +n1n2
                                     ((264, 10, 7), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"])])      # two tiles along x (Float32), ragged rows
def test_z_marching_product_is_bit_identical(sipx, monkeypatch, TF, n, kinds):
    """k_cds_march (x and the +n1n2 band of the previous plane in registers, rows through LDS: every band value crosses the
    fabric once) against k_cds and against the oracle's CDS_MVp (src/CDS_MVp_MT_subfunc.jl:6-20: same summation order): y = Q x
    bit for bit, before and after an incremental Q update; forced onto these small grids in chunks of 4 planes so that chunk
    and tile edges are exercised.  A whole solve with it ends where the solve with k_cds ends."""
    h = (25.0, 20.0, 10.0)
    m = model(n, TF, seed=4)
    go, oo, Po, Ao, propo, AtAo = _problem(O, n, h, TF, kinds, m)
    p = len(Ao)
    rho = list(np.linspace(0.5, 11.0, p))
    x = np.random.default_rng(9).standard_normal(m.size).astype(TF)
    Qo, offo = O.assemble_Q(AtAo, propo.AtA_offsets, np.array(rho, TF), TF)
    want = O.Ax_CDS(x, Qo, offo)
    out = {}
    for sw in ("0", "2"):
        monkeypatch.setenv("SIPX_CDS_MARCH", sw)
        monkeypatch.setenv("SIPX_CDS_MARCH_ZCHUNK", "4")
        gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m, dict(maxit=25))
        os_.rho_ini = rho
        ctx = sipx.host.build_context(m, AtAs, As, props, Ps, gs, os_)
        y0 = ctx.apply_Q(x)
        rho2 = [r * (1.5 if i % 2 else 0.75) for i, r in enumerate(rho)]
        ctx.q_update(rho2, rho)
        y1 = ctx.apply_Q(x)
        ctx.kernel_stats(2)
        log, _ = ctx.parsdmm(os_)
        names = {k["name"] for k in ctx.kernel_stats_all(0)["kernels"]}
        xs, _, _ = ctx.download()
        ctx.close()
        out[sw] = (y0, y1, xs, log)
    assert np.array_equal(out["0"][0], want) and np.array_equal(out["2"][0], want)
    assert np.array_equal(out["0"][1], out["2"][1])
    (a0, a1, xa, la), (b0, b1, xb, lb) = out["0"], out["2"]
    assert len(la.obj) == len(lb.obj) and np.array_equal(la.cg_it, lb.cg_it)
    assert np.linalg.norm(xa.astype(np.float64) - xb) <= (2e-6 if TF == np.float32 else 1e-11) * np.linalg.norm(xa)


@pytest.mark.timeout(400)
def test_slab_search_refines_until_the_bracket_fits(sipx, tmp_path):
    """A tiny exchange segment (256 values per rank: 1 / 72 of the vector) with
    every refinement round enqueued (SIPX_L1_ROUNDS_MIN=6; by default the engine enqueues two more than the previous search
    of the set used): the searches narrow their brackets round by round (one all-reduce each) until they fit, then the usual
    all-gather; the solve ends where the serial solve ends, identically on both ranks."""
    import torch.multiprocessing as mp
    world, kinds, n = 2, ["bounds", "l1:D_x", "l1:D_z"], (32, 24, 16)
    mp.spawn(_slab_worker, args=(world, 30400 + os.getpid() % 1000, str(tmp_path), kinds, n, {"SIPX_GATHER_CAP": "256", "SIPX_L1_ROUNDS_MIN": "6"}),
             nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert str(r0["err"]) == "" and str(r1["err"]) == ""
    for k in ("x", "obj", "cg_it", "rho", "r_pri"):
        assert np.array_equal(r0[k], r1[k], equal_nan=True), k
    TF, h = np.float32, (25.0, 25.0, 25.0)
    m = model(n, TF, seed=5)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m, dict(maxit=40))
    xs, ls, _, _ = sipx.PARSDMM(m.copy(), AtAs, As, props, Ps, gs, os_)
    assert np.linalg.norm(r0["x"] - xs) / np.linalg.norm(xs) < 5e-4


@pytest.mark.timeout(600)
def test_slab_searches_settle_in_one_exchange(sipx, tmp_path, monkeypatch):
    """Slab-decomposed threshold searches through the speculative exchange (one all-gather carrying every rank's probe sums and the
    magnitudes it gathered inside the speculative range): most searches of a solve are settled by it -- the fallback
    (refinement rounds + full-size exchange) runs for the others -- and the iterates are the ones the staged protocol
    (SIPX_SPEC_EXCHANGE=0: all-reduce, ..., all-gather for every search) arrives at, bit for bit, on every rank."""
    import torch.multiprocessing as mp
    world, kinds, n = 2, ["bounds", "l1:D_x", "l1:D_z"], (48, 32, 24)
    res = {}
    for tag, env in (("spec", {}), ("staged", {"SIPX_SPEC_EXCHANGE": "0"}), ("tiny", {"SIPX_GATHER_FAST_CAP": "8"})):
        d = tmp_path / tag
        d.mkdir()
        mp.spawn(_slab_worker, args=(world, 30900 + os.getpid() % 1000 + len(res), str(d), kinds, n, env), nprocs=world, join=True)
        r0, r1 = np.load(d / "r0.npz"), np.load(d / "r1.npz")
        assert str(r0["err"]) == "" and str(r1["err"]) == ""
        for k in ("x", "obj", "cg_it", "rho", "r_pri", "searches"):
            assert np.array_equal(r0[k], r1[k], equal_nan=True), (tag, k)
        res[tag] = r0
    n_s, n_f = res["spec"]["searches"]
    assert n_s > 0 and n_f < n_s, (n_s, n_f)                        # searches were settled by the exchange (while rho, gamma are
                                                                     # still adapted theta leaves the range on many iterations)
    assert res["staged"]["searches"][0] == 0
    assert res["tiny"]["searches"][1] > n_f                          # fast segments of 8 values: (almost) every l1 search falls back ...
    for k in ("x", "obj", "cg_it", "rho", "r_pri"):                  # ... and all three protocols end with the same bits
        assert np.array_equal(res["spec"][k], res["staged"][k], equal_nan=True), k
        assert np.array_equal(res["spec"][k], res["tiny"][k], equal_nan=True), k


@pytest.mark.parametrize("TF,n", [(np.float32, (64, 40, 24)), (np.float64, (64, 40, 24)), (np.float32, (40, 24, 16))])
def test_fused_z_marching_cg_iteration_is_bit_identical(sipx, monkeypatch, TF, n):
    """The fused form of the z-marching product (k_cds_march<MODE 3>: the scalar step of CG iteration k and the product of iteration
    k + 1 on p = r + beta p_old formed wherever it is loaded; the default wherever the march applies) against product + p-update
    (SIPX_CG_FUSED=0), both with the march forced on a small grid in chunks of 5 planes (tile edges, chunk edges and the last,
    shorter chunk inside the grid): same arithmetic, same summation order of the dot products -- the same bits in x, the CG
    iteration counts and residuals and every log."""
    from tests.test_gpu_parity import _c3_problem
    m, g, opt, P, A, prop, AtA = _c3_problem(sipx, n, TF, maxit=30)
    monkeypatch.setenv("SIPX_CDS_MARCH", "2")
    monkeypatch.setenv("SIPX_CDS_MARCH_ZCHUNK", "5")
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


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,n", [(4, (128, 128, 96)), (5, (96, 80, 47))])       # (the parent holds the GPU too: at most 5 ranks)
def test_slab_decomposed_mid_size_ranks(sipx, tmp_path, world, n):
    """Four / five ranks on one GPU at a size where the exchange segments, the speculative ranges and the sampled prediction (forced: 4096
    runs) work on realistic populations (128 x 128 x 96, the headline's set list): every rank ends with identical iterates and
    logs, equal to the serial solve to the reference's serial-vs-parallel tolerance; most searches are settled by the exchange."""
    import torch.multiprocessing as mp
    kinds = ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"]            # (5 ranks on 47 planes: four slabs of 10 and one of 7)
    mp.spawn(_slab_worker, args=(world, 31300 + os.getpid() % 1000 + world, str(tmp_path), kinds, n, {"SIPX_L1_SAMPLE_RUNS": "4096"}),
             nprocs=world, join=True)
    r0 = np.load(tmp_path / "r0.npz")
    assert str(r0["err"]) == ""
    for r in range(1, world):
        r1 = np.load(tmp_path / f"r{r}.npz")
        for k in ("x", "obj", "cg_it", "rho", "r_pri", "searches"):
            assert np.array_equal(r0[k], r1[k], equal_nan=True), (r, k)
    n_s, n_f = r0["searches"]
    assert n_s > 0 and n_f < n_s, (n_s, n_f)
    TF, h = np.float32, (25.0, 25.0, 25.0)
    m = model(n, TF, seed=5)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m, dict(maxit=40))
    xs, ls, _, _ = sipx.PARSDMM(m.copy(), AtAs, As, props, Ps, gs, os_)
    assert len(ls.obj) == len(r0["obj"])
    assert np.linalg.norm(r0["x"] - xs) / np.linalg.norm(xs) < 5e-4


@pytest.mark.parametrize("TF,n", [(np.float32, (64, 40, 24)), (np.float64, (40, 24, 16))])
def test_q_update_inside_the_residual_product_is_bit_identical(sipx, monkeypatch, TF, n):
    """A change of rho decided at the end of an iteration is applied by the residual product that opens the next x-step
    (k_cds_march<MODE 4>: the four stored bands of the old matrix read once, the changed sets' rho differences added in
    k_q_update's order, the result used in the product and written into the second copy of Q) instead of by k_q_update
    (SIPX_Q_FUSED=0): the same Q bit for bit -- read back after 11 steps, i.e. with an update still pending, and at the end of the
    solve -- and the same x, CG counts and logs.  The march is forced on this small grid in chunks of 5 planes."""
    from tests.test_gpu_parity import _c3_problem
    m, g, opt, P, A, prop, AtA = _c3_problem(sipx, n, TF, maxit=30)
    monkeypatch.setenv("SIPX_CDS_MARCH", "2")
    monkeypatch.setenv("SIPX_CDS_MARCH_ZCHUNK", "5")
    out = {}
    for tag in ("0", "1"):
        monkeypatch.setenv("SIPX_Q_FUSED", tag)
        ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt)
        ctx.parsdmm_begin(opt)
        ctx.parsdmm_steps(11)
        Qmid, off = ctx.get_Q()
        ctx.parsdmm_steps(30)
        Qend, _ = ctx.get_Q()
        log = ctx.parsdmm_log()
        x, l, y = ctx.download()
        ctx.close()
        out[tag] = (Qmid, Qend, x, log)
    a, b = out["0"], out["1"]
    assert len({tuple(r) for r in a[3].rho}) > 2                      # rho did change along the way
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3].cg_it, b[3].cg_it) and np.array_equal(a[3].obj, b[3].obj)
    assert np.array_equal(a[3].rho, b[3].rho) and np.array_equal(a[3].r_pri, b[3].r_pri)


@pytest.mark.parametrize("TF,n,kinds", [(np.float32, (64, 48, 40), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"]),
                                         (np.float64, (48, 40, 24), ["bounds", "l1:D_x", "l1:D_z"]),
                                         (np.float32, (256, 192), ["bounds", "l1:D_x", "l1:D_z"])])
def test_lean_first_passes_in_one_sweep_are_bit_identical(sipx, monkeypatch, TF, n, kinds):
    """k_lean_multi (the lean first passes of two or three l1 searches in one sweep, x read once) against one k_pass<M_LEAN> per
    set (SIPX_LEAN_MULTI=0): the same partial sums, the same gathered values -- theta, x, the CG counts and every log bit for bit
    over 60 iterations (the group pass runs on the iterations where no search is rescaled or sampled first; the kernel
    statistics confirm that it did run)."""
    h = (25.0, 25.0, 25.0)[:len(n)]
    m = model(n, TF, seed=7)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m, dict(maxit=60))
    os_.evol_rel_tol = os_.feas_tol = os_.obj_tol = 0.0
    # the group pass needs every set's own scratch, which exists with the set streams: those of the large grids it is made for
    # (a grid this small runs on the engine stream alone by default)
    monkeypatch.setenv("SIPX_SERIAL_SETS", "0")
    monkeypatch.setenv("SIPX_SEARCH_BATCH", "0")        # (round 4's batched chain always takes the group pass: the per-set chains here)
    out = {}
    for tag in ("0", "1"):
        monkeypatch.setenv("SIPX_LEAN_MULTI", tag)
        ctx = sipx.host.build_context(m, AtAs, As, props, Ps, gs, os_)
        ctx.kernel_stats(2)
        ctx.parsdmm_begin(os_)
        ctx.parsdmm_steps(60)
        stats = {k["name"]: k for k in ctx.kernel_stats_all(0)["kernels"]}
        log = ctx.parsdmm_log()
        x, l, y = ctx.download()
        ctx.close()
        out[tag] = (x, log, stats)
    (x0, l0, s0), (x1, l1, s1) = out["0"], out["1"]
    lean = [k for k in s0 if "LEAN" in k or "lean" in k]
    assert lean and s1[lean[0]]["launches"] < s0[lean[0]]["launches"]          # fewer lean launches: the group pass took them
    assert np.array_equal(x0, x1) and np.array_equal(l0.cg_it, l1.cg_it) and np.array_equal(l0.obj, l1.obj)
    assert np.array_equal(l0.rho, l1.rho) and np.array_equal(l0.r_pri, l1.r_pri) and np.array_equal(l0.r_dual, l1.r_dual)


@pytest.mark.parametrize("TF,n,kinds", [(np.float32, (64, 40, 24), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"]),
                                         (np.float64, (40, 24, 16), ["bounds", "l1:TV"]),
                                         (np.float32, (36, 20, 13), ["bounds", "l1:D_z", "l1:D_y"])])
def test_z_marching_rhs_is_bit_identical(sipx, monkeypatch, TF, n, kinds):
    """k_rhs_march (rhs = sum_i A_i'(rho_i y_i + l_i) marched along z: w of the previous plane in registers, of the row above in
    LDS, of the point to the left from the lane next door) against k_rhs (SIPX_RHS_MARCH=0), forced on a small grid in chunks of 5
    planes (tile edges, chunk edges, a last shorter chunk, a grid line shorter than a wave): the same products in the same order --
    the right-hand side bit for bit, for multipliers and auxiliary vectors taken from a solve that has run 9 iterations."""
    h = (25.0, 25.0, 25.0)
    m = model(n, TF, seed=11)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m, dict(maxit=12))
    os_.evol_rel_tol = os_.feas_tol = os_.obj_tol = 0.0
    ctx = sipx.host.build_context(m, AtAs, As, props, Ps, gs, os_)
    ctx.parsdmm_begin(os_)
    ctx.parsdmm_steps(9)
    rho = [0.7 + 0.9 * i for i in range(len(Ps) + 1)]
    out = {}
    for tag in ("0", "2"):
        monkeypatch.setenv("SIPX_RHS_MARCH", tag)
        monkeypatch.setenv("SIPX_RHS_MARCH_ZCHUNK", "5")
        ctx.rhs_compose(rho)
        out[tag] = ctx.get_rhs()
    ctx.close()
    assert np.isfinite(out["0"]).all() and np.abs(out["0"]).max() > 0
    assert np.array_equal(out["0"], out["2"])


@pytest.mark.timeout(600)
def test_slab_decomposed_with_every_z_marching_kernel_forced(sipx, tmp_path):
    """Three ranks (slabs of 8, 8 and 6 planes) with the z-marching product, the z-marching right-hand side and short chunks of the
    sweep forced on the small grid -- the forms a rank's share of 512^3 runs on eight GPUs: row ranges that are slabs, the planes of
    the neighbours read through the copies a rank keeps.  Every rank ends with identical iterates and logs, equal to the serial
    solve to the reference's serial-vs-parallel tolerance."""
    import torch.multiprocessing as mp
    world, kinds, n = 3, ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"], (64, 40, 22)
    env = {"SIPX_CDS_MARCH": "2", "SIPX_CDS_MARCH_ZCHUNK": "5", "SIPX_RHS_MARCH": "2", "SIPX_RHS_MARCH_ZCHUNK": "5", "SIPX_MULTI_ZCHUNK": "5"}
    mp.spawn(_slab_worker, args=(world, 31700 + os.getpid() % 1000, str(tmp_path), kinds, n, env), nprocs=world, join=True)
    r0 = np.load(tmp_path / "r0.npz")
    assert str(r0["err"]) == ""
    for r in range(1, world):
        r1 = np.load(tmp_path / f"r{r}.npz")
        for k in ("x", "obj", "cg_it", "rho", "r_pri"):
            assert np.array_equal(r0[k], r1[k], equal_nan=True), (r, k)
    TF, h = np.float32, (25.0, 25.0, 25.0)
    m = model(n, TF, seed=5)
    gs, os_, Ps, As, props, AtAs = _problem(sipx, n, h, TF, kinds, m, dict(maxit=40))
    xs, ls, _, _ = sipx.PARSDMM(m.copy(), AtAs, As, props, Ps, gs, os_)
    assert len(ls.obj) == len(r0["obj"])
    assert np.linalg.norm(r0["x"] - xs) / np.linalg.norm(xs) < 5e-4


@pytest.mark.timeout(900)
def test_full_size_512_large_grid_forms_are_bit_identical(sipx, monkeypatch):
    """At BASELINE's 512^3 (where they are the default): the z-marching right-hand side and the shared lean first pass against the
    flat right-hand side and one lean pass per set -- x, the CG counts and every log of 14 iterations bit for bit."""
    from tests.test_gpu_parity import _c3_problem
    TF, n = np.float32, (512, 512, 512)
    m, g, opt, P, A, prop, AtA = _c3_problem(sipx, n, TF, maxit=14)
    opt.evol_rel_tol = opt.feas_tol = opt.obj_tol = 0.0
    out = {}
    for tag, env in (("large", {}), ("flat", {"SIPX_RHS_MARCH": "0", "SIPX_LEAN_MULTI": "0"})):
        for k in ("SIPX_RHS_MARCH", "SIPX_LEAN_MULTI"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        x, log, l, y = sipx.PARSDMM(m.copy(), AtA, A, prop, P, g, opt)
        out[tag] = (x, log)
        del l, y
    (x0, l0), (x1, l1) = out["large"], out["flat"]
    assert len(l0.obj) == 14 and np.isfinite(x0).all()
    assert np.array_equal(l0.cg_it, l1.cg_it) and np.array_equal(l0.rho, l1.rho) and np.array_equal(l0.obj, l1.obj)
    assert np.array_equal(l0.r_pri, l1.r_pri) and np.array_equal(x0, x1)


@pytest.mark.parametrize("n,h,kinds", [((96, 80), (25.0, 6.0), ["bounds", "l1:TV"]),
                                       ((40, 24, 20), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"])])
def test_set_streams_or_one_stream_is_bit_identical(sipx, monkeypatch, n, h, kinds):
    """Up to 2^22 grid points every set runs on the engine stream (the default there: the event waits between streams cost more
    than the overlap of such short kernels gains); SIPX_SERIAL_SETS=0 keeps the set / search streams of the larger grids.
    Which stream a kernel runs on changes no arithmetic: x, y, l and the rho / gamma histories are identical bit for bit."""
    TF = np.float32
    m = model(n, TF, seed=11)
    out = {}
    for tag in ("default", "0", "1"):
        if tag == "default":
            monkeypatch.delenv("SIPX_SERIAL_SETS", raising=False)
        else:
            monkeypatch.setenv("SIPX_SERIAL_SETS", tag)
        g, opt, P, A, prop, AtA = _problem(sipx, n, h, TF, kinds, m, dict(maxit=24, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0))
        ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt)
        try:
            ctx.parsdmm_begin(opt)
            ctx.parsdmm_steps(24)
            log = ctx.parsdmm_log()
            x, l, y = ctx.download()
        finally:
            ctx.close()
        out[tag] = (x, l, y, log)
    x0, l0, y0, g0 = out["default"]
    for tag in ("0", "1"):
        x1, l1, y1, g1 = out[tag]
        assert np.array_equal(x0, x1), tag
        for a, b in zip(y0 + l0, y1 + l1):
            assert np.array_equal(a, b), tag
        assert np.array_equal(g0.rho, g1.rho) and np.array_equal(g0.gamma, g1.gamma) and np.array_equal(g0.cg_it, g1.cg_it)


@pytest.mark.timeout(400)
def test_bench_two_ranks_share_the_gpu():
    """The N > 1 flow of bench.py with the REAL engine, rehearsed on a one-GPU box (`SIPX_BENCH_SHARE_GPU=1 python bench.py --gpus 2`:
    the script spawns its two ranks, both run on device 0, the engine's collectives go through gloo callbacks): every world > 1
    branch of the script runs -- rendezvous, both decompositions under fixed keys, max-over-ranks timing, what the communicator
    reports, the ranks comparing their x, a 512^3 leg, the comm probe -- and rank 0 prints ONE line, marked as not a measurement."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SIPX_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(prefix="sipx_bench_"), "detail.json")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-c4", "--no-c5",
                        "--detail", detail], capture_output=True, text=True, timeout=380, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    assert len(lines[0]) < 4096, len(lines[0])                  # the driver keeps a few KB of stdout: round 3's 24 KB line was cut
    h = json.loads(lines[0])
    assert h["n_gpus"] == 2 and h["steps"] == 4 and h["value"] > 0 and h["invalid_as_measurement"] is True
    assert set(h["decompositions"]) == {"slab", "sets"} and h["decomposition"] == "slab" and h["value"] == h["decompositions"]["slab"]["value"]
    assert all(v["ranks_agree_on_x"] is True for v in h["decompositions"].values()) and h["comm"]["rccl_nranks"] == 2
    assert h["c3_512"]["value"] > 0 and h["c3_512"]["comm"]["decomposition"] == "slab" and h["c3_512"]["comm"]["ranks_agree_on_x"] is True
    assert 0 < h["c3_512"]["comm"]["device_bytes_per_rank"] < 20e9          # sparse arrays: half of the 27 GB a single GPU holds, plus halo planes
    assert isinstance(h["comm_probe_us"], dict) and len(h["comm_probe_us"]) >= 8
    assert h["comm"]["device_bytes_per_rank"] > 0
    d = json.load(open(detail))                                 # everything else: the side file
    assert "rehearsal" in d and abs(d["value"] - h["value"]) <= 1e-4 * h["value"]
    for v in list(d["decompositions"].values()) + [d["c3_512"]]:
        assert "error" not in v, v
        assert v["value"] > 0 and v["comm"]["rccl_nranks"] == 2 and v["comm"]["ranks_agree_on_x"] is True
    assert d["c3_512"]["comm"]["decomposition"] == "slab" and d["config"]["all_logs_finite"]
    assert isinstance(d["comm_probe_us"], dict) and "error" not in d["comm_probe_us"]
