"""The N>1 path on CPU: the sharded solve (setintersectionprojection.jl_amd/sharded.py) over 2 / 4 / 6 gloo ranks.
The HIP engine cannot run here, so the phase driver is given a CPU stand-in engine that answers the same phase calls with
the ORACLE's functions and performs the engine's exchange steps through the product's own communicator class
(sharded.TorchComm: reduce-scatter of rhs by slab, slab CG with halo exchange and all-reduced dot partials, all-gather of
x, one all-reduce of the packed per-set scalars) -- test infrastructure injected by the test; the product never does this.
Checks:
  * world=1: the driver's control flow reproduces the oracle's PARSDMM loop bit for bit;
  * world=2/4/6: sharded == serial up to the summation order of rhs and of the dot products (reference tolerance
    5e-4 Float32, test/test_PARSDMM_parallel.jl:72; much tighter here); every rank ends with identical results;
  * the slab arithmetic (ragged and empty slabs) and the four collectives themselves."""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import parsdmm_oracle as O  # noqa: E402

YL_FEAS, YL_BB, YL_FIRST = 1, 2, 4


class OracleEngine:
    """CPU stand-in for host.Context: same phase methods, oracle arithmetic.  With a communicator it is one rank of the
    sharded solve: owned-set mask, z-slab of the x-step, and the collectives at the points where the engine has them."""

    def __init__(self, m, AtA, TD_OP, prop, P_sub, opt, owned, comm=None, n=None):
        TF = self.TF = m.dtype.type
        self.m, self.AtA, self.A, self.prop, self.P = m, AtA, TD_OP, prop, P_sub
        self.p, self.pp, self.N = len(TD_OP), len(P_sub), len(m)
        self.comm = comm
        self.owned = [i for i in range(self.p) if owned[i]]
        z = lambda: [np.zeros(self.A[i].shape[0], TF) for i in range(self.p)]
        (self.y, self.l, self.y_old, self.l_old, self.x_hat, self.r_pri, self.s, self.y_0, self.l_0, self.s_0,
         self.l_hat_0, self.l_hat) = (z() for _ in range(12))
        self.x, self.x_old, self.rhs = np.zeros(self.N, TF), np.zeros(self.N, TF), np.zeros(self.N, TF)
        rho = np.full(self.p, TF(opt.rho_ini[0]), TF)
        self.Q, self.Qo = O.assemble_Q(AtA, prop.AtA_offsets, rho, TF)
        self.rho_prox = rho.copy()
        self.prox = list(P_sub) + [lambda v: O.prox_l2s(v, self.rho_prox[self.p - 1], self.m)]
        eps = np.finfo(TF).eps
        f0 = np.zeros(self.pp)
        for i in range(self.pp):
            if owned[i]:
                Am = O.csc_mul(self.A[i], m)
                f0[i] = TF(O.nrm2(P_sub[i](Am.copy()) - Am, TF) / TF(O.nrm2(Am, TF) + TF(100) * eps))
        if comm is not None:
            from sipx import sharded
            self.plane = self.N // n[-1]
            self.chunk, slabs = sharded.slab_partition(n[-1], self.plane, comm.world)
            self.r0, self.r1 = slabs[comm.rank]
            self.prev = comm.rank - 1 if 0 < self.r0 < self.N else -1
            self.next = comm.rank + 1 if self.r0 < self.r1 < self.N else -1
            f0 = self._allreduce(f0)
        self.feasibility_initial = f0

    def _allreduce(self, v):
        t = torch.from_numpy(np.ascontiguousarray(v, np.float64).copy())
        self.comm.allreduce_sum_(t)
        return t.numpy()

    def _padded(self, v):
        buf = np.zeros(self.chunk * self.comm.world, self.TF)
        buf[:self.N] = v
        return buf

    def rhs_compose(self, rho):
        self.rhs[:] = O.rhs_compose(self.l, self.y, np.asarray(rho, self.TF), self.A, self.p, self.N, only=self.owned)
        if self.comm is not None:                     # reduce-scatter by slab: only the own rows are meaningful afterwards
            buf = torch.from_numpy(self._padded(self.rhs))
            self.comm.reduce_scatter_sum_(buf, self.chunk)
            own = buf.numpy()[self.r0:self.r1].copy()
            self.rhs[:] = np.nan
            self.rhs[self.r0:self.r1] = own

    # ---- the x-step on the slab: src/cg.jl:44-128 with the dot products summed over the ranks ------------------
    def _slab_Ax(self, v_slab):
        """rows [r0, r1) of Q v, v known on the slab: one halo plane from each neighbour first."""
        r0, r1, P = self.r0, self.r1, self.plane
        full = np.zeros(self.N + 2 * P, self.TF)                  # the engine's p buffer with its halo
        full[P + r0:P + r1] = v_slab
        t = torch.from_numpy(full)
        self.comm.halo_exchange(t[P + r0:P + r0 + P], t[r0:P + r0], self.prev, t[r1:P + r1], t[P + r1:P + r1 + P], self.next)
        return O.Ax_CDS(full[P:P + self.N], self.Q, self.Qo)[r0:r1]

    def _gsum(self, *parts):
        return self._allreduce(np.array(parts, np.float64))

    def _slab_argmin_x(self, it, tol_ref):
        TF, r0, r1 = self.TF, self.r0, self.r1
        b = self.rhs[r0:r1]
        x = self.x[r0:r1].copy()
        eps = np.finfo(TF).eps
        # full x is known everywhere (all-gather of the previous iteration): its halo needs no exchange
        r = b - O.Ax_CDS(self.x, self.Q, self.Qo)[r0:r1]
        ss_r, ss_b = self._gsum(O._sumsq64(r), O._sumsq64(b))
        nr0 = TF(math.sqrt(ss_b))
        with np.errstate(all="ignore"):
            relres0 = float(np.float64(0.1) * np.float64(TF(math.sqrt(ss_r))) / np.float64(nr0))
        cand = O._julia_max(relres0, float(TF(10) * eps))
        tol = TF(cand) if it < 3 else TF(O._julia_min(cand, float(tol_ref)))
        if nr0 == 0:
            return np.zeros(r1 - r0, TF), -9, TF(0), 0, tol
        p = r.copy()
        ss = ss_r
        rr = TF(ss)
        if TF(TF(math.sqrt(ss)) / nr0) <= tol:
            return x, 0, TF(0), 1, tol
        flag, last, res_last = -1, 0, TF(0)
        for k in range(1, 1001):
            last = k
            Ap = self._slab_Ax(p)
            gamma = rr
            (pAp,) = self._gsum(float(np.dot(p.astype(np.float64), Ap.astype(np.float64))))
            with np.errstate(all="ignore"):
                alpha = TF(gamma / TF(pAp))
            if np.isposinf(alpha) or alpha < 0:
                flag, res_last = -2, TF(0)
                break
            x += alpha * p
            r -= alpha * Ap
            (ss,) = self._gsum(O._sumsq64(r))
            rr = TF(ss)
            res_last = TF(TF(math.sqrt(ss)) / nr0)
            if res_last <= tol:
                flag = 0
                break
            beta = TF(rr / gamma)
            p = r + beta * p
        return x, flag, res_last, last, tol

    def argmin_x(self, it, tol_ref):
        if self.comm is None:
            self.x_old[:] = self.x
            x, n_it, relres, tol = O.argmin_x(self.Q, self.rhs, self.x, self.TF(tol_ref), it, self.Qo)
            self.x = x
            return float(tol), int(n_it), float(relres), 0
        xs, flag, relres, n_it, tol = self._slab_argmin_x(it, tol_ref)
        r0, r1 = self.r0, self.r1
        d, e = xs - self.m[r0:r1], self.x[r0:r1] - xs                     # obj / evol_x sums over the slab
        self._log_parts = np.array([O._sumsq64(d), O._sumsq64(e), O._sumsq64(xs)])
        buf = self._padded(self.x)
        buf[r0:r1] = xs
        t = torch.from_numpy(buf)
        self.comm.allgather_(t, self.chunk)
        self.x = t.numpy()[:self.N].copy()
        return float(tol), int(n_it), float(relres), int(flag)

    def update_y_l(self, it, flags, rho, gamma):
        TF, p = self.TF, self.p
        rho, gamma = np.asarray(rho, TF), np.asarray(gamma, TF)
        self.rho_prox = rho

        class L: pass
        log = L(); log.r_pri = np.zeros((it, p)); log.r_dual = np.zeros((it, p)); log.set_feasibility = np.zeros((2, self.pp))
        assert bool(flags & YL_FEAS) == (it % 10 == 0)
        O.update_y_l(self.x, p, it, self.y, self.y_old, self.l, self.l_old, rho, gamma, self.prox, self.A, log, self.P, 2,
                     self.x_hat, self.r_pri, self.s, False, only=self.owned)
        self._flags, self._rho = flags, rho
        if flags & YL_FIRST:
            for ii in self.owned:
                self.l_hat[ii][:] = self.l_old[ii] + TF(rho[ii]) * (-self.s[ii] + self.y_old[ii])
                self.l_hat_0[ii][:] = self.l_hat[ii]; self.y_0[ii][:] = self.y[ii]
                self.s_0[ii][:] = self.s[ii]; self.l_0[ii][:] = self.l[ii]
        rp, rd, fe = log.r_pri[it - 1].copy(), log.r_dual[it - 1].copy(), log.set_feasibility[1].copy()
        if self.comm is not None:                     # ONE all-reduce of the packed scalars; non-owners contributed zeros
            pack = self._allreduce(np.concatenate([rp, rd, fe, self._log_parts]))
            rp, rd, fe, self._log_sums = pack[:p], pack[p:2 * p], pack[2 * p:2 * p + self.pp], pack[2 * p + self.pp:]
        return rp, rd, fe

    def log_scalars(self):
        TF = self.TF
        with np.errstate(all="ignore"):
            if self.comm is not None:
                so, se, sx = self._log_sums
                nd = TF(math.sqrt(so))
                return float(TF(0.5) * TF(nd * nd)), float(TF(TF(math.sqrt(se)) / TF(math.sqrt(sx))))
            nd = O.nrm2(self.x - self.m, TF)
            return float(TF(0.5) * TF(nd * nd)), float(TF(O.nrm2(self.x_old - self.x, TF) / O.nrm2(self.x, TF)))

    def adapt_rho_gamma(self, adjust_rho, adjust_gamma, rho, gamma):
        TF = self.TF
        rho, gamma = np.array(rho, TF), np.array(gamma, TF)
        first = bool(self._flags & YL_FIRST)
        O.adapt_rho_gamma(gamma, rho, adjust_gamma, adjust_rho, self.y, self.y_old, self.s, self.s_0, self.l,
                          self.l_hat_0, self.l_0, self.l_old, self.y_0, self.p, self.l_hat, only=self.owned)
        if not first:     # engine semantics: snapshots refresh together with the BB sums (PARSDMM.jl:192-206)
            for ii in self.owned:
                self.l_hat_0[ii][:] = self.l_hat[ii]; self.y_0[ii][:] = self.y[ii]
                self.s_0[ii][:] = self.s[ii]; self.l_0[ii][:] = self.l[ii]
        rho, gamma = rho.astype(np.float64), gamma.astype(np.float64)
        if self.comm is not None:     # the engine all-reduces the six BB sums and applies the rule everywhere; same outcome
            own = np.zeros(self.p); own[self.owned] = 1.0
            pack = self._allreduce(np.concatenate([rho * own, gamma * own]))
            rho, gamma = pack[:self.p], pack[self.p:]
        return rho, gamma

    def q_update(self, rho_new, rho_old):
        class L: pass
        log = L(); log.rho = np.asarray(rho_old, np.float64)[None, :]
        upd = [i for i in range(self.p) if rho_new[i] != rho_old[i]]
        O.Q_update(self.Q, self.AtA, self.prop, np.asarray(rho_new, self.TF), upd, log, 0, self.Qo)


def _setup(TF, n=(24, 18), h=(25.0, 6.0)):
    rng = np.random.default_rng(20240601)
    z = np.linspace(0, 1, n[-1])[None, :]
    m = (1500 + 2500 * z + 150 * rng.standard_normal(n)).astype(TF).reshape(-1, order="F")
    g = O.compgrid(h, n)
    Dz = O.get_TD_operator(g, "D_z", TF)[0]; TV = O.get_TD_operator(g, "TV", TF)[0]
    c = [O.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", "")),
         O.set_definitions("bounds", "D_z", float(0.5 * (Dz @ m).min()), float(0.5 * (Dz @ m).max()), ("matrix", "")),
         O.set_definitions("l1", "TV", 0.0, float(0.5 * np.abs(TV @ m).sum()), ("matrix", ""))]
    opt = O.PARSDMM_options(FL=TF, maxit=45)
    P, A, prop = O.setup_constraints(c, g, TF)
    A, AtA, l, y = O.PARSDMM_precompute_distribute(A, prop, g, opt)
    return m, g, opt, P, A, prop, AtA


def _drive(sharded, m, opt, P, A, prop, AtA, comm, owned, n=(24, 18)):
    eng = OracleEngine(m, AtA, A, prop, P, opt, owned, comm, n)
    drv = sharded.PhaseDriver(eng, opt, any(prop.ncvx[:len(P)]))
    while not drv.step():
        pass
    return eng.x.copy(), drv.result_log()


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_phase_driver_equals_oracle_loop(sipx, TF):
    from sipx import sharded
    m, g, opt, P, A, prop, AtA = _setup(TF)
    xo, lo, _, _ = O.PARSDMM(m.copy(), AtA, A, prop, P, g, O.PARSDMM_options(FL=TF, maxit=45))
    xs, ls = _drive(sharded, m, opt, P, A, prop, AtA, None, [1] * len(A))
    assert len(ls.obj) == len(lo.obj)
    assert np.array_equal(xs, xo)
    for f in ("obj", "r_pri", "r_dual", "r_pri_total", "rho", "gamma", "cg_it", "cg_relres", "set_feasibility"):
        a, b = np.asarray(getattr(ls, f), np.float64), np.asarray(getattr(lo, f), np.float64)
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), f
    assert sharded.shard_sets(5, 2, 0) == [1, 0, 1, 0, 1] and sharded.shard_sets(5, 2, 1) == [0, 1, 0, 1, 0]


def test_slab_partition():
    from __graft_entry__ import load_package
    load_package()
    from sipx import sharded
    P = 7
    assert sharded.slab_partition(16, P, 2) == (8 * P, [(0, 8 * P), (8 * P, 16 * P)])
    assert sharded.slab_partition(16, P, 3) == (6 * P, [(0, 6 * P), (6 * P, 12 * P), (12 * P, 16 * P)])       # ragged
    assert sharded.slab_partition(5, P, 4) == (2 * P, [(0, 2 * P), (2 * P, 4 * P), (4 * P, 5 * P), (5 * P, 5 * P)])   # an empty slab
    assert sharded.slab_partition(3, P, 8)[1][2:] == [(2 * P, 3 * P)] + [(3 * P, 3 * P)] * 5
    for n_last, world in ((256, 8), (512, 8), (17, 5), (2048, 6)):
        chunk, slabs = sharded.slab_partition(n_last, P, world)
        assert chunk * world >= n_last * P and slabs[0][0] == 0 and slabs[-1][1] == n_last * P
        assert all(a[1] == b[0] for a, b in zip(slabs, slabs[1:])) and all(r1 - r0 <= chunk for r0, r1 in slabs)


def _collectives_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        from __graft_entry__ import load_package
        load_package()
        from sipx import sharded
        comm = sharded.TorchComm(dist)
        chunk = 5
        base = np.arange(world * chunk, dtype=np.float64)
        t = torch.from_numpy(base * (rank + 1))
        comm.allreduce_sum_(t)
        ok = np.array_equal(t.numpy(), base * sum(range(1, world + 1)))
        t = torch.from_numpy(base * (rank + 1))
        comm.reduce_scatter_sum_(t, chunk)
        sl = slice(rank * chunk, (rank + 1) * chunk)
        ok &= np.array_equal(t.numpy()[sl], (base * sum(range(1, world + 1)))[sl])
        t = torch.zeros(world * chunk, dtype=torch.float32)
        t[sl] = float(rank + 1)
        comm.allgather_(t, chunk)
        ok &= np.array_equal(t.numpy(), np.repeat(np.arange(1, world + 1, dtype=np.float32), chunk))
        buf = torch.full((4, 3), -1.0)            # rows: halo_prev, first plane, last plane, halo_next
        buf[1], buf[2] = 10.0 * rank + 1, 10.0 * rank + 2
        prev, nxt = (rank - 1 if rank > 0 else -1), (rank + 1 if rank < world - 1 else -1)
        comm.halo_exchange(buf[1], buf[0], prev, buf[2], buf[3], nxt)
        ok &= bool((buf[0] == (10.0 * (rank - 1) + 2 if prev >= 0 else -1.0)).all())
        ok &= bool((buf[3] == (10.0 * (rank + 1) + 1 if nxt >= 0 else -1.0)).all())
        t = torch.arange(world * chunk, dtype=torch.float64) * (100.0 if rank == 1 else 1.0)     # scatter from rank 1
        comm.scatter_(t, chunk, 1)
        ok &= np.array_equal(t.numpy()[sl], (np.arange(world * chunk) * 100.0)[sl])
        t = torch.zeros(world * chunk, dtype=torch.float64)
        t[sl] = float(rank + 7)
        comm.gather_(t, chunk, 2)                                                                 # gather to rank 2
        if rank == 2:
            ok &= np.array_equal(t.numpy(), np.repeat(np.arange(7, 7 + world, dtype=np.float64), chunk))
        ok &= comm.calls == {"allreduce": 1, "reduce_scatter": 1, "allgather": 1, "halo": 1, "scatter": 1, "gather": 1}
        open(os.path.join(out, f"ok{rank}"), "w").write(str(bool(ok)))
    finally:
        dist.destroy_process_group()


def test_collectives_over_gloo(tmp_path):
    """sharded.TorchComm's four in-place operations (the semantics sipx_comm asks of its callbacks), 3 ranks."""
    world = 3
    mp.spawn(_collectives_worker, args=(world, 30500 + (os.getpid() % 2000), str(tmp_path)), nprocs=world, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(world)] == ["True"] * world


def _worker(rank, world, port, TF, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")          # the container hostname may not resolve
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        from __graft_entry__ import load_package
        load_package()
        from sipx import sharded
        m, g, opt, P, A, prop, AtA = _setup(TF)
        owned = sharded.shard_sets(len(A), world, rank)
        x, log = _drive(sharded, m, opt, P, A, prop, AtA, sharded.TorchComm(dist), owned)
        np.savez(os.path.join(out, f"r{rank}.npz"), x=x, obj=log.obj, rho=log.rho, gamma=log.gamma, cg_it=log.cg_it,
                 r_pri=log.r_pri, feas=log.set_feasibility)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("TF", [np.float32, np.float64])
def test_two_rank_sharding_matches_serial(TF, tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000) + (7 if TF == np.float64 else 0)
    mp.spawn(_worker, args=(world, port, TF, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    for k in r0.files:                                # every rank holds identical replicated results
        assert np.array_equal(r0[k], r1[k], equal_nan=True), k
    m, g, opt, P, A, prop, AtA = _setup(TF)
    xo, lo, _, _ = O.PARSDMM(m.copy(), AtA, A, prop, P, g, O.PARSDMM_options(FL=TF, maxit=45))
    K = min(8, len(lo.obj), len(r0["obj"]))
    rt = 5e-4 if TF == np.float32 else 1e-9
    assert np.array_equal(r0["cg_it"][:K], lo.cg_it[:K])
    assert np.allclose(r0["obj"][:K], lo.obj[:K], rtol=rt) and np.allclose(r0["rho"][:K], lo.rho[:K], rtol=rt)
    assert np.allclose(r0["r_pri"][:K], lo.r_pri[:K], rtol=rt, atol=1e-12)      # r_pri of every set reaches every rank
    err = np.linalg.norm(r0["x"].astype(np.float64) - xo) / np.linalg.norm(xo)
    assert err < (5e-4 if TF == np.float32 else 1e-6), err


def test_four_ranks_ragged_slabs(tmp_path):
    """18 planes over 4 ranks: slabs of 5, 5, 5 and 3 planes; 4 terms, one per rank."""
    TF, world = np.float64, 4
    mp.spawn(_worker, args=(world, 30900 + (os.getpid() % 2000), TF, str(tmp_path)), nprocs=world, join=True)
    rs = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    for r in rs[1:]:
        for k in rs[0].files:
            assert np.array_equal(rs[0][k], r[k], equal_nan=True), k
    m, g, opt, P, A, prop, AtA = _setup(TF)
    xo, lo, _, _ = O.PARSDMM(m.copy(), AtA, A, prop, P, g, O.PARSDMM_options(FL=TF, maxit=45))
    K = min(8, len(lo.obj), len(rs[0]["obj"]))
    assert np.array_equal(rs[0]["cg_it"][:K], lo.cg_it[:K]) and np.allclose(rs[0]["obj"][:K], lo.obj[:K], rtol=1e-9)
    assert np.linalg.norm(rs[0]["x"] - xo) / np.linalg.norm(xo) < 1e-6


def test_more_ranks_than_terms(tmp_path):
    """8 GPUs, 5 terms: some ranks own no set at all and only take part in the collectives and the replicated x-step.
    Rehearsed with 6 gloo ranks on 4 terms."""
    TF, world = np.float64, 6
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, TF, str(tmp_path)), nprocs=world, join=True)
    rs = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    for r in rs[1:]:
        for k in rs[0].files:
            assert np.array_equal(rs[0][k], r[k], equal_nan=True), k
    m, g, opt, P, A, prop, AtA = _setup(TF)
    assert len(A) < world
    xo, lo, _, _ = O.PARSDMM(m.copy(), AtA, A, prop, P, g, O.PARSDMM_options(FL=TF, maxit=45))
    K = min(8, len(lo.obj), len(rs[0]["obj"]))
    assert np.array_equal(rs[0]["cg_it"][:K], lo.cg_it[:K]) and np.allclose(rs[0]["obj"][:K], lo.obj[:K], rtol=1e-9)
    err = np.linalg.norm(rs[0]["x"] - xo) / np.linalg.norm(xo)
    assert err < 1e-6, err


# ---- slab decomposition of the whole iteration: the exchange steps of a threshold search --------------------------------
def _slab_search_worker(rank, world, port, out):
    """One l1 threshold search as the slab-decomposed iteration runs it (csrc/kernels_proj.hip, chain_stage): every rank sweeps
    ITS planes of v, ONE all-reduce makes the probe sums global, the bracket decision is the same scalar code on every rank,
    ONE all-gather strings the magnitudes inside the bracket together, every rank solves the same small problem.  numpy
    stand-in for the kernels (test infrastructure), the product's communicator class for the two collectives."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        from __graft_entry__ import load_package
        load_package()
        from sipx import sharded
        comm = sharded.TorchComm(dist)
        rng = np.random.default_rng(11)
        n_last, plane = 13, 40                                    # ragged slabs: ceil(13 / world) planes per rank
        v = (rng.standard_normal(n_last * plane) * np.exp(rng.standard_normal(n_last * plane))).astype(np.float64)
        b = 0.3 * np.abs(v).sum()
        chunk, slabs = sharded.slab_partition(n_last, plane, world)
        r0, r1 = slabs[rank]
        mine = np.abs(v[r0:r1])
        # stage 0: sums at eight probes (and ||v||_1, count) over the rank's planes, all-reduced
        t = np.quantile(np.abs(v), [0.2, 0.4, 0.55, 0.7, 0.8, 0.9, 0.96, 0.99])        # any probes: the search does not depend on them
        red = np.concatenate([[mine.sum(), float((mine > 0).sum())], [mine[mine > tk].sum() for tk in t],
                              [float((mine > tk).sum()) for tk in t]])
        red_t = torch.from_numpy(red)
        comm.allreduce_sum_(red_t)
        red = red_t.numpy()
        S, C = red[2:10], red[10:18]
        f = S - t * C - b
        lo_k = max(k for k in range(8) if f[k] >= 0) if (f >= 0).any() else -1       # f decreases: bracket between two probes
        lo = t[lo_k] if lo_k >= 0 else 0.0
        hi = t[lo_k + 1] if lo_k + 1 < 8 else np.inf
        s_above = S[lo_k + 1] if lo_k + 1 < 8 else 0.0
        c_above = C[lo_k + 1] if lo_k + 1 < 8 else 0.0
        # stage 2: the rank's magnitudes inside (lo, hi] into its segment (count in front), all-gathered
        gcap = n_last * plane
        seg = torch.zeros(world * (gcap + 1), dtype=torch.float64)
        g = mine[(mine > lo) & (mine <= hi)]
        seg[rank * (gcap + 1)] = float(len(g))
        seg[rank * (gcap + 1) + 1:rank * (gcap + 1) + 1 + len(g)] = torch.from_numpy(g)
        comm.allgather_(seg, gcap + 1)
        seg = seg.numpy()
        vals = np.concatenate([seg[r * (gcap + 1) + 1:r * (gcap + 1) + 1 + int(seg[r * (gcap + 1)])] for r in range(world)])
        # stage 3: Michelot from the lower end of the bracket (kernels_proj.hip, k_l1_solve)
        theta, cprev = lo, -1.0
        for _ in range(200):
            act = vals[vals > theta]
            cnt = c_above + len(act)
            tn = (s_above + math.fsum(act) - b) / cnt
            if len(act) == cprev:
                break
            theta, cprev = max(tn, theta), len(act)
        ref = O.project_l1_Duchi(v.copy(), b)
        got = np.sign(v) * np.maximum(np.abs(v) - theta, 0.0)
        ok = np.linalg.norm(got - ref) <= 1e-12 * np.linalg.norm(ref) and abs(np.abs(got).sum() - b) <= 1e-10 * b
        ok &= comm.calls["allreduce"] == 1 and comm.calls["allgather"] == 1 and comm.calls["reduce_scatter"] == 0
        np.save(os.path.join(out, f"theta{rank}.npy"), np.array([theta]))
        open(os.path.join(out, f"ok{rank}"), "w").write(str(bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_threshold_search_over_gloo(tmp_path, world):
    """The exchange pattern of a slab-decomposed l1 search: one all-reduce + one all-gather, theta identical (bits) on every
    rank and equal to the oracle's sort-based threshold."""
    mp.spawn(_slab_search_worker, args=(world, 30900 + (os.getpid() % 2000) + world, str(tmp_path)), nprocs=world, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(world)] == ["True"] * world
    th = [np.load(tmp_path / f"theta{r}.npy") for r in range(world)]
    assert all(np.array_equal(th[0], t) for t in th[1:])


def test_slab_decomposable_classifier(sipx):
    """Which set lists may be decomposed by slab for the whole iteration (sharded.slab_decomposable): C3 yes, C4 (DFT, slice
    rank, cardinality) no."""
    from sipx import sharded
    TF = np.float32
    g = sipx.compgrid((25.0, 25.0, 25.0), (8, 8, 8))

    def build(defs):
        P, A, prop = sipx.setup_constraints(defs, g, TF)
        return P, A

    sd = sipx.set_definitions
    c3 = [sd("bounds", "identity", 0.0, 1.0, ("matrix", "")), sd("l1", "D_x", 0.0, 1.0, ("matrix", "")),
          sd("l1", "D_z", 0.0, 1.0, ("matrix", "")), sd("annulus", "identity", 1.0, 2.0, ("matrix", "")), sd("l1", "TV", 0.0, 1.0, ("matrix", ""))]
    assert sharded.slab_decomposable(*build(c3))
    for bad in (sd("cardinality", "D_z", 0, 10, ("matrix", "")), sd("rank", "identity", 0, 2, ("slice", "z")),
                sd("l1", "DFT", 0.0, 1.0, ("matrix", "")), sd("bounds", "identity", 0.0, 1.0, ("fiber", "z"))):
        assert not sharded.slab_decomposable(*build(c3[:2] + [bad]))
        # ... but the slab decomposition TAKES such lists when it is asked for (round 5: slab-local slices, a search over the slab
        # collectives, a slab-decomposed transform, an owner rank): "auto" stays with the lists whose every projector works from sums
        assert sharded.slab_admissible(*build(c3[:2] + [bad]))
    assert sharded.slab_admissible(*build(c3))
