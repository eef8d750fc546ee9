"""The N>1 path on CPU: the phase-level driver (setintersectionprojection.jl_amd/sharded.py) with
set sharding over 2 gloo ranks.  The HIP engine cannot run here, so the driver is given a CPU
stand-in engine that answers the same phase calls with the ORACLE's functions (test
infrastructure injected by the test; the product never does this).  Checks:
  * world=1: the driver's control flow reproduces the oracle's PARSDMM loop bit for bit;
  * world=2: sharded == serial up to the summation order of rhs (reference tolerance
    5e-4 Float32, test/test_PARSDMM_parallel.jl:72; much tighter here)."""
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
    """CPU stand-in for host.Context: same phase methods, oracle arithmetic, owned-set mask."""

    def __init__(self, m, AtA, TD_OP, prop, P_sub, opt, owned):
        TF = self.TF = m.dtype.type
        self.m, self.AtA, self.A, self.prop, self.P = m, AtA, TD_OP, prop, P_sub
        self.p, self.pp, self.N = len(TD_OP), len(P_sub), len(m)
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
        self.feasibility_initial = f0

    def rhs_host_view(self):
        return self.rhs

    def rhs_compose(self, rho):
        self.rhs[:] = O.rhs_compose(self.l, self.y, np.asarray(rho, self.TF), self.A, self.p, self.N, only=self.owned)

    def argmin_x(self, it, tol_ref):
        self.x_old[:] = self.x
        x, n_it, relres, tol = O.argmin_x(self.Q, self.rhs, self.x, self.TF(tol_ref), it, self.Qo)
        self.x = x
        return float(tol), int(n_it), float(relres), 0

    def update_y_l(self, it, flags, rho, gamma):
        TF, p = self.TF, self.p
        rho, gamma = np.asarray(rho, TF), np.asarray(gamma, TF)
        self.rho_prox = rho

        class L: pass
        log = L(); log.r_pri = np.zeros((it, p)); log.r_dual = np.zeros((it, p)); log.set_feasibility = np.zeros((2, self.pp))
        i_eff = it if (flags & YL_FEAS) else (it if it % 10 else it + 1)
        assert bool(flags & YL_FEAS) == (it % 10 == 0)
        O.update_y_l(self.x, p, it, self.y, self.y_old, self.l, self.l_old, rho, gamma, self.prox, self.A, log, self.P, 2,
                     self.x_hat, self.r_pri, self.s, False, only=self.owned)
        self._flags, self._rho = flags, rho
        if flags & YL_FIRST:
            for ii in self.owned:
                self.l_hat[ii][:] = self.l_old[ii] + TF(rho[ii]) * (-self.s[ii] + self.y_old[ii])
                self.l_hat_0[ii][:] = self.l_hat[ii]; self.y_0[ii][:] = self.y[ii]
                self.s_0[ii][:] = self.s[ii]; self.l_0[ii][:] = self.l[ii]
        return log.r_pri[it - 1].copy(), log.r_dual[it - 1].copy(), log.set_feasibility[1].copy()

    def log_scalars(self):
        TF = self.TF
        nd = O.nrm2(self.x - self.m, TF)
        with np.errstate(all="ignore"):
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
        return rho.astype(np.float64), gamma.astype(np.float64)

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


def _drive(sharded, m, opt, P, A, prop, AtA, comm, owned):
    eng = OracleEngine(m, AtA, A, prop, P, opt, owned)
    drv = sharded.PhaseDriver(eng, opt, comm, owned, any(prop.ncvx[:len(P)]))
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
