"""Phase-level PARSDMM driver: the reference's main loop (src/PARSDMM.jl:97-254) restated over
the phase entry points of the C ABI -- exactly what a Julia shim keeping PARSDMM.jl's own loop
would ccall -- plus the set-sharded multi-GPU mode.

Sharding (reference parallel mode: one worker per set, src/PARSDMM.jl:114-131,183-197;
SURVEY 8e).  One process per GPU.  Rank r owns the sets {i : i mod world == r} (their y_i, l_i,
snapshots and projector).  Per iteration:
  1. every rank forms the partial rhs of its own sets,
  2. ONE all-reduce (RCCL over xGMI) sums the N-vector rhs in place      [rhs_compose.jl:17-20],
  3. every rank runs the same warm-started CG on the replicated Q, so x never has to be
     broadcast (the reference ships x to every worker each iteration, PARSDMM.jl:117-119),
  4. every rank updates its own sets; the per-set scalars (r_pri, r_dual, feasibility,
     obj/evol, adapted rho/gamma) travel in one small all-reduce of a packed vector in which
     non-owners contribute zeros  [the reference fetches whole r_pri vectors, PARSDMM.jl:122-125],
  5. stop rule, rho heuristics and the Q update are replicated scalar / local work.
All ranks see identical reduced data, hence take identical decisions.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from .host import YL_BB, YL_FEAS, YL_FIRST, TIMING_SECTIONS, log_type_PARSDMM


def shard_sets(p: int, world: int, rank: int) -> List[int]:
    """owned[i] = 1 iff term i (constraint sets, then the distance term) lives on `rank`."""
    return [1 if (i % world) == rank else 0 for i in range(p)]


def _nanmax(v) -> float:
    v = np.asarray(v, np.float64)
    return float("nan") if np.isnan(v).any() else float(v.max())


def _argmax_julia(row) -> int:
    row = np.asarray(row, np.float64)
    nan = np.nonzero(np.isnan(row))[0]
    return int(nan[0]) if len(nan) else int(np.argmax(row))


class LocalComm:
    """world == 1: nothing to exchange."""
    world, rank = 1, 0

    def allreduce_rhs(self, ctx):
        pass

    def allreduce_scalars(self, vec: np.ndarray) -> np.ndarray:
        return vec


class TorchComm:
    """torch.distributed collectives (backend "nccl" is RCCL on ROCm).  The engine's HIP stream is
    made torch's current stream while a collective is enqueued, so kernels and collectives are
    ordered on the device without host synchronisation."""

    def __init__(self, dist, device=None):
        import torch
        self.torch, self.dist = torch, dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.device = device
        self._rhs = None
        self._ext = None

    def _bind(self, ctx):
        torch = self.torch
        if self._rhs is not None:
            return
        if hasattr(ctx, "rhs_host_view"):          # CPU stand-in engine of the gloo tests
            self._rhs = torch.from_numpy(ctx.rhs_host_view())
            return
        from .host import lib

        class _Alias:                              # device buffer of the engine as a tensor, zero copy
            def __init__(self, ptr, n, TF):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": np.dtype(TF).str, "data": (ptr, False),
                                                 "version": 2}
        self._rhs = torch.as_tensor(_Alias(lib().sipx_dev_rhs(ctx.h), ctx.N, ctx.TF), device=self.device)
        self._ext = torch.cuda.ExternalStream(lib().sipx_stream(ctx.h), device=self.device)

    def allreduce_rhs(self, ctx):
        self._bind(ctx)
        if self._ext is not None:
            with self.torch.cuda.stream(self._ext):
                self.dist.all_reduce(self._rhs)
        else:
            self.dist.all_reduce(self._rhs)

    def allreduce_scalars(self, vec: np.ndarray) -> np.ndarray:
        t = self.torch.from_numpy(np.ascontiguousarray(vec, np.float64))
        if self.device is not None:
            t = t.to(self.device)
        self.dist.all_reduce(t)
        return t.cpu().numpy()


class PhaseDriver:
    """One PARSDMM solve advanced iteration by iteration (``step()``), serial or set-sharded."""

    def __init__(self, ctx, options, comm=None, owned: Optional[Sequence[int]] = None, any_ncvx=False):
        self.ctx, self.o, self.comm = ctx, options, comm or LocalComm()
        TF = self.TF = ctx.TF
        self.p, self.pp = ctx.p, ctx.pp
        self.owned = list(owned) if owned is not None else [1] * self.p
        self.maxit = int(options.maxit)
        self.evol_rel_tol, self.feas_tol, self.obj_tol = TF(options.evol_rel_tol), TF(options.feas_tol), TF(options.obj_tol)
        self.adjust_rho, self.adjust_gamma = bool(options.adjust_rho), bool(options.adjust_gamma)
        self.adjust_feas_rho = bool(options.adjust_feasibility_rho)
        self.freq = int(options.rho_update_frequency)
        gamma_ini = TF(options.gamma_ini)
        if any_ncvx:                                                   # PARSDMM_initialize.jl:107-114
            self.freq, self.adjust_gamma, gamma_ini = 3, False, TF(0.75)
        rho_ini = [TF(r) for r in options.rho_ini]
        self.rho = np.full(self.p, rho_ini[0], np.float64) if len(rho_ini) == 1 else np.array(rho_ini, np.float64)
        self.gamma = np.full(self.p, float(gamma_ini), np.float64)
        m, p, pp = self.maxit, self.p, self.pp
        self.log = log_type_PARSDMM(np.zeros((m, pp)), np.zeros((m, p)), np.zeros((m, p)), np.zeros(m), np.zeros(m),
                                    np.zeros(m), np.zeros(m), np.zeros((m, p)), np.zeros((m, p)),
                                    np.zeros(m, np.int64), np.zeros(m))
        feas0 = np.asarray(ctx.feasibility_initial, np.float64).copy()
        if not isinstance(self.comm, LocalComm):
            feas0 = self.comm.allreduce_scalars(feas0)
        self.log.set_feasibility[0, :] = feas0
        self.stopped_feasible = bool(pp > 0 and _nanmax(feas0) < float(self.feas_tol))   # PARSDMM.jl:63-82
        self.counter, self.ind_ref, self.tol_ref, self.i = 2, self.maxit, 1.0, 0
        self.done = self.stopped_feasible
        self.cg_total = 0

    # stop_PARSDMM.jl:23-52
    def _stop(self, i):
        log, TF = self.log, self.TF
        stop = False
        with np.errstate(all="ignore"):
            if i > 6 and self.pp > 0 and _nanmax(log.set_feasibility[self.counter - 2, :]) < float(self.feas_tol):
                a = log.obj[i - 6:i].astype(TF); b = log.obj[i - 7:i - 1].astype(TF)
                if _nanmax(np.abs((a - b) / b)) < float(self.obj_tol):
                    stop = True
            if i > 5 and _nanmax(log.evol_x[i - 6:i]) < float(self.evol_rel_tol):
                stop = True
            if i > 20 and self.adjust_rho:
                lo = max(i - 50, 1)
                if log.r_pri_total[i - 1] > _nanmax(log.r_pri_total[lo - 1:i - 1]):
                    self.adjust_rho = self.adjust_feas_rho = self.adjust_gamma = False
                    self.ind_ref = i
            if (not self.adjust_rho) and i > self.ind_ref + 25:
                lo = max(self.ind_ref, max(i - 50, 1))
                if log.r_pri_total[i - 1] > _nanmax(log.r_pri_total[lo - 1:i - 1]):
                    stop = True
        return stop

    def step(self) -> bool:
        """One pass of the loop body; returns True when the solve has stopped."""
        if self.done:
            return True
        ctx, log, TF, p, pp = self.ctx, self.log, self.TF, self.p, self.pp
        self.i += 1
        i = self.i
        ctx.rhs_compose(self.rho)                                                    # PARSDMM.jl:101
        if not isinstance(self.comm, LocalComm):
            self.comm.allreduce_rhs(ctx)
        self.tol_ref, cg_it, relres, _ = ctx.argmin_x(i, self.tol_ref)               # :106-107
        log.cg_it[i - 1], log.cg_relres[i - 1] = cg_it, relres
        self.cg_total += int(cg_it)
        flags = (YL_FEAS if i % 10 == 0 else 0) | (YL_FIRST if i == 1 else 0)
        bb_due = (self.adjust_rho or self.adjust_gamma) and i % self.freq == 0        # :182
        if bb_due:
            flags |= YL_BB
        rp, rd, fe = ctx.update_y_l(i, flags, self.rho, self.gamma)                  # :133
        own_dist = bool(self.owned[p - 1]) or pp == p
        obj, evol = ctx.log_scalars() if (own_dist or isinstance(self.comm, LocalComm)) else (0.0, 0.0)
        rho_new, gam_new = self.rho.copy(), self.gamma.copy()
        if bb_due:                               # speculative: discarded if the stop rule freezes rho below
            rho_new, gam_new = ctx.adapt_rho_gamma(self.adjust_rho, self.adjust_gamma, self.rho, self.gamma)
        if not isinstance(self.comm, LocalComm):
            own = np.asarray(self.owned, np.float64)
            lead = 1.0 if (own_dist and (pp < p or self.comm.rank == 0)) else 0.0
            pack = np.concatenate([rp * own, rd * own, fe * own[:pp], [obj * lead, evol * lead], rho_new * own,
                                   gam_new * own])
            pack = self.comm.allreduce_scalars(pack)
            rp, rd, fe = pack[:p], pack[p:2 * p], pack[2 * p:2 * p + pp]
            obj, evol = pack[2 * p + pp], pack[2 * p + pp + 1]
            rho_new, gam_new = pack[2 * p + pp + 2:3 * p + pp + 2], pack[3 * p + pp + 2:]
        log.r_pri[i - 1], log.r_dual[i - 1] = rp, rd
        sp, sd = TF(rp[0]), TF(rd[0])
        for k in range(1, p):
            sp, sd = TF(sp + TF(rp[k])), TF(sd + TF(rd[k]))
        log.r_pri_total[i - 1], log.r_dual_total[i - 1] = sp, sd                     # :134,138
        if i % 10 == 0:                                                              # update_y_l.jl:90-105
            log.set_feasibility[self.counter - 1, :] = fe
            self.counter += 1
        log.obj[i - 1], log.evol_x[i - 1] = obj, evol
        log.rho[i - 1], log.gamma[i - 1] = self.rho, self.gamma                      # :146-147
        if self._stop(i):                                                            # :153-158
            self.done = True
            return True
        rho = self.rho.copy()
        if bb_due and (self.adjust_rho or self.adjust_gamma):                        # :182-207
            rho, self.gamma = rho_new, gam_new
        if self.adjust_feas_rho and i % 10 == 0 and i > 10 and pp > 0:                # :213-223
            k = _argmax_julia(log.set_feasibility[self.counter - 2, :])
            rho[k] = float(TF(2.0) * TF(rho[k]))
        rho = np.maximum(np.minimum(rho.astype(TF), TF(1e4)), TF(1e-2)).astype(np.float64)   # :226
        ctx.q_update(rho, self.rho)                                                  # :230-243
        self.rho = rho
        if i == self.maxit:
            self.done = True
        return self.done

    def result_log(self) -> log_type_PARSDMM:
        """Truncation of output_check_PARSDMM (src/PARSDMM.jl:261-278)."""
        log = self.log
        if self.stopped_feasible:
            i, c = 1, 1
        else:
            i, c = self.i, self.counter
        return log_type_PARSDMM(log.set_feasibility[:c], log.r_dual[:i], log.r_pri[:i], log.r_dual_total[:i],
                                log.r_pri_total[:i], log.obj[:i], log.evol_x[:i], log.rho[:i], log.gamma[:i],
                                log.cg_it[:i], log.cg_relres[:i], dict.fromkeys(TIMING_SECTIONS, float("nan")))


def PARSDMM_sharded(m, AtA, TD_OP, set_Prop, P_sub, comp_grid, options, dist=None, device=0, x=None, l=None, y=None):
    """PARSDMM with the constraint sets sharded over the ranks of ``dist`` (torch.distributed, one
    process per GPU).  Returns (x, log, l, y); l/y hold the locally owned sets, zeros elsewhere."""
    from .host import build_context
    import torch
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    p = len(TD_OP)
    owned = shard_sets(p, world, rank)
    from .host import set_default_device
    set_default_device(device)
    ctx = build_context(m, AtA, TD_OP, set_Prop, P_sub, comp_grid, options, x, l, y, device, owned)
    try:
        comm = TorchComm(dist, torch.device("cuda", device)) if world > 1 else LocalComm()
        drv = PhaseDriver(ctx, options, comm, owned, any(set_Prop.ncvx[:len(P_sub)]))
        if drv.stopped_feasible:
            xo = np.array(m, copy=True)
            return xo, drv.result_log(), None, None
        while not drv.step():
            pass
        xo, lo, yo = ctx.download()
    finally:
        ctx.close()
    return xo, drv.result_log(), lo, yo
