"""Sharded PARSDMM: one process per GPU.

The reference's parallel mode gives every constraint set its own Julia worker, sums the partial
right-hand sides with a (+) reduction and ships x to every worker (src/PARSDMM.jl:114-131,183-197,
src/rhs_compose.jl:17-20, src/update_y_l_parallel.jl:6-90); its x-step stays on the master.  Here a
rank is one GPU and the ENGINE enqueues the collectives itself (csrc/comm.cpp, csrc/engine.cpp):

  1. every rank forms the partial rhs of the sets it owns (set i lives on rank i mod world),
  2. reduce-scatter of rhs by z-slab: rank r receives the summed rows of ITS slab of the grid,
  3. CG on the slab rows of Q: one halo plane of p from each neighbour per product, the dot products
     through an all-reduce of the float64 block partials (identical bits on every rank, hence
     identical decisions); x is completed with an all-gather,
  4. every rank updates its own sets; ONE all-reduce of the packed per-set sums gives every rank the
     r_pri / r_dual / feasibility / Barzilai-Borwein sums of every set,
  5. stop rule, rho / gamma rules and the Q update (slab rows only) are replicated scalar / local work.

This module is the host side: the communicator a context is given (native RCCL, or torch.distributed
collectives handed to the engine as C callbacks -- what the tests use to run several ranks on ONE
GPU over gloo), and the phase-level driver that restates the reference's main loop over the phase
entry points of the C ABI (what a Julia shim keeping PARSDMM.jl's own loop would ccall).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from .host import YL_BB, YL_FEAS, YL_FIRST, TIMING_SECTIONS, log_type_PARSDMM


def shard_sets(p: int, world: int, rank: int) -> List[int]:
    """owned[i] = 1 iff term i (constraint sets, then the distance term) lives on `rank`."""
    return [1 if (i % world) == rank else 0 for i in range(p)]


def slab_partition(n_last: int, plane: int, world: int):
    """The engine's z-slabs of the x-step: ceil(n_last / world) planes per rank; returns (chunk, [(row0, row1)] per rank).
    Ranks past the end of the grid hold an empty slab; the exchange buffers are padded to world * chunk elements."""
    planes = -(-n_last // world)
    chunk = planes * plane
    N = n_last * plane
    return chunk, [(min(N, r * chunk), min(N, (r + 1) * chunk)) for r in range(world)]


def _nanmax(v) -> float:
    v = np.asarray(v, np.float64)
    return float("nan") if np.isnan(v).any() else float(v.max())


def _argmax_julia(row) -> int:
    row = np.asarray(row, np.float64)
    nan = np.nonzero(np.isnan(row))[0]
    return int(nan[0]) if len(nan) else int(np.argmax(row))


class _SipxComm(C.Structure):
    """sipx_comm of include/sipx.h."""
    _AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)
    _HX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                      C.c_int32, C.c_void_p)
    _BC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p)
    _fields_ = [("user", C.c_void_p), ("world", C.c_int32), ("rank", C.c_int32), ("allreduce_sum", _AR),
                ("reduce_scatter_sum", _AR), ("allgather", _AR), ("halo_exchange", _HX), ("scatter", _BC), ("gather", _BC)]


class TorchComm:
    """The four collectives of the sharded solve over torch.distributed.

    * backend "nccl" (= RCCL on ROCm): device buffers of the engine are aliased as tensors and the collectives are
      enqueued with the engine's HIP stream as torch's current stream -- no host synchronisation;
    * backend "gloo": the buffer is staged through the host (stream synchronised first).  This is how the tests run
      several ranks on ONE GPU, and how the CPU tests drive the same class on numpy arrays.
    The tensor-level methods are usable on their own (CPU tests); ``c_struct()`` wraps them as the sipx_comm callbacks.
    """

    def __init__(self, dist, device=None):
        import torch
        self.torch, self.dist = torch, dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.device = device
        self.nccl = dist.get_backend() == "nccl"
        self._keep = None
        self.calls = {"allreduce": 0, "reduce_scatter": 0, "allgather": 0, "halo": 0, "scatter": 0, "gather": 0}

    # ---- tensor level (in place) ---------------------------------------------------------------------------------
    def allreduce_sum_(self, t):
        self.calls["allreduce"] += 1
        self.dist.all_reduce(t)

    def reduce_scatter_sum_(self, t, chunk):
        """t[rank*chunk:(rank+1)*chunk] <- sum over ranks of that range."""
        self.calls["reduce_scatter"] += 1
        mine = t[self.rank * chunk:(self.rank + 1) * chunk]
        if self.nccl:
            self.dist.reduce_scatter_tensor(mine, t)
        else:                                  # gloo has no reduce-scatter: reduce everything, keep the own range
            self.dist.all_reduce(t)

    def allgather_(self, t, chunk):
        """t[r*chunk:(r+1)*chunk] <- that range of rank r."""
        self.calls["allgather"] += 1
        mine = t[self.rank * chunk:(self.rank + 1) * chunk]
        if self.nccl:
            self.dist.all_gather_into_tensor(t, mine)
        else:
            parts = [self.torch.empty_like(mine) for _ in range(self.world)]
            self.dist.all_gather(parts, mine.clone())
            for r, v in enumerate(parts):
                t[r * chunk:(r + 1) * chunk] = v

    def halo_exchange(self, send_prev, recv_prev, prev, send_next, recv_next, nxt):
        self.calls["halo"] += 1
        ops = []
        P2P = self.dist.P2POp
        if prev >= 0:
            ops += [P2P(self.dist.isend, send_prev.contiguous(), prev), P2P(self.dist.irecv, recv_prev, prev)]
        if nxt >= 0:
            ops += [P2P(self.dist.isend, send_next.contiguous(), nxt), P2P(self.dist.irecv, recv_next, nxt)]
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()

    def scatter_(self, t, chunk, root):
        """rank r receives t[r*chunk:(r+1)*chunk] of rank `root` (in place; the root keeps its own range)."""
        self.calls["scatter"] += 1
        self._fan(t, chunk, root, True)

    def gather_(self, t, chunk, root):
        """rank `root` receives t[r*chunk:(r+1)*chunk] of every rank r (in place)."""
        self.calls["gather"] += 1
        self._fan(t, chunk, root, False)

    def _fan(self, t, chunk, root, out):
        if self.world == 1:
            return
        P2P, ops = self.dist.P2POp, []
        if self.rank == root:
            for p in range(self.world):
                if p != root:
                    v = t[p * chunk:(p + 1) * chunk]
                    ops.append(P2P(self.dist.isend if out else self.dist.irecv, v, p))
        else:
            v = t[self.rank * chunk:(self.rank + 1) * chunk]
            ops.append(P2P(self.dist.irecv if out else self.dist.isend, v, root))
        for w in self.dist.batch_isend_irecv(ops):
            w.wait()

    # ---- C callbacks ------------------------------------------------------------------------------------------------
    def _alias(self, ptr, count, dtype):
        torch = self.torch
        TF = np.float64 if dtype == 1 else np.float32

        class _A:
            __cuda_array_interface__ = {"shape": (int(count),), "typestr": np.dtype(TF).str, "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(_A(), device=self.device)

    def _run(self, stream, bufs, fn):
        """bufs: [(ptr, count, dtype, is_output)] device ranges; fn(tensors) performs the operation in place."""
        torch = self.torch
        ext = torch.cuda.ExternalStream(int(stream) if stream else 0, device=self.device)
        dev = [self._alias(p, n, dt) if p else None for p, n, dt, _ in bufs]
        if self.nccl:
            with torch.cuda.stream(ext):
                fn(dev)
            return
        ext.synchronize()
        host = [None if t is None else t.cpu() for t in dev]
        fn(host)
        for (p, n, dt, out), d, h in zip(bufs, dev, host):
            if out and d is not None:
                d.copy_(h)
        torch.cuda.synchronize(self.device)

    def c_struct(self) -> _SipxComm:
        def guard(f):
            def g(*a):
                try:
                    f(*a)
                    return 0
                except Exception:                      # an exception must not cross the C boundary
                    import traceback
                    traceback.print_exc()
                    return 1
            return g

        def ar(user, buf, count, dtype, stream):
            self._run(stream, [(buf, count, dtype, True)], lambda t: self.allreduce_sum_(t[0]))

        def rs(user, buf, chunk, dtype, stream):
            self._run(stream, [(buf, chunk * self.world, dtype, True)], lambda t: self.reduce_scatter_sum_(t[0], chunk))

        def ag(user, buf, chunk, dtype, stream):
            self._run(stream, [(buf, chunk * self.world, dtype, True)], lambda t: self.allgather_(t[0], chunk))

        def hx(user, sp, rp, prev, sn, rn, nxt, count, dtype, stream):
            self._run(stream, [(sp if prev >= 0 else None, count, dtype, False), (rp if prev >= 0 else None, count, dtype, True),
                               (sn if nxt >= 0 else None, count, dtype, False), (rn if nxt >= 0 else None, count, dtype, True)],
                      lambda t: self.halo_exchange(t[0], t[1], prev, t[2], t[3], nxt))
        def fan(buf, chunk, dtype, root, stream, out):
            # (a rank other than the root may touch its own range of the exchange layout only -- sipx.h: with sparse arrays the
            #  rest is not backed there -- so that range alone is aliased, shifted to where _fan looks for it)
            if self.rank == root:
                self._run(stream, [(buf, chunk * self.world, dtype, True)], lambda t: self._fan(t[0], chunk, root, out))
                return
            item = 8 if dtype == 1 else 4
            mine = buf + self.rank * chunk * item

            class _Shift:                       # t[rank*chunk:(rank+1)*chunk] of the whole layout == the aliased range
                def __init__(self, v, r, c):
                    self.v, self.r, self.c = v, r, c

                def __getitem__(self, sl):
                    assert sl.start == self.r * self.c and sl.stop == (self.r + 1) * self.c
                    return self.v
            self._run(stream, [(mine, chunk, dtype, out)], lambda t: self._fan(_Shift(t[0], self.rank, chunk), chunk, root, out))

        def sc(user, buf, chunk, dtype, root, stream):
            self.calls["scatter"] += 1
            fan(buf, chunk, dtype, root, stream, True)

        def ga(user, buf, chunk, dtype, root, stream):
            self.calls["gather"] += 1
            fan(buf, chunk, dtype, root, stream, False)
        cs = _SipxComm(None, self.world, self.rank, _SipxComm._AR(guard(ar)), _SipxComm._AR(guard(rs)), _SipxComm._AR(guard(ag)),
                       _SipxComm._HX(guard(hx)), _SipxComm._BC(guard(sc)), _SipxComm._BC(guard(ga)))
        self._keep = cs                                   # the engine holds the function pointers
        return cs


def attach_comm(ctx, dist, device=None, mode: Optional[str] = None):
    """Gives a context (before finalize) the communicator of this rank.  mode "rccl": RCCL inside the engine, the
    ncclUniqueId travels through torch.distributed; mode "torch": torch.distributed collectives as callbacks (any
    backend).  Default: "rccl" when the process group is nccl, else "torch".  Returns what must stay alive with the context."""
    import os
    import torch
    from .host import lib, _chk
    mode = mode or os.environ.get("SIPX_COMM") or ("rccl" if dist.get_backend() == "nccl" else "torch")
    world, rank = dist.get_world_size(), dist.get_rank()
    if mode == "rccl":
        uid = (C.c_char * 128)()
        if rank == 0:
            _chk(lib().sipx_rccl_unique_id(uid))
        box = [bytes(uid.raw)]
        dist.broadcast_object_list(box, src=0, device=device if dist.get_backend() == "nccl" else None)
        _chk(lib().sipx_set_comm_rccl(ctx.h, box[0], world, rank))
        return None
    if mode != "torch":
        raise ValueError(f"unknown communicator mode {mode!r}")
    comm = TorchComm(dist, device if device is not None else torch.device("cuda", 0))
    cs = comm.c_struct()
    _chk(lib().sipx_set_comm(ctx.h, C.byref(cs)))
    return comm


class PhaseDriver:
    """One PARSDMM solve advanced iteration by iteration (``step()``) over the phase entry points -- the reference's main
    loop (src/PARSDMM.jl:97-254) as a Julia shim would keep it.  The context may be a rank of a sharded solve: its phase
    calls then contain the collectives and return the scalars of every set, so nothing here knows about ranks."""

    def __init__(self, ctx, options, any_ncvx=False):
        self.ctx, self.o = ctx, options
        TF = self.TF = ctx.TF
        self.p, self.pp = ctx.p, ctx.pp
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
        self.tol_ref, cg_it, relres, _ = ctx.argmin_x(i, self.tol_ref)               # :106-107
        log.cg_it[i - 1], log.cg_relres[i - 1] = cg_it, relres
        self.cg_total += int(cg_it)
        flags = (YL_FEAS if i % 10 == 0 else 0) | (YL_FIRST if i == 1 else 0)
        bb_due = (self.adjust_rho or self.adjust_gamma) and i % self.freq == 0        # :182
        if bb_due:
            flags |= YL_BB
        rp, rd, fe = ctx.update_y_l(i, flags, self.rho, self.gamma)                  # :133
        obj, evol = ctx.log_scalars()
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
            rho, self.gamma = ctx.adapt_rho_gamma(self.adjust_rho, self.adjust_gamma, self.rho, self.gamma)
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


def slab_decomposable(P_sub, TD_OP) -> bool:
    """Can the WHOLE iteration be decomposed by z-slab (sipx_set_decomp(SIPX_DECOMP_SLAB))?  Only for sets whose projector
    needs nothing but sums over the grid: bounds, l1 / l2 ball, annulus, prox_l1 applied to the whole vector, on the identity
    or on D_x / D_y / D_z / TV (TD_OP: one operator per set, the distance term's identity may follow)."""
    ok_kinds = {"bounds", "bounds_vec", "l1", "l2", "annulus", "prox_l1"}
    ok_ops = {"identity", "D_x", "D_y", "D_z", "TV", "D2D", "D3D"}
    for P, A in zip(P_sub, TD_OP):
        if getattr(P, "kind", None) not in ok_kinds or getattr(P, "mode", 1) != 0 or getattr(P, "transform", 1) != 0:
            return False
        if getattr(A, "kind", None) not in ok_ops or getattr(A, "component", 0) != 0:
            return False
    return True


def slab_admissible(P_sub, TD_OP) -> bool:
    """Does the slab decomposition TAKE this list (round 5)?  Beyond the lists of `slab_decomposable`: a slice-wise rank /
    nuclear-norm set on the z-slices of the identity (every rank projects the slices of its own slab) and sets whose projector
    needs the whole array -- l1 / bounds behind the DFT, cardinality, the DCT sets, histogram, subspace -- which an owner rank
    projects on the gathered vector (two fan exchanges of N w bytes per such set and iteration).  Not taken: caller-supplied
    sparse operators, Minkowski components (the engine says so at sipx_finalize)."""
    ok_ops = {"identity", "D_x", "D_y", "D_z", "TV", "D2D", "D3D"}
    for P, A in zip(P_sub, TD_OP):
        if getattr(A, "kind", None) not in ok_ops or getattr(A, "component", 0) != 0:
            return False
    return True


def PARSDMM_sharded(m, AtA, TD_OP, set_Prop, P_sub, comp_grid, options, dist=None, device=0, x=None, l=None, y=None,
                    comm_mode: Optional[str] = None, phase_driver: bool = False, decomp: str = "sets"):
    """PARSDMM as one rank of a sharded solve over ``dist`` (torch.distributed, one process per GPU).  Returns
    (x, log, l, y); x and the log are complete and identical on every rank.  decomp "sets": the reference's split by
    constraint set -- l / y hold the locally owned sets (zeros elsewhere); "slab": every rank works on its z-slab of every
    set (sipx.h, sipx_set_decomp) -- l / y are complete on every rank; "auto": "slab" where every projector works from sums
    over the grid (`slab_decomposable`; a list with gathered sets, `slab_admissible`, is slab-decomposed when asked for).
    phase_driver=True runs the loop over the phase entry points instead of sipx_parsdmm."""
    from .host import build_context, set_default_device
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    p = len(TD_OP)
    if decomp == "auto":
        decomp = "slab" if (dist is not None and slab_decomposable(P_sub, TD_OP) and getattr(options, "Q_mode", "cds") == "cds") else "sets"
    owned = shard_sets(p, world, rank)
    set_default_device(device)
    import torch
    keep = []
    attach = None
    if dist is not None:
        def attach(ctx):
            keep.append(attach_comm(ctx, dist, torch.device("cuda", device), comm_mode))
            if decomp == "slab":
                ctx.set_decomp("slab")
    ctx = build_context(m, AtA, TD_OP, set_Prop, P_sub, comp_grid, options, x, l, y, device, owned, attach)
    try:
        if phase_driver:
            drv = PhaseDriver(ctx, options, any(set_Prop.ncvx[:len(P_sub)]))
            while not drv.step():
                pass
            log, feasible = drv.result_log(), drv.stopped_feasible
        else:
            log, feasible = ctx.parsdmm(options)
        try:        # slab-decomposed: how many threshold searches the speculative exchange settled (engine counters; diagnostics)
            log.slab_searches = ctx.kernel_stats_all(-1).get("slab_searches")       # (-1: read-only, a caller's collection survives)
        except Exception:
            log.slab_searches = None
        if feasible:
            return np.array(m, copy=True), log, None, None
        xo, lo, yo = ctx.download()
    finally:
        ctx.close()
    return xo, log, lo, yo
