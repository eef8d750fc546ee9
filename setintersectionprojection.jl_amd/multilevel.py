"""Coarse-to-fine wrapper around the engine: host-side mirror of PARSDMM_multi_level
(src/PARSDMM_multi_level.jl:8-89) with setup_multi_level_PARSDMM (src/setup_multi_level_PARSDMM.jl:7-137),
constraint2coarse (src/constraint2coarse.jl:8-104) and interpolate_y_l (src/interpolate_y_l.jl:7-97).

Every level is one native solve (sipx_parsdmm) on its own context; x, l and y pass from a level to the next finer one
ON THE DEVICE (sipx_warm_start_from: the coarse context stays alive until the fine one has taken its start from it), nothing
visits the host between levels.  PARITY UNPINNED for the transfers: the reference's
multilevel test is disabled (test/runtests.jl:47) and no test fixes the tie rounding of
Interpolations.BSpline(Constant()) (Interpolations.jl 0.13); half-way positions on range(1, stop=nc, length=nf)
are taken to round up (floor(x + 1/2)).
The TV branch of interpolate_y_l splits the multipliers with [D_x; D_y; D_z] block sizes although the
storage order is [D_z; D_y; D_x] -- replicated as written.
"""
from __future__ import annotations

import copy
import math

import os
import numpy as np

from . import host


def constraint2coarse(constraint, comp_grid, cf):
    n = tuple(int(v) for v in comp_grid.n)
    dim3 = len(n) == 3 and n[2] > 1
    for c in constraint:
        if c.set_type == "rank":
            c.max = min(c.max, min(n))
        if c.set_type == "cardinality":
            c.max = min(c.max, int(np.prod(n)))
        if c.set_type == "l1":
            c.max = c.max / (cf ** 3 if dim3 else cf ** 2)
        if c.set_type == "l2":
            c.max = c.max / (math.sqrt(cf ** 3) if dim3 else cf)
        if c.set_type == "nuclear" and not dim3:
            c.max = c.max / 2.7
    return constraint


def setup_multi_level_PARSDMM(m, n_levels, coarsening_factor, comp_grid, constraint, options):
    TF = m.dtype.type
    cf = coarsening_factor
    P_sub, TD_OP, prop = host.setup_constraints(copy.deepcopy(constraint), comp_grid, TF)
    TD_OP, AtA, l, y = host.PARSDMM_precompute_distribute(TD_OP, prop, comp_grid, options)
    TD_OP_levels, AtA_levels, P_sub_levels, prop_levels, grid_levels = [TD_OP], [AtA], [P_sub], [prop], [comp_grid]
    constraint_level = copy.deepcopy(constraint)
    n0 = tuple(int(v) for v in comp_grid.n)
    for i in range(2, n_levels + 1):
        n = tuple(int(np.rint(v / cf ** (i - 1))) for v in n0)
        d = tuple((a / b) * dd for a, b, dd in zip(n0, n, comp_grid.d))
        g = host.compgrid(d, n)
        constraint_level = constraint2coarse(constraint_level, g, cf)
        P, A, pr = host.setup_constraints(copy.deepcopy(constraint_level), g, TF)
        A, AtA_l, _, _ = host.PARSDMM_precompute_distribute(A, pr, g, options)
        TD_OP_levels.append(A); AtA_levels.append(AtA_l); P_sub_levels.append(P); prop_levels.append(pr); grid_levels.append(g)
    return TD_OP_levels, AtA_levels, P_sub_levels, prop_levels, grid_levels, constraint_level


def interpolate_y_l(l, y, set_Prop_levels, comp_grid_levels, dim3, i):
    nc = tuple(int(v) for v in comp_grid_levels[i + 1].n)
    nf = tuple(int(v) for v in comp_grid_levels[i].n)
    rs = host.resample_nn
    for j in range(len(l)):
        tag = set_Prop_levels[i].tag[j][1]
        if tag in ("TV", "D2D", "D3D"):
            if dim3:
                shapes_c = [(nc[0] - 1, nc[1], nc[2]), (nc[0], nc[1] - 1, nc[2]), (nc[0], nc[1], nc[2] - 1)]
                shapes_f = [(nf[0] - 1, nf[1], nf[2]), (nf[0], nf[1] - 1, nf[2]), (nf[0], nf[1], nf[2] - 1)]
            else:
                shapes_c = [(nc[0] - 1, nc[1]), (nc[0], nc[1] - 1)]
                shapes_f = [(nf[0] - 1, nf[1]), (nf[0], nf[1] - 1)]
            ends = np.cumsum([int(np.prod(s)) for s in shapes_c])
            starts = np.concatenate(([0], ends[:-1]))
            l[j] = np.concatenate([rs(l[j][a:b], sc, sf) for a, b, sc, sf in zip(starts, ends, shapes_c, shapes_f)])
            y[j] = np.concatenate([rs(y[j][a:b], sc, sf) for a, b, sc, sf in zip(starts, ends, shapes_c, shapes_f)])
        else:
            tdn_f = tuple(int(v) for v in set_Prop_levels[i].TD_n[j])
            tdn_c = tuple(int(v) for v in set_Prop_levels[i + 1].TD_n[j])
            s = tuple(a - b for a, b in zip(nf, tdn_f))
            fine = tuple(a - b for a, b in zip(nf, s))
            l[j] = rs(l[j], tdn_c, fine)
            y[j] = rs(y[j], tdn_c, fine)
    return l, y


def _carry_rho(options, log):
    """options.rho_ini = log.rho[end, :] (PARSDMM_multi_level.jl:57,83).  DEVIATION: when a level returned through
    the feasible-input exit (PARSDMM.jl:63-82) its one log row of rho is all zero and the reference would start the
    next level with rho = 0 (Q = 0, 1/rho = Inf, NaN iterates); the previous rho_ini is kept instead."""
    last = [float(v) for v in np.atleast_2d(log.rho)[-1, :]]
    if all(v > 0 for v in last):
        options.rho_ini = last


# The level contexts of the last multilevel call, kept for the next one (round 5, late).  A caller that projects again and again
# (the reference's examples wrap the call as a projector just like PARSDMM's: examples/constrained_freq_FWI_simple.jl:468) pays
# more than the allocations for building the levels anew: the driver wipes released device memory asynchronously and an allocation
# that follows a large release waits for the wipe -- 512^3 Float64: 2.4 s per call back to back where a call on a device at rest
# takes 1.0.  One entry (the levels of ONE problem: device, precision, grids, set descriptors -- host._context_key per level), only
# for lists whose descriptors carry no arrays, one rank, and only while the levels together hold at most 40 % of the device; a call
# with another problem releases it first.  A hit costs sipx_reset per level (the bits of a newly built context,
# tests/test_gpu_round5.py).  SIPX_MULTILEVEL_CACHE=0 switches it off; clear_level_cache() -- also reached through
# host.clear_context_cache() -- frees it.
_level_cache: "dict" = {}


def clear_level_cache():
    for ctxs in list(_level_cache.values()):
        for c in ctxs:
            c.close()
    _level_cache.clear()


host._extra_caches.append(clear_level_cache)


def _level_keys(m_levels, TD_OP_levels, AtA_levels, P_sub_levels, set_Prop_levels, comp_grid_levels, options, device):
    if os.environ.get("SIPX_MULTILEVEL_CACHE") == "0" or host._cache_limit() <= 0:
        return None
    keys = []
    for i in range(len(TD_OP_levels)):
        k = host._context_key(m_levels[i], AtA_levels[i], TD_OP_levels[i], set_Prop_levels[i], P_sub_levels[i], comp_grid_levels[i], options, device)
        if k is None:
            return None
        keys.append(k)
    return tuple(keys)


def PARSDMM_multi_level(m, TD_OP_levels, AtA_levels, P_sub_levels, set_Prop_levels, comp_grid_levels, options,
                        device=None, timings=None, host_transfers=False, dist=None, comm_mode=None, outputs="all"):
    """src/PARSDMM_multi_level.jl:8-89.  `timings` (a dict) receives per-level wall times: context set-up, the device-side
    warm start, the solve, its iteration count.  host_transfers=True keeps the round-1 path (download, resample through
    host.resample_nn / interpolate_y_l, upload at sipx_finalize) for A/B comparison.
    dist (torch.distributed, one process per GPU): every level is solved slab-decomposed over the ranks (sharded.py,
    sipx_set_decomp -- the set lists of the multilevel examples, bounds and l1 / TV, allow it); between levels the coarse
    slabs are all-gathered on the device one block at a time (a coarse-sized temporary) and every rank resamples the grid points
    it stores.  Every rank makes the same calls and
    returns the same x, log, l, y.  outputs="x": l and y stay on the device and come back as None (a caller that uses the projection
    alone does not wait for their copies: 8 of the 9 vectors of the {bounds, l1 TV} list)."""
    if outputs not in ("all", "x"):
        raise host.SipxError(f"outputs must be 'all' or 'x', not {outputs!r}")
    import time
    attach = None
    keep = []
    if dist is not None:
        if host_transfers:
            raise host.SipxError("the sharded multilevel solve passes the iterate between levels on the device")
        from . import sharded
        if not sharded.slab_decomposable(P_sub_levels[0], TD_OP_levels[0]):
            raise host.SipxError("the sharded multilevel solve needs a set list that can be decomposed by slab (sharded.slab_decomposable)")
        import torch

        def attach(ctx):
            keep.append(sharded.attach_comm(ctx, dist, torch.device("cuda", 0 if device is None else device), comm_mode))
            # (sparse arrays on every level since round 5: sipx_warm_start_from completes one coarse block at a time in a
            #  coarse-sized temporary and writes the grid points the fine rank stores; SIPX_MULTILEVEL_SLAB_FULL=1: whole arrays)
            ctx.set_decomp("slab_full" if os.environ.get("SIPX_MULTILEVEL_SLAB_FULL") == "1" else "slab")
    n_levels = len(TD_OP_levels)
    n0 = tuple(int(v) for v in comp_grid_levels[0].n)
    dim3 = len(n0) == 3 and n0[2] > 1
    rho_orig = list(options.rho_ini)
    m_levels = [m] + [host.resample_nn(m, n0, tuple(int(v) for v in comp_grid_levels[i].n)) for i in range(1, n_levels)]
    rec = timings if timings is not None else {}
    rec["levels"] = []
    prev = None
    x = l = y = None
    log = None
    key = None
    if dist is None and not host_transfers:
        key = _level_keys(m_levels, TD_OP_levels, AtA_levels, P_sub_levels, set_Prop_levels, comp_grid_levels, options, device)
    cached = _level_cache.pop(key, None) if key is not None else None
    if cached is None:
        clear_level_cache()                    # (another problem's levels: released before this one's are allocated)
    used = [None] * n_levels                   # the contexts of this call, by level
    rec["contexts_reused"] = cached is not None
    done = False
    try:
        for i in range(n_levels - 1, -1, -1):
            t0 = time.perf_counter()
            first = i == n_levels - 1
            options.zero_ini_guess = first or not host_transfers            # PARSDMM_multi_level.jl:53,81 (device path: start set below)
            if host_transfers and not first:
                nc = tuple(int(v) for v in comp_grid_levels[i + 1].n)
                nf = tuple(int(v) for v in comp_grid_levels[i].n)
                x = host.resample_nn(x, nc, nf)
                l, y = interpolate_y_l(list(l), list(y), set_Prop_levels, comp_grid_levels, dim3, i)
                options.zero_ini_guess = False
            if cached is not None:
                ctx = cached[i]
                TF = np.dtype(m.dtype).type
                ctx.reset(m_levels[i], [float(TF(r)) for r in options.rho_ini], float(TF(options.gamma_ini)), True)
            else:
                ctx = host.build_context(m_levels[i], AtA_levels[i], TD_OP_levels[i], set_Prop_levels[i], P_sub_levels[i],
                                         comp_grid_levels[i], options, x if host_transfers else None, l if host_transfers else None,
                                         y if host_transfers else None, device, None, attach)
            used[i] = ctx
            t1 = time.perf_counter()
            if prev is not None:
                ctx.warm_start_from(prev)                                    # x, l_i, y_i: coarse -> fine on the device
                if key is None:
                    prev.close()
                    used[i + 1] = None
                prev = None
            t2 = time.perf_counter()
            log, feasible = ctx.parsdmm(options)
            t3 = time.perf_counter()
            rec["levels"].append({"grid": [int(v) for v in comp_grid_levels[i].n], "context_s": t1 - t0, "warm_start_s": t2 - t1,
                                  "solve_s": t3 - t2, "iterations": int(len(log.obj)), "cg_iterations": int(np.sum(log.cg_it))})
            try:                                                             # what this level's context holds on this rank's GPU
                rec["levels"][-1]["device_bytes"] = int(ctx.device_bytes()["context"])
                rec["levels"][-1]["sparse_arrays"] = bool(ctx.kernel_stats_all(-1).get("sparse_arrays"))
            except Exception:
                pass
            _carry_rho(options, log)                                         # :57,83
            if host_transfers:
                x, l, y = ctx.download()
                ctx.close()
                used[i] = None
            else:
                prev = ctx
        if not host_transfers:
            if outputs == "x":
                t0 = time.perf_counter()
                x, l, y = prev.download(want_ly=False)
                rec["download_s"] = rec["download_x_only_s"] = time.perf_counter() - t0
                done = True
                return x, log, None, None
            if timings is not None and dist is None:
                # (what a caller that only wants x would wait for: sipx_download with x alone -- timed apart, the caller of this
                #  function subtracts it from its own clock)
                t0 = time.perf_counter()
                prev.download(want_ly=False)
                rec["download_x_only_s"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            x, l, y = prev.download()
            rec["download_s"] = time.perf_counter() - t0
        done = True
    finally:
        keep_them = False
        if done and key is not None and all(c is not None for c in used):
            try:                                   # (kept only while the levels together leave most of the device to others)
                held = sum(int(c.device_bytes()["context"]) for c in used)
                keep_them = held <= 0.4 * int(used[0].device_bytes()["device_total"])
            except Exception:
                keep_them = False
        if keep_them:
            _level_cache[key] = used
        else:
            for c in used:
                if c is not None:
                    c.close()
            if cached is not None:
                for c in cached:                   # (a reused context that this call did not get to is not left behind)
                    if c is not None and not any(c is u for u in used):
                        c.close()
        options.rho_ini = rho_orig                                           # :87
        options.zero_ini_guess = n_levels == 1                               # :53,81 leave it false after a warm-started level
    return x, log, l, y
