#!/usr/bin/env python3
"""bench.py -- PARSDMM iterations/sec on synthetic grids (BASELINE.json metric).

One "step" = one PARSDMM iteration (rhs_compose -> CG x-minimisation -> y/l update of every
set -> logs -> stop rule -> rho/gamma adaptation -> Q update) on the configuration the metric
is quoted on: 3-D 256^3 Float32, sets {bounds on I, l1-ball on D_x, D_y, D_z} + the distance
term (BASELINE.json configs[2]; SURVEY 8d "C3").  Inputs are resident in HBM before the timed
region.  With --gpus N>1 (run plainly: the script spawns its own N ranks; or under torch.distributed.run)
the SAME projection problem is divided over the ranks (one process per GPU, RCCL inside the engine), so
scaling is "strong": the headline set list is decomposed by z-slab for the whole iteration (--decomp
slab: every rank holds every set, halo planes and small all-reduces only); set lists with a transform
or a factorisation (the c4 leg) are sharded by constraint set, the x-step on z-slabs (--decomp sets).

Prints ONE JSON line (rank 0).  roofline = the dominant kernel (cds_spmv fused with the CG dot
product), timed with HIP events on the engine stream over the timed steps; cpu_baseline = the
oracle's C/OpenMP port of the reference algorithm on the host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s measured copy ceiling

CONFIGS = {
    # name: (n, h, set kinds)
    "c3": ((256, 256, 256), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"]),
    "c2": ((2048, 2048), (25.0, 6.0), ["bounds", "l1:TV"]),
    "c3-512": ((512, 512, 512), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"]),
    "c3-768": ((768, 768, 768), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"]),     # ~90 GB of device state
    "c3-small": ((64, 64, 64), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"]),
    # one rank's share of c3 / c3-512 on eight GPUs (a slab of 1 / 8 of the planes): with SIPX_FORCE_DIST=1 --decomp slab, what the
    # slab-decomposed iteration costs a rank before any collective has a latency -- the floor of the strong-scaling curve
    "c3-slab8": ((256, 256, 32), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"]),
    "c3-512-slab8": ((512, 512, 64), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z"]),
    # BASELINE configs[3] (SURVEY 8d "C4"): 8 constraint sets, two of them non-convex
    "c4": ((512, 512, 512), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z", "annulus", "l1dft", "rank:32", "card:D_z"]),
    "c4-256": ((256, 256, 256), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z", "annulus", "l1dft", "rank:32", "card:D_z"]),
    # one rank's share of c4 on eight GPUs with the WHOLE iteration on z-slabs (SIPX_FORCE_DIST=1 --decomp slab): 64 of the 512 slices
    # for the rank set; the gathered sets (l1-DFT, cardinality) at 1 / 8 of their size -- on eight GPUs their owners project the whole
    # array while the others wait at the scatter (DESIGN 5 adds that to this figure)
    "c4-slab8": ((512, 512, 64), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z", "annulus", "l1dft", "rank:32", "card:D_z"]),
    "c4-small": ((64, 64, 64), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z", "annulus", "l1dft", "rank:8", "card:D_z"]),
}


def synthetic_model(n, TF, seed):
    """m = 1500 + 2500*z/(n_last-1) + 150*N(0,1), PCG64 seed 20240601+config# (SURVEY 8d)."""
    rng = np.random.default_rng(seed)
    z = np.linspace(0.0, 1.0, n[-1]).reshape((1,) * (len(n) - 1) + (-1,))
    return (1500.0 + 2500.0 * z + 150.0 * rng.standard_normal(n)).astype(TF).reshape(-1, order="F")


def build_problem(mod, n, h, kinds, m, TF, radius_of):
    g = mod.compgrid(h, n)
    c = []
    for k in kinds:
        if k == "bounds":
            c.append(mod.set_definitions("bounds", "identity", 1600.0, 3900.0, ("matrix", "")))
        elif k.startswith("l1:"):
            c.append(mod.set_definitions("l1", k[3:], 0.0, radius_of(k[3:]), ("matrix", "")))
        elif k == "annulus":                       # [0.9 ||m||, 0.98 ||m||]
            nm = float(np.linalg.norm(m.astype(np.float64)))
            c.append(mod.set_definitions("annulus", "identity", 0.9 * nm, 0.98 * nm, ("matrix", "")))
        elif k == "l1dft":                         # sigma = 0.5 ||F m||_1, F unitary
            z = np.abs(np.fft.fftn(m.reshape(n, order="F"), norm="ortho"))
            c.append(mod.set_definitions("l1", "DFT", 0.0, float(0.5 * z.astype(np.float64).sum()), ("matrix", "")))
        elif k.startswith("rank:"):
            c.append(mod.set_definitions("rank", "identity", 0, int(k[5:]), ("slice", "z") if len(n) == 3 else ("matrix", "")))
        elif k.startswith("card:"):                # keep 10% of the entries of D m
            rows = int(np.prod(n)) // n[-1] * (n[-1] - 1)
            c.append(mod.set_definitions("cardinality", k[5:], 0, int(0.1 * rows), ("matrix", "")))
    return g, c


def c5_model(n, TF, kind="survey", seed=20240601 + 5):
    """Models of the multilevel workload (BASELINE configs[4], SURVEY 8d "C5").
    "survey": SURVEY 8(d)'s generic model, m = 1500 + 2500 z + 150 N(0,1).  Its TV norm is all noise, which nearest-neighbour
    coarsening shrinks by 1/16 per level where constraint2coarse's rule assumes 1/8 (constraint2coarse.jl:44-48) -- with the
    radius 0.5 ||TV m||_1 both coarse problems are feasible on entry and the multilevel path does no work; C5 therefore uses
    0.1 ||TV m||_1 with this model (ratio 0.2 / 0.4 on the coarse levels: every level iterates).
    "layered": a velocity-model-like field of the normalised coordinates (tilted, undulating layers + 4 N(0,1)), the kind of
    model the reference times its multilevel scheme on (examples/test_scaling_3D.jl:78-84, the overthrust model): its TV norm
    scales like the 1/cf^3 rule assumes, so 0.5 ||TV m||_1 stays about half the norm on every level."""
    rng = np.random.default_rng(seed)
    if kind == "survey":
        z = np.linspace(0.0, 1.0, n[-1]).reshape((1,) * (len(n) - 1) + (-1,))
        return (1500.0 + 2500.0 * z + 150.0 * rng.standard_normal(n)).astype(TF).reshape(-1, order="F")
    if kind != "layered":
        raise ValueError(f"unknown model {kind!r}")
    ax = [np.linspace(0.0, 1.0, k) for k in n]
    X, Y, Z = np.meshgrid(*ax, indexing="ij", sparse=True)
    zeta = Z + 0.06 * np.sin(2 * np.pi * (1.5 * X + 0.5 * Y)) + 0.04 * np.cos(2 * np.pi * (2 * Y - X))
    m = 1500.0 + 2500.0 * zeta + 250.0 * np.sin(2 * np.pi * 4 * zeta) + 4.0 * rng.standard_normal(n)
    return m.astype(TF).reshape(-1, order="F")


C5_SIGMA = {"survey": 0.1, "layered": 0.5}


def c5_problem(mod, n, h, TF, model="survey", sigma=None):
    """(m, comp_grid, constraints) of C5: {bounds on I, l1 ball on TV}, radius sigma ||TV m||_1 (test_scaling_3D.jl:144-148
    runs PARSDMM_multi_level on whatever constraint list the serial run used; BASELINE names this pair)."""
    m = c5_model(n, TF, model)
    g = mod.compgrid(h, n)
    s = mod.get_TD_operator(g, "TV", TF)[0] @ m
    frac = C5_SIGMA[model] if sigma is None else float(sigma)
    c = [mod.set_definitions("bounds", "identity", 1600.0, 3900.0, ("tensor", "")),
         mod.set_definitions("l1", "TV", 0.0, float(frac * np.abs(np.asarray(s, np.float64)).sum()), ("tensor", ""))]
    return m, g, c


def run_c5(sipx, n, h=(25.0, 25.0, 25.0), TF=np.float64, maxit=100, model="survey", sigma=None, levels=3, cf=2, device=None,
           dist=None, host_transfers=False, single_level=True, keep_solution=False, settle=3.0):
    """BASELINE configs[4]: PARSDMM_multi_level (src/PARSDMM_multi_level.jl:8-89) timed as ONE call, the way the reference
    times it (examples/test_scaling_3D.jl:144-148: `@timed PARSDMM_multi_level(...)` after the set-up), default stop rules
    and evol_rel_tol = 10 eps (:25).  single_level: the same projection by one cold-started PARSDMM on the finest grid -- the
    iterations the coarse levels save are the only thing the multilevel scheme is for (:150-163)."""
    from sipx import multilevel as ML
    t0 = time.perf_counter()
    m, g, c = c5_problem(sipx, n, h, TF, model, sigma)
    opt = sipx.PARSDMM_options(FL=TF, maxit=maxit, evol_rel_tol=10 * float(np.finfo(TF).eps))
    L = ML.setup_multi_level_PARSDMM(m, levels, cf, g, c, opt)
    t1 = t_setup = time.perf_counter()
    # One rank: the call is made TWICE, each after a pause of `settle` seconds, and the faster one reported, both listed
    # (whole_solve_runs_s).  A whole call of this size depends on the state the box is in, not only on the code: the driver wipes
    # device memory when it is released, asynchronously, and an allocation that follows a large release waits for the wipe -- the
    # finest level's 67 GB take 0.04 s on a device at rest, 1.3-1.4 s right behind the release of a context of that size (five calls
    # back to back in one process: 1.61 / 2.36 / 1.03 after 4 s / 2.41 / 1.10 after 8 s; round 5) -- and in this bench the call
    # follows the 512^3 legs.  The reference times the call on a machine at rest (examples/test_scaling_3D.jl:144-148); so does this.
    runs, best = [], None
    for rep in range(2 if (dist is None and not host_transfers) else 1):
        if dist is None and settle > 0:
            import gc
            sipx.clear_context_cache()          # (both runs build their levels anew: multilevel.py keeps those of the last call)
            gc.collect()
            time.sleep(settle)
        Tr = {}
        ta = time.perf_counter()
        xr, logr, lr, yr = ML.PARSDMM_multi_level(m, *L[:5], opt, device=device, timings=Tr, host_transfers=host_transfers, dist=dist)
        tb = time.perf_counter()
        del lr, yr
        runs.append(tb - ta - (Tr.get("download_x_only_s") or 0.0))
        if best is None or runs[-1] < best[0]:
            best = (runs[-1], xr, logr, Tr, ta, tb)
        del xr
    _, x, log, T, t1, t2 = best
    del best
    repeated = None
    if dist is None and not host_transfers and settle > 0:
        # ... and once more at once, the way a caller's loop calls it: the level contexts of the last call are still there
        # (multilevel.py, _level_cache: sipx_reset per level instead of allocations that would wait for the driver's wipe)
        Tr = {}
        ta = time.perf_counter()
        xr, _, lr, yr = ML.PARSDMM_multi_level(m, *L[:5], opt, device=device, timings=Tr)
        tb = time.perf_counter()
        repeated = {"whole_solve_s": tb - ta - (Tr.get("download_x_only_s") or 0.0), "contexts_reused": bool(Tr.get("contexts_reused")),
                    "same_x": bool(np.array_equal(xr, x))}
        del xr, lr, yr
        sipx.clear_context_cache()
    fin = T["levels"][-1]
    out = {"workload": f"c5: PARSDMM_multi_level {'x'.join(map(str, n))} {'Float64' if TF == np.float64 else 'Float32'}, {levels} levels "
                       f"(coarsening {cf}), sets {{bounds, l1:TV}} + distance term, model '{model}', radius "
                       f"{C5_SIGMA[model] if sigma is None else sigma} ||TV m||_1, maxit {maxit} per level, default stop rules",
           "grid": list(n), "levels": [[int(v) for v in gg.n] for gg in L[4]],
           "transfers": "host (round-1 path)" if host_transfers else "device (sipx_warm_start_from)",
           # whole_solve_s: the call as the reference returns it, x and every l_i, y_i on the host (9 vectors of 512^3 Float64 for this
           # list: most of the difference to solve_only_s is their PCIe copy); whole_solve_x_only_s: what a caller that asks for x
           # alone waits for (the x-only copy is timed apart inside the call and taken out of whole_solve_s)
           "setup_s": t_setup - t0, "whole_solve_s": t2 - t1 - (T.get("download_x_only_s") or 0.0), "whole_solve_runs_s": runs,
           "pause_before_each_run_s": settle if dist is None else 0.0, "repeated_call": repeated,
           "whole_solve_x_only_s": (t2 - t1 - (T.get("download_s") or 0.0)) if T.get("download_x_only_s") is not None else None,
           "download_x_only_s": T.get("download_x_only_s"),
           "solve_only_s": sum(v["solve_s"] for v in T["levels"]), "warm_start_total_s": sum(v["warm_start_s"] for v in T["levels"]),
           "context_total_s": sum(v["context_s"] for v in T["levels"]), "download_s": T.get("download_s"),
           "per_level": T["levels"], "iterations_per_level": [v["iterations"] for v in T["levels"]],
           # what each level's context holds on this rank's GPU (N > 1: the rank's planes + halo planes, sparse arrays)
           "device_bytes_per_level": [v.get("device_bytes") for v in T["levels"]],
           "sparse_arrays_per_level": [v.get("sparse_arrays") for v in T["levels"]],
           "every_level_iterates": bool(all(v["iterations"] > 1 for v in T["levels"])),
           "finest_iterations": fin["iterations"], "finest_cg": fin["cg_iterations"],
           "finest_level_it_per_s": fin["iterations"] / fin["solve_s"],
           "finest_first_r_pri_total": float(log.r_pri_total[0]),
           "obj_last": float(log.obj[-1]), "feas_last": [float(v) for v in log.set_feasibility[-1]],
           "finite": bool(np.isfinite(x).all() and np.isfinite(log.obj).all())}
    if keep_solution:
        out["_x"], out["_log"] = x, log
    del x
    if single_level and dist is None:
        opt.zero_ini_guess = True
        t3 = time.perf_counter()
        ctx = sipx.host.build_context(m, L[1][0], L[0][0], L[3][0], L[2][0], g, opt, device=device)
        t4 = time.perf_counter()
        log1, _ = ctx.parsdmm(opt)
        t5 = time.perf_counter()
        ctx.close()
        out["single_level"] = {"iterations": int(len(log1.obj)), "cg_iterations": int(np.sum(log1.cg_it)), "context_s": t4 - t3,
                               "solve_s": t5 - t4, "it_per_s": len(log1.obj) / (t5 - t4), "first_r_pri_total": float(log1.r_pri_total[0]),
                               "obj_last": float(log1.obj[-1]), "feas_last": [float(v) for v in log1.set_feasibility[-1]]}
        out["finest_iterations_saved"] = int(len(log1.obj)) - int(fin["iterations"])
    return out


def whole_call(sipx, config, TF, calls=4):
    """PARSDMM(m, AtA, TD_OP, set_Prop, P_sub, comp_grid, options) -> (x, log, l, y) as a caller sees it: wall clock of the first call
    (context built: sipx_create ... sipx_finalize), of a second call with another model of the same kind (context reset: sipx_reset,
    everything returned; made twice, the faster one is second_call_s, both are listed) and of a last one that asks for x only, into
    the caller's own array (what a projector inside an outer loop does: examples/constrained_freq_FWI_simple.jl:468).  Default
    options: maxit 200, the stop rules decide."""
    n, h, kinds = CONFIGS[config]
    gs = sipx.compgrid(h, n)
    ms = [synthetic_model(n, TF, 20240601 + 3 + k) for k in range(calls)]

    def radius_of(opname):
        s = sipx.get_TD_operator(gs, opname, TF)[0] @ ms[0]
        return float(0.5 * np.abs(s.astype(np.float64)).sum())
    g, c = build_problem(sipx, n, h, kinds, ms[0], TF, radius_of)
    P, A, prop = sipx.setup_constraints(c, g, TF)
    opt = sipx.PARSDMM_options(FL=TF)
    A, AtA, _, _ = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
    sipx.clear_context_cache()
    xbuf = np.zeros_like(ms[0])
    rows = []
    # (single samples of 30-60 ms each: the interpreter's collector is kept out of them, as timeit does, and the reset call is made
    #  twice -- one of them has come back with 57 ms of "initialization" where 3.4 is the rule, right behind the release of the previous
    #  leg's context)
    import gc
    gc.collect()
    gc_on = gc.isenabled()
    gc.disable()
    for k in range(calls):
        kw = dict(x=xbuf, outputs="x") if k == calls - 1 else {}
        t0 = time.perf_counter()
        x, log, l, y = sipx.PARSDMM(ms[k], AtA, A, prop, P, g, opt, **kw)
        dt = time.perf_counter() - t0
        solve = float(sum(v for kk, v in log.timing.items() if kk != "initialization"))
        rows.append({"call": k + 1, "whole_call_s": dt, "initialization_s": float(log.timing["initialization"]), "solve_s": solve,
                     "download_and_rest_s": dt - solve - float(log.timing["initialization"]), "iterations": int(len(log.obj)),
                     "context_reused": bool(getattr(log, "context_reused", False)), "outputs": "x" if k == calls - 1 else "x, l, y",
                     "finite": bool(np.isfinite(x).all())})
        del x, l, y
    if gc_on:
        gc.enable()
    sipx.clear_context_cache()
    again = min(rows[1:calls - 1], key=lambda r: r["whole_call_s"])          # the faster of the calls on the reused context
    last = rows[-1]
    return {"workload": f"{config}: whole calls of PARSDMM(...), default options (stop rules active)", "calls": rows,
            "first_call_s": rows[0]["whole_call_s"], "second_call_s": again["whole_call_s"],
            "second_call_runs_s": [r["whole_call_s"] for r in rows[1:calls - 1]],
            "second_call_overhead": (again["whole_call_s"] - again["solve_s"]) / again["solve_s"],
            "x_only_call_s": last["whole_call_s"], "x_only_call_overhead": (last["whole_call_s"] - last["solve_s"]) / last["solve_s"]}


_LIB_SHA = None


def lib_sha16():
    """sha256[:16] of the libsipx.so this process runs: what a profiles/ summary must carry to be quoted beside a live number."""
    global _LIB_SHA
    if _LIB_SHA is None:
        import hashlib
        path = os.environ.get("SIPX_LIBRARY") or os.path.join(ROOT, "setintersectionprojection.jl_amd", "libsipx.so")
        _LIB_SHA = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    return _LIB_SHA


def dominant_kernel(per_kernel):
    """From sipx_kernel_stats_json over the all-kernel window: the kernel with the largest total time (chosen, not hard-coded)
    with its launches, average duration, algorithmic bytes and the fraction of the HBM peak they amount to; and the table of
    every kernel above 1 % of the kernel-time sum.  `inclusive` entries (library-backed projectors: their interval contains
    other listed kernels) are not candidates."""
    if not per_kernel or not per_kernel.get("kernels"):
        return None, None
    ks = [k for k in per_kernel["kernels"] if k["launches"]]
    tot = sum(k["total_ms"] for k in ks if not k["inclusive"]) or 1.0
    rows = []
    for k in sorted(ks, key=lambda k: -k["total_ms"]):
        avg_ms = k["total_ms"] / k["launches"]
        row = {"kernel": k["name"], "launches": int(k["launches"]), "avg_launch_ms": avg_ms, "share_of_kernel_time": k["total_ms"] / tot,
               "algorithmic_bytes_per_launch": k["bytes_moved"] / k["launches"],
               "frac": (k["bytes_moved"] / (k["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if k["total_ms"] > 0 else None}
        if k["bytes_survey"] != k["bytes_moved"]:
            row["survey_bytes_per_launch"] = k["bytes_survey"] / k["launches"]
            row["frac_survey"] = (k["bytes_survey"] / (k["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if k["total_ms"] > 0 else None
        if k["gated"]:
            row["noop_launches"] = int(k.get("noop_launches", 0))
            row["note"] = ("some launches return at once on a device-side condition (noop_launches: those that finished sooner than "
                           "their bytes could cross the fabric at twice the HBM peak): their bytes are not booked")
        if k["inclusive"]:
            row["note"] = "library-backed projector (hipFFT / rocSOLVER / rocBLAS calls): its interval contains other listed kernels; not a candidate"
        rows.append(row)
    dom = dict(next(r for r in rows if "library-backed" not in r.get("note", "")))
    dom.update({"bound": "hbm" if dom["algorithmic_bytes_per_launch"] > 0 else "latency (a scalar / one-workgroup step: no algorithmic bytes)",
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "achieved": (dom["frac"] or 0.0) * HBM_PEAK_GBS,
                "definition": "largest total time among the engine's kernels in the statistics window that follows the timed steps "
                              "(HIP events around every launch on the launch's own stream; with two set streams a kernel's interval "
                              "includes the time it waits for CUs the other stream holds); bytes = the kernel's algorithmic bytes "
                              "(DESIGN 3), summed over its launches / summed time / peak"})
    return dom, [r for r in rows if r["share_of_kernel_time"] >= 0.01]


LINE_LIMIT = 4096        # the driver keeps a few KB of stdout tail: the ONE line must fit with room to spare (round 3's grew to 24 KB and was cut)


def _sig(v, sig=5):
    """Numbers of the headline line carry five significant digits: enough for every figure it quotes, and a third of the bytes."""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float(f"{v:.{sig}g}")
    if isinstance(v, dict):
        return {k: _sig(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, sig) for x in v]
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def headline(out, detail_path=None):
    """The ONE stdout line: the contract's keys plus the handful of figures a review reads first; everything else (kernel tables,
    timing buckets, definitions, search counters, per-level C5 detail, comm probe names) lives in the detail file written beside it.
    Mirrors what the reference's own measurement is -- one wall clock per call (examples/test_scaling_3D.jl:116-117)."""
    ROOF = ("kernel", "bound", "peak", "achieved", "unit", "frac", "frac_survey", "traffic", "frac_traffic", "launches", "avg_launch_ms",
            "algorithmic_bytes_per_launch")
    DOM = ("kernel", "bound", "frac", "launches", "avg_launch_ms", "algorithmic_bytes_per_launch", "share_of_kernel_time")
    COMM = ("rccl_nranks", "rccl_version", "decomposition", "ranks_agree_on_x", "device_bytes_per_rank", "sparse_arrays", "comm_mode", "fell_back_from")
    h = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                    "dtype", "data"))
    h["config"] = _pick(out.get("config"), ("workload", "grid", "parallelism", "cg_iterations_in_timed_steps", "all_logs_finite"))
    if isinstance(h["config"].get("parallelism"), str):
        h["config"]["parallelism"] = h["config"]["parallelism"][:60]
    if out.get("roofline") is not None:
        h["roofline"] = _pick(out["roofline"], ROOF)
        h["roofline"]["kernel"] = str(h["roofline"].get("kernel", "")).split(" (")[0]
    else:
        h["roofline"] = None
    if out.get("dominant_kernel"):
        h["dominant_kernel"] = _pick(out["dominant_kernel"], DOM)
    if out.get("iteration_roofline"):
        h["iteration_roofline"] = _pick(out["iteration_roofline"], ("frac", "bytes_moved_per_step", "frac_survey"))
    if out.get("cpu_baseline"):
        h["cpu_baseline"] = _pick(out["cpu_baseline"], ("value", "unit", "cores", "kind", "sample"))
        if isinstance(h["cpu_baseline"].get("sample"), str):
            h["cpu_baseline"]["sample"] = h["cpu_baseline"]["sample"][:160]
    for key in ("c3_512", "c4_512"):
        v = out.get(key)
        if not isinstance(v, dict):
            continue
        if "error" in v:
            h[key] = {"error": str(v["error"])[:160]}
            continue
        o = _pick(v, ("value", "ms_per_step", "steps", "warmup"))
        if v.get("roofline"):
            o["roofline"] = _pick(v["roofline"], ("frac", "frac_survey", "avg_launch_ms"))
        if v.get("dominant_kernel"):
            o["dominant_kernel"] = _pick(v["dominant_kernel"], ("kernel", "frac", "avg_launch_ms", "share_of_kernel_time"))
        if v.get("iteration_roofline"):
            o["iteration_roofline"] = _pick(v["iteration_roofline"], ("frac",))
        if v.get("rank_route"):
            o["rank_route"] = _pick(v["rank_route"], ("calls", "warm_started_subspace", "full_decomposition"))
        if out.get("n_gpus", 1) > 1 and v.get("comm"):
            o["comm"] = _pick(v["comm"], ("decomposition", "ranks_agree_on_x", "device_bytes_per_rank"))
        h[key] = o
    v = out.get("c4_512_slab")
    if isinstance(v, dict):      # (N > 1: the same list with the WHOLE iteration on z-slabs, the rank set slab-local, DFT / cardinality gathered)
        h["c4_512_slab"] = ({"error": str(v["error"])[:160]} if "error" in v else
                            dict(_pick(v, ("value", "ms_per_step")), **_pick(v.get("comm") or {}, ("ranks_agree_on_x", "device_bytes_per_rank")),
                                 collectives_per_step=((v.get("comm") or {}).get("collectives_per_step") or {}).get("total")))
    v = out.get("c2_2048")
    if isinstance(v, dict):
        h["c2_2048"] = {"error": str(v["error"])[:160]} if "error" in v else _pick(v, ("value", "ms_per_step"))
    v = out.get("whole_call")
    if isinstance(v, dict):
        h["whole_call"] = ({"error": str(v["error"])[:160]} if "error" in v else
                           _pick(v, ("first_call_s", "second_call_s", "second_call_overhead", "x_only_call_s", "x_only_call_overhead")))
    for key in ("c5", "c5_layered"):
        v = out.get(key)
        if isinstance(v, dict):
            h[key] = ({"error": str(v["error"])[:160]} if "error" in v else
                      dict(_pick(v, ("whole_solve_s", "whole_solve_runs_s", "whole_solve_x_only_s", "solve_only_s", "iterations_per_level", "finest_level_it_per_s", "finest_iterations_saved", "finite")),
                           **({"finest_device_bytes": (v.get("device_bytes_per_level") or [None])[-1]} if out.get("n_gpus", 1) > 1 else {})))
    if out.get("comm"):
        h["comm"] = _pick(out["comm"], COMM)
    if out.get("decompositions"):
        h["decompositions"] = {k: ({"error": str(v["error"])[:120]} if "error" in v else
                                   dict(_pick(v, ("value", "ms_per_step")), ranks_agree_on_x=(v.get("comm") or {}).get("ranks_agree_on_x")))
                               for k, v in out["decompositions"].items()}
        h["decomposition"] = out.get("decomposition")
        h["faster_decomposition"] = out.get("faster_decomposition")
    if isinstance(out.get("comm_probe_us"), dict):        # medians only, short keys (the descriptions are in the detail file)
        h["comm_probe_us"] = {k.split(" ")[0]: v for k, v in out["comm_probe_us"].items() if isinstance(v, (int, float))}
    if out.get("headline_attempts") and any(not a.get("ok") for a in out["headline_attempts"]):
        h["headline_attempts"] = [{"attempt": a["attempt"][:48], "ok": a["ok"], **({"error": a["error"][:100]} if a.get("error") else {})}
                                  for a in out["headline_attempts"]]
    for k in ("dry_comm", "invalid_as_measurement", "libsipx_sha16", "error", "status", "legs_abandoned_at"):
        if k in out:
            h[k] = out[k]
    if detail_path:
        h["detail"] = os.path.basename(detail_path)
    line = json.dumps(_sig(h), separators=(",", ":"))
    if len(line) >= LINE_LIMIT:                           # never again a line the driver cannot parse: shed the optional objects
        for k in ("comm_probe_us", "c5_layered", "whole_call", "c4_512_slab", "iteration_roofline", "c2_2048", "decompositions", "c4_512", "c5", "c3_512", "dominant_kernel"):
            h.pop(k, None)
            line = json.dumps(_sig(h), separators=(",", ":"))
            if len(line) < LINE_LIMIT:
                break
    return line


def emit(out, detail_path):
    """Detail to the side file (and a copy under gpurun_out/ when that directory exists, so that it travels back from a GPU box)
    and to stderr; the compact headline LAST and alone on stdout."""
    paths = [detail_path] if detail_path else []
    scratch = os.path.join(ROOT, "gpurun_out")
    if detail_path and os.path.isdir(scratch) and os.path.dirname(os.path.abspath(detail_path)) != scratch:
        paths.append(os.path.join(scratch, os.path.basename(detail_path)))
    written = None
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(out, f)
            written = written or p
        except OSError as e:                              # a read-only tree must not take the headline with it
            print(f"bench: could not write {p}: {e}", file=sys.stderr)
    print("bench detail: " + json.dumps(out), file=sys.stderr, flush=True)
    line = headline(out, written)
    assert len(line) < LINE_LIMIT, len(line)
    print(line)
    sys.stdout.flush()


def bench_options(mod, TF, maxit):
    # tolerances at zero: the stop rules never fire, so exactly `steps` iterations are timed
    return mod.PARSDMM_options(FL=TF, maxit=maxit, evol_rel_tol=0.0, feas_tol=0.0, obj_tol=0.0)


def cpu_baseline(name, n, h, kinds, steps_budget_s=20.0):
    """Times the oracle's C/OpenMP port (structure of the reference CPU path) on this host."""
    try:
        from oracle import port
    except Exception as e:                                              # pragma: no cover
        return {"value": None, "unit": "it/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    return port.time_baseline(name, n, h, kinds, steps_budget_s)


# ---- progress marks and the watchdog: the first run on more than one GPU is also the first test of that path, and a rank
# stuck in a collective says nothing by itself.  Every rank notes what it is about to do (stderr with SIPX_BENCH_DEBUG=1, a
# per-rank file when the launcher gave it a directory); when the deadline passes, the rank prints its last mark and exits
# non-zero, which ends the other ranks through their launcher.
_PROGRESS = {"last": "start", "t0": time.time(), "file": None}


def progress(msg):
    line = f"[bench rank {os.environ.get('RANK', '0')} +{time.time() - _PROGRESS['t0']:.1f}s] {msg}"
    _PROGRESS["last"] = line
    if os.environ.get("SIPX_BENCH_DEBUG"):
        print(line, file=sys.stderr, flush=True)
    d = os.environ.get("SIPX_BENCH_PROGRESS_DIR")
    if d:
        try:
            with open(os.path.join(d, f"rank{os.environ.get('RANK', '0')}.log"), "a") as f:
                f.write(line + "\n")
        except OSError:
            pass


class _Watchdog:
    def __init__(self):
        self.timer = None

    def start(self, seconds):
        import threading

        def fire():
            print(f"bench watchdog: no bench line after {seconds:.0f} s; last mark: {_PROGRESS['last']}", file=sys.stderr, flush=True)
            cb = _PROGRESS.get("on_deadline")
            if cb is not None:                     # rank 0: a parseable line with what there is (the headline and the legs done so far,
                try:                               # or value 0 and the mark the run stopped at) -- never an empty stdout
                    cb(f"deadline of {seconds:.0f} s passed; last mark: {_PROGRESS['last']}")
                except Exception as e:             # pragma: no cover
                    print(f"bench watchdog: could not print a line: {e!r}", file=sys.stderr, flush=True)
            os._exit(3)
        if seconds and seconds > 0:
            self.timer = threading.Timer(seconds, fire)
            self.timer.daemon = True
            self.timer.start()

    def cancel(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


WATCHDOG = _Watchdog()


def spawn_ranks(n_gpus, deadline_s=0.0):
    """`python bench.py --gpus N` run plainly (no launcher): start N rank processes of this script, one per GPU, as fresh
    children -- this parent never touches the GPU -- with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as
    torch.distributed.run would.  Rank 0 inherits stdout (the ONE JSON line); the other ranks' stdout goes to stderr.
    Returns the worst exit code; when one rank fails the others are ended (by PID)."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    marks = tempfile.mkdtemp(prefix="sipx_bench_")
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SIPX_BENCH_CHILD="1", SIPX_BENCH_PROGRESS_DIR=marks)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))

    def last_marks():
        for r in range(n_gpus):
            try:
                with open(os.path.join(marks, f"rank{r}.log")) as f:
                    lines = f.read().strip().splitlines()
                print(f"  rank {r}: {lines[-1] if lines else '(no mark)'}", file=sys.stderr)
            except OSError:
                print(f"  rank {r}: (no mark)", file=sys.stderr)
    rc = 0
    pending = set(range(n_gpus))
    t0 = time.time()
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0:
                rc = rc or code
                if pending:
                    print(f"bench launcher: rank {r} exited with code {code}; ending ranks {sorted(pending)}.  Last marks:", file=sys.stderr)
                    last_marks()
                for q in pending:                      # a dead rank leaves the others stuck in a collective
                    procs[q].terminate()
        if pending and deadline_s and time.time() - t0 > deadline_s + 15.0:      # the ranks' own watchdogs should have fired by now
            print(f"bench launcher: no bench line after {time.time() - t0:.0f} s; ending ranks {sorted(pending)} by PID.  Last marks:",
                  file=sys.stderr)
            last_marks()
            for q in pending:
                procs[q].terminate()
            time.sleep(2.0)
            for q in pending:
                if procs[q].poll() is None:
                    procs[q].kill()
            rc = rc or 3
            break
        time.sleep(0.05)
    import shutil
    shutil.rmtree(marks, ignore_errors=True)
    return rc


def comm_probe(dist, torch, world, rank, dev=None, scale=1):
    """What the collectives of the two decompositions cost on THIS node: torch.distributed's communicator on the same devices
    (RCCL on GPUs; gloo on CPU tensors in a dry run), microseconds per call averaged over back-to-back calls, the slowest
    rank's figure.  A probe that cannot be set up on some rank is skipped on ALL ranks (the ranks agree on it through an
    all-reduce before anyone enters the probe's collective), so a local failure cannot leave the others waiting."""
    dev = dev if dev is not None else torch.device("cuda", torch.cuda.current_device())
    on_gpu = dev.type == "cuda"
    out = {}

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    def timed(name, setup, reps):
        fn, err = None, None
        try:
            fn = setup()
        except Exception as e:                               # e.g. an allocation failure on this rank only
            err = repr(e)
        ok = torch.tensor([0.0 if fn is None else 1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 1.0:
            out[name] = {"skipped": err or "set-up failed on another rank"}
            return
        for _ in range(3):
            fn()
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        sync()
        t = torch.tensor([(time.perf_counter() - t0) * 1e6 / reps], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out[name] = round(float(t.item()), 2)

    def all_reduce_of(n, dtype):
        def setup():
            t = torch.zeros(n, dtype=dtype, device=dev)
            return lambda: dist.all_reduce(t)
        return setup

    def gather_of(chunk, dtype):
        def setup():
            buf = torch.zeros(world * chunk, dtype=dtype, device=dev)
            parts = list(buf.split(chunk))
            if on_gpu:
                return lambda: dist.all_gather_into_tensor(buf, buf[rank * chunk:(rank + 1) * chunk])
            return lambda: dist.all_gather(parts, parts[rank].clone())
        return setup

    def reduce_scatter_of(chunk, dtype):
        def setup():
            buf = torch.zeros(world * chunk, dtype=dtype, device=dev)
            if on_gpu:
                return lambda: dist.reduce_scatter_tensor(buf[rank * chunk:(rank + 1) * chunk], buf)
            return lambda: dist.all_reduce(buf)                # gloo has no reduce-scatter (sharded.TorchComm does the same)
        return setup

    def halo_of(plane):
        def setup():
            sp, sn = torch.zeros(plane, device=dev), torch.zeros(plane, device=dev)
            rp, rn = torch.zeros(plane, device=dev), torch.zeros(plane, device=dev)

            def halo():
                ops = []
                if rank > 0:
                    ops += [dist.P2POp(dist.isend, sp, rank - 1), dist.P2POp(dist.irecv, rp, rank - 1)]
                if rank < world - 1:
                    ops += [dist.P2POp(dist.isend, sn, rank + 1), dist.P2POp(dist.irecv, rn, rank + 1)]
                if ops:
                    for w_ in dist.batch_isend_irecv(ops):
                        w_.wait()
            return halo
        return setup

    timed("allreduce_36_f64_us (probe sums of one set)", all_reduce_of(36, torch.float64), 100)
    timed("allreduce_2048_f64_us (dot partials of a CG step)", all_reduce_of(2048, torch.float64), 100)
    timed("allreduce_6153_f64_us (sampled histograms of three sets)", all_reduce_of(3 * 2051, torch.float64), 50)
    timed("allgather_197KB_per_rank_us (speculative exchange of three l1 searches: sums + gathered magnitudes)",
          gather_of(3 * (16384 + 48) // scale, torch.float32), 50)
    timed("allgather_1.5MB_per_rank_us (fallback: bracket segments of three l1 sets)", gather_of(3 * (131072 + 8) // scale, torch.float32), 50)
    for n1, tag in ((256 // scale, "256^3"), (512 // scale, "512^3")):
        timed(f"halo_plane_{tag}_f32_us (one plane to each neighbour)", halo_of(n1 * n1), 50)

        def both(n1=n1):          # what the engine issues as ONE RCCL group per CG iteration; here one after the other (an upper bound)
            ar, halo = all_reduce_of(2048, torch.float64)(), halo_of(n1 * n1)()
            return lambda: (ar(), halo())
        timed(f"allreduce_2048_f64_then_halo_plane_{tag}_us (ungrouped upper bound of the grouped CG call)", both, 50)
    n1 = 256 // scale
    chunk = -(-n1 // world) * n1 * n1
    timed("reduce_scatter_256^3_f32_us (rhs of the set decomposition)", reduce_scatter_of(chunk, torch.float32), 10)
    timed("allgather_256^3_f32_us (x of the set decomposition)", gather_of(chunk, torch.float32), 10)
    if scale != 1:
        out["sizes_divided_by"] = scale
    return out


# Collectives of ONE PARSDMM iteration of the headline set list (C3: p = 5 terms, three l1 searches) with k CG iterations,
# per decomposition, as the engine issues them -- what --dry-comm replays on dummy buffers.
SLAB_SMALL_COLLECTIVES_AT_2_CG = 7         # DESIGN 5 (round 2: 12, then 13 with the refinement rounds of the exchange segments)


def iteration_skeleton(decomp, k, world):
    """Per CG iteration: all-reduce (p.Ap), then ONE grouped call {all-reduce (||r||^2), boundary planes of r}; before the first,
    {all-reduce (||r_0||^2, ||rhs||^2), planes of r_0}.  Slab: the l1 searches of all sets through ONE all-gather (speculative
    exchange; the fallback with its all-reduces is not part of the skeleton), no exchange of x."""
    NBp = 2048
    cg = [("allreduce+halo", 2 * NBp)] + [("allreduce", NBp), ("allreduce+halo", NBp)] * k
    sums = ("allreduce", 6 * 16)
    if decomp == "sets":
        return [("reduce_scatter", "N")] + cg + [("allgather", "N"), sums]
    return cg + [("allgather", "fast_segments"), sums]


def dry_comm(args):
    """--dry-comm: see the flag's help.  One process per rank as in the real run; no libsipx.so, no oracle, no arithmetic."""
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    load_package()
    from sipx import sharded
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    on_gpu = torch.cuda.device_count() >= world and not os.environ.get("SIPX_DRY_COMM_CPU")
    if world == 1 and "MASTER_ADDR" not in os.environ:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)                                   # whatever the communicator prints while it comes up goes to stderr
    progress("dry-comm: rendezvous")
    if on_gpu:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", device_id=dev)
    else:
        import datetime
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dev = torch.device("cpu")
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=max(30.0, args.deadline or 120.0)))
    comm = sharded.TorchComm(dist, dev)
    if os.environ.get("SIPX_BENCH_TEST_HANG") == str(rank):      # tests: a rank that never reaches its collectives
        progress("test hook: this rank stalls here")
        time.sleep(3600)
    scale = 1 if on_gpu else 8
    n1 = 256 // scale
    plane = n1 * n1
    chunk = -(-n1 // world) * plane
    bufs = {"N": torch.zeros(world * chunk, dtype=torch.float32, device=dev),
            "fast_segments": torch.zeros(world * 3 * ((16384 // scale) + 48), dtype=torch.float32, device=dev)}
    halo = [torch.zeros(plane, dtype=torch.float32, device=dev) for _ in range(4)]
    small = {}

    def run(op, arg):
        if op in ("allreduce", "allreduce+halo"):
            t = small.setdefault(arg, torch.zeros(arg, dtype=torch.float64, device=dev))
            comm.allreduce_sum_(t)
            if op == "allreduce+halo":          # (one ncclGroup in the engine)
                comm.halo_exchange(halo[0], halo[1], rank - 1 if rank > 0 else -1, halo[2], halo[3], rank + 1 if rank < world - 1 else -1)
        elif op == "reduce_scatter":
            comm.reduce_scatter_sum_(bufs[arg], bufs[arg].numel() // world)
        elif op == "allgather":
            comm.allgather_(bufs[arg], bufs[arg].numel() // world)
        else:
            comm.halo_exchange(halo[0], halo[1], rank - 1 if rank > 0 else -1, halo[2], halo[3], rank + 1 if rank < world - 1 else -1)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()
    res = {}
    for decomp in ("slab", "sets"):
        progress(f"dry-comm: {decomp} skeleton, {args.warmup} warm-up + {args.steps} timed steps")
        sk = iteration_skeleton(decomp, 2, world)
        for _ in range(args.warmup):
            for op, arg in sk:
                run(op, arg)
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            for op, arg in sk:
                run(op, arg)
        sync()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        res[decomp] = {"value": args.steps / dt, "ms_per_step": dt / args.steps * 1e3, "collectives_per_step": len(sk),
                       "small_collectives_per_step": sum(1 for op, a in sk if a not in ("N",)),
                       "comm": {"rccl_nranks": dist.get_world_size(), "rccl_rank": rank,
                                "rccl_version": ("torch.distributed nccl " + ".".join(map(str, torch.cuda.nccl.version()))) if on_gpu else "torch.distributed gloo",
                                "decomposition": decomp}}
    progress("dry-comm: comm probe")
    probe = comm_probe(dist, torch, world, rank, dev, scale)
    sync()
    sys.stdout.flush()
    os.dup2(saved, 1)
    os.close(saved)
    out = {"metric": "PARSDMM iterations/sec", "value": res["slab"]["value"], "unit": "it/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": res["slab"]["ms_per_step"], "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "none: --dry-comm replays the collective skeleton of the iteration on dummy buffers",
           "dry_comm": True, "invalid_as_measurement": True,
           "config": {"workload": f"dry-comm: collectives of one c3 iteration (2 CG iterations) at {n1}^3, backend {'nccl' if on_gpu else 'gloo'}; "
                                  "NO PARSDMM arithmetic -- a rehearsal of launcher, rendezvous and stdout contract, and the latency floor of "
                                  "each decomposition on this node"},
           "roofline": None, "decomposition": "slab", "decompositions": res,
           "faster_decomposition": max(res, key=lambda k: res[k]["value"]), "comm": res["slab"]["comm"], "comm_probe_us": probe}
    if rank == 0:
        emit(out, args.detail)
    progress("done")
    WATCHDOG.cancel()
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-512", action="store_true", help="skip the short 512^3 leg that follows the default 256^3 headline run")
    ap.add_argument("--no-c4", action="store_true", help="skip the short leg on BASELINE config 4 (512^3, eight sets) that follows the default run")
    ap.add_argument("--no-kernel-table", action="store_true",
                    help="skip the all-kernel statistics window that follows the timed steps (profiling runs: its event records would show up as gaps)")
    ap.add_argument("--no-c2", action="store_true", help="skip the leg on BASELINE config 2 (2048^2, {bounds, l1:TV})")
    ap.add_argument("--no-whole-call", action="store_true", help="skip the whole-call leg (first and later calls of PARSDMM(...) with default options)")
    ap.add_argument("--no-c5", action="store_true", help="skip the leg on BASELINE config 5 (PARSDMM_multi_level, 512^3 Float64, 3 levels) that follows the default run")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"],
                    help="f32 = the contract workload; f64 = the same sets in Float64 (BASELINE config 5 computes in Float64)")
    ap.add_argument("--decomp", default="auto", choices=["auto", "sets", "slab"],
                    help="with more than one rank: sets = the reference's split by constraint set (x-step on z-slabs); slab = the WHOLE "
                         "iteration on z-slabs (every rank holds every set; no N-vector crosses the fabric); auto = slab where the sets allow it")
    ap.add_argument("--q-mode", default="cds", choices=["cds", "stencil"],
                    help="cds = the reference's banded Q (the contract workload); stencil = generated coefficients (SURVEY 8f-2)")
    ap.add_argument("--deadline", type=float, default=float(os.environ.get("SIPX_BENCH_DEADLINE_S", "560")),
                    help="seconds after which a rank that has not finished prints its last progress mark and exits with code 3 (0 = never)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="file for everything that is not the headline: kernel tables, timing buckets, definitions, counters ('' = none)")
    ap.add_argument("--dry-comm", action="store_true",
                    help="rehearsal of the N > 1 flow WITHOUT the engine: launcher, rendezvous, stdout contract, both decompositions and the "
                         "comm probe, each step being the collective skeleton of one PARSDMM iteration (DESIGN 5) on dummy buffers; "
                         "gloo on CPU tensors when no GPU is visible.  What it prints is NOT a measurement of the metric")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # plain invocation: become the launcher (before any GPU call)
        raise SystemExit(spawn_ranks(args.gpus, args.deadline))
    WATCHDOG.start(args.deadline)
    if args.dry_comm:
        raise SystemExit(dry_comm(args))

    import torch
    from __graft_entry__ import load_package
    sipx = load_package()
    from sipx import sharded  # noqa: E402  (registered by load_package)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or run `python bench.py --gpus N` "
                         "without a launcher and let it spawn its own ranks)")
    dist = None
    # SIPX_BENCH_SHARE_GPU=1: a REHEARSAL of the N > 1 flow on a box with one GPU -- every rank runs the real engine on device 0,
    # the engine's collectives go through torch.distributed's gloo group as callbacks (RCCL refuses two ranks on one device).
    # Every world > 1 branch of this script, every leg and both decompositions run as they will on N GPUs; the numbers mean
    # nothing (N ranks share one card) and the line says so (`invalid_as_measurement`).
    share_gpu = bool(os.environ.get("SIPX_BENCH_SHARE_GPU")) and world > 1
    if share_gpu:
        local_rank = 0
    red_dev = "cpu" if share_gpu else "cuda"                  # where the script's own small all-reduces live
    torch.cuda.set_device(local_rank)
    force_dist = bool(os.environ.get("SIPX_FORCE_DIST"))      # exercise the RCCL path even with one rank
    saved_stdout = None
    if world > 1 or force_dist:
        # RCCL prints a version banner on the C-level stdout when its first communicator comes up; the contract is ONE
        # JSON line on stdout, so fd 1 points at stderr until the warm-up (first collectives) is over
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        if force_dist and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if share_gpu:
            import datetime
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=max(60.0, args.deadline or 600.0)))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    sipx.set_default_device(local_rank)
    TF = np.float32 if args.dtype == "f32" else np.float64
    w = np.dtype(TF).itemsize

    def restore_stdout():
        nonlocal saved_stdout
        if saved_stdout is not None:
            torch.cuda.synchronize()
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
            saved_stdout = None

    def measure(config, steps, warmup, decomp=None, comm_mode=None, slab_full=False, mid_hook=None):
        """One workload: builds the context, runs `warmup` untimed and `steps` timed PARSDMM iterations of the native loop
        (sipx_parsdmm_begin / _steps; sharded: the same loop with the engine's collectives inside), returns the numbers.
        comm_mode: "rccl" (native, inside the engine) or "torch" (torch.distributed collectives as callbacks); slab_full: whole
        arrays on every rank of a slab-decomposed context instead of the rank's planes (the headline's fallbacks)."""
        n, h, kinds = CONFIGS[config]
        N = int(np.prod(n))
        m = synthetic_model(n, TF, 20240601 + 3)
        gs = sipx.compgrid(h, n)

        def radius_of(opname):                      # sigma = 0.5 ||A m||_1 (the reference tests' own rule)
            s = sipx.get_TD_operator(gs, opname, TF)[0] @ m
            return float(0.5 * np.abs(s.astype(np.float64)).sum())

        g, c = build_problem(sipx, n, h, kinds, m, TF, radius_of)
        P, A, prop = sipx.setup_constraints(c, g, TF)
        stat_steps = min(steps, 10)                   # iterations of the all-kernel statistics window that follows the timed region
        maxit = warmup + steps + stat_steps + 1
        opt = bench_options(sipx, TF, maxit)
        opt.Q_mode = args.q_mode
        A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
        p = len(A)
        owned = sharded.shard_sets(p, world, rank)
        keep = []
        attach = None
        want = decomp or args.decomp
        slab = (dist is not None and want != "sets" and args.q_mode == "cds" and
                (sharded.slab_admissible(P, A) if want == "slab" else sharded.slab_decomposable(P, A)))
        if want == "slab" and dist is not None and not slab:
            raise SystemExit(f"--decomp slab: the sets of {config} cannot be decomposed by slab")
        if dist is not None:
            def attach(cx):
                keep.append(sharded.attach_comm(cx, dist, torch.device("cuda", local_rank), mode=comm_mode))
                if slab:
                    cx.set_decomp("slab_full" if slab_full else "slab")
        ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt, device=local_rank, owned=owned, attach=attach)
        ctx.parsdmm_begin(opt)
        logs = ctx._run[2]
        for _ in range(warmup):
            ctx.parsdmm_steps(1)
        if mid_hook is not None:
            mid_hook()
        restore_stdout()
        ctx.kernel_stats(True)

        def collective_counts():            # the engine's own counters of the calls it made on its communicator (read-only)
            if dist is None:
                return None
            try:
                return ctx.kernel_stats_all(-1).get("collectives")
            except Exception:
                return None
        coll0 = collective_counts()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps_asked = steps
        if os.environ.get("SIPX_BENCH_DEBUG"):
            ended = False
            for k in range(steps):
                ended = ctx.parsdmm_steps(1)
                i = warmup + k
                print(i + 1, "cg", logs["cg_it"][i], "obj %.4e" % logs["obj"][i], "rpri", logs["r_pri"][i], "rho", logs["rho"][i],
                      file=sys.stderr, flush=True)
                if ended:
                    break
        else:
            # ONE call for the K timed steps: the native loop runs them back to back, as sipx_parsdmm does in a real solve (a
            # call per step from Python put 20-30 us of interpreter time between two iterations)
            ended = ctx.parsdmm_steps(steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        coll1 = collective_counts()
        coll_per_step = None
        if coll0 and coll1:
            coll_per_step = {k: round((coll1[k] - coll0.get(k, 0)) / max(steps, 1), 2) for k in coll1}
            coll_per_step["total"] = round(sum(coll_per_step.values()), 2)
        # the product of the CG iteration over the timed steps; then EVERY kernel over a window of its own (two event records
        # around each launch cost about 5 us: not something to have inside the timed region)
        want_table = not ended and not args.no_kernel_table
        launches, kms = ctx.kernel_stats(2 if want_table else 0)
        per_kernel = None
        moved = None
        if want_table:
            torch.cuda.synchronize()
            ts0 = time.perf_counter()
            n_stat = 0
            for _ in range(stat_steps):
                n_stat += 1
                if ctx.parsdmm_steps(1):
                    break
            torch.cuda.synchronize()
            dts = time.perf_counter() - ts0
            per_kernel = ctx.kernel_stats_all(0)
            tot_moved = sum(k["bytes_moved"] for k in per_kernel["kernels"] if not k["inclusive"])
            moved = {"bytes_per_step": tot_moved / max(n_stat, 1), "steps": n_stat, "ms_per_step": dts / max(n_stat, 1) * 1e3,
                     "achieved": tot_moved / dts / 1e9, "frac": tot_moved / dts / 1e9 / HBM_PEAK_GBS,
                     "definition": "sum over the engine's kernels of their algorithmic bytes (what each has to move at least) in the "
                                   "all-kernel statistics window / wall time of that window (the event records of the window cost "
                                   "a few percent) / peak: the bandwidth the iteration as BUILT sustains"}
        comm_info = ctx.comm_info()
        try:
            counters = ctx.kernel_stats_all(-1)
        except Exception:
            counters = {}
        try:
            dev_bytes = ctx.device_bytes()
        except AttributeError:                               # (an older build of the library under SIPX_LIBRARY: A/B runs)
            dev_bytes = {"context": None, "device_used": None}
        # slab-decomposed: threshold searches that went through the speculative exchange since the context was built, how many of
        # them needed their fallback, and the all-reduces (refinement rounds) those took -- the engine's own counters
        searches = per_kernel.get("slab_searches") if per_kernel else None
        if searches is None and dist is not None:
            try:
                searches = ctx.kernel_stats_all(-1).get("slab_searches")
            except Exception:
                searches = None
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        log = ctx.parsdmm_log()
        if ended and len(log.obj) < warmup + steps_asked + (stat_steps if want_table else 0):
            # stop rules 3 / 4 of stop_PARSDMM do not depend on the tolerances: a very long run may end by itself; what was
            # executed inside the timed region is what is reported
            steps = max(1, min(steps_asked, len(log.obj) - warmup))
        if steps != steps_asked:
            print(f"bench: the solve stopped by itself after {warmup + steps} iterations; {steps} of the {steps_asked} requested steps were timed",
                  file=sys.stderr)
        cg_its = int(np.asarray(log.cg_it)[warmup:warmup + steps].sum())
        row0, row1, _ = ctx.slab()
        d = len(np.unique(np.concatenate([np.asarray(o) for o in prop.AtA_offsets])))
        # B_spmv = (d+2) N w per launch -- over the rows this rank's product covers (all N on one GPU)
        rows_here = row1 - row0
        spmv_bytes = (d + 2) * rows_here * w
        sym_bytes = ((d + 1) // 2 + 2) * rows_here * w            # what the kernel has to move: the bands with offset >= 0, x, y
        if args.q_mode == "stencil":
            spmv_bytes = sym_bytes = 2 * rows_here * w            # reads p, writes Ap; coefficients are generated
        elif os.environ.get("SIPX_CDS_FULL"):
            sym_bytes = spmv_bytes
        avg_ms = (kms / launches) if launches else None
        achieved = (spmv_bytes / (avg_ms * 1e-3) / 1e9) if launches else 0.0
        # HBM traffic of the dominant kernel from the PMC counters: separate rocprofv3 --pmc passes, summarised in profiles/
        # The PMC counters cannot be read inside this process (separate rocprofv3 --pmc passes): `traffic` is carried only when
        # profiles/ holds a summary taken on THIS build of libsipx.so (its hash is recorded by tools/summarize_pmc.py), else null
        traffic, traffic_src = None, None
        if args.q_mode == "cds" and args.dtype == "f32" and world == 1:
            tag = 'c3_256' if config == 'c3' else config.replace('-', '_')
            pmc = next((q for q in (os.path.join(ROOT, "profiles", f"r{rr:02d}_{tag}_pmc.json") for rr in (6, 5, 4)) if os.path.exists(q)), "")
            try:                                             # (a profile file must never take the bench line with it)
                rec = json.load(open(pmc)) if pmc else {}
                if rec.get("libsipx_sha16") == lib_sha16() and "dominant_kernel" in rec:
                    traffic = rec["dominant_kernel"]["hbm_bytes_per_launch_corrected"]
                    traffic_src = (f"profiles/{os.path.basename(pmc)}: separate rocprofv3 --pmc passes on this build (libsipx.so sha256[:16] "
                                   f"{rec['libsipx_sha16']}), 2*FETCH_SIZE + WRITE_SIZE per launch (gfx950 correction); not measured in this run")
            except Exception:
                traffic, traffic_src = None, None
        finite = bool(np.isfinite(log.obj).all() and np.isfinite(log.r_pri_total).all())
        # Whole-iteration roofline with SURVEY 8(d)'s algorithmic bytes: B_rhs + B_resid0 + k B_cg_iter + B_yl + B_log
        # (+ B_adapt + B_Q when rho / gamma are re-adapted, + B_feas every 10th iteration), summed over the timed steps.
        rows = [int(op.shape[0]) for op in A]
        d_i = [len(np.asarray(o)) for o in prop.AtA_offsets]
        pp = len(P)
        i0 = warmup + 1
        it_bytes = 0.0
        for i in range(i0, i0 + steps):                       # 1-based PARSDMM iteration numbers of the timed steps
            k = int(log.cg_it[i - 1])
            b = (sum(2 * r for r in rows) + N) + (d + 4) * N + k * ((d + 2) * N + 9 * N) + sum(N + 7 * r for r in rows) + 3 * N
            if i % int(opt.rho_update_frequency) == 0:
                b += sum(13 * r for r in rows)
            if i < len(log.rho):
                changed = np.nonzero(np.asarray(log.rho[i]) != np.asarray(log.rho[i - 1]))[0]
                b += sum(3 * d_i[j] * N for j in changed)
            if i % 10 == 0:
                b += sum(2 * r for r in rows[:pp])
            it_bytes += b * w
        it_gbs = it_bytes / dt / 1e9 / max(world, 1)                # per GPU
        # Sharded: do the ranks hold the same x?  Every rank completes x (a collective in the slab decomposition), takes an exact
        # integer checksum of its bits and the ranks compare -- the first thing to look at in a first multi-GPU run.
        agree = None
        if dist is not None:
            try:
                xs, _, _ = ctx.download(want_ly=False)
                bits = xs.view(np.uint32 if xs.dtype == np.float32 else np.uint64)
                chk = float(int(bits.astype(np.uint64).sum() % (1 << 52)))
                tt = torch.tensor([chk, -chk], dtype=torch.float64, device=red_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                agree = bool(float(tt[0].item()) == -float(tt[1].item())) and bool(np.isfinite(xs).all())
            except Exception as e:                                  # (diagnostics only)
                agree = f"check failed: {e!r}"[:200]
        ctx.close()
        frac_moved = (sym_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if launches else None
        frac_traffic = (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (launches and traffic) else None
        dominant, table = dominant_kernel(per_kernel)
        return {
            "comm": {"rccl_nranks": comm_info["nranks"], "rccl_rank": comm_info["rank"], "rccl_version": comm_info["version"],
                     "decomposition": comm_info["decomposition"] if dist is not None else None,
                     "slab_searches": searches if (dist is not None and slab) else None,
                     "ranks_agree_on_x": agree,
                     # the rank's planes only (hipMemMap-backed sparse arrays) or whole arrays; the verdict of sipx_finalize's
                     # communicator self-test; which communicator carried the collectives
                     "sparse_arrays": counters.get("sparse_arrays"), "comm_selftest": counters.get("comm_selftest"),
                     # calls the engine made on its communicator per timed step, by kind (scatter_gather: the fan exchanges of the
                     # gathered sets of a slab-decomposed long list); which sets of such a list are slab-local / gathered
                     "collectives_per_step": coll_per_step, "slab_loose": counters.get("slab_loose"),
                     # the set updated on a stream and host thread of its own beside the rest (C4's slice-rank set; -1: none)
                     "lane_set": counters.get("lane_set"),
                     "comm_mode": (comm_mode or os.environ.get("SIPX_COMM") or ("rccl" if dist.get_backend() == "nccl" else "torch")) if dist is not None else None,
                     # what this context allocated on its GPU (slab-decomposed: the rank's planes + halo planes only)
                     "device_bytes_per_rank": dev_bytes["context"], "device_used_bytes": dev_bytes["device_used"]},
            "dominant_kernel": dominant, "kernels": table,
            # one rank: threshold searches through the batched chain since the context was built, and how many needed their fallback sweeps
            "batched_searches": (per_kernel.get("batched_searches") if per_kernel else None),
            # a slice-rank set in the list: which route its projector took since the context was built (engine counters: calls,
            # calls served by the warm-started filtered subspace iteration, full decompositions, products with the Gram matrices)
            "rank_route": (per_kernel.get("rank_route") if per_kernel and (per_kernel.get("rank_route") or {}).get("calls") else None),
            "value": steps / dt, "ms_per_step": dt / steps * 1e3, "decomposition": ("slab" if slab else "sets") if dist is not None else None,
            # log.timing of the whole run (warm-up included), per iteration: where the time of an iteration goes on this rank
            "timing_ms_per_iteration": {k: round(float(v) * 1e3 / max(len(log.obj), 1), 4) for k, v in (log.timing or {}).items()},
            "config": {"workload": f"{config}: {'x'.join(map(str, n))} {'Float32' if args.dtype == 'f32' else 'Float64'}, sets {{{', '.join(kinds)}}} + distance term",
                       "grid": list(n), "sets": kinds, "q_mode": args.q_mode,
                       "parallelism": ((f"whole iteration on z-slabs over {world} ranks (every rank holds every set): slab CG (halo plane per "
                                        "product, all-reduced dot partials), one halo plane of x per neighbour, threshold searches with all-reduced "
                                        "probe sums and all-gathered bracket, one all-reduce of the per-set sums; no N-vector exchange") if slab else
                                       (f"sets sharded over {world} ranks (set i on rank i mod {world}); x-step on z-slabs: reduce-scatter(rhs) -> "
                                        "slab CG (halo plane per product, all-reduced dot partials) -> all-gather(x)")) if world > 1 else "single GPU",
                       "cg_iterations_in_timed_steps": cg_its, "all_logs_finite": finite,
                       "driver": "native loop (sipx_parsdmm_begin/_steps)" + (", collectives inside the engine (RCCL)" if dist is not None else "")},
            "roofline": {"bound": "hbm", "kernel": "k_cds<MODE=1> (cds_spmv + p.Ap partials)" if args.q_mode == "cds" else
                         "k_sq<MODE=1> (stencil Q product + p.Ap partials)",
                         "algorithmic_bytes_definition": ("SURVEY 8(d): B_spmv = (d+2) N w (d bands + x read, y written)" +
                                                          ("" if world == 1 else " over the rows of this rank's slab"))
                         if args.q_mode == "cds" else "2 N w (p read, Ap written)",
                         # `achieved` / `frac`: the bytes the kernel HAS TO MOVE -- the (d+1)/2 bands with a non-negative offset
                         # (symmetric-partner read), the vector, the result -- over the measured launch time.  SURVEY 8(d)'s count,
                         # (d+2) N w, which also prices the bands the kernel never reads, is carried as `frac_survey`.
                         "achieved": (frac_moved or 0.0) * HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac_moved or 0.0,
                         "frac_survey": achieved / HBM_PEAK_GBS, "achieved_survey": achieved,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "frac_traffic": frac_traffic,                  # PMC bytes / time / peak (only with a profile of this build)
                         "frac_bytes_moved": frac_moved,
                         "bands_from_hbm": (int((d + 1) // 2) if not os.environ.get("SIPX_CDS_FULL") else int(d)) if args.q_mode == "cds" else 0,
                         "bytes_with_symmetric_band_read": int(sym_bytes) if args.q_mode == "cds" else None,
                         "launches": int(launches), "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": spmv_bytes},
            # `frac`: the bytes the engine's own kernels have to move (the fused iteration as built) -- an HBM-utilisation figure.
            # `frac_survey`: SURVEY 8(d)'s B_iter, the byte count of the reference's UNFUSED pass structure, over the measured time:
            # how much faster the iteration is than the reference's passes would be at peak; it may exceed 1 and is not a utilisation.
            "iteration_roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "achieved": None if moved is None else moved["achieved"], "frac": None if moved is None else moved["frac"],
                                   "bytes_moved_per_step": None if moved is None else moved["bytes_per_step"],
                                   "window": None if moved is None else {k: moved[k] for k in ("steps", "ms_per_step", "definition")},
                                   "achieved_survey": it_gbs, "frac_survey": it_gbs / HBM_PEAK_GBS, "survey_bytes_per_step": it_bytes / steps,
                                   "definition_survey": "SURVEY 8(d) B_iter (unfused passes of the reference) summed over the timed steps / wall time" +
                                                        ("" if world == 1 else " / n_gpus")},
        }

    def leg(r, steps, warmup, full=True):
        o = {"value": r["value"], "unit": "it/s", "ms_per_step": r["ms_per_step"], "steps": steps, "warmup": warmup, "n_gpus": world,
             "scaling": "strong", "config": r["config"], "comm": r["comm"], "dominant_kernel": r["dominant_kernel"], "kernels": r["kernels"],
             "timing_ms_per_iteration": r["timing_ms_per_iteration"]}
        if full:
            o.update({"roofline": r["roofline"], "iteration_roofline": r["iteration_roofline"]})
        if r.get("rank_route"):
            o["rank_route"] = r["rank_route"]
        if r.get("batched_searches"):
            o["batched_searches"] = r["batched_searches"]
        return o

    # ---- secondary legs that cannot take the headline with them ----------------------------------------------------------
    # One rank, or an engine error that every rank raises alike (a threshold search that does not fit its exchange segments on
    # some node): the error is recorded under the leg's key and the run goes on.  A failure on ONE rank only (round 4's rehearsal
    # with four ranks on one GPU: the fourth ran out of device memory in the c5 leg) used to send that rank on into the next
    # leg's collectives while the others waited in this one's -- a mismatch, an abort, no bench line at all.  Now: the ranks that
    # fail tell the others through the process group's store and wait for them; if not every rank arrives within AGREE_S the
    # failure is rank-local -- rank 0 prints the line it has (headline + the legs done so far) and exits 0, the others park.  A rank
    # still waiting in the leg's collectives is released by the leg's own deadline the same way.
    partial = {"out": None}
    LEG_KEY = {"comm probe": "comm_probe_us", "sets-decomposed headline": "decompositions"}
    leg_deadline_env = float(os.environ.get("SIPX_BENCH_LEG_DEADLINE", "0") or 0)        # (tests)
    agree_s = float(os.environ.get("SIPX_BENCH_AGREE_S", "20") or 20)
    store = None
    if dist is not None and world > 1:
        try:
            from torch.distributed import distributed_c10d as _c10d
            store = _c10d._get_default_store()
        except Exception:
            store = None

    def give_up(name, why):
        if rank == 0 and partial["out"] is not None:
            o = partial["out"]
            err = {"error": str(why)[:600], "rank_local_failure": True}
            if name == "sets-decomposed headline":
                o["decompositions"] = {"sets": err}
            else:
                o[LEG_KEY.get(name, name)] = err
            o["legs_abandoned_at"] = name
            o["status"] = "incomplete"                  # (exit code 0 so that the line counts; the line itself says what is missing)
            progress(f"{name}: giving the remaining legs up, printing the line as it stands")
            try:
                if saved_stdout is not None:
                    os.dup2(saved_stdout, 1)
            except OSError:
                pass
            emit(o, args.detail)
        else:
            progress(f"{name}: giving the remaining legs up")
        os._exit(0 if (rank != 0 or partial["out"] is not None) else 3)

    def failed_alike(name):
        if store is None:
            return False
        import datetime
        key = f"sipx_bench/fail/{name}/"
        try:
            store.set(key + str(rank), "1")
            store.wait([key + str(r) for r in range(world)], datetime.timedelta(seconds=agree_s))
            return True
        except Exception:
            return False

    def safe(name, fn, deadline=300.0):
        timer = None
        dl = leg_deadline_env or deadline
        if world > 1:
            import threading
            timer = threading.Timer(dl, lambda: give_up(name, f"no result after {dl:.0f} s: a failure on one rank leaves the others in the leg's collectives"))
            timer.daemon = True
            timer.start()
        try:
            if os.environ.get("SIPX_BENCH_FAIL_LEG") == f"{name}:{rank}":        # tests: a failure on this rank only
                raise RuntimeError("test hook: this rank fails this leg")
            return fn()
        except Exception as e:
            import gc
            progress(f"{name} leg failed: {e!r}")
            if world > 1 and not failed_alike(name):
                if timer is not None:
                    timer.cancel()
                if rank == 0:
                    give_up(name, repr(e))
                progress(f"{name}: failed on this rank only; parked until the others give the leg up")
                time.sleep(dl + agree_s + 5.0)
                os._exit(0)
            gc.collect()                                        # contexts of the failed leg give their device memory back
            return {"error": repr(e)[:600]}
        finally:
            if timer is not None:
                timer.cancel()

    # ---- the headline (round 5): it cannot come back empty either -------------------------------------------------------------
    # One rank: one attempt, an error is the run's error.  More ranks: a chain of attempts, each safer than the one before --
    #   1. the decomposition asked for (auto: z-slabs, the rank's planes only in hipMemMap-backed arrays), RCCL inside the engine;
    #   2. the same with whole arrays on every rank (SIPX_DECOMP_SLAB_FULL: no mapped memory anywhere near RCCL);
    #   3. the same through torch.distributed's collectives as callbacks (another implementation of every operation: RcclComm's
    #      in-place offsets and grouped calls have never run on more than one GPU here);
    #   4. the reference's own split by constraint set, callbacks.
    # Before an attempt the ranks tell each other through the process group's store that they are ready (a rank whose set-up
    # failed says so and EVERY rank moves on to the next attempt: no collective has been entered yet); sipx_finalize makes a
    # rank-local allocation failure an error on every rank (engine.cpp); after an attempt every rank posts its outcome and waits
    # for the others -- all fine: done; an error somewhere: next attempt; a rank that never answers (it sits in a collective the
    # failed rank left): the line is printed with what there is.  `headline_attempts` and `comm.fell_back_from` say what happened.
    n, h, kinds = CONFIGS[args.config]
    attempts_log = []
    hl_deadline = float(os.environ.get("SIPX_BENCH_HEADLINE_DEADLINE", "0") or 0) or 240.0

    def post_and_wait(key, mine, timeout_s):
        """every rank posts `mine` under key/<rank>; returns {rank: text} of the ranks that answered within timeout_s"""
        import datetime
        got = {}
        if store is None:
            return {rank: mine}
        try:
            store.set(f"{key}/{rank}", mine)
        except Exception:
            return {rank: mine}
        t_end = time.time() + timeout_s
        for rr in range(world):
            try:
                store.wait([f"{key}/{rr}"], datetime.timedelta(seconds=max(0.5, t_end - time.time())))
                got[rr] = store.get(f"{key}/{rr}").decode()
            except Exception:
                pass
        return got

    def headline_attempt(k, label, kw):
        hook = os.environ.get("SIPX_BENCH_FAIL_LEG", "")
        err = None
        try:
            if hook == f"headline:{rank}" and k == 0:                      # tests: this rank's set-up of the first attempt fails
                raise RuntimeError("test hook: this rank fails the headline's first attempt")
        except Exception as e:
            err = repr(e)
        ready = post_and_wait(f"sipx_bench/headline/{k}/ready", "ok" if err is None else "fail: " + err[:300], agree_s)
        bad = {rr: v for rr, v in ready.items() if v != "ok"}
        if len(ready) < world or bad:
            why = ("; ".join(f"rank {rr}: {v}" for rr, v in sorted(bad.items())) or
                   f"ranks {sorted(set(range(world)) - set(ready))} did not report ready within {agree_s:.0f} s")
            return None, why

        def mid():
            if hook == f"headline-mid:{rank}" and k == 0:                  # tests: this rank fails inside the first attempt
                raise RuntimeError("test hook: this rank fails in the middle of the headline's first attempt")
        res, err = None, None
        try:
            res = measure(args.config, args.steps, args.warmup, mid_hook=mid, **kw)
        except BaseException as e:                                         # (SystemExit of --decomp slab on an unsuitable list included)
            if isinstance(e, KeyboardInterrupt):
                raise
            import gc
            err = repr(e)
            progress(f"headline attempt {k} ({label}) failed on this rank: {err[:300]}")
            gc.collect()
        done = post_and_wait(f"sipx_bench/headline/{k}/done", "ok" if err is None else "fail: " + err[:300], hl_deadline)
        bad = {rr: v for rr, v in done.items() if v != "ok"}
        if len(done) == world and not bad:
            return res, None
        why = "; ".join(f"rank {rr}: {v}" for rr, v in sorted(bad.items()))
        if len(done) < world:
            why += f"{'; ' if why else ''}ranks {sorted(set(range(world)) - set(done))} did not finish the attempt within {hl_deadline:.0f} s"
        return None, why

    def on_deadline(why):
        if rank != 0:
            return
        o = partial["out"] or {"metric": "PARSDMM iterations/sec", "value": 0.0, "unit": "it/s", "n_gpus": world, "steps": args.steps,
                               "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                               "dtype": args.dtype, "data": "synthetic", "config": {"workload": f"{args.config}: the headline did not finish"},
                               "roofline": None, "libsipx_sha16": lib_sha16()}
        o["error"] = str(why)[:300]
        if attempts_log:
            o["headline_attempts"] = attempts_log
        try:
            if saved_stdout is not None:
                os.dup2(saved_stdout, 1)
        except OSError:
            pass
        emit(o, args.detail)
    _PROGRESS["on_deadline"] = on_deadline

    progress(f"headline {args.config}: {args.warmup} warm-up + {args.steps} timed steps")
    r = None
    if dist is None or (world == 1 and not os.environ.get("SIPX_BENCH_HEADLINE_CHAIN")):
        r = measure(args.config, args.steps, args.warmup)
    else:
        native = None if share_gpu else "rccl"                              # (the rehearsal on one GPU runs on callbacks throughout)
        chain = []
        if args.decomp in ("auto", "slab"):
            chain.append(("slab, sparse arrays, " + ("callbacks" if share_gpu else "RCCL in the engine"), dict(decomp=args.decomp, comm_mode=native)))
            chain.append(("slab, full arrays, " + ("callbacks" if share_gpu else "RCCL in the engine"), dict(decomp=args.decomp, comm_mode=native, slab_full=True)))
            if not share_gpu:
                chain.append(("slab, full arrays, torch.distributed callbacks", dict(decomp=args.decomp, comm_mode="torch", slab_full=True)))
        else:
            chain.append(("sets, " + ("callbacks" if share_gpu else "RCCL in the engine"), dict(decomp="sets", comm_mode=native)))
        chain.append(("sets, torch.distributed callbacks", dict(decomp="sets", comm_mode="torch")))
        for k, (label, kw) in enumerate(chain):
            progress(f"headline attempt {k}: {label}")
            r, why = headline_attempt(k, label, kw)
            attempts_log.append({"attempt": label, "ok": r is not None, **({} if r is not None else {"error": (why or "")[:400]})})
            if r is not None:
                break
            unanswered = why and "did not" in why
            if unanswered:                                                  # some rank is beyond reach: no further collective can end
                break
        if r is None:
            restore_stdout()
            if rank == 0:
                emit({"metric": "PARSDMM iterations/sec", "value": 0.0, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                      "config": {"workload": f"{args.config}: no attempt of the headline finished on every rank"}, "roofline": None,
                      "headline_attempts": attempts_log, "error": "every attempt of the headline failed", "libsipx_sha16": lib_sha16()}, args.detail)
            progress("headline: every attempt failed")
            WATCHDOG.cancel()
            os._exit(0 if rank != 0 else 4)
        if len(attempts_log) > 1:
            r["comm"]["fell_back_from"] = [a["attempt"] for a in attempts_log[:-1]]
    out = {
        "metric": "PARSDMM iterations/sec", "value": r["value"], "unit": "it/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": r["config"], "roofline": r["roofline"], "dominant_kernel": r["dominant_kernel"], "kernels": r["kernels"],
        "iteration_roofline": r["iteration_roofline"], "timing_ms_per_iteration": r["timing_ms_per_iteration"],
        "comm": r["comm"], "decomposition": r["decomposition"], "libsipx_sha16": lib_sha16(),
    }
    if attempts_log:
        out["headline_attempts"] = attempts_log
    if r.get("rank_route"):
        out["rank_route"] = r["rank_route"]
    if r.get("batched_searches"):
        out["batched_searches"] = r["batched_searches"]
    if share_gpu:
        out["invalid_as_measurement"] = True
        out["rehearsal"] = f"SIPX_BENCH_SHARE_GPU: {world} ranks share GPU 0, the engine's collectives over gloo callbacks -- a rehearsal of the N > 1 flow, not a measurement"
    partial["out"] = out
    both = None
    if (world > 1 or force_dist) and args.decomp == "auto" and r.get("decomposition") == "slab":
        # (the set-sharded run keeps the communicator the headline ended up with)
        sets_mode = (r["comm"].get("comm_mode") if r["comm"].get("fell_back_from") else None)
        # Which of the two decompositions is faster on a given node depends on RCCL's small-message latency (slab: small
        # collectives only) against its bandwidth (sets: two N-vector exchanges).  Both are timed for the same K steps and
        # reported under fixed keys; the headline `value` is ALWAYS the slab decomposition -- the one DESIGN 5 chooses for this
        # set list -- so that it means the same thing from run to run and from node to node.
        progress("headline workload again, sharded by constraint set")
        r2 = safe("sets-decomposed headline", lambda: measure(args.config, args.steps, args.warmup, decomp="sets", comm_mode=sets_mode))
        both = {k: (v if "error" in v else
                    {"value": v["value"], "ms_per_step": v["ms_per_step"], "parallelism": v["config"]["parallelism"], "comm": v["comm"],
                     "timing_ms_per_iteration": v["timing_ms_per_iteration"]}) for k, v in (("slab", r), ("sets", r2))}
    if both is not None:
        out["decompositions"] = both
        out["faster_decomposition"] = max(both, key=lambda k: both[k].get("value", 0.0))
    elif dist is not None and r.get("decomposition"):
        # (a headline that ended up set-sharded, or --decomp sets / slab: the one decomposition that ran, under the same key)
        out["decompositions"] = {r["decomposition"]: {"value": r["value"], "ms_per_step": r["ms_per_step"], "comm": r["comm"],
                                                      "parallelism": r["config"]["parallelism"], "timing_ms_per_iteration": r["timing_ms_per_iteration"]}}
    if args.config == "c3" and args.dtype == "f32" and not args.no_512:
        # the honest HBM point (Q = 3.5 GiB, nothing fits the 256 MiB Infinity Cache): a short run of the same sets at 512^3
        progress("c3_512 leg")
        out["c3_512"] = safe("c3_512", lambda: leg(measure("c3-512", 10, 5), 10, 5))
    if args.config == "c3" and args.dtype == "f32" and args.q_mode == "cds" and not args.no_c4:
        # BASELINE configs[3]: 512^3, the eight constraint sets + distance term, at every N (the scaling target of the
        # contract is quoted on THIS set list: ">= 3.5x at 8 GPUs when 8 constraint sets are sharded").  Sets one per rank;
        # the slice-rank set, half of the single-GPU time (94 % before its warm-started subspace route), is projected by all ranks (each its slab of slices).
        progress("c4_512 leg")
        out["c4_512"] = safe("c4_512", lambda: leg(measure("c4", 6, 2), 6, 2, full=False))
        if dist is not None:
            # ... and with the whole iteration on z-slabs (round 5): y, l of every set on the ranks' planes, the slice-rank set
            # projected by every rank on its own slices, cardinality searched through the slab collectives, l1-DFT projected by an
            # owner rank on the gathered vector
            progress("c4_512 leg, slab-decomposed")
            out["c4_512_slab"] = safe("c4_512_slab", lambda: leg(measure("c4", 6, 2, decomp="slab", comm_mode=("torch" if share_gpu else None)), 6, 2, full=False))
    if args.config == "c3" and args.dtype == "f32" and args.q_mode == "cds" and not args.no_c5:
        # BASELINE configs[4]: PARSDMM_multi_level, 512^3 Float64, 3 levels, {bounds, l1:TV}, timed as ONE call
        # (examples/test_scaling_3D.jl:144-148); at N > 1 every level is slab-decomposed over the ranks.  `c5`: SURVEY 8(d)'s
        # model; `c5_layered`: a velocity-model-like field, the kind the reference's own timing runs on (c5_model).
        for key, model in (("c5", "survey"), ("c5_layered", "layered")):
            progress(f"{key} leg (PARSDMM_multi_level 512^3 Float64, 3 levels)")
            # (rehearsal with the ranks on ONE GPU: rounds 3-4 ran 256^3 there -- every level held whole arrays and four ranks' Float64
            #  arrays of 512^3 did not fit beside each other; every level holds sparse arrays since round 5: the real size, 17 GB per
            #  rank of four at the finest level.  SIPX_BENCH_C5_N: another size)
            n5 = (512, 512, 512)
            if os.environ.get("SIPX_BENCH_C5_N"):
                n5 = (int(os.environ["SIPX_BENCH_C5_N"]),) * 3
            r5m = safe(key, lambda: run_c5(sipx, n5, maxit=100, model=model, device=local_rank, dist=dist))
            r5m["n_gpus"] = world
            out[key] = r5m
    if dist is not None and (world > 1 or force_dist):
        progress("comm probe")
        out["comm_probe_us"] = safe("comm probe", lambda: comm_probe(dist, torch, world, rank, dev=torch.device("cpu") if share_gpu else None))
    if args.config == "c3" and args.dtype == "f32" and args.q_mode == "cds" and world == 1 and dist is None and not args.no_c2:
        # BASELINE configs[1]: 2048^2 Float32, {bounds, l1:TV} -- launch bound, Infinity-Cache resident (Q = 80 MiB): it/s only
        progress("c2_2048 leg")
        out["c2_2048"] = safe("c2_2048", lambda: leg(measure("c2", 20, 5), 20, 5, full=False))
    if args.config == "c3" and args.dtype == "f32" and args.q_mode == "cds" and world == 1 and dist is None and not args.no_whole_call:
        # the reference's own measurement is one wall clock around the call (examples/test_scaling_3D.jl:116-117: `@timed PARSDMM(...)`
        # after a first call): PARSDMM(m, AtA, TD_OP, set_Prop, P_sub, comp_grid, options) with default options, stop rules active,
        # first call (the context is built) and the calls after it (the context is reset: sipx_reset)
        progress("whole-call leg")
        out["whole_call"] = safe("whole_call", lambda: whole_call(sipx, args.config, TF))
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.dtype == "f32":
        progress("cpu baseline")
        out["cpu_baseline"] = safe("cpu_baseline", lambda: cpu_baseline(args.config, n, h, kinds))
    restore_stdout()
    if rank == 0:
        emit(out, args.detail)
    progress("done")
    WATCHDOG.cancel()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
