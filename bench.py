#!/usr/bin/env python3
"""bench.py -- PARSDMM iterations/sec on synthetic grids (BASELINE.json metric).

One "step" = one PARSDMM iteration (rhs_compose -> CG x-minimisation -> y/l update of every
set -> logs -> stop rule -> rho/gamma adaptation -> Q update) on the configuration the metric
is quoted on: 3-D 256^3 Float32, sets {bounds on I, l1-ball on D_x, D_y, D_z} + the distance
term (BASELINE.json configs[2]; SURVEY 8d "C3").  Inputs are resident in HBM before the timed
region.  With --gpus N>1 the constraint sets are sharded over the ranks (one process per GPU,
RCCL all-reduce of the right-hand side): the same projection problem, so scaling is "strong".

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
    # BASELINE configs[3] (SURVEY 8d "C4"): 8 constraint sets, two of them non-convex
    "c4": ((512, 512, 512), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z", "annulus", "l1dft", "rank:32", "card:D_z"]),
    "c4-256": ((256, 256, 256), (25.0, 25.0, 25.0), ["bounds", "l1:D_x", "l1:D_y", "l1:D_z", "annulus", "l1dft", "rank:32", "card:D_z"]),
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"],
                    help="f32 = the contract workload; f64 = the same sets in Float64 (BASELINE config 5 computes in Float64)")
    ap.add_argument("--q-mode", default="cds", choices=["cds", "stencil"],
                    help="cds = the reference's banded Q (the contract workload); stencil = generated coefficients (SURVEY 8f-2)")
    args = ap.parse_args()

    import torch
    from __graft_entry__ import load_package
    sipx = load_package()
    from sipx import sharded  # noqa: E402  (registered by load_package)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N")
    dist = None
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
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    sipx.set_default_device(local_rank)
    TF = np.float32 if args.dtype == "f32" else np.float64
    n, h, kinds = CONFIGS[args.config]
    N = int(np.prod(n))
    m = synthetic_model(n, TF, 20240601 + 3)
    gs = sipx.compgrid(h, n)

    def radius_of(opname):                      # sigma = 0.5 ||A m||_1 (the reference tests' own rule)
        s = sipx.get_TD_operator(gs, opname, TF)[0] @ m
        return float(0.5 * np.abs(s.astype(np.float64)).sum())

    g, c = build_problem(sipx, n, h, kinds, m, TF, radius_of)
    P, A, prop = sipx.setup_constraints(c, g, TF)
    maxit = args.warmup + args.steps + 1
    opt = bench_options(sipx, TF, maxit)
    opt.Q_mode = args.q_mode
    A, AtA, l, y = sipx.PARSDMM_precompute_distribute(A, prop, g, opt)
    p = len(A)
    owned = sharded.shard_sets(p, world, rank)
    ctx = sipx.host.build_context(m, AtA, A, prop, P, g, opt, device=local_rank, owned=owned)
    comm = sharded.TorchComm(dist, torch.device("cuda", local_rank)) if dist is not None else sharded.LocalComm()
    native = dist is None        # one GPU: the native driver loop (sipx_parsdmm_begin/_steps); sharded: phase API

    class NativeDriver:            # same stepping interface as sharded.PhaseDriver
        def __init__(self):
            ctx.parsdmm_begin(opt)
            self.i = 0

        def step(self):
            self.i += 1
            return ctx.parsdmm_steps(1)

        def result_log(self):
            return ctx.parsdmm_log()

        @property
        def cg_total(self):
            return int(ctx._run[2]["cg_it"][:self.i].sum())

        @property
        def log(self):
            return type("L", (), {k: v for k, v in ctx._run[2].items()})

    drv = NativeDriver() if native else sharded.PhaseDriver(ctx, opt, comm, owned, any(prop.ncvx[:len(P)]))

    for _ in range(args.warmup):
        drv.step()
    if saved_stdout is not None:
        torch.cuda.synchronize()
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    ctx.kernel_stats(True)
    cg0 = drv.cg_total
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        drv.step()
        if os.environ.get("SIPX_BENCH_DEBUG"):
            i = drv.i - 1
            print(i + 1, "cg", drv.log.cg_it[i], "obj %.4e" % drv.log.obj[i], "rpri", drv.log.r_pri[i], "rho", drv.log.rho[i],
                  file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    launches, kms = ctx.kernel_stats(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    cg_its = drv.cg_total - cg0
    Q, offs = (None, None)
    d = len(np.unique(np.concatenate([np.asarray(o) for o in prop.AtA_offsets])))
    w = np.dtype(TF).itemsize
    spmv_bytes = (d + 2) * N * w                 # B_spmv = (d+2) N w per launch
    if args.q_mode == "stencil":
        spmv_bytes = 2 * N * w                   # reads p, writes Ap; coefficients are generated
    achieved = (spmv_bytes / (kms / launches * 1e-3) / 1e9) if launches else 0.0
    # HBM traffic of the dominant kernel from the PMC counters: separate rocprofv3 --pmc passes, summarised in profiles/
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "r01_c3_256_pmc.json")
    if args.config == "c3" and args.q_mode == "cds" and args.dtype == "f32" and os.path.exists(pmc):
        traffic = json.load(open(pmc))["dominant_kernel"]["hbm_bytes_per_launch_corrected"]
        traffic_src = "profiles/r01_c3_256_pmc.json (2*FETCH_SIZE + WRITE_SIZE, gfx950 correction)"
    log = drv.result_log()
    finite = bool(np.isfinite(log.obj).all() and np.isfinite(log.r_pri_total).all())

    # Whole-iteration roofline with SURVEY 8(d)'s algorithmic bytes: B_rhs + B_resid0 + k B_cg_iter + B_yl + B_log
    # (+ B_adapt + B_Q when rho / gamma are re-adapted, + B_feas every 10th iteration), summed over the timed steps.
    rows = [int(op.shape[0]) for op in A]
    d_i = [len(np.asarray(o)) for o in prop.AtA_offsets]
    pp = len(P)
    i0 = args.warmup + 1
    it_bytes = 0.0
    for i in range(i0, i0 + args.steps):                       # 1-based PARSDMM iteration numbers of the timed steps
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
    it_gbs = it_bytes / dt / 1e9 / max(world, 1)                # per GPU: every rank moves (at most) its share plus the x-step

    out = {
        "metric": "PARSDMM iterations/sec", "value": args.steps / dt, "unit": "it/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.config}: {'x'.join(map(str, n))} {'Float32' if args.dtype == 'f32' else 'Float64'}, sets {{{', '.join(kinds)}}} + distance term",
                   "grid": list(n), "sets": kinds, "q_mode": args.q_mode, "parallelism": f"set-sharded x{world}" if world > 1 else "single GPU",
                   "cg_iterations_in_timed_steps": int(cg_its), "all_logs_finite": finite,
                   "driver": "native loop (sipx_parsdmm_begin/_steps)" if native else
                             "phase-level C ABI (sipx_rhs_compose/argmin_x/update_y_l/...) + torch.distributed"},
        "roofline": {"bound": "hbm", "kernel": "k_cds<MODE=1> (cds_spmv + p.Ap partials)" if args.q_mode == "cds" else
                     "k_sq<MODE=1> (stencil Q product + p.Ap partials)",
                     "algorithmic_bytes_definition": "SURVEY 8(d): B_spmv = (d+2) N w (d bands + x read, y written)"
                     if args.q_mode == "cds" else "2 N w (p read, Ap written)",
                     "bands_from_hbm": (int((d + 1) // 2) if not os.environ.get("SIPX_CDS_FULL") else int(d)) if args.q_mode == "cds" else 0,
                     "bytes_with_symmetric_band_read": int(((d + 1) // 2 + 2) * N * w) if args.q_mode == "cds" else None, "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src, "launches": int(launches), "avg_launch_ms": (kms / launches) if launches else None,
                     "algorithmic_bytes_per_launch": spmv_bytes},
        "iteration_roofline": {"bound": "hbm", "achieved": it_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": it_gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_step": it_bytes / args.steps,
                               "definition": "SURVEY 8(d) B_iter summed over the timed steps / wall time" +
                                             ("" if world == 1 else " / n_gpus (the replicated x-step is not counted twice)")},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.dtype == "f32":
        out["cpu_baseline"] = cpu_baseline(args.config, n, h, kinds)
    ctx.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
