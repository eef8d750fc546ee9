"""BASELINE config 5 pattern: Float64, {bounds, l1 on TV}, 3 levels, coarsening factor 2 (test_scaling_3D.jl:144-145).
usage: python tools/c5_multilevel.py [n=256] [maxit=30] [host]     one GPU (host: the round-1 path with host transfers)
       python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/c5_multilevel.py 512 30
           one rank per GPU: every level slab-decomposed over the ranks (RCCL inside the engine); rank 0 prints the line
       SIPX_FORCE_DIST=1 python tools/c5_multilevel.py ...           the same path with a world of one"""
import os, sys, time, json
import numpy as np
if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("SIPX_FORCE_DIST"):
    import torch                       # before libsipx: torch brings its own HIP runtime and wants to initialise it first
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from __graft_entry__ import load_package
sipx = load_package()
from sipx import multilevel as ML
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
maxit = int(sys.argv[2]) if len(sys.argv) > 2 else 30
TF = np.float64
n, h = (n1, n1, n1), (25.0, 25.0, 25.0)
rng = np.random.default_rng(20240601 + 5)
m = (1500 + 2500 * np.linspace(0, 1, n[2])[None, None, :] + 150 * rng.standard_normal(n)).astype(TF).reshape(-1, order="F")
g = sipx.compgrid(h, n)
t0 = time.perf_counter()
s = sipx.get_TD_operator(g, "TV", TF)[0] @ m
c = [sipx.set_definitions("bounds", "identity", 1600.0, 3900.0, ("tensor", "")),
     sipx.set_definitions("l1", "TV", 0.0, float(0.5 * np.abs(s).sum()), ("tensor", ""))]
del s
opt = sipx.PARSDMM_options(FL=TF, maxit=maxit, evol_rel_tol=10 * np.finfo(TF).eps)
L = ML.setup_multi_level_PARSDMM(m, 3, 2, g, c, opt)
t1 = time.perf_counter()
host_path = len(sys.argv) > 3 and sys.argv[3] == "host"
T = {}
dist = None
world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
if world > 1 or os.environ.get("SIPX_FORCE_DIST"):
    import torch
    import torch.distributed as dist
    if "MASTER_ADDR" not in os.environ:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(local_rank)
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)                      # RCCL's banner goes to stderr: ONE JSON line on stdout
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    sipx.set_default_device(local_rank)
t1 = time.perf_counter()
x, log, l, y = ML.PARSDMM_multi_level(m, *L[:5], opt, device=local_rank, timings=T, host_transfers=host_path, dist=dist)
t2 = time.perf_counter()
if dist is not None:
    import torch
    torch.cuda.synchronize()
    sys.stdout.flush()
    os.dup2(saved, 1)
    dist.destroy_process_group()
    if rank != 0:
        sys.exit(0)
fin = T["levels"][-1]
print(json.dumps({"grid": n, "levels": [list(gg.n) for gg in L[4]], "transfers": "host (round-1 path)" if host_path else "device (sipx_warm_start_from)",
                  "n_gpus": world, "decomposition": "every level slab-decomposed over the ranks" if dist is not None else "single GPU",
                  "setup_s": t1 - t0, "whole_solve_s": t2 - t1,
                  "solve_only_s": sum(v["solve_s"] for v in T["levels"]), "warm_start_total_s": sum(v["warm_start_s"] for v in T["levels"]),
                  "context_total_s": sum(v["context_s"] for v in T["levels"]), "download_s": T.get("download_s"),
                  "per_level": T["levels"],
                  "finest_iterations": fin["iterations"], "finest_cg": fin["cg_iterations"],
                  "finest_level_it_per_s": fin["iterations"] / fin["solve_s"],
                  "obj_last": float(log.obj[-1]), "feas_last": [float(v) for v in log.set_feasibility[-1]],
                  "finite": bool(np.isfinite(x).all())}))
