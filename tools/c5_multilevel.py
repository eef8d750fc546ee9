"""BASELINE config 5: PARSDMM_multi_level, Float64, {bounds, l1 on TV}, 3 levels, coarsening factor 2
(examples/test_scaling_3D.jl:144-148).  The workload itself (model, radius, options) is bench.run_c5 -- the `c5` leg of the
default bench line; this tool runs it at other sizes / with the other model / over several ranks.
usage: python tools/c5_multilevel.py [n=256] [maxit=100] [host] [model=survey|layered] [sigma=<fraction of ||TV m||_1>]
           one GPU (host: the round-1 path with host transfers)
       python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/c5_multilevel.py 512 100
           one rank per GPU: every level slab-decomposed over the ranks (RCCL inside the engine); rank 0 prints the line
       SIPX_FORCE_DIST=1 python tools/c5_multilevel.py ...           the same path with a world of one
Exits non-zero when a level did not iterate (the round-2 measurement was degenerate that way)."""
import json
import os
import sys

import numpy as np

if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("SIPX_FORCE_DIST"):
    import torch                       # before libsipx: torch brings its own HIP runtime and wants to initialise it first
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                            # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


def main():
    sipx = load_package()
    pos = [a for a in sys.argv[1:] if "=" not in a]
    kw = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
    n1 = int(pos[0]) if len(pos) > 0 else 256
    maxit = int(pos[1]) if len(pos) > 1 else 100
    host_path = len(pos) > 2 and pos[2] == "host"
    model = kw.get("model", "survey")
    sigma = float(kw["sigma"]) if "sigma" in kw else None
    dist = None
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    saved = None
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
    r = bench.run_c5(sipx, (n1, n1, n1), TF=np.float64, maxit=maxit, model=model, sigma=sigma, device=local_rank, dist=dist,
                     host_transfers=host_path)
    r["n_gpus"] = world
    r["decomposition"] = "every level slab-decomposed over the ranks" if dist is not None else "single GPU"
    if dist is not None:
        import torch
        torch.cuda.synchronize()
        sys.stdout.flush()
        os.dup2(saved, 1)
        dist.destroy_process_group()
        if rank != 0:
            return 0
    print(json.dumps(r))
    if not r["every_level_iterates"]:
        print("c5: a level returned without iterating: " + str(r["iterations_per_level"]), file=sys.stderr)
        return 3
    return 0


if __name__ == "__main__":
    sys.exit(main())
