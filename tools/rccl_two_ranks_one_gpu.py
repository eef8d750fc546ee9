"""Does this RCCL accept two ranks on ONE device?  (NCCL answers "Duplicate GPU detected"; asked once on the GPU box so that the
statement in DESIGN 5 is a measured one.)  usage: python tools/rccl_two_ranks_one_gpu.py   -> prints one JSON line"""
import json
import os
import sys
import datetime


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    res = {"rank": rank}
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=40), device_id=torch.device("cuda", 0))
        t = torch.ones(1024, device="cuda") * (rank + 1)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        res["all_reduce"] = float(t[0].item())
        dist.destroy_process_group()
    except Exception as e:
        res["error"] = repr(e)[:400]
    q.put(res)


if __name__ == "__main__":
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, 29733, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = []
    for p in ps:
        p.join(70)
    for p in ps:
        if p.is_alive():
            p.terminate()
            out.append({"hung": p.pid})
    while not q.empty():
        out.append(q.get())
    print(json.dumps({"two_ranks_on_one_device": out}))
