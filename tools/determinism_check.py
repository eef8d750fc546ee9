"""Do repeated runs of a slab-decomposed solve on several ranks of one GPU give the same bits?  (Round 5: a zero-fill on the null
stream raced with a kernel on the non-blocking engine stream at set-up, and the runs differed; every collective protocol of the slab
decomposition rests on all ranks deciding alike from identical state.)  usage: python tools/determinism_check.py  (on a GPU box)"""
import sys, os, tempfile, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch.multiprocessing as mp
import test_gpu_parity as TP
def run(kinds, n, world, reps, tag, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    res = []
    for rep in range(reps):
        out = tempfile.mkdtemp()
        mp.spawn(TP._sharded_worker, args=(world, 29800 + rep, out, kinds, n, "gloo", "torch", False, "slab", "f32"), nprocs=world, join=True)
        res.append(dict(np.load(os.path.join(out, "r0.npz"))))
    bad = []
    for rep in range(1, reps):
        for k in ("x", "r_pri"):
            if not np.array_equal(res[0][k], res[rep][k], equal_nan=True):
                d = np.abs(res[0][k].astype(np.float64) - res[rep][k].astype(np.float64))
                bad.append((rep, k, float(np.nanmax(d)), np.argwhere(d > 0)[0].tolist()))
    print(tag, "->", "deterministic" if not bad else bad, flush=True)
    for k in (env or {}):
        os.environ.pop(k)
if __name__ == "__main__":
    C4 = TP.C4_KINDS
    run(C4, (16, 12, 8), 4, 5, "C4")
    run(["bounds", "l1:D_x", "l1:D_y", "l1:D_z", "annulus", "l1dft"], (16, 12, 8), 4, 4, "C4's convex sets")
    run(C4, (20, 18, 10), 3, 3, "C4 ragged 3 ranks")
