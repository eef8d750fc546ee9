"""A/B of library builds on one box: runs bench.py for each variant in turn, `rounds` times, and prints it/s per variant.
usage: python tools/ab.py <rounds> <config> <variant> [<variant> ...]   (variant = suffix of libsipx_<variant>.so, or "base")"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds, config, variants = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
extra = os.environ.get("AB_ARGS", "").split()
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        env = dict(os.environ)
        if v != "base":
            env["SIPX_LIBRARY"] = os.path.join(root, "setintersectionprojection.jl_amd", f"libsipx_{v}.so")
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-c4", "--config", config] + extra,
                             env=env, capture_output=True, text=True, timeout=300)
        if out.returncode != 0:
            print(v, "FAILED", out.stderr[-400:], flush=True)
            sys.exit(1)
        res[v].append(json.loads(out.stdout.strip().splitlines()[-1])["value"])
        print(r, v, round(res[v][-1], 2), flush=True)
for v in variants:
    a = sorted(res[v])
    print(f"{v:12s} median {a[len(a)//2]:8.2f}  all {[round(x, 1) for x in res[v]]}")
