"""A/B of library builds on one box: runs bench.py for each variant in turn, `rounds` times, and prints it/s per variant plus
the average duration of the kernels named in AB_KERNELS (comma separated; from the bench line's per-kernel table).
usage: python tools/ab.py <rounds> <config> <variant> [<variant> ...]
   variant = "base", a suffix of setintersectionprojection.jl_amd/libsipx_<variant>.so or scratch/libsipx_<variant>.so, or
   NAME=VALUE[,NAME=VALUE...] (environment switches on the base library)"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds, config, variants = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
extra = os.environ.get("AB_ARGS", "").split()
watch = [k for k in os.environ.get("AB_KERNELS", "").split(",") if k]
res = {v: [] for v in variants}
kern = {v: {} for v in variants}
for r in range(rounds):
    for v in variants:
        env = dict(os.environ)
        if "=" in v:
            env.update(dict(kv.split("=", 1) for kv in v.split(",")))
        elif v != "base":
            for d in ("setintersectionprojection.jl_amd", "scratch"):
                p = os.path.join(root, d, f"libsipx_{v}.so")
                if os.path.exists(p):
                    env["SIPX_LIBRARY"] = p
        detail = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"sipx_ab_detail_{os.getpid()}.json")
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-c4", "--no-c5", "--config", config,
                              "--detail", detail] + extra, env=env, capture_output=True, text=True, timeout=300)
        if out.returncode != 0:
            print(v, "FAILED", out.stderr[-400:], flush=True)
            sys.exit(1)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        res[v].append(d["value"])
        if "c3_512" in d:
            res.setdefault(v + " [c3_512]", []).append(d["c3_512"]["value"])
        try:                                   # the kernel table lives in the detail file since round 4
            d = json.load(open(detail))
        except Exception:
            pass
        for row in d.get("kernels") or []:
            if row["kernel"] in watch:
                kern[v].setdefault(row["kernel"], []).append(row["avg_launch_ms"])
        print(r, v, round(res[v][-1], 2), {k: round(a[-1], 4) for k, a in kern[v].items()}, flush=True)
for v in list(res):
    kern.setdefault(v, {})
    a = sorted(res[v])
    print(f"{v:24s} median {a[len(a)//2]:8.2f}  all {[round(x, 1) for x in res[v]]}  " +
          "  ".join(f"{k} {sorted(t)[len(t)//2]:.4f} ms" for k, t in kern[v].items()))
