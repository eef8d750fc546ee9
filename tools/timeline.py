"""Where does the wall time of an iteration go?  Reads a rocprofv3 --kernel-trace CSV and, over the dispatches of the
last `frac` of the run (the timed steps), reports
  * the union of all kernel intervals (GPU busy) and the gaps between them (launch / host latency),
  * the time during which ONLY one-workgroup kernels (decide / solve / finalize) are running: latency-bound tail,
  * per kernel: launches, median / mean duration of the launches that did work (>= 5 us).

usage: python tools/timeline.py <dir with *_kernel_trace.csv> [frac=0.6 | iters=A:B]
  frac=0.6     the last 60 % of the dispatches (rounds 1-4; with bench.py's one-call-per-step warm-up in front of the timed steps this
               window holds warm-up iterations too -- round 4's r04_c3_*_timeline.txt did)
  iters=A:B    PARSDMM iterations A..B (1-based, inclusive) of the LAST context of the run: an iteration opens with its k_cg_begin
               launch (one per x-step).  bench.py --warmup W --steps K: iters=W+1:W+K is exactly the timed region."""
import csv
import glob
import os
import statistics
import sys

SMALL = ("k_decide", "k_l1_solve", "k_fin", "k_cg_fin", "k_cg_begin", "k_card_decide", "k_card_select", "k_ps_init",
         "k_stop", "k_bb", "k_log")


def main():
    src = sys.argv[1]
    arg = sys.argv[2] if len(sys.argv) > 2 else "0.6"
    frac = None if arg.startswith("iters=") else float(arg)
    files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name") or r.get("Name")))
    rows.sort()
    if frac is not None:
        rows = rows[int(len(rows) * (1 - frac)):]
    else:
        a, b = (int(v) for v in arg[len("iters="):].split(":"))
        opens = [i for i, r in enumerate(rows) if "k_cg_begin" in r[2]]
        if len(opens) < b:
            raise SystemExit(f"the trace holds {len(opens)} x-steps, iters={a}:{b} needs {b}")
        # (the window closes where iteration b + 1 opens, or -- b being the run's last iteration -- with the trace)
        rows = rows[opens[a - 1]:(opens[b] if b < len(opens) else len(rows))]
        print(f"PARSDMM iterations {a}..{b} (x-steps {a}..{b} of {len(opens)} in the trace)")
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    # sweep
    ev = []
    for s, e, n in rows:
        small = any(k in n for k in SMALL)
        ev.append((s, 1, small))
        ev.append((e, -1, small))
    ev.sort()
    busy = small_only = 0
    nbig = nsmall = 0
    last = ev[0][0]
    for t, d, small in ev:
        if nbig + nsmall > 0:
            busy += t - last
            if nbig == 0:
                small_only += t - last
        last = t
        if small:
            nsmall += d
        else:
            nbig += d
    wall = t1 - t0
    print(f"window {wall/1e6:.2f} ms, {len(rows)} dispatches")
    print(f"GPU busy (union)        {busy/1e6:8.2f} ms  {100*busy/wall:5.1f} %")
    print(f"  only 1-workgroup work {small_only/1e6:8.2f} ms  {100*small_only/wall:5.1f} %")
    print(f"idle gaps               {(wall-busy)/1e6:8.2f} ms  {100*(wall-busy)/wall:5.1f} %")
    # idle gaps by the pair of kernels around them
    gaps = {}
    cur_end, cur_name = rows[0][1], rows[0][2]
    for s, e, n in rows[1:]:
        if s > cur_end:
            key = (cur_name.split("(")[0].replace("void sipx::", "")[:34], n.split("(")[0].replace("void sipx::", "")[:34])
            g = gaps.setdefault(key, [0, 0.0])
            g[0] += 1
            g[1] += (s - cur_end) * 1e-3
        if e > cur_end:
            cur_end, cur_name = e, n
    print("idle gaps by (kernel before -> kernel after):")
    for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {a:34s} -> {b:34s} n={c:4d} total={t/1e3:6.2f}ms avg={t/c:6.1f}us")
    per = {}
    for s, e, n in rows:
        per.setdefault(n.split("(")[0][:60], []).append((e - s) * 1e-3)
    out = []
    for n, d in per.items():
        w = [v for v in d if v >= 5.0] or d
        out.append((sum(d), n, len(d), len(w), statistics.median(w), sum(w) / len(w)))
    out.sort(reverse=True)
    for tot, n, c, cw, med, mean in out[:24]:
        print(f"{n:60s} n={c:4d} work={cw:4d} median={med:8.1f}us mean={mean:8.1f}us total={tot/1e3:7.2f}ms")


if __name__ == "__main__":
    main()
