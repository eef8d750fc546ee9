"""How much of the slice-rank set's lane runs beside the engine stream?  Reads a rocprofv3 --kernel-trace CSV of a C4 run and,
over PARSDMM iterations A..B (as tools/timeline.py cuts them), reports per hardware queue the union of its kernel intervals,
the time two queues are busy together, and per kernel the mean duration of the launches that ran alone against those that
overlapped a kernel of another queue (contention shows as the second column growing).

usage: python tools/lane_overlap.py <dir with *_kernel_trace.csv> iters=A:B"""
import csv
import glob
import os
import sys


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def inter(a, b):
    i = j = 0
    out = []
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            out.append([s, e])
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


def total(iv):
    return sum(e - s for s, e in iv)


def main():
    src, arg = sys.argv[1], sys.argv[2]
    a, b = (int(v) for v in arg[len("iters="):].split(":"))
    rows = []
    for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name") or r.get("Name"),
                         r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    opens = [i for i, r in enumerate(rows) if "k_cg_begin" in r[2]]
    rows = rows[opens[a - 1]:(opens[b] if b < len(opens) else len(rows))]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    print(f"PARSDMM iterations {a}..{b}: window {(t1 - t0) / 1e6:.2f} ms, {len(rows)} dispatches")
    per_q = {}
    for s, e, n, q, st in rows:
        per_q.setdefault((q, st), []).append([s, e])
    keys = sorted(per_q, key=lambda k: -total(union(per_q[k])))
    un = {k: union(per_q[k]) for k in keys}
    for k in keys:
        print(f"  queue {k[0]} stream {k[1]}: {len(per_q[k]):5d} dispatches, busy {total(un[k]) / 1e6:8.2f} ms")
    allu = union([iv for k in keys for iv in per_q[k]])
    print(f"  any queue busy {total(allu) / 1e6:8.2f} ms; idle {(t1 - t0 - total(allu)) / 1e6:8.2f} ms")
    if len(keys) >= 2:
        both = inter(un[keys[0]], un[keys[1]])
        print(f"  the two busiest queues busy together {total(both) / 1e6:8.2f} ms")
    # per kernel: alone vs overlapped with another queue's kernel (by more than half of its own duration)
    stat = {}
    for s, e, n, q, st in rows:
        other = union([iv for k in keys if k != (q, st) for iv in per_q[k]])
        # (linear scan is fine for a few thousand dispatches)
        ov = sum(max(0, min(e, oe) - max(s, os_)) for os_, oe in other if oe > s and os_ < e)
        name = n.split("(")[0].replace("void sipx::", "")[:56]
        d = stat.setdefault((name, (q, st) == keys[0]), [0, 0.0, 0, 0.0])
        if ov * 2 > (e - s):
            d[2] += 1
            d[3] += (e - s) * 1e-3
        else:
            d[0] += 1
            d[1] += (e - s) * 1e-3
    print(f"{'kernel':56s} {'main':>4s} {'alone n':>8s} {'mean us':>9s} {'overl n':>8s} {'mean us':>9s}")
    for (name, main_q), (n0, t0_, n1, t1_) in sorted(stat.items(), key=lambda kv: -(kv[1][1] + kv[1][3]))[:40]:
        print(f"{name:56s} {'y' if main_q else 'n':>4s} {n0:8d} {t0_ / max(n0, 1):9.1f} {n1:8d} {t1_ / max(n1, 1):9.1f}")


if __name__ == "__main__":
    main()
