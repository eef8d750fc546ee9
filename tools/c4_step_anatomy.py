"""Anatomy of one C4 iteration from a rocprofv3 --kernel-trace CSV: the interval between the last two inertia certificates of
the slice-rank projector (k_cert_or) is one iteration; inside it the span of the rank projector's own call (k_sub_fro_part ..
k_cert_or), the union of kernel intervals inside and outside that span, and the largest kernels outside it.
usage: python tools/c4_step_anatomy.py <dir with *kernel_trace.csv> [iterations from the end = 3]"""
import csv, glob, os, sys, collections
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (r.get("Kernel_Name") or r.get("Name"))))
rows.sort()
certs = [i for i, r in enumerate(rows) if "k_cert_or" in r[2]]
fros = [i for i, r in enumerate(rows) if "k_sub_fro_part" in r[2]]
if len(certs) < 3:
    sys.exit("fewer than three certified calls in the trace")
def union(rs):
    tot, end = 0, 0
    for s, e, _ in rs:
        if e > end:
            tot += e - max(s, end)
            end = e
    return tot
last = min(int(sys.argv[2]) if len(sys.argv) > 2 else 3, len(certs) - 1)
for a, b in zip(certs[-last - 1:-1], certs[-last:]):
    step = rows[a + 1:b + 1]
    t0, t1 = rows[a][1], rows[b][1]
    fro = max(i for i in fros if i <= b)
    call0 = rows[fro][0]
    inside = [r for r in step if r[0] >= call0]
    outside = [r for r in step if r[0] < call0]
    agg = collections.Counter()
    for s, e, n in outside:
        agg[n.replace("void sipx::", "").split("(")[0][:48]] += e - s
    print("iteration %.1f ms: rank call %.1f ms (busy %.1f), before it %.1f ms (busy %.1f); outside, by kernel: %s" % (
        (t1 - t0) / 1e6, (t1 - call0) / 1e6, union(inside) / 1e6, (call0 - t0) / 1e6, union(outside) / 1e6,
        ", ".join("%s %.1f" % (k, v / 1e6) for k, v in agg.most_common(8))))
