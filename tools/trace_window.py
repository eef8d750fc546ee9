"""Prints the dispatches around the n-th last launch of a kernel (name substring) from a rocprofv3 --kernel-trace CSV:
start time relative to the window, duration, idle gap to the previous end, stream/queue id, kernel name.
usage: python tools/trace_window.py <dir> <kernel substring> [n-th last = 2] [dispatches before = 40] [after = 6]"""
import csv, glob, os, sys
src, pat = sys.argv[1], sys.argv[2]
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
before = int(sys.argv[4]) if len(sys.argv) > 4 else 40
after = int(sys.argv[5]) if len(sys.argv) > 5 else 6
rows = []
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name") or r.get("Name"), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
idx = [i for i, r in enumerate(rows) if pat in r[2]]
k = idx[-nth]
lo, hi = max(0, k - before), min(len(rows), k + after)
t0 = rows[lo][0]
end = rows[lo][0]
for s, e, n, q, st in rows[lo:hi]:
    name = n.replace("void sipx::", "").split("(")[0][:60]
    print("%9.1f us  dur %8.1f  gap %7.1f  q %s st %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - end) / 1e3, q, st, name))
    end = max(end, e)
