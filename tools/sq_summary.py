"""Per-kernel SQ counter summary of one rocprofv3 --pmc pass (SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_VALU SQ_INSTS_VALU): where the wave cycles of each kernel go (parked on memory / issue stall / issuing).

usage: python tools/sq_summary.py <dir of the pass>"""
import csv
import glob
import os
import sys

per = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:48]
        per.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
rows = []
for k, c in per.items():
    wc = sum(c.get("SQ_WAVE_CYCLES", [0]))
    if wc <= 0:
        continue
    n = len(c["SQ_WAVE_CYCLES"])
    g = lambda name: sum(c.get(name, [0]))
    rows.append((wc, k, n, g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_ACTIVE_INST_ANY") / wc,
                 g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_INSTS_VALU") / n))
rows.sort(reverse=True)
print(f"{'kernel':48s} {'n':>4s} {'wait_mem':>8s} {'stall':>6s} {'issue':>6s} {'valu':>6s} {'valu insts/launch':>18s}")
for wc, k, n, w, st, ac, va, iv in rows[:20]:
    print(f"{k:48s} {n:4d} {w:8.2f} {st:6.2f} {ac:6.2f} {va:6.2f} {iv:18.3e}")
