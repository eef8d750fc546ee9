"""Per-kernel averages of every counter of one rocprofv3 --pmc pass (counter_collection.csv), with the dispatch count.
usage: python tools/pmc_table.py <dir>"""
import csv
import glob
import os
import sys

per = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void sipx::", "")
        per.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
names = sorted({c for v in per.values() for c in v})
print("kernel".ljust(34), "n".rjust(5), *[c[:22].rjust(23) for c in names])
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1].get(names[0], [0]))):
    n = max(len(x) for x in v.values())
    print(k[:34].ljust(34), str(n).rjust(5), *[("%.4g" % (sum(v.get(c, [0])) / max(len(v.get(c, [0])), 1))).rjust(23) for c in names])
