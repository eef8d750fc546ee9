"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel launch count / average duration, and for the dominant kernel
(k_cds<MODE 1>) the average over the launches that did work.  Since round 2 the engine launches a CG iteration only once
the verdict of the one before it is back (ticket word, engine.cpp argmin_x), so there are no launches that return at once
and rocprofv3's own --stats row needs no post-processing; launches under 5 us are still listed separately as a check.

usage: python tools/summarize_kernel_trace.py <dir with *_kernel_trace.csv> <out.json>"""
import csv
import glob
import json
import re
import os
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no kernel trace found under " + src)
    per = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name") or row.get("Name")
            dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3       # us
            per.setdefault(name, []).append(dur)
    summary = []
    for name, d in per.items():
        summary.append({"kernel": name[:160], "launches": len(d), "total_us": sum(d), "avg_us": sum(d) / len(d)})
    summary.sort(key=lambda r: -r["total_us"])
    tot = sum(r["total_us"] for r in summary)
    for r in summary:
        r["percent"] = 100.0 * r["total_us"] / tot
    def is_cg_product(n):      # MODE = 1 is the 4th template argument of k_cds and of k_cds_march
        m = re.search(r"k_cds(?:_march)?<([^>]*)>", n)
        a = [v.strip() for v in m.group(1).split(",")] if m else []
        return len(a) >= 4 and a[3] == "1"
    dom = [(n, d) for n, d in per.items() if is_cg_product(n)]
    res = {"trace_files": [os.path.basename(f) for f in files], "kernels": summary[:40]}
    if dom:
        name, d = max(dom, key=lambda t: sum(t[1]))
        work = [v for v in d if v >= 5.0]
        res["dominant_kernel"] = {"kernel": name[:160], "launches_total": len(d), "launches_with_work": len(work),
                                  "early_exit_launches": len(d) - len(work),
                                  "avg_us_with_work": sum(work) / max(len(work), 1),
                                  "avg_us_all": sum(d) / len(d),
                                  "note": "early exits = launches under 5 us; none are expected (no CG iteration is launched past convergence)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res.get("dominant_kernel", {}), indent=1))


if __name__ == "__main__":
    main()
