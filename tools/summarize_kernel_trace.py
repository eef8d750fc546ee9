"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel launch count / average duration, and for the dominant kernel
(k_cds<MODE 1>) the average over the launches that did work.  CG iterations are enqueued one ahead of the host
(engine.cpp, argmin_x): the speculative launch past convergence returns at once and would otherwise drag the average
of the same kernel name down; those early exits (< 5 us) are listed separately.

usage: python tools/summarize_kernel_trace.py <dir with *_kernel_trace.csv> <out.json>"""
import csv
import glob
import json
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
    dom = [(n, d) for n, d in per.items() if "k_cds<" in n and ", 1>" in n.split("(")[0]]
    res = {"trace_files": [os.path.basename(f) for f in files], "kernels": summary[:40]}
    if dom:
        name, d = max(dom, key=lambda t: sum(t[1]))
        work = [v for v in d if v >= 5.0]
        res["dominant_kernel"] = {"kernel": name[:160], "launches_total": len(d), "launches_with_work": len(work),
                                  "early_exit_launches": len(d) - len(work),
                                  "avg_us_with_work": sum(work) / max(len(work), 1),
                                  "avg_us_all": sum(d) / len(d),
                                  "note": "early exits = speculative CG iteration enqueued past convergence (returns on the device-side done flag)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res.get("dominant_kernel", {}), indent=1))


if __name__ == "__main__":
    main()
