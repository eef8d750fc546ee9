"""One PARSDMM iteration as the GPU ran it: from a rocprofv3 kernel-trace CSV, the dispatches between two residual products
(the kernel that opens an x-step) near the end of the run -- start offset, duration, gap to the previous kernel's end, grid.
usage: python tools/iter_anatomy.py <kernel_trace.csv> [iterations=2] [skip_from_end=2] [opener substring]"""
import csv
import sys


def short(n):
    n = n.replace("void sipx::", "").replace("sipx::", "")
    head = n.split("(")[0]
    return head[:64]


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name") or r.get("Name"),
                     r.get("Grid_Size") or r.get("Grid_Size_X") or ""))
    rows.sort()
    nit = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    opener = sys.argv[4] if len(sys.argv) > 4 else "NoExtra>"
    # residual product: k_cds_march<float, 4, ORD, 2, ...> or k_cds<..., 2>
    opens = [i for i, r in enumerate(rows) if ("k_cds_march<" in r[2] and ", 2, sipx::" in r[2]) or ("k_cds<" in r[2] and ", 2>" in r[2])]
    if len(opens) < nit + skip + 1:
        print("too few iterations in the trace")
        return
    a, b = opens[-(nit + skip + 1)], opens[-(skip + 1)]
    t0 = rows[a][0]
    end = t0
    for s, e, n, g in rows[a:b]:
        if ("k_cds_march<" in n and ", 2, sipx::" in n) or ("k_cds<" in n and ", 2>" in n):
            print(f"--- iteration: +{(s - t0) / 1e3:9.1f} us")
        print(f"  +{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - end) / 1e3:7.1f}  {short(n)}  grid {g}")
        end = max(end, e)
    print(f"span {((rows[b][0] - t0) / 1e3):.1f} us for {nit} iterations")


if __name__ == "__main__":
    main()
