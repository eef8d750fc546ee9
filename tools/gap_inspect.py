"""What sits in an idle gap of the kernel timeline?  From a rocprofv3 run with --kernel-trace --hip-trace --memory-copy-trace: for every
gap of at least `min_us` between the end of a kernel whose name contains `before` and the start of the next kernel whose name contains
`after`, the HIP API calls and memory copies that overlap the gap (name, start offset from the gap's start, duration).
usage: python tools/gap_inspect.py <rocprof output dir> <before substring> <after substring> [min_us=100] [max gaps=3]"""
import csv
import glob
import os
import sys


def load(src, pattern, cols):
    out = []
    for f in glob.glob(os.path.join(src, "**", pattern), recursive=True):
        for r in csv.DictReader(open(f)):
            try:
                out.append(tuple(r[c] for c in cols))
            except KeyError:
                pass
    return out


def main():
    src, before, after = sys.argv[1:4]
    min_us = float(sys.argv[4]) if len(sys.argv) > 4 else 100.0
    top = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    k = sorted((int(s), int(e), n) for s, e, n in load(src, "*kernel_trace.csv", ("Start_Timestamp", "End_Timestamp", "Kernel_Name")))
    api = sorted((int(s), int(e), n) for s, e, n in load(src, "*hip_api_trace.csv", ("Start_Timestamp", "End_Timestamp", "Function")))
    cp = sorted((int(s), int(e), n) for s, e, n in load(src, "*memory_copy_trace.csv", ("Start_Timestamp", "End_Timestamp", "Direction")))
    print(f"{len(k)} kernels, {len(api)} HIP API calls, {len(cp)} memory copies")
    shown = 0
    end, name = k[0][1], k[0][2]
    for s, e, n in k[1:]:
        if s > end and before in name and after in n and (s - end) / 1e3 >= min_us:
            print(f"\ngap of {(s - end) / 1e3:.1f} us between {name.split('(')[0][-60:]} and {n.split('(')[0][-60:]}")
            for a0, a1, an in api:
                if a1 >= end - 20000 and a0 <= s:
                    print(f"   api  {an:40s} +{(a0 - end) / 1e3:9.1f} us  dur {(a1 - a0) / 1e3:9.1f} us")
            for c0, c1, cn in cp:
                if c1 >= end - 20000 and c0 <= s:
                    print(f"   copy {cn:40s} +{(c0 - end) / 1e3:9.1f} us  dur {(c1 - c0) / 1e3:9.1f} us")
            shown += 1
            if shown >= top:
                break
        if e > end:
            end, name = e, n


if __name__ == "__main__":
    main()
