"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) into per-kernel HBM-side
bytes per launch.  gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE tallies a wide coalesced read
at half its bytes -> bytes_read = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.  Launches of the dominant kernel that exit at
once (speculative CG iteration past convergence) are excluded by their negligible FETCH_SIZE.

usage: python tools/summarize_pmc.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [note]"""
import csv
import glob
import json
import re
import os
import sys


def load(d, counter):
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            per.setdefault(row["Kernel_Name"].split("(")[0], []).append(float(row["Counter_Value"]))
    return per


def is_cg_product(k):
    """k_cds<T, V, D, MODE = 1> or its z-marching form k_cds_march<T, V, ORD, MODE = 1>: the product fused with the dot product"""
    # the MODE argument is the 4th template argument of both (k_cds_march carries one more behind it)
    m = re.search(r"k_cds(?:_march)?<([^>]*)>", k)
    if not m:
        return False
    args = [a.strip() for a in m.group(1).split(",")]
    return len(args) >= 4 and args[3] == "1"


def main():
    fd, wd, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only); " + note,
           "units": "FETCH_SIZE/WRITE_SIZE in KiB per dispatch; gfx950 correction: bytes_read = 2*FETCH_SIZE*1024; WRITE_SIZE exact",
           "kernels": {}}
    for k in sorted(F):
        f, w = F[k], W.get(k, [0.0])
        if is_cg_product(k):
            keep = [v for v in f if v > 1024.0]            # launches that did work (early exits fetch a few KiB)
            wk = sorted(w)[len(w) - len(keep):] if keep else w
            f, w = (keep or f), (wk or w)
        fa, wa = sum(f) / len(f), sum(w) / len(w)
        res["kernels"][k] = {"FETCH_SIZE_KiB_avg": fa, "WRITE_SIZE_KiB_avg": wa, "dispatches": len(f),
                             "hbm_bytes_per_launch_corrected": 2 * fa * 1024 + wa * 1024}
    dom = [k for k in res["kernels"] if is_cg_product(k)]
    if dom:
        dom.sort(key=lambda k: -res["kernels"][k]["dispatches"])
        res["dominant_kernel"] = dict(res["kernels"][dom[0]], kernel=dom[0])     # the product of the CG iteration (bench `roofline`)
    # the build the counters were taken on: bench.py quotes `traffic` only for this very library
    import hashlib
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "setintersectionprojection.jl_amd", "libsipx.so")
    if os.path.exists(lib):
        res["libsipx_sha16"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res.get("dominant_kernel", {}), indent=1))


if __name__ == "__main__":
    main()
