"""C4 (eight sets, 512^3) iteration by iteration from a rocprofv3 kernel-trace CSV: the dispatches between two k_cg_begin launches
are one PARSDMM iteration; per iteration the wall span, the busy time and the kernel time by group -- the slice-rank projector's
products with the Gram matrices, its Rayleigh-Ritz steps, its certificate, a full decomposition, the DFT set, the cardinality
set, the one-sweep y/l update, the threshold searches, the x-step, the rest.
usage: python tools/c4_iter_groups.py <kernel_trace.csv>"""
import collections
import csv
import sys

GROUPS = (
    ("rank: full decomposition", ("rocsolver", "syr2", "stedc", "latrd", "substitution", "iota", "trsm", "gemv", "rocblas_")),
    ("rank: certificate", ("k_cert", "potrf", "potf2", "chol_", "trtri")),
    ("rank: Rayleigh-Ritz", ("k_chol_inv", "k_ritz", "k_sub_", "k_cheb_plan")),
    ("rank: recurrence", ("k_cheb_step", "k_cheb_mask")),
    ("rank: GEMM", ("Cijk_",)),
    ("rank: gather/scatter", ("k_seg_gather", "k_seg_scatter")),
    ("dft set", ("fft_", "k_pack", "k_cabs", "k_csoft", "k_unpack", "transpose")),
    ("cardinality", ("k_seg_card", "k_card", "k_kth", "k_hist")),
    ("sweep y/l", ("k_yl_multi",)),
    ("per-set y/l", ("k_yl<", "k_adj", "k_fwd", "k_apply")),
    ("searches", ("k_pass", "k_lean", "k_spec", "k_sample", "k_l1_solve", "k_slot", "k_ps_", "k_proj")),
    ("x-step", ("k_cds", "k_cg_")),
    ("rhs / Q", ("k_rhs", "k_q_update")),
    ("copies", ("copyBuffer", "fillBuffer", "memset")),
)


def group_of(name):
    for g, keys in GROUPS:
        if any(k in name for k in keys):
            return g
    return "other: " + name.replace("void sipx::", "").split("(")[0][:40]


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name") or r.get("Name")))
rows.sort()
opens = [i for i, r in enumerate(rows) if "k_cg_begin" in r[2]]
for n, (a, b) in enumerate(zip(opens[:-1], opens[1:])):
    step = rows[a:b]
    span = (rows[b][0] - rows[a][0]) / 1e6
    busy, end = 0, 0
    agg = collections.Counter()
    gemms = 0
    for s, e, name in step:
        if e > end:
            busy += e - max(s, end)
            end = e
        agg[group_of(name)] += e - s
        gemms += 1 if "Cijk_" in name else 0
    print("iteration %2d: %6.1f ms (busy %6.1f), %3d GEMMs | %s" % (
        n + 1, span, busy / 1e6, gemms, ", ".join("%s %.1f" % (k, v / 1e6) for k, v in agg.most_common(12))))
