#!/bin/bash
# Regenerates the round's profile files on a GPU box (run through gpurun from the repo root); outputs under gpurun_out/prof/,
# to be copied into profiles/ by the caller.  usage: tools/refresh_profiles.sh [round tag, default r05] [part: a | b | c | all]
#   a: PMC passes, kernel traces + timelines of c3 / c3-512, the default bench run;  b: C4 traces, its 80-iteration run, C2;
#   c: steady-state, Float64, C5, RCCL-world-of-one, rank-share and 768^3 runs.  (One gpurun call holds 20 minutes: a part each.)
set -e -o pipefail
R=${1:-r05}
PART=${2:-all}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; mkdir -p $O
T="timeout -k 10 500"
# profiled runs: the headline workload only, without the all-kernel statistics window (its event records would show up as gaps)
B="--no-cpu-baseline --no-512 --no-c4 --no-c5 --no-c2 --no-whole-call --no-kernel-table"
S="--no-cpu-baseline --no-c4 --no-c5 --no-c2 --no-whole-call"      # plain runs of one workload (the 512^3 leg kept where the config is c3)
pmc_pair() {   # <config> <tag>: the two HBM-side passes (FETCH_SIZE / WRITE_SIZE in separate runs, --kernel-trace only)
  $T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o t --output-format csv -- python3 bench.py $B --config $1 --steps 6 --warmup 2 > /dev/null 2>$O/pmc.err
  $T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o t --output-format csv -- python3 bench.py $B --config $1 --steps 6 --warmup 2 > /dev/null 2>>$O/pmc.err
  python tools/summarize_pmc.py $O/pmc_f $O/pmc_w $O/${R}_$2_pmc.json "bench.py $B --config $1 --steps 6 --warmup 2" > /dev/null
  rm -rf $O/pmc_f $O/pmc_w
}
trace() {      # <config> <tag> <first:last timed iteration> [extra bench args]: rocprofv3 --kernel-trace --stats of the bench command + per-dispatch summary + timeline
  local cfg=$1 tag=$2 win=$3; shift 3
  $T rocprofv3 --kernel-trace --stats -d $O/kt -o t --output-format csv -- python3 bench.py $B --config $cfg "$@" > $O/${R}_${tag}_bench_under_rocprof.json 2>$O/kt.err
  python tools/summarize_kernel_trace.py $O/kt $O/${R}_${tag}_kernel_trace_summary.json > /dev/null
  python tools/timeline.py $O/kt iters=$win > $O/${R}_${tag}_timeline.txt
  cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/${R}_${tag}_kernel_stats.csv
  if [[ $tag == c4* ]]; then python tools/c4_iter_groups.py $(find $O/kt -name "*kernel_trace.csv" | head -1) > $O/${R}_${tag}_iteration_groups.txt; fi
  rm -rf $O/kt
}
if [[ $PART == a || $PART == all ]]; then
pmc_pair c3 c3_256
pmc_pair c3-512 c3_512
trace c3 c3_256 6:25
trace c3-512 c3_512 6:15 --steps 10 --warmup 5
# the product kernel alone, k_cds against its z-marching form: time (stand-alone microbenchmark) and fabric-side traffic
if [ -x scratch/spmv_bench ]; then
  SPMV_QUICK=1 $T scratch/spmv_bench 256 > $O/${R}_spmv_march_256.txt 2>&1
  SPMV_QUICK=1 $T scratch/spmv_bench 512 > $O/${R}_spmv_march_512.txt 2>&1
  for which in engine march; do
    for c in FETCH_SIZE WRITE_SIZE; do
      SPMV_QUICK=1 SPMV_ONLY=$which $T rocprofv3 --kernel-trace --pmc $c -d $O/sp_${which}_$c -o t --output-format csv -- scratch/spmv_bench 512 > /dev/null 2>>$O/pmc.err
    done
  done
  python - "$O" "$R" <<'PY'
import csv, glob, json, os, sys
O, R = sys.argv[1], sys.argv[2]
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) around `SPMV_ONLY=<engine|march> scratch/spmv_bench 512`: "
                 "the LAST 20 dispatches of each kernel (the isolated launches); bytes = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950 correction)",
       "minimum_bytes": 6 * 512 ** 3 * 4}
for which, pat in (("engine", "k_cds<"), ("march", "k_spmv_march")):
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = []
        for f in glob.glob(os.path.join(O, f"sp_{which}_{c}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and pat in r["Kernel_Name"]:
                    rows.append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
        rows.sort()
        last = [v for _, v in rows[-20:]]
        vals[c] = sum(last) / max(len(last), 1)
    out[which] = {"FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
                  "hbm_bytes_per_launch_corrected": 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024}
    out[which]["over_minimum"] = out[which]["hbm_bytes_per_launch_corrected"] / out["minimum_bytes"]
json.dump(out, open(os.path.join(O, f"{R}_spmv_march_512_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
  rm -rf $O/sp_engine_* $O/sp_march_*
fi
cp $O/${R}_c3_256_pmc.json $O/${R}_c3_512_pmc.json profiles/      # (this box's copy: the bench line quotes `traffic` from a profile of the running build)
$T python bench.py --detail $O/${R}_bench_default_detail.json > $O/${R}_bench_default.json 2>$O/bench.err
fi
if [[ $PART == b || $PART == all ]]; then
  # C4 (eight sets, 512^3) iteration by iteration: with the slice-rank set on its lane (the product), and in turn on the engine
  # stream (kernel times that do not overlap: what each group costs alone)
  trace c4 c4_512 3:8 --steps 6 --warmup 2
  export SIPX_RANK_LANE=0
  trace c4 c4_512_in_turn 3:8 --steps 6 --warmup 2
  unset SIPX_RANK_LANE
  # ... and over 80 iterations: what a long solve of that list sustains (the slice-rank projector's route on inputs that keep moving)
  $T python bench.py $S --no-512 --config c4 --steps 80 --warmup 2 > $O/${R}_c4_512_bench_80_iterations.json 2>>$O/bench.err
  # fabric-side traffic of C4's kernels (the rank projector's GEMMs included): two counter passes of the c4 command
  B4="--no-cpu-baseline --no-512 --no-c4 --no-c5 --no-c2 --no-whole-call --no-kernel-table"
  $T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o t --output-format csv -- python3 bench.py $B4 --config c4 --steps 6 --warmup 2 > /dev/null 2>$O/pmc.err
  $T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o t --output-format csv -- python3 bench.py $B4 --config c4 --steps 6 --warmup 2 > /dev/null 2>>$O/pmc.err
  python tools/summarize_pmc.py $O/pmc_f $O/pmc_w $O/${R}_c4_512_pmc.json "bench.py $B4 --config c4 --steps 6 --warmup 2" > /dev/null
  rm -rf $O/pmc_f $O/pmc_w
  trace c2 c2_2048 6:25
  $T python bench.py $S --config c2 > $O/${R}_c2_2048_bench.json 2>>$O/bench.err
fi
if [[ $PART == c || $PART == all ]]; then
  $T python bench.py $S --no-512 --warmup 100 --steps 100 > $O/${R}_c3_256_bench_steady_it101_200.json 2>>$O/bench.err
  $T python bench.py $S --config c3-512 --warmup 100 --steps 60 > $O/${R}_c3_512_bench_steady_it101_160.json 2>>$O/bench.err
  $T python bench.py $S --no-512 --dtype f64 > $O/${R}_c3_256_f64_bench.json 2>>$O/bench.err
  $T python bench.py $S --dtype f64 --config c3-512 > $O/${R}_c3_512_f64_bench.json 2>>$O/bench.err
  $T python tools/c5_multilevel.py 512 100 > $O/${R}_c5_512_f64_multilevel.json 2>>$O/bench.err
  $T python tools/c5_multilevel.py 512 100 model=layered > $O/${R}_c5_512_f64_multilevel_layered.json 2>>$O/bench.err
  SIPX_FORCE_DIST=1 $T python tools/c5_multilevel.py 512 100 > $O/${R}_c5_512_f64_multilevel_rccl_world1_slab.json 2>>$O/bench.err
  SIPX_FORCE_DIST=1 $T python bench.py $S --decomp sets > $O/${R}_c3_bench_rccl_world1_sets.json 2>>$O/bench.err
  SIPX_FORCE_DIST=1 $T python bench.py $S --decomp slab > $O/${R}_c3_bench_rccl_world1_slab.json 2>>$O/bench.err
  # one rank's share of c3 / c3-512 on eight GPUs, slab-decomposed through RCCL with one rank, the sampled prediction forced as the
  # whole grid's size would switch it on: what the iteration costs a rank before any collective has a latency
  SIPX_L1_SAMPLE_RUNS=2048 SIPX_FORCE_DIST=1 $T python bench.py $S --no-512 --config c3-slab8 --decomp slab > $O/${R}_c3_slab8_share_rccl_world1.json 2>>$O/bench.err
  SIPX_L1_SAMPLE_RUNS=4096 SIPX_FORCE_DIST=1 $T python bench.py $S --no-512 --config c3-512-slab8 --decomp slab > $O/${R}_c3_512_slab8_share_rccl_world1.json 2>>$O/bench.err
  # BASELINE config 4's list with the WHOLE iteration on z-slabs (round 5): through RCCL with one rank at full size, and a rank's
  # share of eight (512 x 512 x 64: 64 slices for the rank set, every other set at 1 / 8); the same share over 40 iterations
  SIPX_FORCE_DIST=1 $T python bench.py $S --no-512 --config c4 --decomp slab --steps 6 --warmup 2 > $O/${R}_c4_512_slab_rccl_world1.json 2>>$O/bench.err
  SIPX_FORCE_DIST=1 $T python bench.py $S --no-512 --config c4 --decomp sets --steps 6 --warmup 2 > $O/${R}_c4_512_sets_rccl_world1.json 2>>$O/bench.err
  SIPX_FORCE_DIST=1 $T python bench.py $S --no-512 --config c4-slab8 --decomp slab --steps 6 --warmup 2 --detail $O/${R}_c4_slab8_share_rccl_world1_detail.json > $O/${R}_c4_slab8_share_rccl_world1.json 2>>$O/bench.err
  SIPX_FORCE_DIST=1 $T python bench.py $S --no-512 --config c4-slab8 --decomp slab --steps 40 --warmup 2 > $O/${R}_c4_slab8_share_rccl_world1_40its.json 2>>$O/bench.err
  # a chain of dependent small kernels, launched one by one and as a captured graph: what a launch-bound iteration (2048^2) can gain
  if [ -x scratch/launch_gap_bench ]; then $T scratch/launch_gap_bench > $O/${R}_launch_gap.txt 2>&1; fi
  # NOT a measurement: the N > 1 flow of the whole bench with four ranks on this one GPU (headline chain, every leg incl. c4_512_slab)
  SIPX_BENCH_SHARE_GPU=1 timeout -k 10 700 python bench.py --gpus 4 --no-c5 --no-512 --detail $O/${R}_bench_4ranks_share_one_gpu_rehearsal_detail.json > $O/${R}_bench_4ranks_share_one_gpu_rehearsal.json 2>>$O/bench.err
  # the slice-rank projector on slices without a spectral gap (C4's model), filtered subspace route against the full decomposition
  $T python tools/rank_flat_bench.py 512 32 12 > $O/${R}_rank_flat_512.json 2>>$O/bench.err
  $T python tools/rank_flat_bench.py 256 32 12 > $O/${R}_rank_flat_256.json 2>>$O/bench.err
  $T python bench.py $S --no-512 --config c3-768 --steps 6 --warmup 3 > $O/${R}_c3_768_bench.json 2>>$O/bench.err
fi
for f in $O/${R}_*bench*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], d.get("value"), r.get("avg_launch_ms"), r.get("frac"), (d.get("dominant_kernel") or {}).get("kernel"), (d.get("c3_512") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
