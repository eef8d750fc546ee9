#!/bin/bash
# Regenerates the round's profile files on a GPU box (run through gpurun from the repo root); outputs under gpurun_out/prof/,
# to be copied into profiles/ by the caller.  usage: tools/refresh_profiles.sh [round tag, default r02] [light]
set -e -o pipefail
R=${1:-r02}
LIGHT=${2:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; mkdir -p $O
T="timeout -k 10 500"
pmc_pair() {   # <config> <tag>: the two HBM-side passes (FETCH_SIZE / WRITE_SIZE in separate runs, --kernel-trace only)
  $T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o t --output-format csv -- python3 bench.py --no-cpu-baseline --no-512 --no-c4 --config $1 --steps 6 --warmup 2 > /dev/null 2>$O/pmc.err
  $T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o t --output-format csv -- python3 bench.py --no-cpu-baseline --no-512 --no-c4 --config $1 --steps 6 --warmup 2 > /dev/null 2>>$O/pmc.err
  python tools/summarize_pmc.py $O/pmc_f $O/pmc_w $O/${R}_$2_pmc.json "bench.py --no-cpu-baseline --no-512 --no-c4 --config $1 --steps 6 --warmup 2" > /dev/null
  cp $O/${R}_$2_pmc.json profiles/${R}_$2_pmc.json        # bench.py reads it for roofline.traffic / frac_traffic
  rm -rf $O/pmc_f $O/pmc_w
}
trace() {      # <config> <tag> [extra bench args]: rocprofv3 --kernel-trace --stats of the bench command + per-dispatch summary + timeline
  local cfg=$1 tag=$2; shift 2
  $T rocprofv3 --kernel-trace --stats -d $O/kt -o t --output-format csv -- python3 bench.py --no-cpu-baseline --no-512 --no-c4 --config $cfg "$@" > $O/${R}_${tag}_bench_under_rocprof.json 2>$O/kt.err
  python tools/summarize_kernel_trace.py $O/kt $O/${R}_${tag}_kernel_trace_summary.json > /dev/null
  python tools/timeline.py $O/kt 0.75 > $O/${R}_${tag}_timeline.txt
  cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/${R}_${tag}_kernel_stats.csv
  rm -rf $O/kt
}
pmc_pair c3 c3_256
pmc_pair c3-512 c3_512
trace c3 c3_256
trace c3-512 c3_512 --steps 10 --warmup 5
# SQ counters (VALU / LDS / waves) of every kernel, two passes of 8 SQ slots each
tools/pmc_pass.sh ${R}_c3_256_sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
tools/pmc_pass.sh ${R}_c3_256_sq2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
$T python bench.py > $O/${R}_c3_256_bench_final.json 2>$O/bench.err
if [ -z "$LIGHT" ]; then
  trace c2 c2_2048
  $T python bench.py --no-cpu-baseline --no-512 --no-c4 --warmup 100 --steps 100 > $O/${R}_c3_256_bench_steady_it101_200.json 2>>$O/bench.err
  $T python bench.py --no-cpu-baseline --no-c4 --config c3-512 --warmup 100 --steps 60 > $O/${R}_c3_512_bench_steady_it101_160.json 2>>$O/bench.err
  $T python bench.py --no-cpu-baseline --no-c4 --config c2 > $O/${R}_c2_2048_bench.json 2>>$O/bench.err
  $T python bench.py --no-cpu-baseline --no-512 --no-c4 --dtype f64 > $O/${R}_c3_256_f64_bench.json 2>>$O/bench.err
  $T python bench.py --no-cpu-baseline --no-c4 --dtype f64 --config c3-512 > $O/${R}_c3_512_f64_bench.json 2>>$O/bench.err
  $T python bench.py --no-cpu-baseline --no-c4 --config c4-256 --steps 10 --warmup 3 > $O/${R}_c4_256_bench.json 2>>$O/bench.err
  $T python bench.py --no-cpu-baseline --no-c4 --config c4 --steps 6 --warmup 2 > $O/${R}_c4_512_bench.json 2>>$O/bench.err
  $T python tools/c5_multilevel.py 512 30 > $O/${R}_c5_512_f64_multilevel.json 2>>$O/bench.err
  $T python tools/c5_multilevel.py 512 30 host > $O/${R}_c5_512_f64_multilevel_host_transfers.json 2>>$O/bench.err
  SIPX_FORCE_DIST=1 $T python tools/c5_multilevel.py 512 30 > $O/${R}_c5_512_f64_multilevel_rccl_world1_slab.json 2>>$O/bench.err
  SIPX_FORCE_DIST=1 $T python bench.py --no-cpu-baseline --no-512 --no-c4 --decomp sets > $O/${R}_c3_256_bench_rccl_world1.json 2>>$O/bench.err
  SIPX_FORCE_DIST=1 $T python bench.py --no-cpu-baseline --no-512 --no-c4 --decomp slab > $O/${R}_c3_256_bench_rccl_world1_slab.json 2>>$O/bench.err
fi
for f in $O/${R}_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], d.get("value"), r.get("avg_launch_ms"), r.get("frac"), r.get("frac_traffic"), (d.get("c3_512") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
