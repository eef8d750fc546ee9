#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5r; mkdir -p $O
S="--no-cpu-baseline --no-c4 --no-c5 --no-c2 --no-whole-call --no-512"
timeout -k 10 120 scratch/launch_gap_bench > $O/launch_gap.txt 2>&1; cat $O/launch_gap.txt
SIPX_FORCE_DIST=1 timeout -k 10 400 python bench.py $S --config c4-slab8 --decomp slab --steps 6 --warmup 2 --detail $O/c4_slab8_detail.json > $O/c4_slab8_share_rccl_world1.json 2>$O/c4s.err; echo "rc=$?"
SIPX_FORCE_DIST=1 timeout -k 10 400 python bench.py $S --config c4 --decomp slab --steps 6 --warmup 2 --detail $O/c4_slab_w1_detail.json > $O/c4_slab_rccl_world1.json 2>>$O/c4s.err; echo "rc=$?"
SIPX_FORCE_DIST=1 timeout -k 10 400 python bench.py $S --config c4 --decomp sets --steps 6 --warmup 2 --detail $O/c4_sets_w1_detail.json > $O/c4_sets_rccl_world1.json 2>>$O/c4s.err; echo "rc=$?"
tail -5 $O/c4s.err
python - <<'PY'
import json
for f in ("c4_slab8_share_rccl_world1","c4_slab_rccl_world1","c4_sets_rccl_world1"):
    try:
        d=json.load(open(f"gpurun_out/r5r/{f}.json")); print(f, d.get("value"), d.get("ms_per_step"), d.get("comm"))
    except Exception as e: print(f, "unreadable", e)
PY
