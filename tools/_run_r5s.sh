#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5s; mkdir -p $O
S="--no-cpu-baseline --no-c4 --no-c5 --no-c2 --no-whole-call --no-512"
for ov in 1 0; do
SIPX_FAN_OVERLAP=$ov SIPX_FORCE_DIST=1 timeout -k 10 400 python bench.py $S --config c4-slab8 --decomp slab --steps 6 --warmup 2 --detail $O/d.json > $O/c4_slab8_share_ov$ov.json 2>$O/c4s.err; echo "rc=$?"
SIPX_FAN_OVERLAP=$ov SIPX_FORCE_DIST=1 timeout -k 10 400 python bench.py $S --config c4 --decomp slab --steps 6 --warmup 2 --detail $O/d.json > $O/c4_slab_w1_ov$ov.json 2>>$O/c4s.err; echo "rc=$?"
done
SIPX_FORCE_DIST=1 timeout -k 10 400 python bench.py $S --config c4-slab8 --decomp slab --steps 40 --warmup 2 --detail $O/d.json > $O/c4_slab8_share_40its.json 2>>$O/c4s.err; echo "rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5s/c4_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], d.get("value"), d.get("ms_per_step"), (d.get("comm") or {}).get("device_bytes_per_rank"))
    except Exception as e: print(f, "unreadable", e)
PY
