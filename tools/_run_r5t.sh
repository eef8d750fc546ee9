#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5t; mkdir -p $O
S="--no-cpu-baseline --no-c4 --no-c5 --no-c2 --no-whole-call --no-512"
SIPX_FORCE_DIST=1 timeout -k 10 400 python bench.py $S --config c4 --decomp slab --steps 6 --warmup 2 --detail $O/c4_slab_w1_detail.json > $O/c4_slab_rccl_world1.json 2>$O/c4s.err; echo "rc=$?"
SIPX_BENCH_SHARE_GPU=1 timeout -k 10 700 python bench.py --gpus 4 --no-c5 --no-512 --detail $O/rehearsal4_detail.json > $O/rehearsal4.json 2> $O/rehearsal4.err
echo "rehearsal rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5t/c4_slab_w1_detail.json")); print("w1", d["value"], d["comm"].get("sparse_arrays"), d["comm"].get("device_bytes_per_rank"), d["comm"].get("slab_loose"))
d=json.load(open("gpurun_out/r5t/rehearsal4.json"))
print(len(json.dumps(d)), {k:d.get(k) for k in ("value","c4_512_slab")})
d=json.load(open("gpurun_out/r5t/rehearsal4_detail.json"))
print(d["c4_512_slab"]["comm"])
PY
