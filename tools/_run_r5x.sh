#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5x; mkdir -p $O
timeout -k 10 500 python -m pytest tests -x -q -m gpu -k "dft_in_float64 or self_test or sharded_one_rank_through_rccl" > $O/t.txt 2>&1; tail -3 $O/t.txt | cut -c1-300
S="--no-cpu-baseline --no-c4 --no-c5 --no-c2 --no-whole-call --no-512"
SIPX_FORCE_DIST=1 timeout -k 10 400 python bench.py $S --config c4-slab8 --decomp slab --steps 6 --warmup 2 --detail $O/c4_slab8_detail.json > $O/c4_slab8.json 2>$O/c4s.err; echo "rc=$?"
SIPX_FORCE_DIST=1 timeout -k 10 400 python bench.py $S --config c4-slab8 --decomp slab --steps 40 --warmup 2 --detail $O/c4_slab8_40_detail.json > $O/c4_slab8_40.json 2>>$O/c4s.err; echo "rc=$?"
SIPX_FORCE_DIST=1 timeout -k 10 400 python bench.py $S --config c4 --decomp slab --steps 6 --warmup 2 --detail $O/c4_slab_w1_detail.json > $O/c4_slab_w1.json 2>>$O/c4s.err; echo "rc=$?"
SIPX_BENCH_SHARE_GPU=1 timeout -k 10 800 python bench.py --gpus 4 --no-c5 --no-512 --detail $O/rehearsal4_detail.json > $O/rehearsal4.json 2> $O/rehearsal4.err
echo "rehearsal rc=$?"
python - <<'PY'
import json
for f in ("c4_slab8_detail","c4_slab8_40_detail","c4_slab_w1_detail"):
    d=json.load(open(f"gpurun_out/r5x/{f}.json")); print(f, round(d["value"],2), round(d["ms_per_step"],2), d["comm"].get("collectives_per_step"), d["comm"].get("slab_loose"), d["comm"].get("device_bytes_per_rank"))
d=json.load(open("gpurun_out/r5x/rehearsal4.json"))
print(len(json.dumps(d)), {k:d.get(k) for k in ("value","c4_512_slab")})
d=json.load(open("gpurun_out/r5x/rehearsal4_detail.json"))
print(d["c4_512_slab"].get("comm") or d["c4_512_slab"])
PY
