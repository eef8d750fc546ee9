#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5u; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "multilevel or c5" > $O/ml.txt 2>&1; tail -3 $O/ml.txt
timeout -k 10 300 python tools/c5_multilevel.py 256 100 > $O/c5_256_single.json 2>$O/c5.err; echo "rc=$?"
SIPX_BENCH_SHARE_GPU=1 timeout -k 10 700 python bench.py --gpus 4 --no-512 --no-c4 --detail $O/rehearsal4_detail.json > $O/rehearsal4.json 2> $O/rehearsal4.err
echo "rehearsal rc=$?"
python - <<'PY'
import json
a=json.load(open("gpurun_out/r5u/c5_256_single.json")); print("single", a.get("device_bytes_per_level"), a.get("whole_solve_s"), a.get("solve_only_s"))
d=json.load(open("gpurun_out/r5u/rehearsal4_detail.json"))
for k in ("c5","c5_layered"):
    v=d.get(k) or {}; print(k, v.get("error"), v.get("device_bytes_per_level"), v.get("sparse_arrays_per_level"), v.get("iterations_per_level"))
print(json.load(open("gpurun_out/r5u/rehearsal4.json")).get("c5"))
PY
