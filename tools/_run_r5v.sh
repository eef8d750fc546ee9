#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5v; mkdir -p $O
SIPX_BENCH_C5_N=512 SIPX_BENCH_SHARE_GPU=1 timeout -k 10 900 python bench.py --gpus 4 --no-512 --no-c4 --detail $O/rehearsal4_c5_512_detail.json > $O/rehearsal4_c5_512.json 2> $O/rehearsal4.err
echo "rehearsal rc=$?"
tail -3 $O/rehearsal4.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5v/rehearsal4_c5_512_detail.json"))
for k in ("c5","c5_layered"):
    v=d.get(k) or {}; print(k, v.get("error"), v.get("grid"), v.get("device_bytes_per_level"), v.get("sparse_arrays_per_level"), v.get("iterations_per_level"), v.get("whole_solve_s"))
PY
