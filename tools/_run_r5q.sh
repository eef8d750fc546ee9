#!/bin/bash
# round 5: C4's list with the whole iteration on z-slabs -- 4 ranks sharing the GPU (rehearsal), and a rank's share of eight
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5q; mkdir -p $O
SIPX_BENCH_SHARE_GPU=1 timeout -k 10 700 python bench.py --gpus 4 --no-c5 --no-512 --detail $O/rehearsal4_detail.json > $O/rehearsal4.json 2> $O/rehearsal4.err
echo "rehearsal rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5q/rehearsal4.json"))
print({k:d.get(k) for k in ("value","c4_512","c4_512_slab","decompositions")})
PY
