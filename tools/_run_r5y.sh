#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5y; mkdir -p $O
S="--no-cpu-baseline --no-c4 --no-c5 --no-c2 --no-whole-call --no-512 --no-kernel-table"
SIPX_EXT_DEBUG=2 SIPX_FORCE_DIST=1 timeout -k 10 400 python bench.py $S --config c4-slab8 --decomp slab --steps 10 --warmup 4 > $O/c4_slab8.json 2>$O/rank.err
grep "sipx rank" $O/rank.err | tail -24 | cut -c1-360
