#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5w; mkdir -p $O
SIPX_SPEC_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-512 --no-c4 --no-c5 --no-c2 --no-whole-call --no-kernel-table --config c4-256 --steps 14 --warmup 2 > $O/c4_256.json 2> $O/spec.err
grep -c "batched chain" $O/spec.err
grep "set \|theta " $O/spec.err | cut -c1-230 | head -150 > $O/spec_short.txt
wc -l $O/spec_short.txt
