#!/bin/bash
# One rocprofv3 --pmc pass over a short bench run (through gpurun, from the repo root): tools/pmc_pass.sh <tag> "<counters>" [bench args...]
# Counters are collected with --kernel-trace only (never with other trace domains); output: gpurun_out/prof/<tag>_pmc.csv
set -e -o pipefail
TAG=$1; CNT=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; mkdir -p $O
rm -rf $O/pmc_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --pmc $CNT -d $O/pmc_$TAG -o t --output-format csv -- python3 bench.py --no-cpu-baseline --no-512 --no-c4 --steps 6 --warmup 2 "$@" > /dev/null 2>$O/pmc_$TAG.err
python tools/pmc_table.py $O/pmc_$TAG > $O/${TAG}_pmc.txt
rm -rf $O/pmc_$TAG
