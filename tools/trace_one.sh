#!/bin/bash
# One profiled run of the bench command on a GPU box: rocprofv3 --kernel-trace --stats + per-dispatch summary + timeline.
# usage (through gpurun, from the repo root): tools/trace_one.sh <config> <tag> [bench args...]   -> gpurun_out/prof/<tag>_*
set -e -o pipefail
cfg=$1; tag=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; mkdir -p $O
B="--no-cpu-baseline --no-512 --no-c4 --no-c5 --no-kernel-table"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/kt_$tag -o t --output-format csv -- python3 bench.py $B --config $cfg "$@" > $O/${tag}_bench_under_rocprof.json 2>$O/kt_$tag.err
python tools/summarize_kernel_trace.py $O/kt_$tag $O/${tag}_kernel_trace_summary.json > /dev/null
python tools/timeline.py $O/kt_$tag 0.75 > $O/${tag}_timeline.txt
cp $(find $O/kt_$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv
cp $(find $O/kt_$tag -name "*kernel_trace.csv" | head -1) $O/${tag}_kernel_trace.csv
rm -rf $O/kt_$tag
