#!/bin/bash
# Kernel trace + timeline of one bench configuration (run through gpurun from the repo root): tools/quick_trace.sh <tag> [bench args...]
set -e -o pipefail
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; mkdir -p $O
rm -rf $O/kt_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/kt_$TAG -o t --output-format csv -- python3 bench.py --no-cpu-baseline --no-512 --no-c4 "$@" > $O/${TAG}_bench_under_rocprof.json 2>$O/${TAG}_kt.err
python tools/summarize_kernel_trace.py $O/kt_$TAG $O/${TAG}_kernel_trace_summary.json > /dev/null
python tools/timeline.py $O/kt_$TAG 0.75 > $O/${TAG}_timeline.txt
cp $(find $O/kt_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
rm -rf $O/kt_$TAG
cat $O/${TAG}_timeline.txt
