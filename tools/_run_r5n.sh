set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5n
O=gpurun_out/r5n
B="--no-cpu-baseline --no-512 --no-c4 --no-c5 --no-c2 --no-whole-call --no-kernel-table"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o t --output-format csv -- python3 bench.py $B --config c2 > $O/c2_bench_under_rocprof.json 2>$O/kt.err
python tools/timeline.py $O/kt iters=6:25 > $O/c2_timeline.txt
python tools/iter_anatomy.py $(find $O/kt -name "*kernel_trace.csv" | head -1) 2 2 > $O/c2_iteration_anatomy.txt
rm -rf $O/kt
timeout -k 10 300 python bench.py $B --config c2 > $O/c2_bench.json 2>$O/c2.err
SIPX_MARK_STRIDE=7 timeout -k 10 300 python bench.py $B --config c2 > $O/c2_bench_stride7.json 2>>$O/c2.err
head -30 $O/c2_timeline.txt
cat $O/c2_iteration_anatomy.txt | head -70
cut -c1-120 $O/c2_bench.json $O/c2_bench_stride7.json
