set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5g
O=gpurun_out/r5g
B="--no-cpu-baseline --no-512 --no-c4 --no-c5 --no-c2 --no-whole-call --no-kernel-table"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o t --output-format csv -- python3 bench.py $B > $O/c3_256_bench_under_rocprof.json 2>$O/kt.err
python tools/timeline.py $O/kt iters=6:25 > $O/c3_256_timeline_timed_steps.txt
python tools/timeline.py $O/kt 0.75 > $O/c3_256_timeline_last75.txt
rm -rf $O/kt
timeout -k 10 300 rocprofv3 --kernel-trace --hip-trace --memory-copy-trace -d $O/ht -o t --output-format csv -- python3 bench.py $B > $O/c3_256_bench_under_hiptrace.json 2>$O/ht.err
ls -R $O/ht | head -20
python tools/gap_inspect.py $O/ht k_spec_finish k_yl_multi 100 3 > $O/c3_256_gap_inspect.txt 2>&1
python tools/timeline.py $O/ht iters=6:25 > $O/c3_256_timeline_timed_steps_hiptrace.txt
rm -rf $O/ht
export SIPX_RANK_LANE=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt4 -o t --output-format csv -- python3 bench.py --no-cpu-baseline --no-512 --no-c4 --no-c5 --no-c2 --no-whole-call --no-kernel-table --config c4 --steps 6 --warmup 2 > $O/c4_512_in_turn_bench_under_rocprof.json 2>$O/kt4.err
python tools/c4_iter_groups.py $(find $O/kt4 -name "*kernel_trace.csv" | head -1) > $O/c4_512_in_turn_iteration_groups.txt
cp $(find $O/kt4 -name "*kernel_stats.csv" | head -1) $O/c4_512_in_turn_kernel_stats.csv
rm -rf $O/kt4
unset SIPX_RANK_LANE
timeout -k 10 300 python bench.py --config c2 --no-cpu-baseline --no-c4 --no-c5 > $O/c2_bench.json 2>$O/c2.err
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py -m gpu -q -p no:cacheprovider --timeout 600 > $O/pytest_round5.txt 2>&1
tail -12 $O/pytest_round5.txt
cat $O/c3_256_timeline_timed_steps.txt | head -24
cat $O/c3_256_gap_inspect.txt | head -60
cat $O/c4_512_in_turn_iteration_groups.txt
cat $O/c2_bench.json | cut -c1-300
du -sh gpurun_out
