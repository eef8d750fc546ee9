set -x
mkdir -p gpurun_out/r5f
O=gpurun_out/r5f
timeout -k 10 900 python -m pytest tests/test_gpu_round5.py -m gpu -q -p no:cacheprovider -x --timeout 600 > $O/pytest_round5.txt 2>&1
tail -15 $O/pytest_round5.txt
timeout -k 10 600 python bench.py --no-c5 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
cat $O/bench.json
cp bench_detail.json $O/ 2>/dev/null
du -sh gpurun_out
