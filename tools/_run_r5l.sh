set -x
mkdir -p gpurun_out/r5l
O=gpurun_out/r5l
SIPX_BENCH_SHARE_GPU=1 timeout -k 10 560 python bench.py --gpus 4 --detail $O/r05_bench_4ranks_share_one_gpu_rehearsal_detail.json > $O/r05_bench_4ranks_share_one_gpu_rehearsal.json 2> $O/rehearsal4.err
echo "rehearsal rc=$?"
cat $O/r05_bench_4ranks_share_one_gpu_rehearsal.json | cut -c1-3000
tail -5 $O/rehearsal4.err | cut -c1-300
du -sh gpurun_out
