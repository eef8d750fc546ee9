set -x
mkdir -p gpurun_out/r5j
O=gpurun_out/r5j
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x > $O/pytest_gpu.txt 2>&1
tail -8 $O/pytest_gpu.txt
du -sh gpurun_out
