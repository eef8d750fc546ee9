set -x
mkdir -p gpurun_out/r5b
O=gpurun_out/r5b
SIPX_EXT_DEBUG=1 timeout -k 10 300 python tools/rank_probe.py rank 512 12 > $O/rank512_dbg1.json 2> $O/rank512_dbg1.err
SIPX_EXT_DEBUG=2 SIPX_RANK_LANE=0 timeout -k 10 300 python tools/rank_probe.py c4 512 6 > $O/c4_512_dbg2.json 2> $O/c4_512_dbg2.err
SIPX_EXT_DEBUG=2 timeout -k 10 300 python tools/rank_probe.py rank 64 12 > $O/rank64_dbg2.json 2> $O/rank64_dbg2.err
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -k "test_rank_projection_subspace_route or (filtered_route_on_flat and 7-0)" -q -p no:cacheprovider > $O/pytest_rank.txt 2>&1
tail -5 $O/pytest_rank.txt
du -sh gpurun_out
