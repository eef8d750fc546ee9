set -x
mkdir -p gpurun_out/r5d
O=gpurun_out/r5d
SIPX_EXT_DEBUG=4 timeout -k 10 300 python tools/rank_probe.py rank 512 8 > $O/rank512_dbg4.json 2> $O/rank512_dbg4.err
timeout -k 10 300 python tools/rank_probe.py rank 512 16 > $O/rank512_keep1.json 2> $O/rank512_keep1.err
SIPX_RANK_KEEP=0 timeout -k 10 300 python tools/rank_probe.py rank 512 16 > $O/rank512_keep0.json 2> $O/rank512_keep0.err
timeout -k 10 300 python tools/rank_probe.py c4 512 16 > $O/c4_512_keep1.json 2> $O/c4_512_keep1.err
SIPX_RANK_KEEP=0 timeout -k 10 300 python tools/rank_probe.py c4 512 16 > $O/c4_512_keep0.json 2> $O/c4_512_keep0.err
timeout -k 10 300 python tools/rank_probe.py rank 64 24 > $O/rank64.json 2> $O/rank64.err
cat $O/*.json
du -sh gpurun_out
