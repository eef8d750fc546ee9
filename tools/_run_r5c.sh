set -x
mkdir -p gpurun_out/r5c
O=gpurun_out/r5c
SIPX_EXT_DEBUG=4 timeout -k 10 300 python tools/rank_probe.py rank 512 7 > $O/rank512_dbg4.json 2> $O/rank512_dbg4.err
for f in 0.9 0.86 0.82; do
SIPX_RANK_FLOOR=$f timeout -k 10 300 python tools/rank_probe.py rank 512 16 > $O/rank512_floor$f.json 2> $O/rank512_floor$f.err
done
SIPX_RANK_GUARDS=16 timeout -k 10 300 python tools/rank_probe.py rank 512 16 > $O/rank512_guards16.json 2> $O/rank512_guards16.err
SIPX_RANK_GUARDS=32 timeout -k 10 300 python tools/rank_probe.py rank 512 16 > $O/rank512_guards32.json 2> $O/rank512_guards32.err
cat $O/*.json
du -sh gpurun_out
