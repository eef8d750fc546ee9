set -x
mkdir -p gpurun_out/r5h
O=gpurun_out/r5h
for w in 0 4 8 12 16; do
SIPX_RANK_WINDOW=$w timeout -k 10 300 python tools/rank_probe.py c4 512 18 > $O/c4_512_w$w.json 2> $O/c4_512_w$w.err
done
SIPX_RANK_WINDOW=8 SIPX_RANK_GUARDS=32 timeout -k 10 300 python tools/rank_probe.py c4 512 18 > $O/c4_512_w8_g32.json 2> $O/c4_512_w8_g32.err
SIPX_RANK_WINDOW=8 SIPX_RANK_EPS=4.8e-7 timeout -k 10 300 python tools/rank_probe.py c4 512 18 > $O/c4_512_w8_eps21.json 2> $O/c4_512_w8_eps21.err
cat $O/*.json
du -sh gpurun_out
