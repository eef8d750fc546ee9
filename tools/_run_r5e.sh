set -x
mkdir -p gpurun_out/r5e
O=gpurun_out/r5e
for g in 24 32; do for f in 0.9 0.94 0.97; do
SIPX_RANK_GUARDS=$g SIPX_RANK_FLOOR=$f timeout -k 10 300 python tools/rank_probe.py rank 512 16 > $O/rank512_g${g}_f$f.json 2> $O/rank512_g${g}_f$f.err
done; done
SIPX_RANK_GUARDS=32 SIPX_RANK_CHEB_MMAX=24 timeout -k 10 300 python tools/rank_probe.py rank 512 16 > $O/rank512_g32_m24.json 2> $O/rank512_g32_m24.err
SIPX_RANK_GUARDS=32 timeout -k 10 300 python tools/rank_probe.py c4 512 16 > $O/c4_512_g32.json 2> $O/c4_512_g32.err
cat $O/*.json
du -sh gpurun_out
