set -x
mkdir -p gpurun_out/r5m
O=gpurun_out/r5m
for f in 0.9 0.7 0.5; do for g in 2 0; do
SIPX_RANK_FLOOR=$f SIPX_RANK_CHEB_GUARD=$g timeout -k 10 300 python tools/rank_probe.py c4 512 20 > $O/c4_f${f}_g$g.json 2> $O/c4_f${f}_g$g.err
done; done
cat $O/*.json
du -sh gpurun_out
