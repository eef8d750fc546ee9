set -x
mkdir -p gpurun_out/r5i
O=gpurun_out/r5i
SIPX_EXT_DEBUG=1 timeout -k 10 300 python tools/rank_probe.py rank 512 8 > $O/rank512_dbg1.json 2> $O/rank512_dbg1.err
echo "rc=$?"
timeout -k 10 300 python tools/rank_probe.py c4 512 16 512 /tmp/x32.npy > $O/c4_512_f32.json 2> $O/c4_512_f32.err
SIPX_RANK_F32=0 timeout -k 10 300 python tools/rank_probe.py c4 512 16 512 /tmp/x64.npy > $O/c4_512_f64.json 2> $O/c4_512_f64.err
SIPX_RANK_STRICT=1 timeout -k 10 300 python tools/rank_probe.py c4 512 16 512 /tmp/xst.npy > $O/c4_512_strict.json 2> $O/c4_512_strict.err
python -c "
import numpy as np
a=np.load('/tmp/x32.npy').astype(np.float64); b=np.load('/tmp/x64.npy').astype(np.float64); c=np.load('/tmp/xst.npy').astype(np.float64)
print('rel diff of x after 16 iterations: f32 loop vs f64 loop', np.linalg.norm(a-b)/np.linalg.norm(b), ' f32 vs strict', np.linalg.norm(a-c)/np.linalg.norm(c), ' f64 vs strict', np.linalg.norm(b-c)/np.linalg.norm(c))
" > $O/xdiff.txt 2>&1
SIPX_EXT_DEBUG=2 SIPX_RANK_LANE=0 timeout -k 10 300 python tools/rank_probe.py c4 512 6 > $O/c4_512_dbg2.json 2> $O/c4_512_dbg2.err
timeout -k 10 300 python tools/rank_probe.py rank 64 24 > $O/rank64.json 2> $O/rank64.err
timeout -k 10 900 python -m pytest tests -m gpu -k "rank or c4 or C4 or library_backed or nuclear or round5" -q -p no:cacheprovider --timeout 600 > $O/pytest_rank.txt 2>&1
tail -8 $O/pytest_rank.txt
cat $O/*.json $O/xdiff.txt
du -sh gpurun_out
