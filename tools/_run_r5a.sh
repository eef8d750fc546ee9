set -x
mkdir -p gpurun_out/r5a
O=gpurun_out/r5a
SIPX_EXT_DEBUG=1 timeout -k 10 300 python tools/rank_probe.py rank 512 14 > $O/rank512_dbg1.json 2> $O/rank512_dbg1.err
echo "rc=$?" >> $O/rank512_dbg1.err
timeout -k 10 300 python tools/rank_probe.py c4 512 14 512 $O/x_new.npy > $O/c4_512.json 2> $O/c4_512.err
echo "rc=$?" >> $O/c4_512.err
SIPX_RANK_STRICT=1 SIPX_RANK_COLD=0 timeout -k 10 300 python tools/rank_probe.py c4 512 14 512 $O/x_old.npy > $O/c4_512_strict.json 2> $O/c4_512_strict.err
echo "rc=$?" >> $O/c4_512_strict.err
python -c "
import numpy as np
a=np.load('$O/x_new.npy').astype(np.float64); b=np.load('$O/x_old.npy').astype(np.float64)
print('rel diff of x after 14 iterations, new vs strict:', np.linalg.norm(a-b)/np.linalg.norm(b))
" > $O/xdiff.txt 2>&1
SIPX_EXT_DEBUG=2 SIPX_RANK_LANE=0 timeout -k 10 300 python tools/rank_probe.py c4 512 8 > $O/c4_512_dbg2.json 2> $O/c4_512_dbg2.err
timeout -k 10 300 python tools/rank_probe.py rank 64 24 > $O/rank64.json 2> $O/rank64.err
timeout -k 10 600 python -m pytest tests -m gpu -k "rank or c4 or C4 or library_backed or nuclear" -q -p no:cacheprovider > $O/pytest_rank.txt 2>&1
tail -5 $O/pytest_rank.txt
cat $O/c4_512.json $O/c4_512_strict.json $O/rank64.json $O/xdiff.txt
