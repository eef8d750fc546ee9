set -x
mkdir -p gpurun_out/r5k
O=gpurun_out/r5k
for jw in 1 0; do
SIPX_RANK_JACOBI_WAVE=$jw timeout -k 10 300 python tools/rank_probe.py c4 512 16 512 /tmp/x$jw.npy > $O/c4_512_jw$jw.json 2> $O/c4_512_jw$jw.err
SIPX_RANK_JACOBI_WAVE=$jw timeout -k 10 300 python tools/rank_probe.py rank 64 24 > $O/rank64_jw$jw.json 2> $O/rank64_jw$jw.err
done
python -c "
import numpy as np
a=np.load('/tmp/x1.npy').astype(np.float64); b=np.load('/tmp/x0.npy').astype(np.float64)
print('rel diff of x after 16 iterations: wave Jacobi vs 256-thread Jacobi', np.linalg.norm(a-b)/np.linalg.norm(b))
" > $O/xdiff.txt 2>&1
SIPX_EXT_DEBUG=2 SIPX_RANK_LANE=0 timeout -k 10 300 python tools/rank_probe.py c4 512 6 > $O/c4_512_dbg2.json 2> $O/c4_512_dbg2.err
SIPX_TRACE_SEARCHES=1 timeout -k 10 300 python tools/rank_probe.py c4 512 9 > $O/c4_512_trace_searches.json 2> $O/c4_512_trace_searches.err
timeout -k 10 900 python -m pytest tests -m gpu -k "rank or c4 or C4 or library_backed or nuclear or round5" -q -p no:cacheprovider --timeout 600 > $O/pytest_rank.txt 2>&1
tail -6 $O/pytest_rank.txt
cat $O/*.json $O/xdiff.txt
grep -c "sipx search" $O/c4_512_trace_searches.err
du -sh gpurun_out
