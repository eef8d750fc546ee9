#!/bin/bash
# Regenerates the round's profile files on a GPU box (run through gpurun from the repo root); outputs under gpurun_out/prof/.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; mkdir -p $O
T="timeout -k 10 400"
# PMC passes first: bench.py reads profiles/r01_c3_256_pmc.json for roofline.traffic
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o t --output-format csv -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>$O/pmc.err
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o t --output-format csv -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>>$O/pmc.err
python tools/summarize_pmc.py $O/pmc_f $O/pmc_w $O/r01_c3_256_pmc.json "bench.py --no-cpu-baseline --steps 6 --warmup 2" > /dev/null
cp $O/r01_c3_256_pmc.json profiles/r01_c3_256_pmc.json
rm -rf $O/pmc_f $O/pmc_w
$T rocprofv3 --kernel-trace --stats -d $O/kt -o c3 --output-format csv -- python3 bench.py --no-cpu-baseline > $O/r01_c3_256_bench_under_rocprof.json 2>$O/kt.err
python tools/summarize_kernel_trace.py $O/kt $O/r01_c3_256_kernel_trace_summary.json > /dev/null
python tools/timeline.py $O/kt 0.75 > $O/r01_c3_256_timeline.txt
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/r01_c3_256_kernel_stats_final.csv
rm -rf $O/kt
$T python bench.py > $O/r01_c3_256_bench_final.json 2>$O/bench.err
$T python bench.py --no-cpu-baseline --warmup 100 --steps 100 > $O/r01_c3_256_bench_steady_it101_200.json 2>>$O/bench.err
$T python bench.py --no-cpu-baseline --config c3-512 > $O/r01_c3_512_bench.json 2>>$O/bench.err
$T python bench.py --no-cpu-baseline --config c3-512 --warmup 100 --steps 60 > $O/r01_c3_512_bench_steady_it101_160.json 2>>$O/bench.err
$T python bench.py --no-cpu-baseline --config c3-768 --steps 6 --warmup 3 > $O/r01_c3_768_bench.json 2>>$O/bench.err
$T python bench.py --no-cpu-baseline --config c2 > $O/r01_c2_2048_bench.json 2>>$O/bench.err
$T python bench.py --no-cpu-baseline --dtype f64 > $O/r01_c3_256_f64_bench.json 2>>$O/bench.err
$T python bench.py --no-cpu-baseline --dtype f64 --config c3-512 > $O/r01_c3_512_f64_bench.json 2>>$O/bench.err
$T python bench.py --no-cpu-baseline --q-mode stencil > $O/r01_c3_256_bench_stencilQ.json 2>>$O/bench.err
$T python bench.py --no-cpu-baseline --q-mode stencil --config c3-512 > $O/r01_c3_512_bench_stencilQ.json 2>>$O/bench.err
$T python bench.py --no-cpu-baseline --config c4-256 --steps 10 --warmup 3 > $O/r01_c4_256_bench.json 2>>$O/bench.err
$T python bench.py --no-cpu-baseline --config c4 --steps 6 --warmup 2 > $O/r01_c4_512_bench.json 2>>$O/bench.err
$T python tools/c5_multilevel.py 512 30 > $O/r01_c5_512_f64_multilevel.json 2>>$O/bench.err
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], d.get("value"), (d.get("roofline") or {}).get("avg_launch_ms"), (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
