#!/bin/bash
# A/B of one environment switch on one box: tools/ab_env.sh <rounds> <config> VAR=a VAR=b ...   (prints it/s per setting and round)
R=$1; C=$2; shift 2
for r in $(seq $R); do for kv in "$@"; do
  v=$(env $kv python bench.py --no-cpu-baseline --no-512 --no-c4 --config $C 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
  echo "$r $kv $v"
done; done
