#!/bin/bash
cd /root/repo
for ab in none nostore nocompute nostore,nocompute; do
  VW_ABLATE=$ab VW_FWD_TILE=$1 VW_INV_TILE=$2 timeout -k 10 120 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ab $1 $2', 'fwd', d['forward_ms'], 'inv', d['inverse_ms'])" || exit 1
done
