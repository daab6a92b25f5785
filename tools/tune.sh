#!/bin/bash
# usage: tools/tune.sh "<fwd tile>" "<inv tile>" ... pairs; runs bench for each setting
cd /root/repo
for cfg in "$@"; do
  f=${cfg%%:*}; i=${cfg##*:}
  VW_FWD_TILE=$f VW_INV_TILE=$i timeout -k 10 120 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', 'value', d['value'], 'fwd', d['forward_ms'], 'inv', d['inverse_ms'], d['bit_exact_vs_oracle'])" || exit 1
done
