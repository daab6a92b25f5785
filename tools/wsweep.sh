#!/bin/bash
cd /root/repo
for w in "$@"; do
  timeout -k 10 120 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline --wavelet $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', 'value', d['value'], 'fwd', d['forward_ms'], 'inv', d['inverse_ms'], d['bit_exact_vs_oracle'])" || exit 1
done
