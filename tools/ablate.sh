#!/bin/bash
# needs the tuning build (make -C vectorwave_amd/csrc tuning): only that build reads the VW_* overrides
export VW_LIBVWGPU=${VW_LIBVWGPU:-/root/repo/vectorwave_amd/libvwgpu_tuning.so}
cd /root/repo
for ab in $3; do
  VW_ABLATE=$ab VW_FWD_TILE=$1 VW_INV_TILE=$2 timeout -k 10 120 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline $BENCH_EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); a=d.get('alt_mode') or {}; print('$ab $1 $2', 'fwd', d['forward_ms'], 'inv', d['inverse_ms'], '| alt', a.get('forward_ms'), a.get('inverse_ms'))" || exit 1
done
