#!/bin/bash
# needs the tuning build (make -C vectorwave_amd/csrc tuning): only that build reads the VW_* overrides
export VW_LIBVWGPU=${VW_LIBVWGPU:-/root/repo/vectorwave_amd/libvwgpu_tuning.so}
# usage: tools/tune.sh "<fwd tile>:<inv tile>" ...; runs bench for each setting
cd /root/repo
for cfg in "$@"; do
  f=${cfg%%:*}; i=${cfg##*:}
  VW_FWD_TILE=$f VW_INV_TILE=$i timeout -k 10 120 python -u bench.py --steps ${STEPS:-10} --warmup ${WARMUP:-4} --no-cpu-baseline $BENCH_EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); a=d.get('alt_mode') or {}; print('$cfg', d['arith'], 'value', d['value'], 'fwd', d['forward_ms'], 'inv', d['inverse_ms'], d['bit_exact_vs_oracle'], d['max_rel_dev_vs_oracle'], '| alt', a.get('arith'), a.get('forward_ms'), a.get('inverse_ms'), a.get('Gsamples_per_s_per_gpu'), a.get('max_rel_dev_vs_oracle'))" || exit 1
done
