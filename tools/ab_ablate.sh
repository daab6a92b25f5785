#!/bin/bash
# inverse / forward times of the tuning build under a list of VW_ABLATE settings (one process each, same box)
# usage: tools/ab_ablate.sh "none ringnomath ringnoload" [bench args]
export VW_LIBVWGPU=${VW_LIBVWGPU:-/root/repo/vectorwave_amd/libvwgpu_tuning.so}
cd /root/repo
ABS=$1; shift
for ab in $ABS; do
  VW_ABLATE=$ab timeout -k 10 200 python bench.py --no-cpu-baseline --no-alt-mode --steps 8 --warmup 2 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$ab', 'fwd', d['forward_ms'], 'inv', d['inverse_ms'])" || exit 1
done
