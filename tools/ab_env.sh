#!/bin/bash
# A/B of two environment settings on one box (alternating): tools/ab_env.sh "<VAR=a>" "<VAR=b>" [rounds] [bench args]
A=$1; B=$2; R=${3:-2}; shift 3
run() { echo "== $1"; env $1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt-mode --steps 12 --warmup 3 "${@:2}" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['forward_ms'], d['inverse_ms'], d['value'])"; }
for i in $(seq $R); do run "$A" "$@" && run "$B" "$@" || exit 1; done
