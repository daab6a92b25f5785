#!/bin/bash
# A/B of two builds of the library on one box (alternating): tools/ab_lib.sh <libA> <libB> [bench args]
A=$1; B=$2; shift 2
run() { echo "== $1"; VW_LIBVWGPU=$1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-alt-mode --steps 8 --warmup 3 "${@:2}" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['forward_ms'], d['inverse_ms'], d['value'])"; }
run $A "$@" && run $B "$@" && run $A "$@" && run $B "$@"
