#!/bin/bash
# rocprofv3 kernel trace + stats of tools/bench_configs.py restricted to one configuration family.
# usage: tools/prof_cfg_stats.sh <outdir> <only> [steps]
out=$1; only=$2; steps=${3:-20}
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/$out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out -o stats -- python3 /root/repo/tools/bench_configs.py --only "$only" --steps $steps --warmup 2 > /root/repo/$out/run.log 2>&1
echo EXIT $?
f=$(find /root/repo/$out -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-150 "$f" | head -12
