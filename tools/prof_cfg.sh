#!/bin/bash
# kernel durations of one configuration of tools/bench_configs.py: tools/prof_cfg.sh <outdir> "<--only pattern>"
out=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/$out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out -o cfg -- python3 /root/repo/tools/bench_configs.py --only "$1" --steps 5 > /root/repo/$out/cfg.log 2>&1
grep "vw::" /root/repo/$out/cfg_kernel_stats.csv | awk -F'",' '{print substr($1,1,75), $2}' | cut -c1-150
