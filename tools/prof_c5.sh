#!/bin/bash
# kernel durations of the config-5 denoise (8192 x 480, J=4): rocprofv3 kernel trace of tools/bench_configs.py --only C5
out=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/$out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out -o c5 -- python3 /root/repo/tools/bench_configs.py --only C5 --steps 20 > /root/repo/$out/c5.log 2>&1
grep "vw::" /root/repo/$out/c5_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
