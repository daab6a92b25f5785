#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command; CSV summaries land in <outdir>.
out=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/$out
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out -o stats -- python3 /root/repo/bench.py --no-cpu-baseline "$@" > /root/repo/$out/bench.json 2> /root/repo/$out/bench.err
echo EXIT $?
cat /root/repo/$out/bench.json
find /root/repo/$out -name "*.csv" | head
