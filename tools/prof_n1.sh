#!/bin/bash
# kernel stats of the WaveletDenoiser configurations and of C5 (tools/bench_configs.py --only N1 / C5)
R=/root/repo; out=$R/gpurun_out/n1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/n1 -o n1 -- python3 $R/tools/bench_configs.py --only N1 --steps 5 > $out/n1.log 2>&1 || { tail -5 $out/n1.log; exit 1; }
grep "^N1" $out/n1.log | cut -c1-150
grep "vw::" $out/n1/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-220
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5 -o c5 -- python3 $R/tools/bench_configs.py --only C5 --steps 20 > $out/c5.log 2>&1 || { tail -5 $out/c5.log; exit 1; }
grep "^C5" $out/c5.log | cut -c1-200
grep "vw::" $out/c5/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-220
