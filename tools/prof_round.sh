#!/bin/bash
# One round's judged evidence, gathered into gpurun_out/<tag>/ (copy what is to be kept into profiles/<tag>/):
#   bench_default.json            python bench.py (defaults, CPU baselines included)
#   stats/                        rocprofv3 --kernel-trace --stats of the same command (without the CPU legs)
#   traffic/                      FETCH_SIZE and WRITE_SIZE passes (one rocprofv3 run each) of the same command
#   configs.json                  tools/bench_configs.py: every other configuration
#   c5/                           kernel stats of the short-row denoise configuration
# usage: tools/prof_round.sh <tag>
tag=$1
R=/root/repo
out=$R/gpurun_out/$tag
mkdir -p $out $out/traffic
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $out/bench_default.json 2> $out/bench_default.err || { echo "bench failed"; tail -5 $out/bench_default.err; exit 1; }
cut -c1-600 $out/bench_default.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 $R/bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.err || { echo "stats failed"; tail -5 $out/stats.err; exit 1; }
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); cut -c1-170 "$f" | head -8
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $out/traffic/$c -o $c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-mode > $out/traffic/$c.json 2> $out/traffic/$c.err || { echo "pass $c failed"; tail -3 $out/traffic/$c.err; exit 1; }
done
python3 $R/tools/pmc_traffic.py $out/traffic 65536 16384 6 db4 $out/pmc_traffic.json || exit 1
timeout -k 10 400 python3 $R/tools/bench_configs.py --out $out/configs.json > $out/configs.log 2>&1 || { echo "configs failed"; tail -5 $out/configs.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5 -o c5 -- python3 $R/tools/bench_configs.py --only C5 --steps 20 > $out/c5.log 2>&1 || { echo "c5 failed"; tail -5 $out/c5.log; exit 1; }
grep "vw::" $out/c5/*kernel_stats.csv $out/c5/*/*kernel_stats.csv 2>/dev/null | cut -d, -f1-4 | cut -c1-200
