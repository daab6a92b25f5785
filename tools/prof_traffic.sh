#!/bin/bash
# HBM traffic of the two hot kernels at the bench's full size: one rocprofv3 --pmc pass per counter
# (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Summaries land in <outdir>; tools/pmc_traffic.py
# turns them into profiles/pmc_traffic.json (bytes per launch) which bench.py reports as roofline.traffic.
out=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/$out
run() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d /root/repo/$out/$name -o $name -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-mode $BENCH_EXTRA > /root/repo/$out/$name.json 2> /root/repo/$out/$name.err || { echo "pass $name failed"; tail -3 /root/repo/$out/$name.err; return 1; }
}
run fetch FETCH_SIZE && run write WRITE_SIZE && python3 /root/repo/tools/pmc_summary.py /root/repo/$out 'k_inv|k_fwd'
