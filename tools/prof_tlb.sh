#!/bin/bash
# UTCL1 (per-CU translation cache) counters of tools/exp_tlb.py, one rocprofv3 pass
R=/root/repo; out=$R/gpurun_out/tlb; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python3 $R/tools/exp_tlb.py > $out/plain.log 2>&1 || { tail -5 $out/plain.log; exit 1; }
cat $out/plain.log | grep pad
timeout -s KILL 200 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_THRASHING_STALL_sum --output-format csv -d $out/p1 -o p1 -- python3 $R/tools/exp_tlb.py > $out/p1.log 2>&1 || { tail -5 $out/p1.log; exit 1; }
grep pad $out/p1.log
