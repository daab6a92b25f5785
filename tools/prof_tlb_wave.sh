#!/bin/bash
# UTCL1 (per-CU translation cache) counters of the inverse kernels: the wave-private kernel (1280 rows walked at one
# time) against the ring kernel (64 rows at one time)
R=/root/repo; out=$R/gpurun_out/tlbw; mkdir -p $out
export VW_LIBVWGPU=$R/vectorwave_amd/libvwgpu_tuning.so
cd /tmp && export TMPDIR=/tmp
for v in wave ring; do
  if [ $v = wave ]; then export VW_INV_WAVE=32; else export VW_INV_WAVE=-1; fi
  timeout -s KILL 200 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_THRASHING_STALL_sum --output-format csv -d $out/$v -o $v -- python3 $R/bench.py --rows 16384 --steps 2 --warmup 1 --no-cpu-baseline --no-alt-mode > $out/$v.json 2> $out/$v.err || { tail -3 $out/$v.err; exit 1; }
  python3 $R/tools/pmc_summary.py $out/$v 'k_inv|k_fwd'
done
