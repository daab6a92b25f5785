#!/bin/bash
# A/B of the wave-private inverse against the ring kernel, then its ablations (tuning build: wrong results by design).
set -e
OUT=gpurun_out/ab_wave.log
: > $OUT
timeout -k 10 240 python tools/ab_plan.py --rounds 3 base: wave32:inv_wave=32 wave16:inv_wave=16 wave8:inv_wave=8 >> $OUT 2>&1
export VW_LIBVWGPU=$PWD/vectorwave_amd/libvwgpu_tuning.so
echo "--- nomath" >> $OUT
VW_ABLATE=ringnomath timeout -k 10 120 python tools/ab_plan.py --rounds 2 wave32:inv_wave=32 >> $OUT 2>&1
echo "--- noload" >> $OUT
VW_ABLATE=ringnoload timeout -k 10 120 python tools/ab_plan.py --rounds 2 wave32:inv_wave=32 >> $OUT 2>&1
echo "--- nomath ring" >> $OUT
VW_ABLATE=ringnomath timeout -k 10 120 python tools/ab_plan.py --rounds 2 base: >> $OUT 2>&1
cat $OUT
