#!/bin/bash
# ablations of the wave-private inverse (tuning build: wrong results by design), then its SQ counters
OUT=gpurun_out/ab_wave2.log
: > $OUT
export VW_LIBVWGPU=$PWD/vectorwave_amd/libvwgpu_tuning.so
for ab in none ringnomath ringnoload "ringnomath,ringnoload"; do
  echo "--- $ab" >> $OUT
  VW_ABLATE=$ab timeout -k 10 120 python tools/ab_plan.py --rounds 2 wave32:inv_wave=32 >> $OUT 2>&1 || exit 1
done
cat $OUT
VW_INV_WAVE=32 bash tools/prof_inv.sh gpurun_out/prof_wave wave32 "" 
