#!/bin/bash
OUT=gpurun_out/ab_wave4.log
: > $OUT
export VW_LIBVWGPU=$PWD/vectorwave_amd/libvwgpu_tuning.so
for ab in none wavent "wavent,ringnomath" ringnomath; do
  echo "--- $ab" >> $OUT
  VW_ABLATE=$ab timeout -k 10 120 python tools/ab_plan.py --rounds 3 wave:inv_wave=32 >> $OUT 2>&1 || exit 1
done
cat $OUT
