#!/bin/bash
OUT=gpurun_out/ab_wave3.log
: > $OUT
timeout -k 10 300 python tools/ab_plan.py --rounds 3 base: wave32:inv_wave=32 rot:inv_wave=32,inv_wave_rot=1 w4:inv_wave=32,inv_wave_pad=6000 w4rot:inv_wave=32,inv_wave_pad=6000,inv_wave_rot=1 w3:inv_wave=32,inv_wave_pad=20000 >> $OUT 2>&1
export VW_LIBVWGPU=$PWD/vectorwave_amd/libvwgpu_tuning.so
echo "--- nomath" >> $OUT
VW_ABLATE=ringnomath timeout -k 10 300 python tools/ab_plan.py --rounds 2 wave32:inv_wave=32 rot:inv_wave=32,inv_wave_rot=1 w4:inv_wave=32,inv_wave_pad=6000 w4rot:inv_wave=32,inv_wave_pad=6000,inv_wave_rot=1  w3:inv_wave=32,inv_wave_pad=20000 >> $OUT 2>&1
cat $OUT
