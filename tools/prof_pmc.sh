#!/bin/bash
# Collects rocprofv3 counter passes (one run per pass) for a short bench run.
# usage: tools/prof_pmc.sh <outdir> [bench args...]
out=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/$out
run() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d /root/repo/$out/$name -o $name -- python3 /root/repo/bench.py $BENCH_ARGS > /root/repo/$out/$name.json 2> /root/repo/$out/$name.err || { echo "pass $name failed"; tail -3 /root/repo/$out/$name.err; return 1; }
}
BENCH_ARGS="--rows 8192 --steps 2 --warmup 1 --no-cpu-baseline $*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS &&
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE &&
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
find /root/repo/$out -name "*.csv" | head -20
