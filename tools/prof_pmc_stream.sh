#!/bin/bash
# SQ counter passes over tools/exp_stream_one.py (one run per pass).  usage: tools/prof_pmc_stream.sh <outdir> <args of exp_stream_one.py>
out=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/$out
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d /root/repo/$out/$name -o $name -- python3 /root/repo/tools/exp_stream_one.py $ARGS > /root/repo/$out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 /root/repo/$out/$name.log; return 1; }
}
ARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS &&
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
find /root/repo/$out -name "*counter_collection.csv" | head
