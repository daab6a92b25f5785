#!/bin/bash
# A/B counter comparison: tools/prof_ab.sh <outdir> -- runs the SQ passes with the current environment
out=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/$out
run() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d /root/repo/$out/$name -o $name -- python3 /root/repo/bench.py --rows 8192 --steps 2 --warmup 1 --no-cpu-baseline --no-alt-mode > /root/repo/$out/$name.json 2> /root/repo/$out/$name.err || { echo "pass $name failed"; tail -3 /root/repo/$out/$name.err; return 1; }
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS &&
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
run sq3 SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_INT32 &&
python3 /root/repo/tools/pmc_summary.py /root/repo/$out 'k_inv|k_fwd'
