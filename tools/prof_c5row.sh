#!/bin/bash
# SQ counters of the row-resident denoise kernel on the config-5 shape: tools/prof_c5row.sh <outdir>
out=$1
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/$out
run() {
  pass=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d /root/repo/$out/$pass -o $pass -- python3 /root/repo/tools/ab_c5.py --rounds 1 --reps 3 row: > /root/repo/$out/$pass.log 2>&1 || { echo "pass $pass failed"; tail -3 /root/repo/$out/$pass.log; return 1; }
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS &&
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
run sq3 SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVES GRBM_GUI_ACTIVE &&
python3 /root/repo/tools/pmc_summary.py /root/repo/$out 'k_swt_denoise_row'
