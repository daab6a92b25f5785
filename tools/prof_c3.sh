#!/bin/bash
# SQ counters of the long-context kernels on one C3 configuration: tools/prof_c3.sh <outdir> "<--only pattern>"
out=$1; pat=$2
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/$out
run() {
  pass=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d /root/repo/$out/$pass -o $pass -- python3 /root/repo/tools/bench_configs.py --only "$pat" --steps 2 --warmup 1 > /root/repo/$out/$pass.log 2>&1 || { echo "pass $pass failed"; tail -3 /root/repo/$out/$pass.log; return 1; }
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS &&
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
run clk GRBM_GUI_ACTIVE &&
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/kt -o kt -- python3 /root/repo/tools/bench_configs.py --only "$pat" --steps 4 --warmup 1 > /root/repo/$out/kt.log 2>&1 &&
python3 /root/repo/tools/pmc_summary.py /root/repo/$out 'k_inv|k_fwd' > /root/repo/$out/summary.txt &&
grep -h "vw::" /root/repo/$out/kt/*kernel_stats.csv | cut -c1-200 >> /root/repo/$out/summary.txt; cat /root/repo/$out/summary.txt
