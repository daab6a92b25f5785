#!/bin/bash
# SQ counters + effective clock of the inverse kernel variants on an 8192-row slice (tuning build).
# usage: tools/prof_inv.sh <outdir> <name> [VW_ABLATE value] [bench args...]
out=$1; name=$2; export VW_ABLATE=$3; shift 3
export VW_LIBVWGPU=/root/repo/vectorwave_amd/libvwgpu_tuning.so
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/$out
run() {
  pass=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d /root/repo/$out/$name/$pass -o $pass -- python3 /root/repo/bench.py --rows 8192 --steps 2 --warmup 1 --no-cpu-baseline --no-alt-mode $EXTRA > /root/repo/$out/$name.$pass.json 2> /root/repo/$out/$name.$pass.err || { echo "pass $pass failed"; tail -3 /root/repo/$out/$name.$pass.err; return 1; }
}
EXTRA="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS &&
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
run sq3 SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_INT32 &&
run clk GRBM_GUI_ACTIVE &&
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/$name/kt -o kt -- python3 /root/repo/bench.py --rows 8192 --steps 4 --warmup 1 --no-cpu-baseline --no-alt-mode $EXTRA > /root/repo/$out/$name.kt.json 2> /root/repo/$out/$name.kt.err &&
python3 /root/repo/tools/pmc_summary.py /root/repo/$out/$name 'k_inv' > /root/repo/$out/$name.summary.txt &&
grep -h "k_inv" /root/repo/$out/$name/kt/*kernel_stats.csv | cut -c1-200 >> /root/repo/$out/$name.summary.txt; cat /root/repo/$out/$name.summary.txt
