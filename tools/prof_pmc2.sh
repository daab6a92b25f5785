#!/bin/bash
out=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/$out
run() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d /root/repo/$out/$name -o $name -- python3 /root/repo/bench.py $BENCH_ARGS > /root/repo/$out/$name.json 2> /root/repo/$out/$name.err || { echo "pass $name failed"; tail -3 /root/repo/$out/$name.err; return 1; }
}
BENCH_ARGS="--rows 8192 --steps 2 --warmup 1 --no-cpu-baseline $*"
run a SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL &&
run b SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU &&
run c SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LEVEL_WAVES SQ_CYCLES SQ_BUSY_CU_CYCLES
