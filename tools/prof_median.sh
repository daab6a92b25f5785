#!/bin/bash
# SQ counters of the workgroup-resident median kernel (tools/bench_configs.py --only N1), one rocprofv3 pass
R=/root/repo; out=$R/gpurun_out/median; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD --output-format csv -d $out/p1 -o p1 -- python3 $R/tools/bench_configs.py --only N1 --steps 2 --warmup 1 > $out/p1.log 2>&1 || { tail -5 $out/p1.log; exit 1; }
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/root/repo/gpurun_out/median/p1/p1_counter_collection.csv')))
d = collections.OrderedDict()
for r in rows:
    if 'threshold_wg' not in r['Kernel_Name']: continue
    d.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
for k, v in list(d.items())[:2]:
    print(k, {a: "%.4g" % b for a, b in sorted(v.items())})
PY
