#!/bin/bash
# needs the tuning build (make -C vectorwave_amd/csrc tuning): only that build reads the VW_* overrides
export VW_LIBVWGPU=${VW_LIBVWGPU:-/root/repo/vectorwave_amd/libvwgpu_tuning.so}
# effective clock (GRBM_GUI_ACTIVE / 8 / duration) of the fused kernels under the ablations
cd /tmp && export TMPDIR=/tmp
for ab in none nostore nocompute; do
  mkdir -p /root/repo/gpurun_out/clk/$ab
  VW_ABLATE=$ab timeout -s KILL 180 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /root/repo/gpurun_out/clk/$ab -o clk -- python3 /root/repo/bench.py --rows 16384 --steps 3 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/clk/$ab.json 2>/dev/null
done
python3 - <<'PY'
import csv, collections
for ab in ('none','nostore','nocompute'):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open('/root/repo/gpurun_out/clk/%s/clk_counter_collection.csv'%ab)):
        k=r['Kernel_Name']
        if 'fused' not in k: continue
        short='inv' if 'inv_fused' in k else 'fwd'
        dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
        agg[short].append((float(r['Counter_Value'])/8/dur, dur/1e6))
    for k,v in agg.items():
        print(ab,k,'GHz %.3f'%(sum(x[0] for x in v)/len(v)),'ms %.3f'%(sum(x[1] for x in v)/len(v)))
PY
