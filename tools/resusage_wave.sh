#!/bin/bash
# register / scratch usage of every instantiation in vw_launch_wave.hip (cross-compile, no GPU needed)
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --cuda-device-only -Rpass-analysis=kernel-resource-usage -S -o /tmp/wave.s /root/repo/vectorwave_amd/csrc/vw_launch_wave.hip 2>&1 | python3 -c '
import re,sys
name=None
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: name=m.group(1); d={}
    for k in ("TotalSGPRs","VGPRs","ScratchSize \[bytes/lane\]","VGPRs Spill","SGPRs Spill"):
        m=re.search(r"remark:\s+"+k+r": (\d+)",l)
        if m: d[k]=m.group(1)
    if "LDS Size" in l and name:
        print(name[:60], d); name=None
    if "error" in l: print(l.rstrip())
'
