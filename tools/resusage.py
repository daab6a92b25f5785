#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stdin or file) per kernel."""
import re, subprocess, sys
t = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: Function Name: ", t)[1:]
for b in blocks:
    name = b.split()[0]
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    if pat and not re.search(pat, name):
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    print("%-62s VGPR %s SGPRspill %s VGPRspill %s scratch %s occ %s" % (
        name.replace("vw::", "").split("(")[0][:62], g("    VGPRs"), g("SGPRs Spill"), g("VGPRs Spill"),
        g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
