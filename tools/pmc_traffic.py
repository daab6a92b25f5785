#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE passes of tools/prof_traffic.sh into HBM bytes per launch.

usage: pmc_traffic.py <dir of the two passes> <rows> <n> <J> <wavelet> [out.json]

rocprofv3 reports both counters in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide
(16 B per lane) streaming reads at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16 B-per-lane
streaming stores.  The result is keyed "<kernel>:<rows>x<n>:J<J>:<wavelet>" like bench.py looks it up; every
entry carries the full kernel symbol it was measured on and the digest of the libvwgpu.so that ran (the first
16 hex digits of its SHA-256): bench.py reports a figure only when that digest is the one of the library it has
loaded, so a kernel change without a new pass shows as "traffic_stale" instead of as an old number.
"""
import csv, glob, hashlib, json, os, re, sys
from collections import defaultdict

root, rows, n, J, wname = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
out = sys.argv[6] if len(sys.argv) > 6 else os.path.join(os.path.dirname(__file__), "..", "profiles", "pmc_traffic.json")
lib = os.environ.get("VW_LIBVWGPU") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vectorwave_amd", "libvwgpu.so")
digest = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
acc = defaultdict(lambda: [0.0, set()])
symbol = {}
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_fwd_stream|k_fwd_fused|k_fwd_sdry|k_inv_fused|k_inv_blk|k_inv_ring|k_inv_wave|k_inv_sdry)", r["Kernel_Name"])
        if not m:
            continue
        symbol[m.group(1)] = r["Kernel_Name"]
        a = acc[(m.group(1), r["Counter_Name"])]
        a[0] += float(r["Counter_Value"])
        a[1].add(r["Dispatch_Id"])
res = {}
try:
    res = json.load(open(out))
except Exception:
    pass
for k in sorted({k for k, _ in acc}):
    fetch = acc.get((k, "FETCH_SIZE"))
    write = acc.get((k, "WRITE_SIZE"))
    if not fetch or not write:
        continue
    fb = fetch[0] / len(fetch[1]) * 1024.0 * 2.0
    wb = write[0] / len(write[1]) * 1024.0
    key = "%s:%dx%d:J%d:%s" % (k, rows, n, J, wname)
    res[key] = round(fb + wb)
    res[key + ":detail"] = {"fetch_bytes_corrected_x2": round(fb), "write_bytes": round(wb),
                            "algorithmic_bytes": 8 * (J + 2) * rows * n, "launches_averaged": len(fetch[1]),
                            "kernel_symbol": symbol[k], "lib_digest": digest}
    print(key, res[key], "algorithmic", 8 * (J + 2) * rows * n, "ratio %.3f" % ((fb + wb) / (8.0 * (J + 2) * rows * n)))
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
