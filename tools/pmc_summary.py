#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per dispatch, per kernel.
usage: pmc_summary.py <dir with *counter_collection.csv> [kernel-regex]"""
import csv, glob, re, sys
from collections import defaultdict
root = sys.argv[1]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
acc = defaultdict(lambda: [0.0, set()])
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat and not pat.search(k):
            continue
        k = re.sub(r"\(.*", "", k).replace("void vw::", "")
        a = acc[(k, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"])
        a[1].add(r["Dispatch_Id"])
for (k, c), (v, ids) in sorted(acc.items()):
    print("%-60s %-32s %16.0f  (%d dispatches)" % (k[:60], c, v / max(1, len(ids)), len(ids)))
