#!/usr/bin/env python3
"""development helper: sums rocprofv3 --pmc counter_collection CSVs per counter over the dispatches of one kernel
usage: pmc_sum.py <output dir> [kernel substring = render_kernel]"""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "render_kernel"
out = {}
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    tot = collections.defaultdict(float)
    disp = collections.defaultdict(set)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if want not in row.get("Kernel_Name", ""):
                continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            disp[row["Counter_Name"]].add(row.get("Dispatch_Id"))
    for k in tot:
        out[k] = {"sum": tot[k], "dispatches": len(disp[k])}
print(json.dumps(out, indent=1, sort_keys=True))
