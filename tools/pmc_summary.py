#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel name: tools/pmc_summary.py <dir> [kernel-substring]"""
import csv
import glob
import json
import sys
from collections import defaultdict

d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc, launches = defaultdict(lambda: defaultdict(float)), defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if pat and pat not in k:
            continue
        k = k.split("(")[0][-70:] or "kernel"
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r.get("Dispatch_Id"))
out = {k: {"dispatches": len(launches[k]), **{c: v for c, v in sorted(v.items())}} for k, v in acc.items()}
print(json.dumps(out, indent=1))
