#!/usr/bin/env python3
"""Per-kernel means of a few counters from one rocprofv3 --pmc pass:  python3 tools/pmc_one.py <dir> <kernel substring>"""
import csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items()):
    vals = sorted(v.values())
    print(f"{k:28s} launches {len(vals):4d}  median {vals[len(vals)//2]:.4g}  mean {sum(vals)/len(vals):.4g}")
