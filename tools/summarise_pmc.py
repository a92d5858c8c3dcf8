#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: mean counter value per launch for every kernel, skipping each kernel's first
`--skip` launches (initial full-population evaluation, warm-up).

    python3 tools/summarise_pmc.py OUT.json DIR [DIR ...]      # DIRs hold *_counter_collection.csv from separate passes
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    skip = 2
    for a in sys.argv[1:]:
        if a.startswith("--skip="):
            skip = int(a.split("=")[1])
    out, dirs = args[0], args[1:]
    vals = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> [per-dispatch value]
    grid = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = defaultdict(float)  # (dispatch, kernel, counter) -> sum over instances
            order = []
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    key = (int(row["Dispatch_Id"]), row["Kernel_Name"], row["Counter_Name"])
                    if key not in per:
                        order.append(key)
                    per[key] += float(row["Counter_Value"])
                    grid[row["Kernel_Name"]] = int(row["Grid_Size"])
            for key in sorted(order):
                vals[key[1]][key[2]].append(per[key])
    res = {}
    for k, cs in vals.items():
        name = k.replace("void ", "").split("(")[0]
        e = {"grid_size": grid[k]}
        for c, v in cs.items():
            use = v[skip:] if len(v) > skip else v
            e[c + "_mean"] = sum(use) / len(use)
            e["launches_" + c] = len(use)
        res[name] = e
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
