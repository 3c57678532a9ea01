#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per (kernel, grid) over the dispatches of a run.
usage: pmc_summary.py <dir with *_counter_collection.csv> [more dirs] [--match substr]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

match = None
dirs = []
a = sys.argv[1:]
while a:
    x = a.pop(0)
    if x == "--match":
        match = a.pop(0)
    else:
        dirs.append(x)
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        meta = {}
        for row in csv.DictReader(open(f)):
            k = (row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[k] += float(row["Counter_Value"])
            meta[row["Dispatch_Id"]] = (row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:], row["Grid_Size"], row.get("VGPR_Count", ""), row.get("Scratch_Size", ""))
        for (disp, cname), v in per_dispatch.items():
            acc[meta[disp]][cname].append(v)
out = {}
for key, cs in acc.items():
    if match and match not in key[0]:
        continue
    name = "%s grid=%s vgpr=%s scratch=%s" % key
    out[name] = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    out[name]["dispatches"] = max(len(v) for v in cs.values())
print(json.dumps(out, indent=1))
