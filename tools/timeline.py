#!/usr/bin/env python3
"""print the last `calls`-th part of a rocprofv3 kernel trace csv as a timeline (name, stream/queue, start, duration in us)"""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last call: from the last k_part_count on
idx = max(i for i, r in enumerate(rows) if "k_part_count" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-28s q%-3s %9.1f %9.1f" % (r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-28:], r.get("Queue_Id", "?"), (s - t0) / 1e3, (e - s) / 1e3))
