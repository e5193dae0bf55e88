#!/usr/bin/env python3
"""Per-launch durations of the decoder kernels from a rocprofv3 kernel trace of tools/scratch_place.py, grouped by the
13 launches of each scratch position.  usage: tools/scratch_trace.py <kernel_trace.csv>"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    m = re.search(r"lane_forward_grouped|lane_traceback_grouped|msc_history", n)
    if m:
        per[m.group(0)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in per.items():
    groups = [v[i:i + 13] for i in range(0, len(v), 13)]
    print(k, " ".join("%.0f" % (sum(g[3:]) / max(1, len(g[3:]))) for g in groups), "us (mean of the 10 timed launches per scratch position)")
