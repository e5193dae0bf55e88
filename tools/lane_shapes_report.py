#!/usr/bin/env python3
"""Per-kernel durations of tools/lane_shapes.py from a rocprofv3 kernel trace: the five shapes are launched in order,
13 calls each (3 warm-up + 10 timed); prints forward / traceback / other kernel time per shape (mean of the last 10).
usage: tools/lane_shapes_report.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "dabk" not in k:
        continue
    rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                 k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].replace("dabk::", "")))
rows.sort()
# group into calls: a call = consecutive kernels up to and including its last traceback (or history) kernel
names = ["fic", "msc", "msc + msc", "fic + msc", "fic + msc + msc"]
fw = [d for _, d, k in rows if "forward" in k]
tb = [d for _, d, k in rows if "traceback" in k]
per = len(fw) // len(names)
print("%-18s %12s %12s   (us, rocprofv3 kernel trace, mean of the last 10 of %d launches)" % ("shape", "forward", "traceback", per))
for i, n in enumerate(names):
    f = fw[i * per:(i + 1) * per][-10:]; t = tb[i * per:(i + 1) * per][-10:]
    print("%-18s %12.1f %12.1f" % (n, sum(f) / len(f), sum(t) / len(t)))
