#!/usr/bin/env python3
"""Trim a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv to this
repo's kernels (namespace dabk::) and shorten the names.  Usage:
    tools/rocprof_summary.py gpurun_out/prof/x_kernel_stats.csv profiles/r01_x_kernel_stats.csv "command line"
"""
import csv
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(dabk::\w+)(<.*>)?\(", name)
    if m:
        targs = (m.group(2) or "").replace("dabk::", "").replace("(Tail)", "Tail=")
        return m.group(1) + targs
    return name


def main():
    src, dst = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    rows = list(csv.DictReader(open(src)))
    mine = [r for r in rows if "dabk::" in r["Name"]]
    total = sum(int(r["TotalDurationNs"]) for r in mine)
    with open(dst, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- %s\n" % cmd)
        f.write("# rows: this repo's kernels only (torch input-generation kernels dropped); Percentage is of these rows\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
        for r in mine:
            f.write('"%s",%s,%s,%.1f,%.2f,%s,%s,%.1f\n' % (short(r["Name"]), r["Calls"], r["TotalDurationNs"],
                    float(r["AverageNs"]), 100.0 * int(r["TotalDurationNs"]) / total, r["MinNs"], r["MaxNs"],
                    float(r["StdDev"])))


if __name__ == "__main__":
    main()
