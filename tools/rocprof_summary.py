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


def launch_durations(trace_csv, needle):
    """per-launch durations (ms) of the kernels whose name contains `needle`, in launch order"""
    rows = [r for r in csv.DictReader(open(trace_csv)) if needle in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]


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
        # the dominant kernel launch by launch: bench.py's roofline is priced on the K timed steps = the LAST K launches
        # (the ones before are the untimed settling and warm-up calls, which run slower: first touch, clocks ramping)
        trace = src.replace("_kernel_stats.csv", "_kernel_trace.csv")
        try:
            d = launch_durations(trace, "ofdm_wave_kernel<false, false, false, true>")
            if d:
                f.write("# ofdm_wave_kernel<false,false,false,true> per launch, ms: %s\n" % " ".join("%.3f" % x for x in d))
                for k in (10, 20):
                    if len(d) >= k:
                        f.write("# mean of the last %d launches: %.4f ms\n" % (k, sum(d[-k:]) / k))
        except OSError:
            pass


if __name__ == "__main__":
    main()
