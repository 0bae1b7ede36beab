#!/usr/bin/env python3
"""Diagnostics: start / end of every kernel of the last decode step in a rocprofv3 kernel trace (CSV), relative
to the step's first kernel.  usage: step_timeline.py <..._kernel_trace.csv>"""
import csv
import sys


def main():
    rows = [r for r in csv.DictReader(open(sys.argv[1]))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a step begins with the memset that precedes the level kernel; take everything from the last k_levels' fill kernel on
    idx = [i for i, r in enumerate(rows) if "k_levels" in r["Kernel_Name"]]
    if not idx:
        sys.exit("no k_levels launch in the trace")
    last = idx[-1]
    prev_end = max(int(r["End_Timestamp"]) for r in rows[: idx[-2] + 1]) if len(idx) > 1 else 0
    first = last
    while first > 0 and int(rows[first - 1]["Start_Timestamp"]) > prev_end and "k_levels" not in rows[first - 1]["Kernel_Name"]:
        first -= 1
    # kernels of the previous step may still run: start at the first fill kernel after the previous k_levels
    step = [r for r in rows[first:]]
    t0 = int(step[0]["Start_Timestamp"])
    for r in step:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pqd::", "")[:34]
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print("%-34s stream %-3s %9.1f -> %9.1f  (%8.1f us)" % (n, r["Stream_Id"], s, e, e - s))


if __name__ == "__main__":
    main()
