#!/usr/bin/env python3
"""Turns rocprofv3 CSV output (gpurun_out/prof/{kt,fetch,write}) into the committed summaries under
profiles/: the --kernel-trace --stats table and the per-launch HBM traffic of each kernel from the
FETCH_SIZE / WRITE_SIZE passes.

gfx950 corrections: both counters are in KiB; FETCH_SIZE reports half the bytes of a wide coalesced
streaming read (128-byte requests tallied at 64 bytes), so the read side is doubled; WRITE_SIZE is exact
for 16-byte-per-lane streaming stores.
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    src, tag, key = sys.argv[1], sys.argv[2], sys.argv[3]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    lines = ["# rocprofv3 summary %s (workload key %s)" % (tag, key), ""]
    newest = lambda paths: sorted(paths, key=os.path.getmtime)[-1:]  # gpurun_out/ keeps the files of earlier runs
    ks = newest(glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv")))
    if ks:
        lines += ["## --kernel-trace --stats", "", "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
        for r in csv.DictReader(open(ks[0])):
            lines.append("| %s | %s | %.0f | %s | %s | %s |" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]),
                                                             r["MinNs"], r["MaxNs"], r["Percentage"]))
        lines.append("")
    per = collections.defaultdict(dict)
    launches = {}
    for name in ("fetch", "write"):
        fs = newest(glob.glob(os.path.join(src, name, "*", "*_counter_collection.csv")))
        if not fs:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            per[k][c] = sum(v) / len(v)
            launches[k] = len(v)
    traffic = {}
    if per:
        lines += ["## --pmc FETCH_SIZE / WRITE_SIZE (separate passes), average per launch", "",
                  "| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes (2*FETCH + WRITE) * 1024 |", "|---|---|---|---|"]
        for k, d in sorted(per.items()):
            f, w = d.get("FETCH_SIZE", 0.0), d.get("WRITE_SIZE", 0.0)
            b = int((2 * f + w) * 1024)
            traffic[k] = b
            lines.append("| %s | %.1f | %.1f | %d |" % (k, f, w, b))
        lines.append("")
    open(os.path.join(out_dir, "%s_rocprof_summary.md" % tag), "w").write("\n".join(lines))
    tj_path = os.path.join(out_dir, "traffic.json")
    tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
    # HBM bytes of one decode step: every kernel of the decoder and the memsets between them, weighted by how often
    # each runs per step (the profiled command runs STEPS_PROFILED steps; copies and fills of the harness around the
    # timed steps are left out)
    steps = int(os.environ.get("STEPS_PROFILED", "12"))
    step_bytes = 0
    for k, b in traffic.items():
        if "pqd::" in k or "fillBufferAligned" in k:
            step_bytes += b * launches.get(k, 0) / steps
    if step_bytes:
        tj[key] = int(step_bytes)
        lines += ["HBM bytes per decode step (all decoder kernels and memsets, %d steps profiled): %d" % (steps, int(step_bytes)), ""]
        open(os.path.join(out_dir, "%s_rocprof_summary.md" % tag), "w").write("\n".join(lines))
    tj.setdefault("_per_kernel", {})[key] = traffic
    json.dump(tj, open(tj_path, "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
