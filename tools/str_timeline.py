#!/usr/bin/env python3
"""Diagnostics: phase timeline of k_str_fused (strings in one pass) on one bench workload (runs on the GPU box).
usage: python tools/str_timeline.py [workload] [pages (mixed: row groups)] [row groups]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PQD_STR_STAMPS"] = "1"
os.environ.setdefault("PQD_OVERLAP", "0")
import numpy as np  # noqa: E402

import bench  # noqa: E402
from parquet_amd import _native as N  # noqa: E402
from parquet_amd import reader as R  # noqa: E402

NAMES = ["start", "bytes arrived", "guess + proof", "barrier 1", "unit AGG out", "look-back", "unit done", "barrier 2", "masks, offsets, compaction", "stores"]


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "optstr"
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    if workload == "mixed":
        buf, _ = bench.build_file(workload, list(range(k)), 1, 1_000_000)
    else:
        buf, _ = bench.build_file(workload, list(range(int(sys.argv[3]) if len(sys.argv) > 3 else 2)), k, 1_000_000)
    pf = R.ParquetFile(buf)
    ctx = R.Context(0)
    chunks, nc, pgs, npg = pf.descriptors(0, pf.num_row_groups)
    plan = R.Plan(ctx, chunks, nc, pgs, npg, pf.address, pf.size)
    d_src = N.decode.pqd_device_alloc(ctx.handle, pf.size)
    assert N.decode.pqd_copy_to_device(ctx.handle, d_src, pf.address, pf.size) == 0
    for _ in range(3):
        plan.run(d_src)
    plan.sync()
    cap = 1 << 27
    out = np.zeros(cap // 8, np.uint64)
    n = N.decode.pqd_plan_debug_stamps(plan._h, out.ctypes.data_as(C.c_void_p), cap)
    assert n > 0, n
    st = out[: n // 8].reshape(-1, 16).astype(np.int64)
    t0 = st[:, 0].min()
    us = (st - t0) / 100.0  # 100 MHz
    print("%d workgroups; kernel span %.1f us; workgroup life median %.2f us" % (len(st), us[:, 9].max(), np.median(us[:, 9] - us[:, 0])))
    print("%-28s %9s %9s %9s | %9s %9s" % ("phase", "median", "p90", "max", "first at", "last at"))
    for i in range(1, 10):
        d = us[:, i] - us[:, i - 1]
        print("%-28s %9.2f %9.2f %9.2f | %9.1f %9.1f" % (NAMES[i], np.median(d), np.percentile(d, 90), d.max(), us[:, i].min(), us[:, i].max()))
    print("unit record arrived: median %.2f p90 %.2f" % (np.median(us[:, 10] - us[:, 0]), np.percentile(us[:, 10] - us[:, 0], 90)))
    mhz = st[:, 15] / np.maximum(us[:, 9] - us[:, 0], 0.01)
    print("shader clock over the workgroups' lives: median %.0f MHz (p10 %.0f, p90 %.0f)" % (np.median(mhz), np.percentile(mhz, 10), np.percentile(mhz, 90)))
    print("start times: median %.1f p90 %.1f max %.1f" % (np.median(us[:, 0]), np.percentile(us[:, 0], 90), us[:, 0].max()))
    plan.close()
    ctx.close()


if __name__ == "__main__":
    main()
