#!/usr/bin/env python3
"""Diagnostics: phase timeline of k_levels (the single-pass level decoder) on one bench workload.
Runs the plan with PQD_LV_STAMPS set, fetches the per-workgroup wall-clock stamps and prints, per phase,
the median / 90th percentile duration and when the first and last workgroup passed it."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PQD_LV_STAMPS"] = "1"
os.environ.setdefault("PQD_OVERLAP", "0")
import numpy as np  # noqa: E402

import bench  # noqa: E402
from parquet_amd import _native as N  # noqa: E402
from parquet_amd import reader as R  # noqa: E402

NAMES = ["start", "loaded", "sweep", "groups", "exits", "entry known", "published", "group walk", "sub walk",
         "values", "prefix known", "surplus chk", "assembled(last win)", "expanded"]


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "optint"
    pages = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    rows_per_page = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
    buf, _ = bench.build_file(workload, [0, 1], pages, rows_per_page)
    pf = R.ParquetFile(buf)
    ctx = R.Context(0)
    chunks, nc, pgs, npg = pf.descriptors(0, pf.num_row_groups)
    plan = R.Plan(ctx, chunks, nc, pgs, npg, pf.address, pf.size)
    d_src = N.decode.pqd_device_alloc(ctx.handle, pf.size)
    assert N.decode.pqd_copy_to_device(ctx.handle, d_src, pf.address, pf.size) == 0
    for _ in range(3):
        plan.run(d_src)
    plan.sync()
    cap = 1 << 26
    out = np.zeros(cap // 8, np.uint64)
    n = N.decode.pqd_plan_debug_stamps(plan._h, out.ctypes.data_as(C.c_void_p), cap)
    assert n > 0, n
    st = out[: n // 8].reshape(-1, 16).astype(np.int64)
    st = st[st[:, 0] != 0]  # positions of the launch order without a super-tile leave no record
    t0 = st[:, 0].min()
    us = (st - t0) / 100.0  # 100 MHz
    print("%d workgroups; kernel span %.1f us (first start to last expanded)" % (len(st), us[:, 13].max()))
    print("%-14s %9s %9s %9s | %9s %9s" % ("phase", "median", "p90", "max", "first at", "last at"))
    for i in range(1, 14):
        valid = st[:, i] != 0
        if not valid.any():
            continue
        d = us[valid, i] - us[valid, i - 1]
        print("%-14s %9.2f %9.2f %9.2f | %9.1f %9.1f" % (NAMES[i], np.median(d), np.percentile(d, 90), d.max(), us[valid, i].min(), us[valid, i].max()))
    for nm, i, j in (("win zero+main loop", 14, 11), ("slow runs", 15, 14), ("7a' + barriers", 12, 15)):
        d = us[:, i] - us[:, j]
        print("%-20s %9.2f %9.2f %9.2f" % (nm, np.median(d), np.percentile(d, 90), d.max()))
    print("start times: median %.1f p90 %.1f max %.1f" % (np.median(us[:, 0]), np.percentile(us[:, 0], 90), us[:, 0].max()))
    plan.close()
    ctx.close()


if __name__ == "__main__":
    main()
