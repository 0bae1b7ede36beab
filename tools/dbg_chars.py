"""Debug helper (GPU box): where do the GPU chars of a string column differ from the oracle's?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import helpers as H
from parquet_amd import _native as N, reader as R
from parquet_amd.writer import Column

ctx = R.Context(0)
rng = np.random.default_rng(7)
cols = [Column("a", N.INT32, [1]), Column("b", N.INT64, [1]), Column("s", N.BYTE_ARRAY, [1]),
        Column("t", N.BYTE_ARRAY, [0]), Column("d", N.DOUBLE, [1])]
buf = H.build_file(cols, 30011, 12000, 5000, rng, p_null=0.0, str_len=(8, 32), alphabet=(97, 123))
o = H.oracle_file(buf)
g = H.gpu_file(buf, ctx)
for key in sorted(o):
    if o[key].phys_type != N.BYTE_ARRAY:
        continue
    a, b = o[key].values, g[key]["chars"]
    print(key, len(a), len(b), "offsets equal", np.array_equal(o[key].offsets, g[key]["offsets"]))
    n = min(len(a), len(b))
    d = np.nonzero(a[:n] != b[:n])[0]
    print("  mismatches", len(d), d[:20], d[-5:] if len(d) else "")
    if len(d):
        i = d[0]
        print("  oracle", bytes(a[i - 8:i + 24]), "\n  gpu   ", bytes(b[i - 8:i + 24]))
        # runs of mismatches
        br = np.nonzero(np.diff(d) > 1)[0]
        starts = np.concatenate([[d[0]], d[br + 1]]); ends = np.concatenate([d[br], [d[-1]]])
        print("  runs", list(zip(starts[:10], ends[:10] - starts[:10] + 1)))
        offs = o[key].offsets
        print("  string index at first mismatch", np.searchsorted(offs, i, side="right") - 1, "page chars boundaries?")
