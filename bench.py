#!/usr/bin/env python3
"""Benchmark: device-resident Parquet page decode GB/s on MI355X.

A step = one pass of the decoder over one batch of column chunks that is already resident in HBM.  The
headline workload is "mixed": this GPU's share of configs[4] of BASELINE.json, i.e. the columns of configs[1]
(flat required int32 / int64 / float64), configs[2] (optional int32 and string, 10 % nulls) and configs[3]
(Person{Being, Username, Friends []Being}) side by side, row groups and pages of 1M rows.  GB/s counts
algorithmic bytes: page payload bytes read + decoded bytes the reference materialises (levels 1 B each, dense
values, string offsets + chars).  The single-config workloads (flat / optint / optstr / nested) are measured
in the same run on smaller files and reported under "per_workload".

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

Ranks decode independent shards (row groups are independent: weak scaling); the only exchange is one
all_gather of the per-shard row counts per step.  Rank 0 prints one JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pages", type=int, default=None, help="pages per column chunk (default 1 for mixed, 64 for flat, 8 otherwise)")
    ap.add_argument("--row-groups", type=int, default=None, help="row groups per GPU (default 96 for mixed, 4 for flat, 2 otherwise)")
    ap.add_argument("--rows-per-page", type=int, default=1_000_000)
    ap.add_argument("--workload", default="mixed", choices=["mixed", "flat", "optional", "optint", "optstr", "nested", "bools", "wide"])
    ap.add_argument("--no-extras", action="store_true", help="skip the per-workload lines measured beside the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="per CPU baseline leg (one thread, all cores)")
    ap.add_argument("--pcie", action="store_true", help="also measure the rate including H2D and D2H copies")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"))
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even for one rank (rehearses the collective path on a one-GPU box)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo lets several ranks share one GPU (rehearsal of the N>1 path on a 1-GPU box)")
    return ap.parse_args()


class Expect:
    """What a synthetic file must decode to: totals of every chunk and, for the chunks of the first and the last
    row group, the arrays themselves (kept from the generator's inputs)."""

    def __init__(self):
        self.totals = {}   # (rg index in the file, column) -> (n_levels, n_values, char bytes)
        self.arrays = {}   # (rg index, column) -> dict of arrays, first and last row group only
        self._cur = {}

    def page(self, rg, col, keep, values, defs=None, reps=None, offsets=None):
        nl = defs.size if defs is not None else (offsets.size - 1 if offsets is not None else values.size)
        nv = offsets.size - 1 if offsets is not None else values.size
        ch = int(offsets[-1]) if offsets is not None else 0
        t = self.totals.get((rg, col), (0, 0, 0))
        self.totals[(rg, col)] = (t[0] + nl, t[1] + nv, t[2] + ch)
        if keep:
            d = self.arrays.setdefault((rg, col), {"values": [], "defs": [], "reps": [], "lens": []})
            d["values"].append(np.ascontiguousarray(values).view(np.uint8).copy())
            if defs is not None:
                d["defs"].append(defs.copy())
            if reps is not None:
                d["reps"].append(reps.copy())
            if offsets is not None:
                d["lens"].append(np.diff(offsets).astype(np.int64))


def _flat_cols(N, Column):
    return [Column("i32", N.INT32, [0]), Column("i64", N.INT64, [0]), Column("f64", N.DOUBLE, [0])]


def _optional_cols(N, Column):
    return [Column("oi32", N.INT32, [1]), Column("ostr", N.BYTE_ARRAY, [1])]


def _bool_cols(N, Column):
    return [Column("b", N.BOOLEAN, [0]), Column("ob", N.BOOLEAN, [1])]


def _person_cols(N, Column):
    # README.md:146-159 Person{Being{ID int32; Age *int32}; Username string; Friends []Being}
    return [Column("being.id", N.INT32, [0, 0]), Column("being.age", N.INT32, [0, 1]), Column("username", N.BYTE_ARRAY, [0]),
            Column("friends.id", N.INT32, [2, 0]), Column("friends.age", N.INT32, [2, 1])]


def _wide_cols(N, Column):
    """The rest of the type list, and the level widths the Person columns do not reach: bools, float32, uint32 / uint64
    (physical INT32 / INT64 with converted types UINT_32 / UINT_64, template.go:464-481), the Dremel Document
    (cmd/parquetgen/dremel/testcases/doc/doc.go:5-24: definition levels up to 3, repetition levels up to 2: widths
    2 / 2) and a five-deep chain of optional and repeated groups (definition levels up to 5: width 3)."""
    from parquet_amd.writer import UINT_32, UINT_64
    return [Column("b", N.BOOLEAN, [0]), Column("ob", N.BOOLEAN, [1]), Column("f32", N.FLOAT, [0]), Column("of32", N.FLOAT, [1]),
            Column("u32", N.INT32, [0], UINT_32), Column("ou64", N.INT64, [1], UINT_64),
            Column("links.backward", N.INT64, [1, 2]), Column("names.languages.country", N.BYTE_ARRAY, [2, 2, 1]),
            Column("a.b.c.d.e", N.INT32, [1, 2, 1, 2, 1])]


def gen_nested_levels(types, n_rows, rng, p_null=0.25, max_count=3):
    """Dremel levels of one leaf whose path has the given repetition types (0 required, 1 optional, 2 repeated), for
    n_rows records: (defs, reps).  An optional group is absent with p_null, a repeated one has 0..max_count entries; the
    record is cut off where a group is absent or empty (one level with the definition level reached), as the generated
    read functions of the reference shred it (cmd/parquetgen/dremel/write_optional.go, write_repeated.go)."""
    defs = np.zeros(n_rows, np.uint8)
    reps = np.zeros(n_rows, np.uint8)
    alive = np.ones(n_rows, bool)
    r = 0
    for t in types:
        if t == 1:
            gone = alive & (rng.random(defs.size) < p_null)
            defs = defs + (alive & ~gone)
            alive = alive & ~gone
        elif t == 2:
            r += 1
            k = rng.integers(0, max_count + 1, defs.size)
            k[~alive] = 1
            alive = alive & (k > 0)
            cnt = np.maximum(k, 1)
            first = np.zeros(int(cnt.sum()), bool)
            first[np.cumsum(cnt) - cnt] = True
            defs, reps, alive = np.repeat(defs, cnt), np.repeat(reps, cnt), np.repeat(alive, cnt)
            reps = np.where(first, reps, r).astype(np.uint8)
            defs = defs + alive
    return defs.astype(np.uint8), reps


def _gen_page(N, name, rng, n, srng=None):
    """One page of `n` rows of the named column: (values, defs, reps, offsets).  srng: the generator that draws how many
    friends a row has; friends.id and friends.age of one page get the same one (they are fields of the same friends, so
    their repetition levels are the same, as in any file of that schema)."""
    if name in ("f32", "of32", "u32", "ou64"):
        mask = rng.random(n) >= 0.1 if name[0] == "o" else None
        nv = int(mask.sum()) if mask is not None else n
        v = {"f32": lambda: rng.random(nv, dtype=np.float32), "of32": lambda: rng.random(nv, dtype=np.float32),
             "u32": lambda: rng.integers(0, 2**32, nv, dtype=np.uint64).astype(np.uint32),
             "ou64": lambda: rng.integers(0, 2**63, nv, dtype=np.uint64) * np.uint64(2) + np.uint64(1)}[name]()
        return v, (mask.astype(np.uint8) if mask is not None else None), None, None
    if name in ("links.backward", "names.languages.country", "a.b.c.d.e"):
        types = {"links.backward": [1, 2], "names.languages.country": [2, 2, 1], "a.b.c.d.e": [1, 2, 1, 2, 1]}[name]
        defs, reps = gen_nested_levels(types, n, rng)
        nv = int((defs == sum(1 for t in types if t)).sum())
        if name == "links.backward":
            return rng.integers(-2**63, 2**63 - 1, nv, dtype=np.int64), defs, reps, None
        if name == "a.b.c.d.e":
            return rng.integers(-2**31, 2**31 - 1, nv, dtype=np.int64).astype(np.int32), defs, reps, None
        lens = rng.integers(2, 9, nv)  # country codes and names
        offs = np.zeros(nv + 1, np.int32)
        np.cumsum(lens, out=offs[1:])
        return rng.integers(97, 123, int(offs[-1]), dtype=np.uint8), defs, reps, offs
    if name == "i32" or name == "being.id":
        return rng.integers(-2**31, 2**31 - 1, n, dtype=np.int64).astype(np.int32), None, None, None
    if name == "i64":
        return rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64), None, None, None
    if name == "f64":
        return rng.random(n), None, None, None
    if name == "b":
        return rng.integers(0, 2, n, dtype=np.uint8), None, None, None
    if name == "ob":
        mask = rng.random(n) >= 0.1
        return rng.integers(0, 2, int(mask.sum()), dtype=np.uint8), mask.astype(np.uint8), None, None
    if name == "oi32":
        mask = rng.random(n) >= 0.1
        return rng.integers(-2**31, 2**31 - 1, int(mask.sum()), dtype=np.int64).astype(np.int32), mask.astype(np.uint8), None, None
    if name == "ostr":
        mask = rng.random(n) >= 0.1
        nv = int(mask.sum())
        lens = rng.integers(8, 33, nv)
        offs = np.zeros(nv + 1, np.int32)
        np.cumsum(lens, out=offs[1:])
        return rng.integers(97, 123, int(offs[-1]), dtype=np.uint8), mask.astype(np.uint8), None, offs
    if name == "username":
        lens = rng.integers(3, 11, n)
        offs = np.zeros(n + 1, np.int32)
        np.cumsum(lens, out=offs[1:])
        return rng.integers(97, 123, int(offs[-1]), dtype=np.uint8), None, None, offs
    if name == "being.age":
        mask = rng.random(n) >= 0.5
        return rng.integers(0, 100, int(mask.sum()), dtype=np.int64).astype(np.int32), mask.astype(np.uint8), None, None
    # friends.id / friends.age: 0-3 friends per row (as _examples/people/people.go:72), age null 50 %
    k = (srng if srng is not None else rng).integers(0, 4, n)
    nl = np.maximum(k, 1)
    starts = np.zeros(n + 1, np.int64)
    np.cumsum(nl, out=starts[1:])
    total = int(starts[-1])
    reps = np.ones(total, np.uint8)
    reps[starts[:-1]] = 0
    present = np.repeat(k > 0, nl)
    if name == "friends.id":
        defs = present.astype(np.uint8)
        nv = int(present.sum())
    else:
        has_age = rng.random(total) >= 0.5
        defs = present.astype(np.uint8) + (present & has_age).astype(np.uint8)
        nv = int((defs == 2).sum())
    return rng.integers(0, 100, nv, dtype=np.int64).astype(np.int32), defs, reps, None


def build_file(workload, rg_ids, pages, rows_per_page):
    """Synthetic file of one workload: returns (file bytes, Expect)."""
    from parquet_amd import _native as N
    from parquet_amd.writer import Column, ParquetWriter
    cols = {"flat": _flat_cols(N, Column), "optional": _optional_cols(N, Column), "optint": _optional_cols(N, Column)[:1],
            "optstr": _optional_cols(N, Column)[1:], "bools": _bool_cols(N, Column),
            "nested": [c for c in _person_cols(N, Column) if c.path != "username"], "wide": _wide_cols(N, Column),
            "mixed": _flat_cols(N, Column) + _optional_cols(N, Column) + _person_cols(N, Column)}[workload]
    seed0 = {"flat": 1000, "optional": 2000, "optint": 2000, "optstr": 2000, "nested": 3000, "mixed": 5000, "bools": 6000, "wide": 7000}[workload]
    w = ParquetWriter(cols, stats=True)
    w.reserve(len(rg_ids) * pages * rows_per_page * sum(30 if c.phys_type == N.BYTE_ARRAY else 10 for c in cols) + (1 << 20))
    ex = Expect()
    for gi, g in enumerate(rg_ids):
        rng = np.random.default_rng(seed0 + g)
        keep = gi == 0 or gi == len(rg_ids) - 1
        w.begin_row_group()
        for ci, c in enumerate(cols):
            for pi in range(pages):
                v, d, r, o = _gen_page(N, c.path, rng, rows_per_page, np.random.default_rng([seed0 + g, pi, 99]))
                w.write_page(ci, v, defs=d, reps=r, offsets=o)
                ex.page(gi, ci, keep, v, d, r, o)
        w.end_row_group(pages * rows_per_page)
    buf = w.finish()
    w.close()
    return buf, ex


def verify(plan, pf, chunks, nc, ex):
    """Checks a decoded batch against what the generator put into the file: level / value / char totals of
    every chunk, and the full arrays of the first and the last row group."""
    from parquet_amd import _native as N
    ncol = pf.num_columns
    for k in range(nc):
        rg, ci = k // ncol, k % ncol
        info = plan.info(k)
        nl, nv, ch = ex.totals[(rg, ci)]
        cd = chunks[k]
        assert info.n_values == nv, ("values", rg, ci, info.n_values, nv)
        if cd.max_def > 0:
            assert info.n_levels == nl, ("levels", rg, ci, info.n_levels, nl)
        if cd.phys_type == N.BYTE_ARRAY:
            assert info.chars_bytes == ch, ("chars", rg, ci, info.chars_bytes, ch)
        arr = ex.arrays.get((rg, ci))
        if arr is None:
            continue
        if cd.max_def > 0:
            assert np.array_equal(plan.fetch(k, N.OUT_DEFS), np.concatenate(arr["defs"])), ("defs", rg, ci)
        if cd.max_rep > 0:
            assert np.array_equal(plan.fetch(k, N.OUT_REPS), np.concatenate(arr["reps"])), ("reps", rg, ci)
        vals = np.concatenate(arr["values"])
        if cd.phys_type == N.BYTE_ARRAY:
            assert np.array_equal(plan.fetch(k, N.OUT_CHARS), vals), ("chars", rg, ci)
            offs = np.zeros(nv + 1, np.int64)
            np.cumsum(np.concatenate(arr["lens"]), out=offs[1:])
            assert np.array_equal(plan.fetch(k, N.OUT_OFFSETS, np.int32).astype(np.int64), offs), ("offsets", rg, ci)
        else:
            assert np.array_equal(plan.fetch(k, N.OUT_VALUES), vals), ("values", rg, ci)


def _oracle_chunk(args):
    import oracle as O
    buf, m, phys, max_def, max_rep = args
    ch = O.read_chunk(buf, m["start_pos"], phys, max_def, max_rep, m["num_values"], m["total_compressed_size"])
    assert ch.error == 0
    out = ch.values.size + ch.defs.size + ch.reps.size + (ch.offsets.size * 4 if ch.offsets is not None else 0)
    return m["total_compressed_size"] + out  # headers are < 0.001 % of the payload


def cpu_baseline(buf, pf, seconds, threads):
    """The CPU oracle (a C restatement of the reference's Go reader) over the leading row groups of the same
    file: column chunks in parallel on `threads` threads (the C code runs outside the interpreter lock), repeated
    until `seconds` have passed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import concurrent.futures as cf
    n_rg = min(pf.num_row_groups, max(1, (2 * threads + pf.num_columns - 1) // pf.num_columns)) if threads > 1 else 1
    tasks = []
    for g in range(n_rg):
        for c in range(pf.num_columns):
            m = pf.chunk_meta(g, c)
            ci = pf.columns[m["column"]][1]
            tasks.append((buf, m, ci.phys_type, ci.max_def, ci.max_rep))
    t0 = time.perf_counter()
    reps, nbytes = 0, 0
    with cf.ThreadPoolExecutor(max_workers=threads) as pool:
        while True:
            nbytes = sum(pool.map(_oracle_chunk, tasks)) if threads > 1 else sum(_oracle_chunk(t) for t in tasks)
            reps += 1
            el = time.perf_counter() - t0
            if el >= seconds:
                break
    return {"value": round(nbytes * reps / el / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "port",
            "sample": "first %d row group(s) of the same file (%d column chunks, %.0f MB algorithmic) decoded %d times by the C "
                      "restatement of the reference reader, %d thread(s), chunks in parallel, %.1f s" % (n_rg, len(tasks), nbytes / 1e6, reps, threads, el)}


DOMINANT = {"flat": "k_copy (PLAIN value gather)", "optint": "k_levels (single-pass level decode) beside k_copy",
            "optional": "k_str_fused (BYTE_ARRAY pages in one pass)", "optstr": "k_str_fused (BYTE_ARRAY pages in one pass)",
            "nested": "k_levels (single-pass level decode) and k_rows (row offsets)",
            "bools": "k_bools (bit -> byte expansion) and k_levels",
            "wide": "k_levels (widths 1-3), k_rows, k_copy, k_bools, k_str_fused",
            "mixed": "k_str_fused, k_copy and k_levels, side by side on three streams"}
DESCRIBE = {
    "mixed": "this GPU's share of configs[4]: configs[1]+[2]+[3] columns side by side (flat int32/int64/float64, optional int32 + string with "
             "10%% nulls, Person{Being,Username,Friends []Being}), %d rows/page, %d page(s)/chunk, %d row groups per GPU (uncompressed DATA_PAGE v1)",
    "flat": "configs[1]: flat required int32/int64/float64 columns, %d rows/page, %d pages/chunk, %d row groups per GPU (uncompressed DATA_PAGE v1, PLAIN)",
    "optional": "configs[2]: optional int32 + string columns, 10%% nulls, %d rows/page, %d pages/chunk, %d row groups per GPU",
    "nested": "configs[3]: Person{Being,Friends []Being} int32 leaves, %d rows/page, %d pages/chunk, %d row groups per GPU",
    "optint": "configs[2], int32 half: optional int32 column, 10%% nulls, %d rows/page, %d pages/chunk, %d row groups per GPU",
    "bools": "required + optional (10%% nulls) bool columns, %d rows/page, %d pages/chunk, %d row groups per GPU",
    "wide": "the rest of the type list: required + optional bool, float32, uint32 / uint64, the Dremel Document columns links.backward and "
            "names.languages.country (levels up to 3 / 2) and a five-deep chain a.b.c.d.e (definition levels up to 5, width 3), %d rows/page, "
            "%d pages/chunk, %d row groups per GPU",
    "optstr": "configs[2], string half: optional string column (8-32 B), 10%% nulls, %d rows/page, %d pages/chunk, %d row groups per GPU",
}
DEFAULT_SHAPE = {"mixed": (1, 96), "flat": (64, 4), "optional": (8, 2), "optint": (8, 2), "optstr": (8, 2), "nested": (8, 2), "bools": (32, 2), "wide": (2, 2)}  # pages, row groups


class Loaded:
    """A workload resident on the device: file bytes in HBM and a decode plan over all its row groups."""

    def __init__(self, ctx, workload, rg_ids, pages, rows_per_page):
        from parquet_amd import _native as N
        from parquet_amd import reader as R
        self.N, self.ctx = N, ctx
        self.buf, self.expect = build_file(workload, rg_ids, pages, rows_per_page)
        self.pf = R.ParquetFile(self.buf)
        self.chunks, self.nc, self.pgs, self.npg = self.pf.descriptors(0, self.pf.num_row_groups)
        self.plan = R.Plan(ctx, self.chunks, self.nc, self.pgs, self.npg, self.pf.address, self.pf.size)
        self.d_src = N.decode.pqd_device_alloc(ctx.handle, self.pf.size)
        assert self.d_src
        assert N.decode.pqd_copy_to_device(ctx.handle, self.d_src, self.pf.address, self.pf.size) == 0

    def close(self):
        self.plan.close()
        self.N.decode.pqd_device_free(self.ctx.handle, self.d_src)


def timed(ld, steps, warmup, before_step=None, barrier=None):
    """W untimed steps, a check of the decoded batch, then K timed steps: returns (seconds, per-step device ms from the
    HIP events on the decoder's stream, per-step ms of the PLAIN value gather)."""
    import torch
    for _ in range(max(warmup, 1)):
        if before_step:
            before_step()
        ld.plan.run(ld.d_src)
    ld.plan.sync()
    torch.cuda.synchronize()
    if not os.environ.get("PQD_LIB"):  # an experimental build of the library (tools/build_variants.sh) may leave stores out
        verify(ld.plan, ld.pf, ld.chunks, ld.nc, ld.expect)
    if barrier:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if before_step:
            before_step()
        ld.plan.run(ld.d_src)
    ld.plan.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tot_ms, cp_ms = [], []
    for k in range(min(steps, 256)):
        a, b = ld.plan.run_ms(k)
        tot_ms.append(a)
        cp_ms.append(b)
    return elapsed, float(np.mean(tot_ms)), float(np.mean(cp_ms))


def spawn_ranks(args):
    """`--gpus N` without a launcher around it: start N ranks (one per GPU) through torch.distributed.run as a child
    process and pass its output and exit code on.  This process never touches a GPU."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decoder has no CPU path")
    if args.dist_backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL unless --dist-backend gloo was asked for; an initialisation failure ends the run (no silent fallback)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    coll_dev = "cuda" if args.dist_backend == "nccl" else "cpu"

    from parquet_amd import reader as R
    from parquet_amd import shard

    pages = args.pages or DEFAULT_SHAPE[args.workload][0]
    if args.row_groups is None:
        args.row_groups = DEFAULT_SHAPE[args.workload][1]
    t_build = time.perf_counter()
    # The job's file has world * row_groups row groups (weak scaling); whole row groups go to ranks in
    # contiguous ranges balanced by payload bytes, and each rank materialises only its own shard.
    first, count = shard.assign_row_groups([1] * (world * args.row_groups), world)[rank]
    ctx = R.Context(local_rank)
    ld = Loaded(ctx, args.workload, list(range(first, first + count)), pages, args.rows_per_page)
    t_build = time.perf_counter() - t_build

    rows_local = torch.tensor([ld.pf.num_rows], dtype=torch.int64, device=coll_dev)
    rows_all = torch.zeros(world, dtype=torch.int64, device=coll_dev)
    # The exchange of the per-shard row counts.  Over RCCL it is the decoder library's own gather (pqd_rccl_*: its
    # communicator owns stream and buffers); rank 0's id travels through torch.distributed once.  Over gloo (several ranks
    # on one GPU, which RCCL does not allow) it is torch.distributed's all_gather.
    from parquet_amd import _native as N
    comm = None
    collective = "none"
    if use_dist and args.dist_backend == "nccl":
        ident = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            raw = (C.c_char * 128)()
            rc = N.decode.pqd_rccl_unique_id(raw)
            if rc != 0:
                raise SystemExit("pqd_rccl_unique_id failed: %d" % rc)
            ident.copy_(torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8))
        dist.broadcast(ident, 0)
        raw = (C.c_char * 128).from_buffer_copy(bytes(ident.cpu().numpy().tobytes()))
        comm = C.c_void_p()
        rc = N.decode.pqd_rccl_comm_create(local_rank, rank, world, raw, C.byref(comm))
        if rc != 0:
            raise SystemExit("pqd_rccl_comm_create failed: %d" % rc)  # no silent fallback
        collective = "rccl (pqd_rccl_gather_rows: ncclAllGather of one int64 per rank)"
    elif use_dist:
        collective = "gloo (torch.distributed all_gather)"
    all_rows_buf = (C.c_int64 * world)()
    row_off = C.c_int64(0)

    def gather_rows():  # per-shard row counts → global row offsets
        if comm is not None:
            rc = N.decode.pqd_rccl_gather_rows(comm, ld.pf.num_rows, all_rows_buf, C.byref(row_off))
            if rc != 0:
                raise SystemExit("pqd_rccl_gather_rows failed: %d" % rc)
        else:
            dist.all_gather_into_tensor(rows_all, rows_local)

    steps = args.steps
    elapsed, step_ms, cp_ms = timed(ld, steps, args.warmup, gather_rows if use_dist else None, dist.barrier if use_dist else None)
    if use_dist:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    bytes_in, bytes_out, copy_bytes = ld.plan.bytes()
    alg = bytes_in + bytes_out
    # products of the timed step beyond the arrays the reference materialises, listed beside the algorithmic bytes and
    # never part of `value`: validity bitmap (N / 8) and value index (4 bytes per 64 levels) of the optional chunks,
    # the two int32 row-offset arrays of the repeated chunks
    extra = 0
    seen_rl = set()
    for k in range(ld.nc):
        cd, ci = ld.chunks[k], ld.plan.info(k)
        if cd.max_def > 0:
            extra += (ci.n_levels + 7) // 8 + 4 * ((ci.n_levels + 63) // 64 + 1)
        if cd.max_rep > 0:  # a chunk that shares another chunk's repetition levels shares its row_level_off: written once
            rl = ld.plan.device_ptr(k, ld.N.OUT_ROW_LEVEL_OFF)
            extra += (4 if rl in seen_rl else 8) * (ci.n_rows + 1)
            seen_rl.add(rl)
    alg_t = torch.tensor([alg], dtype=torch.int64, device=coll_dev)
    if use_dist:
        dist.all_reduce(alg_t)
    total_alg = int(alg_t.item())

    if rank == 0:
        traffic = None
        try:
            tj = json.load(open(args.traffic_json))
            traffic = tj.get("%s_p%d_g%d" % (args.workload, pages, args.row_groups))
        except Exception:
            pass
        achieved = alg / (step_ms * 1e-3) / 1e9
        res = {
            "metric": "device-resident page decode GB/s (levels+PLAIN values), 1/2/4/8 MI355X",
            "value": round(total_alg * steps / elapsed / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": DESCRIBE[args.workload] % (args.rows_per_page, pages, args.row_groups),
                "file_bytes_per_gpu": ld.pf.size,
                "algorithmic_bytes_per_step_per_gpu": alg,
                "sharding": "row groups per rank, no data-path collective; all_gather of row counts per step" if world > 1 else "single GPU",
            },
            "collective": {"backend": collective, "world": world, "rows_per_rank": list(all_rows_buf) if comm is not None else rows_all.tolist()},
            # whole step: algorithmic bytes of one plan run / its device time (HIP events on the decoder's stream, average
            # over the timed steps); a step is a fixed sequence of launches on three streams, so there is no single launch
            "roofline": {
                "bound": "hbm",
                "kernel": "whole step; dominant: " + DOMINANT[args.workload],
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": "PMC passes of an earlier profile run of this workload (profiles/traffic.json), not collected in this run" if traffic else None,
                "step_ms_avg": round(step_ms, 5),
                "algorithmic_bytes_per_step": alg,
                "extra_products": {"bytes_per_step": int(extra), "what": "validity bitmaps, value index, row offsets: written in the timed step, not counted in achieved",
                                   "achieved_counting_them": round((alg + extra) / (step_ms * 1e-3) / 1e9, 1)},
                "plain_gather": {"kernel": "k_copy", "launch_ms_avg": round(cp_ms, 5), "algorithmic_bytes_per_launch": copy_bytes,
                                 "achieved": round(copy_bytes / (cp_ms * 1e-3) / 1e9, 1) if cp_ms > 0 and copy_bytes else 0.0,
                                 "note": "runs beside the level and string kernels on a stream of its own, so its launch spans most of the step"},
            },
            "device_ms_per_step_events": round(step_ms, 4),
            "string_chunks_through_two_passes": ld.plan.string_fallbacks(),
            "chunks_sharing_repetition_levels": int(ld.N.decode.pqd_plan_shared_reps(ld.plan._h)),
            "setup_seconds": round(t_build, 1),
        }
        if world == 1 and not args.no_cpu_baseline:  # the CPU legs are timed on a single-GPU run only
            threads = max(1, min(16, len(os.sched_getaffinity(0))))
            res["cpu_baseline"] = cpu_baseline(ld.buf, ld.pf, args.cpu_seconds, 1)
            res["cpu_baseline_all_cores"] = cpu_baseline(ld.buf, ld.pf, args.cpu_seconds, threads)
        if args.pcie:
            res["pcie_inclusive"] = pcie_inclusive(ctx, ld.plan, ld.pf, ld.d_src, alg)
            res["pcie_overlapped"] = pcie_overlapped(ld.pf, local_rank, alg)
            res["pcie_via_pqd_decode"] = pcie_via_decode(ld.pf, local_rank, alg)
    ld.close()
    if rank == 0:
        if world == 1 and not args.no_extras and args.workload == "mixed":
            # the single-config workloads on smaller files, same run, same checks
            extras = {}
            # two shapes each: a file of two row groups (a step of 0.15-0.6 ms, where the hand-over chains inside a level block
            # or a column chunk are as long as the step) and one of 16-32 row groups, where the kernels run at their rate
            for key, wl, (pg, rgs) in (("flat", "flat", (32, 2)), ("optint", "optint", (32, 2)), ("optstr", "optstr", (16, 2)),
                                       ("nested", "nested", (16, 2)), ("wide", "wide", (2, 2)),
                                       ("optint_32rg", "optint", (8, 32)), ("optstr_32rg", "optstr", (4, 32)),
                                       ("nested_32rg", "nested", (4, 32)), ("wide_16rg", "wide", (2, 16))):
                e = Loaded(ctx, wl, list(range(rgs)), pg, args.rows_per_page)
                el, sm, _ = timed(e, 50, 3)
                bi, bo, _ = e.plan.bytes()
                extras[key] = {"value": round((bi + bo) * 50 / el / 1e9, 1), "unit": "GB/s", "ms_per_step": round(el / 50 * 1e3, 4),
                              "roofline_frac": round((bi + bo) / (sm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "shape": "%d pages/chunk, %d row groups" % (pg, rgs)}
                e.close()
            res["per_workload"] = extras
        print(json.dumps(res), flush=True)
    ctx.close()
    if comm is not None:
        N.decode.pqd_rccl_comm_destroy(comm)
    if use_dist:
        dist.destroy_process_group()


def pcie_inclusive(ctx, plan, pf, d_src, alg):
    """H2D of the file bytes from pinned memory + decode + D2H of every decoded array into pinned memory."""
    from parquet_amd import _native as N
    hsrc = N.decode.pqd_host_alloc(ctx.handle, pf.size)
    C.memmove(hsrc, pf.address, pf.size)
    outs = []
    total_out = 0
    for c in range(plan.n_chunks):
        ci = plan.info(c)
        for which, nb in ((N.OUT_DEFS, ci.n_levels), (N.OUT_REPS, ci.n_levels), (N.OUT_VALUES, ci.value_bytes),
                          (N.OUT_OFFSETS, (ci.n_values + 1) * 4), (N.OUT_CHARS, ci.chars_bytes)):
            if nb and plan.device_ptr(c, which):
                outs.append((c, which, nb, N.decode.pqd_host_alloc(ctx.handle, nb)))
                total_out += nb
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        assert N.decode.pqd_copy_to_device(ctx.handle, d_src, hsrc, pf.size) == 0
        plan.run(d_src)
        plan.sync()
        for c, which, nb, hp in outs:
            n = N.decode.pqd_plan_fetch(plan._h, c, which, hp, nb)
            assert n == nb
        el = time.perf_counter() - t0
        best = el if best is None else min(best, el)
    for _, _, _, hp in outs:
        N.decode.pqd_host_free(ctx.handle, hp)
    N.decode.pqd_host_free(ctx.handle, hsrc)
    return {"value": round(alg / best / 1e9, 2), "unit": "GB/s", "seconds": round(best, 4),
            "note": "pinned H2D of %d file bytes + decode + pinned D2H of %d decoded bytes, serial, best of 3" % (pf.size, total_out)}


def pcie_overlapped(pf, device, alg):
    """The same with the row groups pipelined over two contexts (two streams, two host threads): the H2D of
    one row group runs beside the decode and the D2H of another."""
    import threading
    from parquet_amd import _native as N
    from parquet_amd import reader as R
    n_ctx = 2
    ctxs = [R.Context(device) for _ in range(n_ctx)]
    hsrc = N.decode.pqd_host_alloc(ctxs[0].handle, pf.size)
    C.memmove(hsrc, pf.address, pf.size)
    jobs = []
    total_out = 0
    for g in range(pf.num_row_groups):
        ctx = ctxs[g % n_ctx]
        chunks, nc, pgs, npg = pf.descriptors(g, 1)
        lo = min(pgs[i].src_off for i in range(npg))
        hi = max(pgs[i].src_off + pgs[i].payload_len for i in range(npg))
        plan = R.Plan(ctx, chunks, nc, pgs, npg, pf.address, pf.size)
        jobs.append({"ctx": ctx, "plan": plan, "lo": lo, "len": hi - lo, "outs": None})
    d_src = [N.decode.pqd_device_alloc(c.handle, pf.size) for c in ctxs]
    assert all(d_src)
    # one untimed pass sizes the outputs and allocates their pinned buffers
    for g, j in enumerate(jobs):
        d = d_src[g % n_ctx]
        assert N.decode.pqd_copy_to_device(j["ctx"].handle, d + j["lo"], hsrc + j["lo"], j["len"]) == 0
        j["plan"].run(d)
        j["plan"].sync()
        outs = []
        for c in range(j["plan"].n_chunks):
            ci = j["plan"].info(c)
            for which, nb in ((N.OUT_DEFS, ci.n_levels), (N.OUT_REPS, ci.n_levels), (N.OUT_VALUES, ci.value_bytes),
                              (N.OUT_OFFSETS, (ci.n_values + 1) * 4), (N.OUT_CHARS, ci.chars_bytes)):
                if nb and j["plan"].device_ptr(c, which):
                    outs.append((c, which, nb, N.decode.pqd_host_alloc(j["ctx"].handle, nb)))
                    total_out += nb
        j["outs"] = outs

    def worker(k):
        for g in range(k, len(jobs), n_ctx):
            j = jobs[g]
            assert N.decode.pqd_copy_to_device(j["ctx"].handle, d_src[k] + j["lo"], hsrc + j["lo"], j["len"]) == 0
            j["plan"].run(d_src[k])
            j["plan"].sync()
            for c, which, nb, hp in j["outs"]:
                assert N.decode.pqd_plan_fetch(j["plan"]._h, c, which, hp, nb) == nb

    best = None
    for _ in range(3):
        th = [threading.Thread(target=worker, args=(k,)) for k in range(n_ctx)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        best = el if best is None else min(best, el)
    for j in jobs:
        for _, _, _, hp in j["outs"]:
            N.decode.pqd_host_free(j["ctx"].handle, hp)
        j["plan"].close()
    for k, c in enumerate(ctxs):
        N.decode.pqd_device_free(c.handle, d_src[k])
    N.decode.pqd_host_free(ctxs[0].handle, hsrc)
    for c in ctxs:
        c.close()
    return {"value": round(alg / best / 1e9, 2), "unit": "GB/s", "seconds": round(best, 4),
            "note": "row groups pipelined over %d contexts: pinned H2D + decode + pinned D2H of %d decoded bytes, best of 3" % (n_ctx, total_out)}


def pcie_via_decode(pf, device, alg):
    """The drop-in path as a reader uses it: one pqd_decode per row group (it uploads the row group's byte range, plans,
    decodes and waits), then pinned D2H of every decoded array, plan destroyed; two host threads on two contexts."""
    import threading
    from parquet_amd import _native as N
    from parquet_amd import reader as R
    n_ctx = 2
    ctxs = [R.Context(device) for _ in range(n_ctx)]
    hsrc = N.decode.pqd_host_alloc(ctxs[0].handle, pf.size)
    C.memmove(hsrc, pf.address, pf.size)
    descs = [pf.descriptors(g, 1) for g in range(pf.num_row_groups)]  # copies of the file object's arrays
    hout = [N.decode.pqd_host_alloc(c.handle, 1 << 28) for c in ctxs]  # per thread: the largest decoded array fits
    moved = [0, 0]

    def worker(k):
        h2d = d2h = 0
        for g in range(k, len(descs), n_ctx):
            ch, nc, pg, npg = descs[g]
            plan = R.decode_host(ctxs[k], ch, nc, pg, npg, hsrc, pf.size)
            h2d += N.decode.pqd_plan_h2d_bytes(plan._h)
            for c in range(nc):
                ci = plan.info(c)
                for which, nb in ((N.OUT_DEFS, ci.n_levels), (N.OUT_REPS, ci.n_levels), (N.OUT_VALUES, ci.value_bytes),
                                  (N.OUT_OFFSETS, (ci.n_values + 1) * 4), (N.OUT_CHARS, ci.chars_bytes)):
                    if nb and plan.device_ptr(c, which):
                        assert N.decode.pqd_plan_fetch(plan._h, c, which, hout[k], 1 << 28) == nb
                        d2h += nb
            plan.close()
        moved[k] = (h2d, d2h)

    best = None
    for _ in range(2):
        th = [threading.Thread(target=worker, args=(k,)) for k in range(n_ctx)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        best = el if best is None else min(best, el)
    for k, c in enumerate(ctxs):
        N.decode.pqd_host_free(c.handle, hout[k])
    N.decode.pqd_host_free(ctxs[0].handle, hsrc)
    for c in ctxs:
        c.close()
    h2d, d2h = sum(m[0] for m in moved), sum(m[1] for m in moved)
    return {"value": round(alg / best / 1e9, 2), "unit": "GB/s", "seconds": round(best, 4),
            "note": "one pqd_decode per row group (upload of its byte range + plan + decode) and pinned D2H of every array, %d contexts: "
                    "%d bytes H2D for a file of %d, %d bytes D2H, best of 2" % (n_ctx, h2d, pf.size, d2h)}


if __name__ == "__main__":
    main()
