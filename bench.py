#!/usr/bin/env python3
"""Benchmark: device-resident Parquet page decode GB/s on MI355X.

A step = one pass of the decoder over one batch of column chunks that is already resident in HBM:
configs[1] of BASELINE.json — flat required int32 / int64 / float64 columns, 1M rows per page (pure
PLAIN decode, bandwidth bound).  GB/s counts algorithmic bytes: page payload bytes read + decoded bytes
the reference materialises (here 8 B per int32 value, 16 B per int64/float64 value).

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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pages", type=int, default=None, help="pages per column chunk (default 64 for flat, 8 otherwise)")
    ap.add_argument("--row-groups", type=int, default=None, help="row groups per GPU (default 4 for flat, 2 otherwise)")
    ap.add_argument("--rows-per-page", type=int, default=1_000_000)
    ap.add_argument("--workload", default="flat", choices=["flat", "optional", "optint", "optstr", "nested"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--pcie", action="store_true", help="also measure the rate including H2D and D2H copies")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"))
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even for one rank (rehearses the collective path on a one-GPU box)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo lets several ranks share one GPU (rehearsal of the N>1 path on a 1-GPU box)")
    return ap.parse_args()


def build_flat_file(rg_ids, pages, rows_per_page):
    """configs[1]: three required columns, `pages` pages of `rows_per_page` rows per chunk."""
    from parquet_amd import _native as N
    from parquet_amd.writer import Column, ParquetWriter
    cols = [Column("i32", N.INT32, [0]), Column("i64", N.INT64, [0]), Column("f64", N.DOUBLE, [0])]
    w = ParquetWriter(cols, stats=True)
    w.reserve(len(rg_ids) * pages * rows_per_page * 20 + (1 << 20))
    for g in rg_ids:
        rng = np.random.default_rng(1000 + g)
        w.begin_row_group()
        for ci, c in enumerate(cols):
            for _ in range(pages):
                if c.phys_type == N.INT32:
                    v = rng.integers(-2**31, 2**31 - 1, rows_per_page, dtype=np.int64).astype(np.int32)
                elif c.phys_type == N.INT64:
                    v = rng.integers(-2**63, 2**63 - 1, rows_per_page, dtype=np.int64)
                else:
                    v = rng.random(rows_per_page)
                w.write_page(ci, v)
        w.end_row_group(pages * rows_per_page)
    buf = w.finish()
    w.close()
    return buf


def build_optional_file(rg_ids, pages, rows_per_page, which="both"):
    """configs[2]: optional int32 and string columns, 10 % nulls (def level width 1)."""
    from parquet_amd import _native as N
    from parquet_amd.writer import Column, ParquetWriter
    cols = [Column("oi32", N.INT32, [1]), Column("ostr", N.BYTE_ARRAY, [1])]
    cols = {"both": cols, "int": cols[:1], "str": cols[1:]}[which]
    w = ParquetWriter(cols, stats=True)
    for g in rg_ids:
        rng = np.random.default_rng(2000 + g)
        w.begin_row_group()
        for ci, c in enumerate(cols):
            for _ in range(pages):
                mask = rng.random(rows_per_page) >= 0.1
                nv = int(mask.sum())
                defs = mask.astype(np.uint8)
                if c.phys_type == N.INT32:
                    w.write_page(ci, rng.integers(-2**31, 2**31 - 1, nv, dtype=np.int64).astype(np.int32), defs=defs)
                else:
                    lens = rng.integers(8, 33, nv)
                    offs = np.zeros(nv + 1, np.int32)
                    np.cumsum(lens, out=offs[1:])
                    chars = rng.integers(97, 123, int(offs[-1]), dtype=np.uint8)
                    w.write_page(ci, chars, defs=defs, offsets=offs)
        w.end_row_group(pages * rows_per_page)
    buf = w.finish()
    w.close()
    return buf


def build_nested_file(rg_ids, pages, rows_per_page):
    """configs[3]: Person{Being{id, age*}, username, Friends []Being}: rep/def levels, ragged offsets."""
    from parquet_amd import _native as N
    from parquet_amd.writer import Column, ParquetWriter
    cols = [Column("being.id", N.INT32, [0, 0]), Column("being.age", N.INT32, [0, 1]),
            Column("friends.id", N.INT32, [2, 0]), Column("friends.age", N.INT32, [2, 1])]
    w = ParquetWriter(cols, stats=True)
    for g in rg_ids:
        rng = np.random.default_rng(3000 + g)
        w.begin_row_group()
        for ci, c in enumerate(cols):
            for _ in range(pages):
                n = rows_per_page
                if ci == 0:
                    w.write_page(ci, rng.integers(0, 2**31 - 1, n, dtype=np.int64).astype(np.int32))
                elif ci == 1:
                    mask = rng.random(n) >= 0.5
                    w.write_page(ci, rng.integers(0, 100, int(mask.sum()), dtype=np.int64).astype(np.int32), defs=mask.astype(np.uint8))
                else:
                    k = rng.integers(0, 4, n)                      # friends per row
                    nl = np.maximum(k, 1)                          # levels per row
                    starts = np.zeros(n + 1, np.int64)
                    np.cumsum(nl, out=starts[1:])
                    total = int(starts[-1])
                    reps = np.ones(total, np.uint8)
                    reps[starts[:-1]] = 0
                    present = np.repeat(k > 0, nl)
                    if ci == 2:
                        defs = present.astype(np.uint8)
                        nv = int(present.sum())
                    else:
                        has_age = rng.random(total) >= 0.5
                        defs = present.astype(np.uint8) + (present & has_age).astype(np.uint8)
                        nv = int((defs == 2).sum())
                    w.write_page(ci, rng.integers(0, 100, nv, dtype=np.int64).astype(np.int32), defs=defs, reps=reps)
        w.end_row_group(pages * rows_per_page)
    buf = w.finish()
    w.close()
    return buf


def cpu_baseline(buf, pf, seconds):
    """The CPU oracle (a restatement of the reference's Go reader, one thread) over the first row group
    of the same file, repeated until `seconds` have passed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    metas = []
    nbytes = 0
    for c in range(pf.num_columns):
        m = pf.chunk_meta(0, c)
        ci = pf.columns[m["column"]][1]
        metas.append((m, ci))
    t0 = time.perf_counter()
    reps = 0
    while True:
        nbytes = 0
        for m, ci in metas:
            ch = O.read_chunk(buf, m["start_pos"], ci.phys_type, ci.max_def, ci.max_rep, m["num_values"],
                              m["total_compressed_size"])
            assert ch.error == 0
            payload = m["total_compressed_size"]  # headers are < 0.001 % of it
            out = ch.values.size + ch.defs.size + ch.reps.size + (ch.offsets.size * 4 if ch.offsets is not None else 0)
            nbytes += payload + out
        reps += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    return {"value": round(nbytes * reps / el / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": "first row group of the same file (%d column chunks, %.0f MB algorithmic) decoded %d times by the "
                      "C restatement of the reference reader, one thread, %.1f s" % (len(metas), nbytes / 1e6, reps, el)}


def main():
    args = parse_args()
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
        if args.dist_backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            except Exception as e:  # the collective is 8 bytes per rank: a CPU backend measures the same decode
                print("bench: RCCL initialisation failed (%s), falling back to gloo" % e, file=sys.stderr, flush=True)
                args.dist_backend = "gloo"
                dist.init_process_group("gloo")
        else:
            dist.init_process_group("gloo")
    coll_dev = "cuda" if args.dist_backend == "nccl" else "cpu"

    from parquet_amd import _native as N
    from parquet_amd import reader as R

    builders = {"flat": build_flat_file, "optional": build_optional_file, "nested": build_nested_file,
                "optint": lambda *a: build_optional_file(*a, which="int"),
                "optstr": lambda *a: build_optional_file(*a, which="str")}
    pages = args.pages or (64 if args.workload == "flat" else 8)
    if args.row_groups is None:
        args.row_groups = 4 if args.workload == "flat" else 2
    t_build = time.perf_counter()
    # The job's file has world * row_groups row groups (weak scaling); whole row groups go to ranks in
    # contiguous ranges balanced by payload bytes, and each rank materialises only its own shard.
    from parquet_amd import shard
    first, count = shard.assign_row_groups([1] * (world * args.row_groups), world)[rank]
    buf = builders[args.workload](list(range(first, first + count)), pages, args.rows_per_page)
    pf = R.ParquetFile(buf)
    t_build = time.perf_counter() - t_build
    ctx = R.Context(local_rank)
    chunks, nc, pgs, npg = pf.descriptors(0, pf.num_row_groups)
    plan = R.Plan(ctx, chunks, nc, pgs, npg, pf.address, pf.size)
    d_src = N.decode.pqd_device_alloc(ctx.handle, pf.size)
    assert d_src
    assert N.decode.pqd_copy_to_device(ctx.handle, d_src, pf.address, pf.size) == 0

    rows_local = torch.tensor([pf.num_rows], dtype=torch.int64, device=coll_dev)
    rows_all = torch.zeros(world, dtype=torch.int64, device=coll_dev)

    def step():
        if use_dist:
            dist.all_gather_into_tensor(rows_all, rows_local)  # per-shard row counts → global row offsets
        plan.run(d_src)

    for _ in range(max(args.warmup, 1)):
        step()
    plan.sync()
    torch.cuda.synchronize()

    # sanity: the decoded first chunk equals the payload bytes it was decoded from
    first_vals = plan.fetch(0, N.OUT_VALUES) if args.workload == "flat" else None
    if first_vals is not None:
        src = np.frombuffer(buf, np.uint8)
        ok = True
        o = 0
        for p in range(chunks[0].first_page, chunks[0].first_page + chunks[0].n_pages):
            a, l = pgs[p].src_off, pgs[p].payload_len
            ok &= bool(np.array_equal(first_vals[o:o + l], src[a:a + l]))
            o += l
        assert ok and o == first_vals.size, "decoded values differ from the page payloads"

    steps = args.steps
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    plan.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    bytes_in, bytes_out, copy_bytes = plan.bytes()
    alg = bytes_in + bytes_out
    alg_t = torch.tensor([alg], dtype=torch.int64, device=coll_dev)
    if use_dist:
        dist.all_reduce(alg_t)
    total_alg = int(alg_t.item())

    # per-launch durations of the dominant kernel (PLAIN value gather) and of the whole run, from HIP
    # events recorded on the decoder's own stream during the timed steps
    n_ev = min(steps, 256)
    tot_ms, cp_ms = [], []
    for k in range(n_ev):
        a, b = plan.run_ms(k)
        tot_ms.append(a)
        cp_ms.append(b)
    cp_avg = float(np.mean(cp_ms))
    achieved = copy_bytes / (cp_avg * 1e-3) / 1e9 if cp_avg > 0 and copy_bytes else 0.0

    if rank == 0:
        traffic = None
        try:
            tj = json.load(open(args.traffic_json))
            key = "%s_p%d_g%d" % (args.workload, pages, args.row_groups)
            traffic = tj.get(key)
        except Exception:
            pass
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
                "workload": {
                    "flat": "configs[1]: flat required int32/int64/float64 columns, %d rows/page, %d pages/chunk, %d row groups per GPU (uncompressed DATA_PAGE v1, PLAIN)",
                    "optional": "configs[2]: optional int32 + string columns, 10%% nulls, %d rows/page, %d pages/chunk, %d row groups per GPU",
                    "nested": "configs[3]: Person{Being,Friends []Being} int32 leaves, %d rows/page, %d pages/chunk, %d row groups per GPU",
                    "optint": "configs[2], int32 half: optional int32 column, 10%% nulls, %d rows/page, %d pages/chunk, %d row groups per GPU",
                    "optstr": "configs[2], string half: optional string column (8-32 B), 10%% nulls, %d rows/page, %d pages/chunk, %d row groups per GPU",
                }[args.workload] % (args.rows_per_page, pages, args.row_groups),
                "file_bytes_per_gpu": pf.size,
                "algorithmic_bytes_per_step_per_gpu": alg,
                "sharding": "row groups per rank, no data-path collective; all_gather of row counts per step" if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_copy (PLAIN value gather)",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "launch_ms_avg": round(cp_avg, 5),
                "algorithmic_bytes_per_launch": copy_bytes,
            },
            "device_ms_per_step_events": round(float(np.mean(tot_ms)), 4),
            "setup_seconds": round(t_build, 1),
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is timed on a single-GPU run only
            res["cpu_baseline"] = cpu_baseline(buf, pf, args.cpu_seconds)
        if args.pcie:
            res["pcie_inclusive"] = pcie_inclusive(ctx, plan, pf, d_src, alg)
            res["pcie_overlapped"] = pcie_overlapped(pf, local_rank, alg)
        print(json.dumps(res), flush=True)

    plan.close()
    N.decode.pqd_device_free(ctx.handle, d_src)
    ctx.close()
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


if __name__ == "__main__":
    main()
