"""Multi-GPU sharding of a file's row groups: one process per GPU, whole row groups per rank.

Row groups are independent in the reference reader (fresh Field objects per group,
cmd/parquetgen/gen/template.go:329), so ranks decode disjoint contiguous ranges with no data exchange.
The only collective is one all_gather of the per-shard row counts, from which every rank derives the
global row offset of its shard (RCCL over xGMI when the process group backend is "nccl"; gloo on CPU).
"""
import numpy as np


def assign_row_groups(weights, world):
    """Contiguous ranges of row groups per rank, balanced by `weights` (payload bytes per row group).
    Returns [(first, count)] * world; ranges are disjoint, ordered and cover every row group."""
    n = len(weights)
    prefix = np.zeros(n + 1, np.float64)
    np.cumsum(np.asarray(weights, np.float64), out=prefix[1:])
    total = prefix[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        i = int(np.searchsorted(prefix, target))
        # nearest boundary to the target, never before the previous cut
        if i > 0 and abs(prefix[i - 1] - target) <= abs(prefix[min(i, n)] - target):
            i -= 1
        cuts.append(min(max(i, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1] - cuts[r]) for r in range(world)]


def row_group_weights(pf):
    """Payload bytes of every row group of a parquet_amd.reader.ParquetFile."""
    out = []
    for g in range(pf.num_row_groups):
        out.append(sum(pf.chunk_meta(g, c)["total_compressed_size"] for c in range(pf.num_columns)))
    return out


def gather_row_counts(local_rows, device=None):
    """all_gather of per-shard row counts → (counts of every rank, global row offset of this rank)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    mine = torch.tensor([int(local_rows)], dtype=torch.int64, device=device)
    if world == 1:
        return mine.clone(), 0
    allc = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allc, mine)
    counts = allc.cpu()
    return counts, int(counts[:dist.get_rank()].sum())
