"""Multi-GPU plumbing: reads shard across ranks, the k-mer table is replicated, nothing is
exchanged on the data path.  The only collective is the end-of-run sum of the two
summary counters (struct _summary, main.cpp:32-36), one all-reduce of two int64."""


def shard_range(n_units, rank, world):
    """Contiguous [lo, hi) of correction units for `rank`.  A unit is a read (single-end) or a
    PAIR (paired / interleaved): mates share the threshold t = min(t1, t2)
    (ErrorCorrection.cpp:97-106) and therefore never split across ranks."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def read_range(mode, n_reads, rank, world):
    """[lo, hi) in READ indices of an interleaved (mode 2) or single (mode 0) arena; for mode 1 the
    same unit range applies to both mate arenas."""
    if mode == 2:
        lo, hi = shard_range(n_reads // 2, rank, world)
        return 2 * lo, 2 * hi
    return shard_range(n_reads, rank, world)


def reduce_summary(total_reads, total_corrections, device=None):
    """Global {totalReads, totalCorrections}: one all-reduce(sum) of two int64 (RCCL on GPUs, gloo
    on CPU).  Returns python ints; a no-op without an initialised process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(total_reads), int(total_corrections)
    t = torch.tensor([int(total_reads), int(total_corrections)], dtype=torch.int64, device=device or "cpu")
    dist.all_reduce(t)
    a, b = t.tolist()
    return int(a), int(b)
