"""CPU, world_size 2 over gloo: the N>1 path -- unit sharding that never splits a pair, results
independent of the rank count, and the one collective (summary reduce).  The per-read compute is
the oracle here (no GPU in this container); on the GPU box the same sharding feeds the HIP path."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    import datasets
    from oracle import pyoracle as po
    from rcorrector_amd.distributed import read_range, reduce_summary
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = datasets.make("il_k23")
    n = len(d["seqs1"])
    lo, hi = read_range(2, n, rank, world)
    assert lo % 2 == 0 and hi % 2 == 0
    sub = dict(d)
    sub["seqs1"], sub["quals1"] = d["seqs1"][lo:hi], d["quals1"][lo:hi]
    ret, l, m, h, arena = datasets.run_oracle(po, sub, threads=2)
    np.savez(os.path.join(outdir, "r%d.npz" % rank), ret=ret, l=l, m=m, h=h, arena=arena, lo=lo, hi=hi)
    reads, cors = reduce_summary(len(ret), int(ret[ret > 0].sum()))
    if rank == 0:
        np.savez(os.path.join(outdir, "sum.npz"), reads=reads, cors=cors)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one(oracle, tmp_path):
    import datasets
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    d = datasets.make("il_k23")
    want = datasets.run_oracle(oracle, d)
    parts = [np.load(tmp_path / ("r%d.npz" % r)) for r in range(world)]
    assert parts[0]["lo"] == 0 and parts[0]["hi"] == parts[1]["lo"] and parts[1]["hi"] == len(d["seqs1"])
    for j, key in enumerate(["ret", "l", "m", "h", "arena"]):
        assert np.array_equal(np.concatenate([p[key] for p in parts]), want[j]), key
    s = np.load(tmp_path / "sum.npz")
    assert int(s["reads"]) == len(want[0]) and int(s["cors"]) == int(want[0][want[0] > 0].sum())


def test_shard_ranges_cover_and_keep_pairs():
    from rcorrector_amd.distributed import read_range, shard_range
    for n in (0, 1, 2, 7, 64, 1001):
        for w in (1, 2, 3, 8):
            edges = [shard_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in edges) - min(h - l for l, h in edges) <= 1
    for r in range(4):
        lo, hi = read_range(2, 1002, r, 4)
        assert lo % 2 == 0 and hi % 2 == 0
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)
