#!/usr/bin/env python3
"""Dev tool: find the read(s) whose correction dominates a launch (the heavy tail of the search).
Regenerates a bench.py workload, times rc_correct_device on shrinking slices and prints the slowest
read with the oracle's lookup count for it."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import rcorrector_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=4000000)
ap.add_argument("--len", type=int, default=192)
ap.add_argument("-k", type=int, default=23)
ap.add_argument("--seed", type=int, default=1001000)
a = ap.parse_args()
dev = torch.device("cuda", 0)
n, L, k = a.reads, a.len, a.k
seq, qual = bench.synth_reads_gpu(a.seed, n, L, 30000, 1500, 0.8, 0.005, dev, paired=True)
ctx = rcorrector_amd.Context(k=k)
ctx.count_reads_device(seq, seq.numel(), 2)
rate = ctx.estimate_error_rate(0.95)
ctx.set_run_params(rate, b"H")
print("ERROR_RATE", rate)


def run(lo, hi):   # single-end run over reads [lo, hi) (threshold pairing does not matter for finding the tail)
    m = hi - lo
    s = seq[lo * (L + 1):hi * (L + 1)].clone()
    q = qual[lo * (L + 1):hi * (L + 1)]
    off = (torch.arange(m + 1, device=dev, dtype=torch.int64) * (L + 1)).to(torch.int32)
    ret = torch.zeros(m, dtype=torch.int32, device=dev)
    l_, m_, h_ = torch.zeros_like(ret), torch.zeros_like(ret), torch.zeros_like(ret)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.correct_device(0, m, s.numel(), L, s, q, off, ret, l_, m_, h_)
    ctx.sync()
    return time.perf_counter() - t0, ret


run(0, 1000)
lo, hi = 0, n
t_all, _ = run(lo, hi)
print("all %d reads: %.1f ms" % (n, t_all * 1e3))
while hi - lo > 1:
    mid = (lo + hi) // 2
    ta, _ = run(lo, mid)
    tb, _ = run(mid, hi)
    print("[%d,%d) %.1f ms | [%d,%d) %.1f ms" % (lo, mid, ta * 1e3, mid, hi, tb * 1e3))
    if ta > tb:
        hi = mid
    else:
        lo = mid
t1, ret = run(lo, lo + 1)
r = bytes(seq[lo * (L + 1):lo * (L + 1) + L].cpu().numpy())
print("slowest read index %d: %.2f ms alone, ret=%d" % (lo, t1 * 1e3, int(ret[0])))
print(r.decode())
cnt = torch.zeros((L + 1), dtype=torch.int32, device=dev)
ctx.probe_device(seq[lo * (L + 1):(lo + 1) * (L + 1)].clone(), L + 1, cnt)
ctx.sync()
print("counts:", cnt[:L - k + 1].cpu().numpy().tolist())
