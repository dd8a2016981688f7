#!/usr/bin/env python3
"""Dev tool (needs a library built with -DRC_EXP_ROUNDS, RC_LIB=...): distribution of gather rounds per read."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, bench, rcorrector_amd
ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=4000000); ap.add_argument("--len", type=int, default=150)
ap.add_argument("-k", type=int, default=23); ap.add_argument("--err", type=float, default=0.005)
a = ap.parse_args()
dev = torch.device("cuda", 0); n, L, k = a.reads, a.len, a.k
seq, qual = bench.synth_reads_gpu(1001000, n, L, 30000, 1500, 0.8, a.err, dev, paired=True)
ctx = rcorrector_amd.Context(k=k); ctx.count_reads_device(seq, seq.numel(), 2)
ctx.set_run_params(ctx.estimate_error_rate(0.95), b"H")
off = (torch.arange(n + 1, device=dev, dtype=torch.int64) * (L + 1)).to(torch.int32)
ret = torch.zeros(n, dtype=torch.int32, device=dev); l_, m_, h_ = torch.zeros_like(ret), torch.zeros_like(ret), torch.zeros_like(ret)
ctx.correct_device(1, n, seq.numel(), L, seq.clone(), qual, off, ret, l_, m_, h_); ctx.sync()
r = l_.cpu().numpy().astype(np.int64)
print("reads %d len %d k %d: rounds/read mean %.2f  p50 %d p90 %d p99 %d p99.9 %d p99.99 %d max %d  | sum %d, top-10 reads hold %.1f%%, reads>1000 rounds: %d" % (
    n, L, k, r.mean(), *np.percentile(r, [50, 90, 99, 99.9, 99.99]).astype(int), r.max(), r.sum(), 100.0 * np.sort(r)[-10:].sum() / r.sum(), (r > 1000).sum()))
print("top 10:", np.sort(r)[-10:].tolist())
