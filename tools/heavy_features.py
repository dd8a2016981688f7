#!/usr/bin/env python3
"""Dev tool (RC_LIB = a library built with -DRC_EXP_ROUNDS): which cheap per-read features predict the
reads with very many gather rounds?"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, bench, rcorrector_amd
ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=4000000); ap.add_argument("--len", type=int, default=192)
ap.add_argument("-k", type=int, default=23)
a = ap.parse_args()
dev = torch.device("cuda", 0); n, L, k = a.reads, a.len, a.k
seq, qual = bench.synth_reads_gpu(1001000, n, L, 30000, 1500, 0.8, 0.005, dev, paired=True)
ctx = rcorrector_amd.Context(k=k); ctx.count_reads_device(seq, seq.numel(), 2)
ctx.set_run_params(ctx.estimate_error_rate(0.95), b"H")
off = (torch.arange(n + 1, device=dev, dtype=torch.int64) * (L + 1)).to(torch.int32)
cnt = torch.zeros(seq.numel(), dtype=torch.int32, device=dev)
ctx.probe_device(seq, seq.numel(), cnt); ctx.sync()
kc = L - k + 1
C = cnt.view(n, L + 1)[:, :kc]
strong = torch.zeros(n, dtype=torch.int32, device=dev)
ctx.strong_threshold_device(seq, off, n, seq.numel(), L, strong); ctx.sync()
ret = torch.zeros(n, dtype=torch.int32, device=dev); l_, m_, h_ = torch.zeros_like(ret), torch.zeros_like(ret), torch.zeros_like(ret)
ctx.correct_device(1, n, seq.numel(), L, seq.clone(), qual, off, ret, l_, m_, h_); ctx.sync()
rounds = l_.long()
zero = (C == 0)
n_zero = zero.sum(1)
weak = (C < strong[:, None].clamp(min=1))
n_weak = weak.sum(1)
run = torch.zeros(n, dtype=torch.int32, device=dev); best = torch.zeros_like(run)
for j in range(kc):
    run = torch.where(zero[:, j], run + 1, torch.zeros_like(run)); best = torch.maximum(best, run)
med = C.median(1).values
feats = {"n_zero": n_zero, "n_weak": n_weak, "zero_run": best, "strong": strong, "median": med}
order = torch.argsort(rounds, descending=True)[:12]
print("top reads by rounds:")
for i in order.tolist():
    print("  rounds %6d ret %3d | " % (int(rounds[i]), int(ret[i])) + " ".join("%s=%d" % (k2, int(v[i])) for k2, v in feats.items()))
for name, v in feats.items():
    vv = v.float(); q = torch.quantile(vv[torch.randperm(n, device=dev)[:1000000]], torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev))
    print("%-9s population p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f" % (name, *q.tolist()))
for thr in (k + 5, k + 10, 2 * k - 4):
    sel = best >= thr
    print("zero_run >= %d: %.3f %% of reads, holding %.1f %% of all rounds, %d of the top-12" % (thr, 100.0 * sel.float().mean(), 100.0 * rounds[sel].sum() / rounds.sum(), int(sel[order].sum())))
