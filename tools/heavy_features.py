#!/usr/bin/env python3
"""Dev tool (RC_LIB = a library built with -DRC_EXP_ROUNDS): which cheap per-read features predict the
reads with very many gather rounds?"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, bench, rcorrector_amd
ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=4000000); ap.add_argument("--len", type=int, default=192)
ap.add_argument("-k", type=int, default=23); ap.add_argument("--err", type=float, default=0.005)
ap.add_argument("--maxcork", type=int, default=4); ap.add_argument("--single", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0); n, L, k = a.reads, a.len, a.k
seq, qual = bench.synth_reads_gpu(1001000, n, L, 30000, 1500, 0.8, a.err, dev, paired=not a.single)
ctx = rcorrector_amd.Context(k=k, max_fix_per_k=a.maxcork); ctx.count_reads_device(seq, seq.numel(), 2)
ctx.set_run_params(ctx.estimate_error_rate(0.95), b"H")
off = (torch.arange(n + 1, device=dev, dtype=torch.int64) * (L + 1)).to(torch.int32)
cnt = torch.zeros(seq.numel(), dtype=torch.int32, device=dev)
ctx.probe_device(seq, seq.numel(), cnt); ctx.sync()
kc = L - k + 1
C = cnt.view(n, L + 1)[:, :kc]
strong = torch.zeros(n, dtype=torch.int32, device=dev)
ctx.strong_threshold_device(seq, off, n, seq.numel(), L, strong); ctx.sync()
ret = torch.zeros(n, dtype=torch.int32, device=dev); l_, m_, h_ = torch.zeros_like(ret), torch.zeros_like(ret), torch.zeros_like(ret)
ctx.correct_device(0 if a.single else 1, n, seq.numel(), L, seq.clone(), qual, off, ret, l_, m_, h_); ctx.sync()
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

# heaviest-first scheduling: how much of the tail would a sort by a feature move to the front?
top = torch.argsort(rounds, descending=True)[:100]
for name in ("strong", "median", "n_weak", "n_zero"):
    v = feats[name].float()
    for frac in (0.01, 0.05, 0.2):
        thr = torch.quantile(v[torch.randperm(n, device=dev)[:1000000]], 1.0 - frac)
        sel = v >= thr
        print("%-7s top %4.0f %% (>= %.0f): %.2f %% of reads, %.1f %% of rounds, %d of the 100 heaviest, %d of the 12 heaviest" % (
            name, frac * 100, float(thr), 100.0 * sel.float().mean(), 100.0 * rounds[sel].sum() / rounds.sum(), int(sel[top].sum()), int(sel[order].sum())))
print("rule: share of reads in H, share of rounds in H, max rounds outside H, reads > 5000 rounds outside H")
big = rounds > 5000
for lo in (60, 80, 90, 100, 105, 110):
    for smax in (1 << 30, 200, 64):
        H = (n_weak >= lo) & (n_weak < kc) & (strong <= smax)
        out = ~H
        print("n_weak in [%d, %d) & strong <= %d: %.1f %% of reads, %.1f %% of rounds, max outside %d, >5000 outside %d of %d" % (
            lo, kc, smax, 100.0 * H.float().mean(), 100.0 * rounds[H].sum() / rounds.sum(), int(rounds[out].max()), int((big & out).sum()), int(big.sum())))
hist = torch.bincount(n_weak[torch.argsort(rounds, descending=True)[:1000]], minlength=kc + 1)
print("n_weak of the 1000 heaviest:", {i: int(c) for i, c in enumerate(hist.tolist()) if c})
