#!/usr/bin/env python3
"""Dev tool: of the reads the threshold kernel hands to k_correct, which come back unchanged, and which of
class Z's two conditions (every count >= t0; more than half of the k-mers >= s) kept them on the list?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, bench, rcorrector_amd
dev = torch.device("cuda", 0); n, L, k = 4000000, 150, 23
seq, qual = bench.synth_reads_gpu(1002, n, L, 30000, 1500, 0.8, 0.005, dev, paired=True)
ctx = rcorrector_amd.Context(k=k); ctx.count_reads_device(seq, seq.numel(), 2)
er = ctx.estimate_error_rate(0.95); ctx.set_run_params(er, b"H")
off = (torch.arange(n + 1, device=dev, dtype=torch.int64) * (L + 1)).to(torch.int32)
cnt = torch.zeros(seq.numel(), dtype=torch.int32, device=dev)
ctx.probe_device(seq, seq.numel(), cnt); ctx.sync()
kc = L - k + 1
C = cnt.view(n, L + 1)[:, :kc]
strong = torch.zeros(n, dtype=torch.int32, device=dev)
ctx.strong_threshold_device(seq, off, n, seq.numel(), L, strong); ctx.sync()
ret = torch.zeros(n, dtype=torch.int32, device=dev); l_, m_, h_ = torch.zeros_like(ret), torch.zeros_like(ret), torch.zeros_like(ret)
ctx.correct_device(1, n, seq.numel(), L, seq.clone(), qual, off, ret, l_, m_, h_); ctx.sync()
half = n // 2
mate = torch.cat([strong[half:], strong[:half]])
pair_t = torch.minimum(strong, mate)
s = torch.where((pair_t >= 1) & (strong > pair_t), pair_t, strong)
smax = int(s.max().item())
bi, _ = ctx.selftest_get_bound(np.arange(smax + 1, dtype=np.int32), er)
t0 = torch.from_numpy(bi).to(dev)[s.clamp(min=0).long()].clamp(min=2)
allge = (C >= t0[:, None]).all(1)
ntr = (C >= s[:, None]).sum(1)
maj = ntr > (kc + 1) // 2
z = allge & maj & (strong >= 0)
print("reads %d: corrected %.1f %%, ret == 0 %.1f %%, ret < 0 %.1f %%; class Z (this script's restatement) %.1f %%" % (
    n, 100.0 * (ret > 0).float().mean(), 100.0 * (ret == 0).float().mean(), 100.0 * (ret < 0).float().mean(), 100.0 * z.float().mean()))
u = (ret <= 0) & ~z
print("listed and unchanged: %.1f %% of all reads" % (100.0 * u.float().mean()))
for name, m in (("ret < 0", u & (ret < 0)), ("ret == 0, some count < t0", u & (ret == 0) & ~allge), ("ret == 0, all counts >= t0, no majority at s", u & (ret == 0) & allge & ~maj)):
    print("  %-45s %.2f %% of all reads" % (name, 100.0 * m.float().mean()))
m = u & (ret == 0) & allge & ~maj
if m.any():
    q = torch.quantile(ntr[m].float(), torch.tensor([0.1, 0.5, 0.9], device=dev))
    print("  k-mers >= s in the last group: p10 %.0f p50 %.0f p90 %.0f of %d" % (*q.tolist(), kc))
m2 = u & (ret == 0) & ~allge
nb = (C < t0[:, None]).sum(1)
q = torch.quantile(nb[m2].float(), torch.tensor([0.1, 0.5, 0.9], device=dev))
print("  k-mers < t0 in the 'some count < t0' group: p10 %.0f p50 %.0f p90 %.0f" % tuple(q.tolist()))
