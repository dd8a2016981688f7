#!/usr/bin/env python3
"""scratch experiment: do the probe kernel of one sub-batch and the correction kernel of another overlap
usefully when two contexts (sharing one table) run on one GPU?"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, numpy as np
import rcorrector_amd, synth_int

L, k = 150, 23
PAIRS = int(os.environ.get("EXP_PAIRS", 6_250_000))   # per arena
NAR = int(os.environ.get("EXP_ARENAS", 4))
dev = torch.device("cuda", 0)
gen = synth_int.Synth(1002, L, 30000, 1500, 0.8, 0.005, True, bias3=False, device=dev)
A = rcorrector_amd.Context(k=k, max_fix_per_k=4, device=0)
A.count_begin()
arenas = []
for i in range(NAR):
    s, q = gen.generate(i * PAIRS, PAIRS)
    A.count_add_device(s, s.numel())
    arenas.append((s, q))
print("kmers", A.count_finish(2), flush=True)
qa = arenas[0][1]
fq = qa[0::(L + 1)][:1000000]; lq = qa[L - 1::(L + 1)][:1000000]
fh = torch.bincount(fq.long(), minlength=300)[:300].cpu().numpy().astype(np.int32)
lh = torch.bincount(lq.long(), minlength=300)[:300].cpu().numpy().astype(np.int32)
bad = A.bad_quality_from_hist(fh, lh, int(fq.numel()))
er = A.estimate_error_rate(0.95)
A.set_run_params(er, bad)
B = rcorrector_amd.Context(k=k, max_fix_per_k=4, device=0)
B.share_table_of(A)
B.set_run_params(er, bad)
nr = 2 * PAIRS
nb = nr * (L + 1)
off = (torch.arange(nr + 1, device=dev, dtype=torch.int64) * (L + 1)).to(torch.int32)
works = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(NAR)]
res = [[torch.zeros(nr, dtype=torch.int32, device=dev) for _ in range(4)] for _ in range(NAR)]

def restore():
    for w, (s, q) in zip(works, arenas):
        w.copy_(s)
    torch.cuda.synchronize()

def run(ctx, idx):
    for i in idx:
        r4 = res[i]
        ctx.correct_device(1, nr, nb, L, works[i], arenas[i][1], off, r4[0], r4[1], r4[2], r4[3])
        ctx.sync()

def seq():
    restore(); t = time.perf_counter(); run(A, range(NAR)); return time.perf_counter() - t

def par():
    restore()
    ta = threading.Thread(target=run, args=(A, range(0, NAR, 2)))
    tb = threading.Thread(target=run, args=(B, range(1, NAR, 2)))
    t = time.perf_counter(); ta.start(); tb.start(); ta.join(); tb.join(); return time.perf_counter() - t

seq(); par()
A.profile(True); A.profile_reset(); seq()
print("per-kernel ms over %d calls: probe %.1f threshold %.1f correct %.1f" % (NAR, A.profile_get(0)[0], A.profile_get(1)[0], A.profile_get(2)[0]), flush=True)
A.profile(False)
ref = [r[0].clone() for r in res]
for rep in range(3):
    ts = seq()
    ok1 = all(torch.equal(a, b[0]) for a, b in zip(ref, res))
    tp = par()
    ok2 = all(torch.equal(a, b[0]) for a, b in zip(ref, res))
    print("K3_GRID_WAVES=%s reads %d: sequential %.1f ms, two contexts %.1f ms (%.3fx) same %s %s" % (
        os.environ.get("RC_K3_GRID_WAVES", "-"), NAR * nr, ts * 1e3, tp * 1e3, ts / tp, ok1, ok2), flush=True)
