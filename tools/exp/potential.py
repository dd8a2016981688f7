#!/usr/bin/env python3
"""dev: what share of a bench preset's reads would finish before the general search under the model with weak
stretches (tools/exp/model_w.py), against what is built (tests/k2s_model.py)?  Builds the preset's table on the GPU as
bench.py does, then runs the oracle and both models on the first N reads of shard 0.  usage: potential.py <config> [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools", "exp")):
    sys.path.insert(0, p)
import numpy as np
import torch
import bench, synth_int, rcorrector_amd
import model_w as M
from oracle import pyoracle as po
po.build()
c = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
P = bench.PRESETS[c]
dev = torch.device("cuda", 0)
L, k, n = P["len"], P["k"], P["reads"]
upr = 2 if P["paired"] else 1
units = n // upr
sub_units = bench.SUB_BATCH // upr
gen = synth_int.Synth(P["seed"], L, 30000, 1500, P["alpha"], P["err"], P["paired"], bias3=P["bias3"], device=dev)
ctx = rcorrector_amd.Context(k=k, max_fix_per_k=P["maxcork"], device=0)
ctx.count_begin()
first = None
for lo in range(0, units, sub_units):
    m = min(sub_units, units - lo)
    s0, q0 = gen.generate(lo, m)
    ctx.count_add_device(s0, s0.numel())
    if first is None: first = (s0, q0, m * upr)
    else: del s0, q0
nk = ctx.count_finish(2)
rate = ctx.estimate_error_rate(0.95)
codes, counts = ctx.table_export()
T = po.Table(k, len(codes)); T.put_many(codes, counts)
Pp = po.make_params(k, P["maxcork"], rate, b"H")
s0, q0, nr = first
rows = s0.view(nr, L + 1)[:, :L]
if P["paired"]:
    half = nr // 2; h = N // 2
    sel = torch.cat([rows[:h], rows[half:half + h]]).cpu().numpy()
    mate = lambda i: i + h if i < h else i - h
else:
    sel = rows[:N].cpu().numpy(); mate = None
seqs = [bytes(r) for r in sel]
quals = [b"I" * L for _ in seqs]
a, off = po.pack_reads(seqs[:len(seqs) // upr if P["paired"] else len(seqs)]); 
if P["paired"]:
    a1, off1 = po.pack_reads(seqs[:h]); q1, _ = po.pack_reads(quals[:h]); a2, off2 = po.pack_reads(seqs[h:]); q2, _ = po.pack_reads(quals[h:])
    res = po.correct_batch(Pp, T, 1, a1, q1, off1, a2, q2, off2, threads=32)
    out = po.unpack_reads(a1, off1) + po.unpack_reads(a2, off2)
else:
    a1, off1 = po.pack_reads(seqs); q1, _ = po.pack_reads(quals)
    res = po.correct_batch(Pp, T, 0, a1, q1, off1, threads=32)
    out = po.unpack_reads(a1, off1)
ret, l, m, hh = res[:4]
# the same sample through the device path: how many of its reads reach k_correct?
ctx.set_run_params(rate, b"H")
nn = len(seqs)
arena = torch.from_numpy(np.concatenate([np.frombuffer(x + b"\0", np.uint8) for x in seqs])).to(dev)
qar = torch.from_numpy(np.concatenate([np.frombuffer(x + b"\0", np.uint8) for x in quals])).to(dev)
offd = (torch.arange(nn + 1, device=dev, dtype=torch.int64) * (L + 1)).to(torch.int32)
r4 = [torch.zeros(nn, dtype=torch.int32, device=dev) for _ in range(4)]
os.environ.setdefault("RC_LOCALITY", "force")
ctx2 = rcorrector_amd.Context(k=k, max_fix_per_k=P["maxcork"], device=0)
ctx2.table_build(codes, counts)
ctx2.set_run_params(rate, b"H")
ctx2.profile(2); ctx2.profile_reset()
ctx2.correct_device(1 if P["paired"] else 0, nn, nn * (L + 1), L, arena, qar, offd, *r4)
ctx2.sync()
listed = ctx2.profile_correct_counters()[0]
dev_ret = r4[0].cpu().numpy()
print("device: %d of %d reads listed for k_correct (%.1f %% finished early); results equal the oracle's: %s" % (listed, nn, 100 - 100.0 * listed / nn, bool(np.array_equal(dev_ret, ret))), flush=True)
strong, info = M.front_end(Pp, T, seqs, k)
acc0 = acc1 = bad = ch1 = accd = pure = 0
t0 = time.time()
for i in range(len(seqs)):
    pt = -1 if mate is None else int(min(strong[i], strong[mate(i)]))
    r0 = M.finished_early(Pp, T, seqs[i], k, P["maxcork"], int(strong[i]), int(info[i]), pt)
    r1 = M.finished_early(Pp, T, seqs[i], k, P["maxcork"], int(strong[i]), int(info[i]), pt, allow_weak=True)
    acc0 += r0 is not None; acc1 += r1 is not None
    pure += r0 is None and r1 is not None and r1[0] == 0
    accd += M.finished_early(Pp, T, seqs[i], k, P["maxcork"], int(strong[i]), int(info[i]), pt, bs_limit=16) is not None
    if r1 is not None:
        ch1 += r1[0] > 0
        if r1 != (int(ret[i]), out[i], int(l[i]), int(m[i]), int(hh[i])): bad += 1
print("config %d: %d reads, changed %.1f %%; finished early: model %.1f %%, model with the 16-entry step table %.1f %%, with weak stretches %.1f %% (of which unchanged reads: %.1f %%) (mismatches %d)" % (c, len(seqs), 100 * float((ret > 0).sum()) / len(seqs), 100 * acc0 / len(seqs), 100 * accd / len(seqs), 100 * acc1 / len(seqs), 100 * pure / len(seqs), bad), flush=True)
