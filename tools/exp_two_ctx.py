#!/usr/bin/env python3
"""dev: the packed host path with the batches dealt to C contexts that share one table (rc_table_share): each context has its
own streams and scratch, so the kernels of consecutive batches overlap on the GPU (a batch's K3 tail -- a few waves on its
slowest reads -- under the next batch's probe kernel).  usage: exp_two_ctx.py [batch_reads] [n_batches] [preset]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import torch
import bench, synth_int, rcorrector_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6_250_000
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 12
P = bench.PRESETS[int(sys.argv[3]) if len(sys.argv) > 3 else 2]
dev = torch.device("cuda", 0)
L, k, MFK = P["len"], P["k"], P["maxcork"]
PE = 2 if P["paired"] else 1     # mates per unit
MODE = 1 if P["paired"] else 0
gen = synth_int.Synth(P["seed"], L, 30000, 1500, P["alpha"], P["err"], P["paired"], bias3=P["bias3"], device=dev)
ctx = rcorrector_amd.Context(k=k, max_fix_per_k=MFK, device=0)
units = 12_500_000
s0, q0 = gen.generate(0, units)
ctx.count_begin(); ctx.count_add_device(s0, s0.numel()); ctx.count_finish(2)
rate = ctx.estimate_error_rate(0.95)
ctx.set_run_params(rate, b"H")
half = units
bu = B // PE
nb1 = bu * (L + 1); nb = PE * nb1
ND = min(NB, units // bu)
off = ctx.host_array(PE * bu + 1, np.uint32); off[:] = (np.arange(PE * bu + 1, dtype=np.int64) * (L + 1)).astype(np.uint32)
bufs = []
for i in range(ND):
    a = np.empty(nb, np.uint8); q = np.empty(nb, np.uint8)
    for j, base in enumerate((0, half)[:PE]):
        lo = (base + i * bu) * (L + 1)
        a[j * nb1:(j + 1) * nb1] = s0[lo:lo + nb1].cpu().numpy(); q[j * nb1:(j + 1) * nb1] = q0[lo:lo + nb1].cpu().numpy()
    bases = ctx.host_array((nb + 15) // 16, np.uint32); ctx.pack_bases(a, bases=bases)
    qb = ctx.host_array((nb + 7) // 8); ctx.pack_quality_bits(q, b"H", out=qb)
    bufs.append(dict(bases=bases, qb=qb))
e = (np.zeros(0, np.uint32), np.zeros(0, np.uint8))
cap = nb // 4
for C in (1, 2, 3, 4, 1, 2):
    ctxs = [ctx]
    for _ in range(C - 1):
        c2 = rcorrector_amd.Context(k=k, max_fix_per_k=MFK, device=0)
        c2.share_table_of(ctx)
        c2.set_run_params(rate, b"H")
        ctxs.append(c2)
    SL = 2 if C > 1 else 3   # slots per context
    nin = C * SL             # batches in flight
    outs = [dict(res=[ctx.host_array(PE * bu, np.int32) for _ in range(4)], fix=(ctx.host_array(cap, np.uint32), ctx.host_array(cap, np.uint8))) for _ in range(nin)]
    def submit(i):
        b, o = bufs[i % ND], outs[i % nin]
        ctxs[i % C].submit_packed((i // C) % SL, MODE, nb, off, b["bases"], b["qb"], e[0], e[1], res=o["res"], fix_pos=o["fix"][0], fix_chr=o["fix"][1])
    def wait(i):
        return ctxs[i % C].wait_packed((i // C) % SL)
    for i in range(min(nin, 2 * C)):   # warm
        submit(i)
    for i in range(min(nin, 2 * C)):
        wait(i)
    t0 = time.perf_counter()
    for i in range(min(nin, NB)):
        submit(i)
    done = []
    for i in range(NB):
        wait(i); done.append(time.perf_counter())
        if i + nin < NB:
            submit(i + nin)
    dt = done[-1] - t0
    print("%d context(s) x %d slots: %6.1f M reads/s whole, %6.1f M steady" % (C, SL, NB * B / dt / 1e6, (NB - 1) * B / (done[-1] - done[0]) / 1e6), flush=True)
    for c2 in ctxs[1:]:
        c2.close()
