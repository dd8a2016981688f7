#!/usr/bin/env python3
"""dev: where the time of the packed host path goes -- per-call wall times of rc_submit_packed / rc_wait_packed against
rc_submit / rc_wait (quality bits) on the same batches.  usage: exp_packed.py [batch_reads] [n_batches]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import torch
import bench, synth_int, rcorrector_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 8
P = bench.PRESETS[2]
dev = torch.device("cuda", 0)
L, k = P["len"], P["k"]
gen = synth_int.Synth(P["seed"], L, 30000, 1500, P["alpha"], P["err"], True, device=dev)
ctx = rcorrector_amd.Context(k=k, max_fix_per_k=4, device=0)
units = 12_500_000
s0, q0 = gen.generate(0, units)
ctx.count_begin(); ctx.count_add_device(s0, s0.numel()); ctx.count_finish(2)
ctx.set_run_params(ctx.estimate_error_rate(0.95), b"H")
half = units
bu = B // 2
nb1 = bu * (L + 1); nb = 2 * nb1
off = ctx.host_array(2 * bu + 1, np.uint32); off[:] = (np.arange(2 * bu + 1, dtype=np.int64) * (L + 1)).astype(np.uint32)
hoff = (np.arange(bu + 1, dtype=np.int64) * (L + 1)).astype(np.uint32)
bufs = []
for i in range(NB):
    a = np.empty(nb, np.uint8); q = np.empty(nb, np.uint8)
    for j, base in enumerate((0, half)):
        lo = (base + i * bu) * (L + 1)
        a[j * nb1:(j + 1) * nb1] = s0[lo:lo + nb1].cpu().numpy(); q[j * nb1:(j + 1) * nb1] = q0[lo:lo + nb1].cpu().numpy()
    bases = ctx.host_array((nb + 15) // 16, np.uint32); ctx.pack_bases(a, bases=bases)
    qb = ctx.host_array((nb + 7) // 8); ctx.pack_quality_bits(q, b"H", out=qb)
    s1 = ctx.host_array(nb1); s1[:] = a[:nb1]; s2 = ctx.host_array(nb1); s2[:] = a[nb1:]
    qb1 = ctx.host_array((nb1 + 7) // 8); ctx.pack_quality_bits(q[:nb1], b"H", out=qb1)
    qb2 = ctx.host_array((nb1 + 7) // 8); ctx.pack_quality_bits(q[nb1:], b"H", out=qb2)
    cap = nb // 32
    bufs.append(dict(bases=bases, qb=qb, s1=s1, s2=s2, qb1=qb1, qb2=qb2, res=[ctx.host_array(2 * bu, np.int32) for _ in range(4)],
                     fix=(ctx.host_array(cap, np.uint32), ctx.host_array(cap, np.uint8))))
e = (np.zeros(0, np.uint32), np.zeros(0, np.uint8))
slots = 3
for what in ([os.environ["RC_EXP_ONLY"]] * 2 if os.environ.get("RC_EXP_ONLY") else ("packed", "bytes+qbits", "packed", "bytes+qbits")):
    ctx.set_quality_bits(what != "packed")
    def submit(i):
        b = bufs[i]
        if what == "packed":
            ctx.submit_packed(i % slots, 1, nb, off, b["bases"], b["qb"], e[0], e[1], res=b["res"], fix_pos=b["fix"][0], fix_chr=b["fix"][1])
        else:
            ctx.submit(i % slots, 1, b["s1"], b["qb1"], hoff, b["s2"], b["qb2"], hoff, res=b["res"])
    wait = (lambda i: ctx.wait_packed(i % slots)) if what == "packed" else (lambda i: ctx.wait(i % slots))
    ts, tw = [], []
    t0 = time.perf_counter()
    for i in range(min(slots, NB)):
        t = time.perf_counter(); submit(i); ts.append(time.perf_counter() - t)
    for i in range(NB):
        t = time.perf_counter(); wait(i); tw.append(time.perf_counter() - t)
        if i + slots < NB:
            t = time.perf_counter(); submit(i + slots); ts.append(time.perf_counter() - t)
    dt = time.perf_counter() - t0
    print("%-12s %6.1f M reads/s; submit ms: %s; wait ms: %s" % (what, NB * B / dt / 1e6, " ".join("%.2f" % (x * 1e3) for x in ts), " ".join("%.2f" % (x * 1e3) for x in tw)), flush=True)
ctx.set_quality_bits(False)
