#!/usr/bin/env python3
"""dev: the packed boundary at the CLI's batch size against the number of batches in flight (bench.host_path_packed,
the headline preset's reads).  usage: exp_cli_slots.py [batch_reads]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch
import bench, synth_int, rcorrector_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
P = bench.PRESETS[2]
L, k = P["len"], P["k"]
gen = synth_int.Synth(P["seed"], L, 30000, 1500, P["alpha"], P["err"], True, device=torch.device("cuda", 0))
ctx = rcorrector_amd.Context(k=k, max_fix_per_k=4, device=0)
units = 12_500_000
s0, q0 = gen.generate(0, units)
ctx.count_begin(); ctx.count_add_device(s0, s0.numel()); ctx.count_finish(2)
ctx.set_run_params(ctx.estimate_error_rate(0.95), b"H")
for rep in range(2):
    for slots in (2, 3, 4):
        r = bench.host_path_packed(ctx, (s0, q0, 2 * units), 1, L, 322e6, b"H", batch_reads=B, n_timed=24, n_warm=4, slots=slots)
        print("batch %d, %d in flight: %.1f M reads/s (steady %.1f M)" % (r["batch_reads"], slots, r["reads_per_s"] / 1e6, r["steady_reads_per_s"] / 1e6), flush=True)
