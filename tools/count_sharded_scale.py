#!/usr/bin/env python3
"""The sharded k-mer count (rc_table_count_finish_sharded) at the bench's scale, every context on this box's one device: the
25 M x 150 bp pairs of the headline shard counted (a) by one context holding all of them and (b) by N contexts holding 1 / N
each -- every occurrence crosses "xGMI" (here: a device-to-device copy on one GPU, or the staged host path with
RC_REPLICATE_STAGED=1) once, as 8 bytes -- so that the exchange's overhead is known before real links are.  The tables must be
digest-equal.  Dev / measurement tool (round 6); prints one JSON object."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import rcorrector_amd  # noqa: E402
import synth_int  # noqa: E402


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
    L, k = 150, 23
    dev = torch.device("cuda", 0)
    gen = synth_int.Synth(1002, L, 30000, 1500, 0.8, 0.005, True, device=dev)
    pairs = reads // 2
    out = {"reads": reads, "read_len": L, "k": k, "device": "one MI355X, every context on it", "runs": []}
    digest0 = None
    for N in (1, 2, 4, 8):
        ctxs = [rcorrector_amd.Context(k=k, device=0) for _ in range(N)]
        per = (pairs + N - 1) // N
        arenas = []
        for g in range(N):
            lo = g * per
            m = max(0, min(per, pairs - lo))
            s0, q0 = gen.generate(lo, m)
            del q0
            arenas.append(s0)
        torch.cuda.synchronize()
        best = None
        for rep in range(2):
            for c in ctxs:
                c.count_begin()
            t0 = time.perf_counter()
            for c, s0 in zip(ctxs, arenas):
                c.count_add_device(s0, s0.numel())
            if N == 1:
                nk = ctxs[0].count_finish(2)
            else:
                nk = ctxs[0].count_finish_sharded(ctxs[1:], 2)
            for c in ctxs:
                c.sync()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        d = ctxs[0].table_digest()
        if digest0 is None:
            digest0 = d
        out["runs"].append({"contexts": N, "count_s": round(best, 4), "kmers": int(nk), "digest": "%016x" % d, "same_table_as_one_context": d == digest0})
        del ctxs, arenas
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
