#!/usr/bin/env python3
"""Dev: what the search's speculation depth costs in table probes and gather rounds, on the CPU -- the lane-serial build of the kernel's
control flow (tests/hostsim) compiled with several RC_SPEC (and the flags given after --), run on reads shaped like the stress
preset (k 31, maxcorK 8, 5 % errors) and like the headline preset (k 23, maxcorK 4, 1 %), results compared with the oracle's.
usage: spec_depth.py [depth ...] [-- extra g++ flags]"""
import os, subprocess, sys, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import numpy as np
import ctypes as C
import conftest, datasets, synth
import oracle.pyoracle as po

args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
depths = [int(x) for x in args] or [4, 6, 8, 12, 16]
subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
datasets.CONFIGS["stress"] = dict(k=31, mfk=8, rate=0.05, mode=0, kw=dict(seed=1234, n=6000, length=150, e=0.05, n_tx=6, l_tx=1500))
datasets.CONFIGS["headline"] = dict(k=23, mfk=4, rate=0.01, mode=1, kw=dict(seed=1235, n=6000, length=150, e=0.01, n_tx=20, l_tx=1500, paired=True))
for name in ("stress", "headline"):
    d = datasets.make(name)
    want = datasets.run_oracle(po, d)
    print("== %s: %d reads, %d corrected by the oracle" % (name, len(want[0]), int((want[0] > 0).sum())))
    for dep in depths:
        so = "/tmp/hostsim_spec%d.so" % dep
        conftest.build_hostsim(so, flags=["-O2", "-DRC_SPEC=%d" % dep] + extra)
        lib = conftest.load_hostsim(so)
        st = (C.c_long * 4)()
        os.environ["HOSTSIM_STATS"] = "1"
        r, w = os.pipe(); saved = os.dup(2); os.dup2(w, 2)
        try:
            got = datasets.run_oracle(po, d, threads=1, fn=lambda p, t, b: lib.hostsim_correct_batch(p, t, b, C.addressof(st)))
        finally:
            os.dup2(saved, 2); os.close(w)
        txt = os.read(r, 1 << 16).decode(); os.close(r)
        same = all(np.array_equal(a, b) for a, b in zip(want, got))
        kv = dict((k, int(v)) for k, v in re.findall(r"(\w+)=(\d+)", txt))
        n = kv.get("reads", 1)
        print("  depth %2d: %s  rounds %.2f / read, probes in rounds %.1f / read (%.1f per round), all gets %.1f / read; keep-runs keep %.2f of %.2f offered"
              % (dep, "same results" if same else "RESULTS DIFFER", kv["refills"] / n, kv["round_probes"] / n, kv["round_probes"] / max(kv["refills"], 1), kv["gets"] / n,
                 kv["sumR"] / max(kv["calls"], 1), kv["sum_avail"] / max(kv["calls"], 1)))
