import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np
import datasets, synth, k2s_model as M
from oracle import pyoracle as po
po.build()
tot = acc = newacc = bad = 0
rng = np.random.default_rng(5)
for trial in range(int(sys.argv[1])):
    k = int(rng.choice([15, 23, 23, 25, 31, 32]))
    length = int(rng.integers(k + 6, 200))
    e = float(rng.choice([0.003, 0.01, 0.03, 0.06]))
    mode = int(rng.integers(0, 2))
    alpha = float(rng.choice([0.8, 1.5]))
    n_tx = int(rng.choice([6, 40, 200]))
    seed = 9000 + trial
    n = 1200
    rate = float(rng.choice([0.004, 0.01, 0.02]))
    mfk = int(rng.choice([4, 4, 8]))
    s1, q1, s2, q2, lens = synth.make_reads(seed, n, length, n_tx=n_tx, l_tx=600, alpha=alpha, e=e, paired=bool(mode), var_len=bool(trial % 3 == 0))
    keys, cnt = synth.count_kmers([s1, s2], k, [lens, lens] if lens is not None else None)
    rows = lambda a: None if a is None else [a[i, :(length if lens is None else lens[i])].tobytes() for i in range(len(a))]
    d = dict(k=k, mfk=mfk, rate=rate, mode=mode, keys=keys, counts=cnt, seqs1=rows(s1), quals1=rows(q1), seqs2=rows(s2) if mode else None, quals2=rows(q2) if mode else None)
    T = po.Table(k, len(keys)); T.put_many(keys, cnt)
    P = po.make_params(k, mfk, rate, b"H")
    want = datasets.run_oracle(po, d)
    ret, l, m, h = want[:4]
    if mode == 1:
        seqs = d["seqs1"] + d["seqs2"]; n1 = len(d["seqs1"]); mate = lambda i: i + n1 if i < n1 else i - n1
        out = po.unpack_reads(want[4], po.pack_reads(d["seqs1"])[1]) + po.unpack_reads(want[5], po.pack_reads(d["seqs2"])[1])
    else:
        seqs = d["seqs1"]; mate = None
        out = po.unpack_reads(want[4], po.pack_reads(d["seqs1"])[1])
    strong, info = M.front_end(P, T, seqs, k)
    for i, s in enumerate(seqs):
        pt = -1 if mate is None else int(min(strong[i], strong[mate(i)]))
        r = M.finished_early(P, T, s, k, mfk, int(strong[i]), int(info[i]), pt)
        tot += 1
        if r is None: continue
        acc += 1
        if r != (int(ret[i]), out[i], int(l[i]), int(m[i]), int(h[i])):
            bad += 1
            print("MISMATCH trial", trial, "read", i, r[:1], ret[i], k, length, e, mode)
    print(trial, k, length, e, mode, "tot", tot, "acc", acc, "bad", bad, flush=True)
