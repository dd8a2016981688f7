"""Seeded synthetic data sets shared by the CPU and GPU parity tests (tools/synth.py, synth-v1)."""
import numpy as np

import synth

# name -> (k, max_fix_per_k, error_rate, synth kwargs, mode) ; mode 0 single, 1 paired, 2 interleaved
CONFIGS = {
    "se_k23": dict(k=23, mfk=4, rate=0.01, mode=0, kw=dict(seed=101, n=3000, length=100, e=0.01)),
    "pe_k23": dict(k=23, mfk=4, rate=0.01, mode=1, kw=dict(seed=102, n=1500, length=150, e=0.005, paired=True)),
    "il_k23": dict(k=23, mfk=4, rate=0.01, mode=2, kw=dict(seed=103, n=1500, length=150, e=0.005, paired=True)),
    "k31_mc8": dict(k=31, mfk=8, rate=0.01, mode=0, kw=dict(seed=104, n=1500, length=150, e=0.05)),
    "skew": dict(k=23, mfk=4, rate=0.004, mode=0, kw=dict(seed=105, n=3000, length=150, e=0.005, alpha=1.5, bias3=True)),
    "nrich": dict(k=23, mfk=4, rate=0.01, mode=0, kw=dict(seed=107, n=3000, length=100, e=0.01, p_n=0.01)),
    "varlen": dict(k=23, mfk=4, rate=0.01, mode=0, kw=dict(seed=108, n=3000, length=100, e=0.02, var_len=True)),
    "k15": dict(k=15, mfk=4, rate=0.01, mode=0, kw=dict(seed=109, n=2000, length=75, e=0.01, n_tx=50)),
    "k32": dict(k=32, mfk=4, rate=0.01, mode=0, kw=dict(seed=110, n=2000, length=150, e=0.01)),
    "long300": dict(k=23, mfk=4, rate=0.01, mode=0, kw=dict(seed=112, n=400, length=300, e=0.01, n_tx=20, l_tx=1500)),
    "long600_k31": dict(k=31, mfk=4, rate=0.01, mode=1, kw=dict(seed=113, n=150, length=600, e=0.01, n_tx=10, l_tx=1500, paired=True)),
    "max1023": dict(k=23, mfk=4, rate=0.01, mode=0, kw=dict(seed=114, n=80, length=1023, e=0.008, n_tx=4, l_tx=1500, var_len=True)),
    "k11": dict(k=11, mfk=4, rate=0.01, mode=0, kw=dict(seed=115, n=1500, length=60, e=0.01, n_tx=6, l_tx=300)),
    "pe_var": dict(k=23, mfk=4, rate=0.02, mode=1, kw=dict(seed=111, n=1500, length=120, e=0.03, paired=True, var_len=True, p_n=0.005)),
    # 129 k-mers per read: one more than eight count registers per lane hold (rc_quarter.h: the <9, 10> instances)
    "se_151": dict(k=23, mfk=4, rate=0.01, mode=0, kw=dict(seed=116, n=3000, length=151, e=0.01)),
    "pe_151": dict(k=23, mfk=4, rate=0.01, mode=1, kw=dict(seed=117, n=1500, length=151, e=0.006, paired=True)),
    # 146 k-mers per read: the <10, 10> instances
    "pe_160_k15": dict(k=15, mfk=4, rate=0.01, mode=1, kw=dict(seed=118, n=1200, length=160, e=0.006, paired=True, n_tx=60)),
}

# read lengths either side of every tier boundary of rc_api_batch.hip (k = 23: S <= 160 < M <= 278 < L), mates drawn independently
TIER_LENGTHS = [60, 100, 128, 150, 150, 150, 151, 151, 160, 161, 200, 250, 278, 279, 320, 321, 400]


def tier_reads(mode, seed=119, n=1800, k=23):
    """Mixed-length batches: most reads short, some either side of the length tiers, the two mates of a pair of different
    lengths more often than not (a unit is routed by its longer read)."""
    length = max(TIER_LENGTHS)
    s1, q1, s2, q2, _ = synth.make_reads(seed, n, length, e=0.008, paired=mode != 0, n_tx=12, l_tx=1500, frag_len=500)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    lens1 = rng.choice(TIER_LENGTHS, n)
    lens2 = rng.choice(TIER_LENGTHS, n)
    keys, cnt = synth.count_kmers([s1, s2], k, [lens1, lens2])
    r1 = [s1[i, :lens1[i]].tobytes() for i in range(n)]
    qq1 = [q1[i, :lens1[i]].tobytes() for i in range(n)]
    r2 = qq2 = None
    if mode != 0:
        r2 = [s2[i, :lens2[i]].tobytes() for i in range(n)]
        qq2 = [q2[i, :lens2[i]].tobytes() for i in range(n)]
    if mode == 2:
        r1 = [x for p in zip(r1, r2) for x in p]
        qq1 = [x for p in zip(qq1, qq2) for x in p]
        r2 = qq2 = None
    return dict(k=k, mfk=4, rate=0.01, mode=mode, keys=keys, counts=cnt, seqs1=r1, quals1=qq1, seqs2=r2, quals2=qq2)


def adversarial_reads(seed=7, n=600, length=100):
    """Edge cases the reference's screens and k-mer state machine care about (SURVEY §11 fx_edge):
    reads shorter than / equal to k, 6+ N, isolated N, IUPAC letters, poly-A/T tails, all-A."""
    rng = np.random.Generator(np.random.PCG64(seed))
    s1, q1, _, _, _ = synth.make_reads(seed, n, length, e=0.01)
    reads = [s1[i].tobytes() for i in range(n)]
    quals = [q1[i].tobytes() for i in range(n)]
    out_r, out_q = [], []
    for i, (r, q) in enumerate(zip(reads, quals)):
        r = bytearray(r)
        kind = i % 12
        if kind == 0:
            r = r[:int(rng.integers(1, 23))]
        elif kind == 1:
            r = r[:23]
        elif kind == 2:
            for p in rng.choice(length, 7, replace=False):
                r[p] = ord('N')
        elif kind == 3:
            r[int(rng.integers(0, length))] = ord('N')
        elif kind == 4:
            for p in rng.choice(length, 3, replace=False):
                r[p] = rng.choice(list(b"RYKMSWBDHV"))
        elif kind == 5:
            t = int(rng.integers(10, 60))
            r[length - t:] = b"A" * t
        elif kind == 6:
            t = int(rng.integers(10, 60))
            r[:t] = b"T" * t
        elif kind == 7:
            r = bytearray(b"A" * length)
        elif kind == 8:
            for p in range(40, 46):
                r[p] = ord("ACGT"[(b"ACGT".index(r[p]) + 1) % 4]) if r[p] in b"ACGT" else r[p]
        elif kind == 9:
            r[0] = ord('N')
            r[-1] = ord('N')
        elif kind == 10:
            r = r[:24 + int(rng.integers(0, 10))]
        q = bytes(q[:len(r)])
        out_r.append(bytes(r))
        out_q.append(q)
    return out_r, out_q


def polya_boundary_reads(k, seed=4242, n_tx=6, l_tx=400, cover=40, length=100):
    """Reads over transcripts that carry A-rich / T-rich stretches whose k-windows sit right at the two IsPolyA
    thresholds the path uses (ErrorCorrection.cpp:53-71): k - 2 A's or T's (the trusted mask, :870-931, and the veto on
    substitutions, :343 / :577) and k - max(7, k/2) (the mask of GetStrongTrustedThreshold, :1530-1541) -- windows with
    exactly the threshold count, one below and one above, for both letters; 0.5 % substitutions on top."""
    rng = np.random.Generator(np.random.PCG64(seed))
    thr7 = max(7, k // 2)
    txs = []
    for t in range(n_tx):
        tx = rng.integers(0, 4, l_tx).astype(np.uint8)
        letter = 0 if t % 2 == 0 else 3
        other = [c for c in range(4) if c != letter]
        pos = 40
        for want in (k - 2, k - 3, k - 1, k, k - thr7, k - thr7 - 1, k - thr7 + 1):
            w = np.full(k, letter, np.uint8)
            holes = rng.choice(k, k - want, replace=False)
            w[holes] = rng.choice(other, len(holes))
            tx[pos:pos + k] = w
            # (the windows around it hold fewer of the letter: the flanks are random)
            pos += k + 17
        txs.append(tx)
    reads, quals = [], []
    for tx in txs:
        for _ in range(cover * l_tx // length):
            a = int(rng.integers(0, l_tx - length + 1))
            r = tx[a:a + length].copy()
            e = rng.random(length) < 0.005
            r[e] = (r[e] + rng.integers(1, 4, int(e.sum()))) % 4
            if rng.random() < 0.5:
                r = (3 - r)[::-1]
            reads.append(np.frombuffer(b"ACGT", np.uint8)[r])
            quals.append(np.full(length, ord("I"), np.uint8))
    return np.stack(reads), np.stack(quals)


def make(name):
    """Returns dict(k, mfk, rate, mode, keys, counts, seqs1, quals1, seqs2, quals2) as python bytes lists."""
    if name.startswith("polya_k"):
        k = int(name[len("polya_k"):])
        s1, q1 = polya_boundary_reads(k)
        keys, cnt = synth.count_kmers([s1], k)
        return dict(k=k, mfk=4, rate=0.0041, mode=0, keys=keys, counts=cnt, seqs1=[r.tobytes() for r in s1],
                    quals1=[q.tobytes() for q in q1], seqs2=None, quals2=None)
    if name in ("tiers_se", "tiers_pe", "tiers_il"):
        return tier_reads({"se": 0, "pe": 1, "il": 2}[name[-2:]])
    if name == "edge":
        r, q = adversarial_reads()
        s1, _, _, _, _ = synth.make_reads(7, 600, 100, e=0.01)
        keys, cnt = synth.count_kmers([s1], 23)
        return dict(k=23, mfk=4, rate=0.01, mode=0, keys=keys, counts=cnt, seqs1=r, quals1=q, seqs2=None, quals2=None)
    c = CONFIGS[name]
    kw = dict(c["kw"])
    seed, n, length = kw.pop("seed"), kw.pop("n"), kw.pop("length")
    s1, q1, s2, q2, lens = synth.make_reads(seed, n, length, **kw)
    keys, cnt = synth.count_kmers([s1, s2], c["k"], [lens, lens] if lens is not None else None)

    def rows(a):
        if a is None:
            return None
        return [a[i, :(length if lens is None else lens[i])].tobytes() for i in range(len(a))]
    r1, qq1, r2, qq2 = rows(s1), rows(q1), rows(s2), rows(q2)
    if c["mode"] == 2:
        r1 = [x for p in zip(r1, r2) for x in p]
        qq1 = [x for p in zip(qq1, qq2) for x in p]
        r2 = qq2 = None
    return dict(k=c["k"], mfk=c["mfk"], rate=c["rate"], mode=c["mode"], keys=keys, counts=cnt,
                seqs1=r1, quals1=qq1, seqs2=r2, quals2=qq2)


def run_oracle(po, d, threads=4, fn=None):
    """Runs the oracle (or any same-signature batch function) on data set d.
    Returns (ret, l, m, h, corrected_arena1[, corrected_arena2])."""
    T = po.Table(d["k"], len(d["keys"]))
    T.put_many(d["keys"], d["counts"])
    P = po.make_params(d["k"], d["mfk"], d["rate"], b"H")
    a, off = po.pack_reads(d["seqs1"])
    qa, _ = po.pack_reads(d["quals1"])
    if d["mode"] == 1:
        a2, off2 = po.pack_reads(d["seqs2"])
        qa2, _ = po.pack_reads(d["quals2"])
        res = po.correct_batch(P, T, 1, a, qa, off, a2, qa2, off2, threads=threads, fn=fn)
        return res + (a, a2)
    res = po.correct_batch(P, T, d["mode"], a, qa, off, threads=threads, fn=fn)
    return res + (a,)


def mason_style_reads(seed=11, n=300, length=100, n_tx=6, l_tx=500, e=0.02):
    """Simulated reads whose headers carry the truth the way the Mason simulator writes it and the
    reference's scorer parses it (verify.cpp:193-232,290-298): haplotype_infix= the true bases on
    the forward strand, edit_string= one letter per read base (M correct, E substituted),
    strand=forward|reverse, exp=high|medium|low|<other>, and for some reads trim=<n>.
    Returns (headers, reads, quals) as lists of bytes; the reads are what a sequencer would give
    (errors included).  Every transcript is covered deeply enough for the corrector to fix most
    substitutions."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tx = synth.NUC[rng.integers(0, 4, size=(n_tx, l_tx), dtype=np.uint8)]
    heads, reads, quals = [], [], []
    for i in range(n):
        t = int(rng.integers(0, n_tx))
        st = int(rng.integers(0, l_tx - length + 1))
        fwd = tx[t, st:st + length].copy()
        rev = bool(rng.integers(0, 2))
        true_read = synth.COMP[fwd[::-1]] if rev else fwd
        m = rng.random(length) < e
        obs = true_read.copy()
        shift = rng.integers(1, 4, size=length, dtype=np.uint8)
        obs[m] = synth.NUC[(synth.CODE[true_read[m]] + shift[m]) & 3]
        edit = np.where(m, ord('E'), ord('M')).astype(np.uint8)
        q = np.where(m, ord('#'), ord('I')).astype(np.uint8)
        exp = [b"high", b"medium", b"low", b"none"][int(rng.integers(0, 4))]
        h = b"@sim.%09d contig=tx%d haplotype=0 length=%d orig_begin=%d strand=%s exp=%s haplotype_infix=%s edit_string=%s" % (
            i, t, length, st, b"reverse" if rev else b"forward", exp, fwd.tobytes(), edit.tobytes())
        if i % 17 == 0:
            h += b" trim=%d" % (i % 5)
        heads.append(h)
        reads.append(obs.tobytes())
        quals.append(q.tobytes())
    return heads, reads, quals


def write_mason_fastq(path, heads, reads, quals):
    with open(path, "wb") as f:
        for h, r, q in zip(heads, reads, quals):
            f.write(h + b"\n" + r + b"\n+\n" + q + b"\n")


def mason_indel_variants(heads, reads, quals):
    """The same reads as a trimming / indel-making corrector could return them: some cut short
    at the 3' end, some with a base dropped or inserted in the middle -- exercises the scorer's
    unequal-length alignment (verify.cpp:58-129) and its -noindel switch."""
    out_r, out_q = [], []
    for i, (r, q) in enumerate(zip(reads, quals)):
        if i % 5 == 1:
            r, q = r[:-3], q[:-3]
        elif i % 5 == 2:
            p = 20 + i % 50
            r, q = r[:p] + r[p + 1:], q[:p] + q[p + 1:]
        elif i % 5 == 3:
            p = 10 + i % 60
            r, q = r[:p] + b"G" + r[p:], q[:p] + b"I" + q[p:]
        out_r.append(r)
        out_q.append(q)
    return heads, out_r, out_q
