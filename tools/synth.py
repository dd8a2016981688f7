#!/usr/bin/env python3
"""synth-v1: deterministic synthetic RNA-seq reads + a stand-in Jellyfish dump.

Test/bench infrastructure (not part of the product path).  Mirrors SURVEY.md §8(d):
  * transcriptome: n_tx transcripts x l_tx iid-uniform ACGT
  * expression weight of transcript i ~ (i+1)^-alpha
  * read: transcript by weight, start uniform, strand uniform, per-base substitution with
    probability e (uniform among the 3 other bases); quality 'I', substituted bases '#';
    optional N at rate p_n (quality '!')
  * paired: fragment of frag_len bases, mate 2 = reverse complement of the fragment tail
  * dump: exact canonical k-mer counts over all emitted reads, count>=2, ascending 2-bit
    canonical code order, text format ">COUNT\nKMER\n" (what `jellyfish dump -L 2` writes and
    what the reference parses at main.cpp:295-307).

All randomness comes from numpy's PCG64 seeded explicitly, so a (seed, parameters) pair
pins the bytes.
"""
import argparse
import os
import sys

import numpy as np

NUC = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    COMP[a] = b
CODE = np.full(256, 255, dtype=np.uint8)
for i, c in enumerate(b"ACGT"):
    CODE[c] = i


def make_reads(seed, n, length, n_tx=200, l_tx=1500, alpha=0.8, e=0.01, p_n=0.0,
               paired=False, frag_len=300, bias3=False, var_len=False):
    """Returns (seq1, qual1, seq2, qual2): uint8 arrays [n, length] (seq2/qual2 None if SE).
    With var_len, a per-read length array is also returned as 5th item (else None)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tx = NUC[rng.integers(0, 4, size=(n_tx, l_tx), dtype=np.uint8)]
    w = (np.arange(n_tx) + 1.0) ** (-alpha)
    w /= w.sum()
    tid = rng.choice(n_tx, size=n, p=w)
    if paired and frag_len < length:
        frag_len = min(l_tx, 2 * length)
    span = frag_len if paired else length
    if span > l_tx:
        raise ValueError("transcripts shorter than fragment")
    if bias3:
        u = rng.random(n) ** (1.0 / 3.0)   # density ~ position^2 (3' bias)
        start = np.minimum((u * (l_tx - span + 1)).astype(np.int64), l_tx - span)
    else:
        start = rng.integers(0, l_tx - span + 1, size=n)
    idx = start[:, None] + np.arange(span)[None, :]
    frag = tx[tid[:, None], idx]
    strand = rng.integers(0, 2, size=n).astype(bool)
    frag_rc = COMP[frag[:, ::-1]]
    frag = np.where(strand[:, None], frag_rc, frag)

    def mutate(s):
        s = s.copy()
        q = np.full(s.shape, ord('I'), dtype=np.uint8)
        m = rng.random(s.shape) < e
        shift = rng.integers(1, 4, size=s.shape, dtype=np.uint8)
        sub = NUC[(CODE[s] + shift) & 3]
        s = np.where(m, sub, s)
        q[m] = ord('#')
        if p_n > 0:
            mn = rng.random(s.shape) < p_n
            s[mn] = ord('N')
            q[mn] = ord('!')
        return s, q

    s1, q1 = mutate(frag[:, :length])
    s2 = q2 = None
    if paired:
        s2, q2 = mutate(COMP[frag[:, ::-1]][:, :length])
    lens = None
    if var_len:
        lens = rng.integers(max(10, length // 3), length + 1, size=n)
    return s1, q1, s2, q2, lens


def canonical_codes(seqs, k, lens=None):
    """All valid canonical k-mer codes (reference 2-bit encoding, KmerCode.cpp:7-21) of the
    reads in `seqs` ([n, L] uint8).  k-mers touching a non-ACGT base are skipped."""
    n, L = seqs.shape
    if L < k:
        return np.zeros(0, dtype=np.uint64)
    c = CODE[seqs]
    bad = (c == 255)
    c = np.where(bad, 0, c).astype(np.uint64)
    kc = L - k + 1
    fwd = np.zeros((n, kc), dtype=np.uint64)
    rev = np.zeros((n, kc), dtype=np.uint64)
    nbad = np.zeros((n, kc), dtype=np.int32)
    for j in range(k):
        fwd = (fwd << np.uint64(2)) | c[:, j:j + kc]
        rev = rev | ((np.uint64(3) - c[:, j:j + kc]) << np.uint64(2 * j))
        nbad += bad[:, j:j + kc]
    ok = nbad == 0
    if lens is not None:
        pos = np.arange(kc)[None, :]
        ok &= (pos + k) <= lens[:, None]
    can = np.minimum(fwd, rev)
    return can[ok]


def count_kmers(seq_arrays, k, lens_list=None):
    codes = []
    for i, s in enumerate(seq_arrays):
        if s is None:
            continue
        ln = None if lens_list is None else lens_list[i]
        codes.append(canonical_codes(s, k, ln))
    allc = np.concatenate(codes) if codes else np.zeros(0, dtype=np.uint64)
    keys, cnt = np.unique(allc, return_counts=True)
    keep = cnt >= 2
    return keys[keep], cnt[keep].astype(np.int64)


def decode_kmers(keys, k):
    out = np.empty((len(keys), k), dtype=np.uint8)
    for j in range(k):
        out[:, j] = NUC[((keys >> np.uint64(2 * (k - 1 - j))) & np.uint64(3)).astype(np.uint8)]
    return out


def write_dump(path, keys, cnt, k, order=None):
    txt = decode_kmers(keys, k)
    idx = np.arange(len(keys)) if order is None else order
    with open(path, "wb") as f:
        chunk = []
        for i in idx:
            chunk.append(b">%d\n%s\n" % (cnt[i], txt[i].tobytes()))
            if len(chunk) >= 65536:
                f.write(b"".join(chunk))
                chunk = []
        f.write(b"".join(chunk))


def write_fastq(path, seqs, quals, tag="", lens=None, start_index=0):
    n, L = seqs.shape
    with open(path, "wb") as f:
        chunk = []
        for i in range(n):
            ln = L if lens is None else int(lens[i])
            chunk.append(b"@r%d%s\n%s\n+\n%s\n" % (i + start_index, tag.encode(),
                                                  seqs[i, :ln].tobytes(), quals[i, :ln].tobytes()))
            if len(chunk) >= 65536:
                f.write(b"".join(chunk))
                chunk = []
        f.write(b"".join(chunk))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("-n", type=int, default=2000)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("-k", type=int, default=23)
    ap.add_argument("--n-tx", type=int, default=200)
    ap.add_argument("--l-tx", type=int, default=1500)
    ap.add_argument("--alpha", type=float, default=0.8)
    ap.add_argument("-e", type=float, default=0.01)
    ap.add_argument("--p-n", type=float, default=0.0)
    ap.add_argument("--paired", action="store_true")
    ap.add_argument("--interleaved", action="store_true", help="with --paired: one interleaved file")
    ap.add_argument("--bias3", action="store_true")
    ap.add_argument("--var-len", action="store_true")
    ap.add_argument("--shuffle-dump", action="store_true", help="dump in a seeded random order")
    ap.add_argument("--out", required=True, help="output prefix")
    a = ap.parse_args()
    s1, q1, s2, q2, lens = make_reads(a.seed, a.n, a.len, a.n_tx, a.l_tx, a.alpha, a.e, a.p_n,
                                      a.paired, bias3=a.bias3, var_len=a.var_len)
    keys, cnt = count_kmers([s1, s2], a.k, [lens, lens] if lens is not None else None)
    order = None
    if a.shuffle_dump:
        order = np.random.Generator(np.random.PCG64(a.seed + 7)).permutation(len(keys))
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    write_dump(a.out + ".jf_dump", keys, cnt, a.k, order)
    if a.paired and a.interleaved:
        n, L = s1.shape
        si = np.empty((2 * n, L), dtype=np.uint8)
        qi = np.empty((2 * n, L), dtype=np.uint8)
        si[0::2], si[1::2], qi[0::2], qi[1::2] = s1, s2, q1, q2
        with open(a.out + "_il.fq", "wb") as f:
            for i in range(n):
                ln = L if lens is None else int(lens[i])
                f.write(b"@r%d/1\n%s\n+\n%s\n@r%d/2\n%s\n+\n%s\n" % (
                    i, s1[i, :ln].tobytes(), q1[i, :ln].tobytes(),
                    i, s2[i, :ln].tobytes(), q2[i, :ln].tobytes()))
    elif a.paired:
        write_fastq(a.out + "_1.fq", s1, q1, "/1", lens)
        write_fastq(a.out + "_2.fq", s2, q2, "/2", lens)
    else:
        write_fastq(a.out + ".fq", s1, q1, "", lens)
    print("reads=%d len=%d k=%d dump_kmers=%d" % (a.n * (2 if a.paired else 1), a.len, a.k, len(keys)),
          file=sys.stderr)


if __name__ == "__main__":
    main()
