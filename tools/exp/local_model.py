#!/usr/bin/env python3
"""Model of a minimiser-homed k-mer table (DESIGN.md §8 item 2, round-5 review item 1) on the bench's own
synthetic reads, CPU only.  Test / experiment infrastructure, not product code.

Builds the k-mer table of a scaled-down preset (same coverage: transcripts and reads scaled together), then for
m in a range of minimiser lengths reports
  * how the table's k-mers distribute over minimisers (group sizes: mean, 99 %, max),
  * how far entries are displaced when groups are placed into regions of R slots at a given load (linear probing by
    region, the build's prefix-max scan), and
  * how many distinct 128-byte lines the k-mers of one read touch: hashed homes against minimiser homes.
"""
import argparse
import sys
import os
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import synth_int  # noqa: E402

U = np.uint64


def revcomp(x, k):
    """reverse complement of 2-bit codes (first base most significant), numpy uint64"""
    y = np.zeros_like(x)
    t = (~x) & U((1 << (2 * k)) - 1)
    for _ in range(k):
        y = (y << U(2)) | (t & U(3))
        t = t >> U(2)
    return y


def mix32(x):
    x = x.astype(np.uint64)
    x = (x * U(0x9E3779B1)) & U(0xFFFFFFFF)
    x ^= x >> U(15)
    x = (x * U(0x85EBCA77)) & U(0xFFFFFFFF)
    x ^= x >> U(13)
    x = (x * U(0xC2B2AE3D)) & U(0xFFFFFFFF)
    x ^= x >> U(16)
    return x


def mix64(z):
    z = z.astype(np.uint64)
    z ^= z >> U(30)
    z = z * U(0xbf58476d1ce4e5b9)
    z ^= z >> U(27)
    z = z * U(0x94d049bb133111eb)
    z ^= z >> U(31)
    return z


def minimiser(codes, k, m):
    """hash of the smallest-hash canonical m-mer inside each k-mer code (strand-symmetric)"""
    best = np.full(codes.shape, np.iinfo(np.uint64).max, dtype=np.uint64)
    mm = U((1 << (2 * m)) - 1)
    for p in range(k - m + 1):
        w = (codes >> U(2 * (k - m - p))) & mm
        c = np.minimum(w, revcomp(w, m))
        best = np.minimum(best, mix32(c) << U(32) | c)
    return best  # (hash << 32 | canonical m-mer): equal iff the same minimiser


def read_kmers(seq2d, k):
    """forward codes of every k-mer of every read: (n, L-k+1) uint64"""
    n, L = seq2d.shape
    lut = np.zeros(256, dtype=np.uint64)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
    b = lut[seq2d]
    out = np.zeros((n, L - k + 1), dtype=np.uint64)
    cur = np.zeros(n, dtype=np.uint64)
    mask = U((1 << (2 * k)) - 1)
    for i in range(L):
        cur = ((cur << U(2)) | b[:, i]) & mask
        if i >= k - 1:
            out[:, i - k + 1] = cur
    return out


def place(group_of_entry, n_regions, R):
    """entries sorted by home region, slot p_j = max(R * home_j, p_{j-1} + 1): displacement in regions"""
    home = np.sort(group_of_entry)
    j = np.arange(home.size, dtype=np.int64)
    q = R * home.astype(np.int64) - j
    p = np.maximum.accumulate(q) + j
    return p // R - home


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.04, help="fraction of the preset (transcripts and reads)")
    ap.add_argument("-k", type=int, default=23)
    ap.add_argument("--err", type=float, default=0.005)
    ap.add_argument("--len", type=int, default=150)
    ap.add_argument("--reads", type=int, default=25_000_000)
    ap.add_argument("--ms", default="11,13,15")
    ap.add_argument("--load", type=float, default=0.45)
    ap.add_argument("--sample", type=int, default=20000)
    a = ap.parse_args()
    k, L = a.k, a.len
    n_tx = max(50, int(30000 * a.scale))
    n_reads = int(a.reads * a.scale) & ~1
    g = synth_int.Synth(1002, L, n_tx, 1500, 0.8, a.err, True)
    seq, _ = g.generate(0, n_reads // 2)
    seq2d = seq.numpy().reshape(n_reads, L + 1)[:, :L]
    print("reads %d x %d, %d transcripts, k = %d, err = %g" % (n_reads, L, n_tx, k, a.err), flush=True)
    canon_all = []
    CH = 200000
    for lo in range(0, n_reads, CH):
        f = read_kmers(seq2d[lo:lo + CH], k).reshape(-1)
        canon_all.append(np.minimum(f, revcomp(f, k)))
    canon_all = np.concatenate(canon_all)
    keys, counts = np.unique(canon_all, return_counts=True)
    del canon_all
    keys = keys[counts >= 2]
    n = keys.size
    print("table: %d k-mers (count >= 2)" % n, flush=True)
    samp = seq2d[:: max(1, n_reads // a.sample)][: a.sample]
    f = read_kmers(samp, k)
    sc = np.minimum(f, revcomp(f.reshape(-1), k).reshape(f.shape))
    in_table = np.isin(sc, keys)
    print("sample: %d reads, %.1f %% of their k-mers in the table" % (samp.shape[0], 100.0 * in_table.mean()))
    # hashed homes: 64-byte buckets of 8 slots, lines of 128 bytes = 2 buckets
    nb = int(n / (8 * a.load)) + 1
    hb = ((mix64(sc) >> U(32)) * U(nb)) >> U(32)
    lines = np.array([np.unique(r >> U(1)).size for r in hb])
    print("hashed homes: %.1f distinct 128-B lines per read (of %d k-mers)" % (lines.mean(), sc.shape[1]))
    for m in [int(x) for x in a.ms.split(",")]:
        mt = minimiser(keys, k, m)
        _, inv, gs = np.unique(mt, return_inverse=True, return_counts=True)
        srt = np.sort(gs)
        print("m = %d: %d minimisers, k-mers per minimiser mean %.2f, 50/90/99/99.9 %% %d/%d/%d/%d, max %d; "
              "entry-weighted share in groups > 16: %.2f %%, > 64: %.2f %%"
              % (m, gs.size, gs.mean(), srt[gs.size // 2], srt[int(gs.size * 0.9)], srt[int(gs.size * 0.99)],
                 srt[int(gs.size * 0.999)], srt[-1], 100.0 * gs[gs > 16].sum() / n, 100.0 * gs[gs > 64].sum() / n))
        for R in (8, 16, 32):
            n_reg = int(n / (R * a.load)) + 1
            reg_of_min = ((mix64(np.unique(mt)) >> U(32)) * U(n_reg)) >> U(32)
            d = place(reg_of_min[inv], n_reg, R)
            # a cap on group size: heavier groups fall back to hashed homes (their k-mers hash individually)
            heavy = gs[inv] > 2 * R
            home2 = np.where(heavy, ((mix64(keys) >> U(32)) * U(n_reg)) >> U(32), reg_of_min[inv])
            d2 = place(home2, n_reg, R)
            print("   regions of %2d slots (%3d B) at load %.2f: displaced > 0: %.1f %%, > 1: %.2f %%, > 3: %.3f %%, max %d | "
                  "with groups > %d hashed (%.2f %% of entries): > 0: %.1f %%, > 1: %.2f %%, > 3: %.3f %%, max %d"
                  % (R, R * 8, a.load, 100.0 * (d > 0).mean(), 100.0 * (d > 1).mean(), 100.0 * (d > 3).mean(), d.max(), 2 * R,
                     100.0 * heavy.mean(), 100.0 * (d2 > 0).mean(), 100.0 * (d2 > 1).mean(), 100.0 * (d2 > 3).mean(), d2.max()))
        # lines per read under minimiser homes: region of 128 B (16 slots) = one line
        ms = minimiser(sc.reshape(-1), k, m).reshape(sc.shape)
        n_reg = int(n / (16 * a.load)) + 1
        rg = ((mix64(ms) >> U(32)) * U(n_reg)) >> U(32)
        lines = np.array([np.unique(r).size for r in rg])
        runs = np.array([1 + np.count_nonzero(r[1:] != r[:-1]) for r in rg])
        print("   minimiser homes: %.1f distinct 128-B lines per read, %.1f runs of equal minimiser" % (lines.mean(), runs.mean()), flush=True)


if __name__ == "__main__":
    main()
