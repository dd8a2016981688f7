#!/usr/bin/env python3
"""synth-v1, integer edition: the bench's read generator as a PURE FUNCTION of (seed, read index).

Test / bench infrastructure (not part of the product path).  SURVEY.md §8(d): transcriptome of
n_tx x l_tx iid-uniform bases, expression weight of transcript i ~ (i+1)^-alpha, read = transcript by
weight, start uniform (or with a 3' bias: density ~ position^2), strand uniform, per-base
substitution with probability e (uniform among the other three), quality 'I' ('#' on substituted
bases); paired: fragment of frag_len bases, mate 2 = reverse complement of the fragment's tail.

Every random draw is splitmix64 of a counter -- mix(key(seed, stream) + index * golden) -- evaluated
in int64 tensor arithmetic (wrapping multiply, logical shifts spelled out), so the bytes of read g
depend on (seed, g) only: not on the device, the chunking, the rank that generates it or any RNG
state.  That is what lets every rank of a multi-GPU run rebuild the SAME replicated k-mer table from
"shard 0" without communication (bench.py checks the table digests against each other anyway), and
what makes the driver's table_kmers reproducible run to run.
"""
import numpy as np
import torch

M64 = (1 << 64) - 1
LMAX = 1024  # stride of the per-base counters (reads hold < 1024 bases, utils.h:7)


def _s(x):
    """python int (mod 2^64) -> the same bits as a signed 64-bit value"""
    x &= M64
    return x - (1 << 64) if x >= (1 << 63) else x


C1, C2, C3 = _s(0x9E3779B97F4A7C15), _s(0xBF58476D1CE4E5B9), _s(0x94D049BB133111EB)


def lsr(z, s):
    """logical shift right of an int64 tensor"""
    return (z >> s) & ((1 << (64 - s)) - 1)


def mix(z):
    """splitmix64 step on an int64 tensor (wrapping arithmetic)"""
    z = z + C1
    z = (z ^ lsr(z, 30)) * C2
    z = (z ^ lsr(z, 27)) * C3
    return z ^ lsr(z, 31)


def mix_py(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def key(seed, stream):
    return _s(mix_py(mix_py(seed & M64) ^ ((stream * 0xD1342543DE82EF95) & M64)))


def rnd(k, idx):
    """64 random bits for every counter value in the int64 tensor idx, stream key k"""
    return mix(idx * C1 + k)


S_TX, S_TID, S_START, S_START2, S_START3, S_STRAND, S_MUT = range(1, 8)


class Synth:
    def __init__(self, seed, length, n_tx=30000, l_tx=1500, alpha=0.8, err=0.005, paired=False, frag_len=300,
                 bias3=False, device="cpu"):
        self.seed, self.L, self.n_tx, self.l_tx, self.err = seed, length, n_tx, l_tx, err
        self.paired, self.bias3, self.dev = paired, bias3, torch.device(device)
        self.span = frag_len if paired else length
        if self.paired and self.span < length:
            raise ValueError("fragment shorter than the reads")
        if self.span > l_tx:
            raise ValueError("transcripts shorter than the fragment")
        # the transcriptome depends on the seed's thousands only (seed 1002 and 1002 + rank share it)
        ktx = key(seed // 1000, S_TX)
        i = torch.arange(n_tx * l_tx, dtype=torch.int64, device=self.dev)
        self.tx = (lsr(rnd(ktx, i), 7) & 3).to(torch.uint8)
        # inverse CDF of the expression weights on a 2^53 grid (float64 on the host: every rank
        # computes the same table from the same numbers)
        w = (np.arange(n_tx, dtype=np.float64) + 1.0) ** (-alpha)
        c = np.cumsum(w / w.sum())
        cdf = np.minimum((c * float(1 << 53)).astype(np.int64), (1 << 53) - 1)
        cdf[-1] = (1 << 53) - 1
        self.cdf = torch.from_numpy(cdf).to(self.dev)
        self.thr = int(err * float(1 << 53))
        self.lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=self.dev)

    def units_per_read(self):
        return 2 if self.paired else 1

    def generate(self, unit0, n_units, chunk=1 << 17):
        """Units (reads, or pairs) [unit0, unit0 + n_units).  Returns (seq, qual): uint8 tensors of
        n_reads * (L+1) bytes, a NUL after every read; paired: all first mates, then all second mates."""
        L, dev = self.L, self.dev
        n_reads = n_units * self.units_per_read()
        seq = torch.zeros((n_reads, L + 1), dtype=torch.uint8, device=dev)
        qual = torch.zeros((n_reads, L + 1), dtype=torch.uint8, device=dev)
        ar = torch.arange(self.span, dtype=torch.int64, device=dev)
        pos = torch.arange(L, dtype=torch.int64, device=dev)
        k_tid, k_s1, k_s2, k_s3 = (key(self.seed, s) for s in (S_TID, S_START, S_START2, S_START3))
        k_str, k_mut = key(self.seed, S_STRAND), key(self.seed, S_MUT)
        R = self.l_tx - self.span + 1

        def emit(codes, g, mate, lo, m):
            e = ((g * 2 + mate) * LMAX)[:, None] + pos[None, :]
            h = rnd(k_mut, e)
            mut = lsr(h, 11) < self.thr
            shift = 1 + lsr(h, 40) % 3
            codes = torch.where(mut, (codes + shift) & 3, codes)
            seq[lo:lo + m, :L] = self.lut[codes]
            q = torch.full((m, L), ord('I'), dtype=torch.uint8, device=dev)
            q[mut] = ord('#')
            qual[lo:lo + m, :L] = q

        for lo in range(0, n_units, chunk):
            m = min(chunk, n_units - lo)
            g = torch.arange(unit0 + lo, unit0 + lo + m, dtype=torch.int64, device=dev)
            tid = torch.searchsorted(self.cdf, lsr(rnd(k_tid, g), 11), right=True).clamp_(max=self.n_tx - 1)
            r = lsr(rnd(k_s1, g), 32)
            if self.bias3:  # max of three uniforms: density ~ position^2
                r = torch.maximum(r, torch.maximum(lsr(rnd(k_s2, g), 32), lsr(rnd(k_s3, g), 32)))
            start = (r * R) >> 32
            codes = self.tx[(tid * self.l_tx + start)[:, None] + ar[None, :]].to(torch.int64)
            rev = lsr(rnd(k_str, g), 63).bool()
            codes = torch.where(rev[:, None], 3 - codes.flip(1), codes)
            emit(codes[:, :L], g, 0, lo, m)
            if self.paired:
                emit((3 - codes.flip(1))[:, :L], g, 1, n_units + lo, m)
        return seq.reshape(-1), qual.reshape(-1)
