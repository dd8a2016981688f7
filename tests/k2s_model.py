"""A restatement, over the oracle's own primitives, of the conditions under which the GPU path finishes a read WITHOUT
the general search (rcorrector_amd/csrc/rc_quarter.h: the clean test on the real counts; rc_single.h: conditions (0)-(6)
of k_single) -- the model those kernels were written from.  `finished_early` returns None when the read has to go
through the search, else (ret, corrected read, l, m, h): what ErrorCorrection + GetKmerInformation
(ErrorCorrection.cpp:682-1480, :1567-1602) must produce for it.  tests/test_k2s_model.py checks exactly that against
the oracle.  `allow_double` adds "class D" (two substitutions less than k bases apart inside the read; DESIGN.md
section 3: validated here, not built on the device).  Test infrastructure: nothing in the product imports this."""
import ctypes as C

import numpy as np

from oracle import pyoracle as po

L_ = None


def _lib():
    global L_
    if L_ is None:
        po.build()
        L_ = po.lib()
    return L_


def bound_i(P, c):
    return _lib().rco_get_bound_int(C.byref(P), int(c))


def bound_d(P, c):
    return _lib().rco_get_bound(C.byref(P), int(c))


def polya(seq, k, thr):
    a = np.frombuffer(seq, np.uint8)
    cA = np.concatenate([[0], np.cumsum(a == 65)]); cT = np.concatenate([[0], np.cumsum(a == 84)])
    na = cA[k:] - cA[:-k]; nt = cT[k:] - cT[:-k]
    return (na >= k - thr) | (nt >= k - thr)
def finished_early(P, T, seq, k, mfk, strong0, info0, pair_t, allow_double=False, max_fix=3):
    Ln = len(seq)
    if Ln < k or (info0 & 4): return None
    if any(ch not in b"ACGT" for ch in seq): return None
    kc = Ln - k + 1
    counts = po.kmer_counts(P, T, seq).astype(np.int64)
    strong = strong0; flag = False
    trust = bound_i(P, strong)
    if info0 & 1:
        if strong >= 20 and (info0 & 2) and trust < 3: flag = True; trust = 3
    if pair_t >= 1 and strong > pair_t:
        if (not flag) or pair_t < 20: trust = bound_i(P, pair_t)
        strong = pair_t
    if trust < 2: trust = 2
    t = trust
    pa2 = polya(seq, k, 2)
    Tm = (counts >= strong) & ~pa2
    if counts.min() >= t and bool((Tm[:-1] & Tm[1:]).any()):     # condition (0): nothing to correct -- a real island (two
        # adjacent trusted k-mers, ErrorCorrection.cpp:870-931) and no count below the weak threshold
        v = np.sort(np.where(counts == 0, 1, counts))
        return 0, bytes(seq), int(v[0]), int(v[len(v) // 2]), int(v[-1])
    d_ = np.diff(np.concatenate([[0], Tm.astype(np.int8), [0]]))
    starts = np.nonzero(d_ == 1)[0]; ends = np.nonzero(d_ == -1)[0] - 1
    if len(starts) < 1: return None
    if ((ends - starts + 1) < 2).any(): return None
    segs = []   # (z0, z1, kind)
    if starts[0] > 0:
        if starts[0] > k: return None
        segs.append((0, int(starts[0]) - 1, 'L'))
    for i in range(len(starts) - 1):
        z0, z1 = int(ends[i]) + 1, int(starts[i + 1]) - 1
        zl = z1 - z0 + 1
        if zl == k: segs.append((z0, z1, 'M'))
        elif allow_double and k < zl <= 2 * k - 1: segs.append((z0, z1, 'D'))
        else: return None
    if ends[-1] < kc - 1:
        z0 = int(ends[-1]) + 1
        if kc - z0 > k: return None
        segs.append((z0, kc - 1, 'R'))
    nfix = sum(2 if s[2] == 'D' else 1 for s in segs)
    if not segs or nfix > min(max_fix, mfk - 1): return None
    cur = bytearray(seq)          # the read with the fixes made so far
    newcounts = counts.copy()
    best_bott = 10**9
    def cnt_of(s, w): return int(po.kmer_counts(P, T, bytes(s))[w])
    def ext4(w, vb):              # the four counts of window w of `cur` with base vb replaced by each letter
        out = []
        for c in range(4):
            s2 = bytearray(cur); s2[vb] = b"ACGT"[c]; out.append(cnt_of(s2, w))
        return out
    prev_z1 = -1
    for si, (z0, z1, kind) in enumerate(segs):
        zl = z1 - z0 + 1
        next_z0 = segs[si + 1][0] if si + 1 < len(segs) else kc
        if kind == 'L': right = False
        elif kind == 'R': right = True
        else: right = (z0 - prev_z1 - 1) >= (next_z0 - z1 - 1)
        prev_z1 = z1
        d = zl - k if kind == 'D' else 0
        U = (lambda i: z0 + i) if right else (lambda i: z1 - i)
        VB = (lambda i: U(i) + k - 1) if right else (lambda i: U(i))
        if kind == 'R': U = lambda i: z0 + i; VB = lambda i: z0 + i + k - 1
        if kind == 'L': U = lambda i: z1 - i; VB = lambda i: z1 - i
        tt = t                    # the t handed down
        bott = 10**9
        def sub_node(i, tt):
            w = U(i); vb = VB(i)
            b = b"ACGT".index(bytes([cur[vb]]))
            c4 = ext4(w, vb)
            mx = max(c4 + [0]); ret = max(1, bound_i(P, mx))
            thr = ret if (tt > ret or tt <= 0) else tt
            if c4[b] >= thr: return None
            if thr == 1 and tt <= 2: return None
            if pa2[w]: return None
            cands = [c for c in range(4) if c != b and c4[c] >= thr]
            if len(cands) != 1: return None
            c = cands[0]
            if c4[c] < t: return None
            return c, c4[c], thr
        r0 = sub_node(0, tt)
        if r0 is None: return None
        c1, x1, thr0 = r0
        if kind == 'D' and not right and thr0 != tt: return None
        cur[VB(0)] = b"ACGT"[c1]; newcounts[U(0)] = x1; bott = min(bott, x1)
        if kind == 'D':
            if counts[z0:z1 + 1].max() > 1: return None       # pairwise veto cannot fire
            for i in range(1, d):
                w = U(i); vb = VB(i)
                b = b"ACGT".index(bytes([cur[vb]]))
                c4 = ext4(w, vb)
                x = c4[b]
                if x < t: return None
                if not right and bound_i(P, x) < t: return None
                mx = max(c4 + [0]); ret = max(1, bound_i(P, mx))
                thr = ret if (tt > ret or tt <= 0) else tt
                if any(c4[c] >= thr for c in range(4) if c != b): return None
                newcounts[w] = x; bott = min(bott, x)
            r1 = sub_node(d, tt)
            if r1 is None: return None
            c2, x2, thr1 = r1
            cur[VB(d)] = b"ACGT"[c2]; newcounts[U(d)] = x2; bott = min(bott, x2)
        # keep-only nodes behind the last substitution; right searches inside the read: the last window is not a node
        for i in range(d + 1, zl):
            w = U(i)
            x = cnt_of(cur, w)
            node = not (right and kind in ('M', 'D') and i == zl - 1)
            if node:
                if x < t: return None
                if not right and bound_i(P, x) < t: return None
                bott = min(bott, x)
            newcounts[w] = x
        if bott < t: return None
        best_bott = min(best_bott, bott)
    if strong >= 0 and float(best_bott) < bound_d(P, strong): return None
    v = np.where(newcounts == 0, 1, newcounts); v.sort()
    return nfix, bytes(cur), int(v[0]), int(v[len(v) // 2]), int(v[-1])


def front_end(P, T, seqs, k):
    """strong threshold and the info bits of rc_front_end() (found / prev == 2 / screened) for every read"""
    strong = np.array([_lib().rco_strong_trusted_threshold(C.byref(P), T.h, s) for s in seqs])
    info = np.zeros(len(seqs), int)
    thr7 = max(7, k // 2)
    for i, s in enumerate(seqs):
        Ln = len(s)
        if Ln < k or s.count(b"N") > 5 or s.count(b"A") > Ln - k or s.count(b"T") > Ln - k:
            info[i] = 4
            continue
        c = po.kmer_counts(P, T, s).astype(np.int64)
        v = np.where(polya(s, k, thr7), -1, c)
        v.sort()
        for j in range(len(v) - 1, 0, -1):
            if v[j] > 2 * v[j - 1] and v[j] > 10:
                info[i] = 1 | (2 if v[j - 1] == 2 else 0)
                break
    return strong, info
