#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ with the UNMODIFIED reference binary
(oracle/_ref/rcorrector_ref, built from /root/reference by oracle/Makefile).  Run in the build
container only; the fixtures (inputs + the reference's outputs) are committed so that the GPU box,
which has no /root/reference, can check byte-exact parity.

Jellyfish is not available offline, so every dump is the stand-in SURVEY.md §8(c) describes: exact
canonical k-mer counts >= 2 in ascending canonical-code order (or a seeded shuffle where noted).
"""
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
import datasets  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "rcorrector_ref")
SAMPLE = "/root/reference/Sample"


def fq(path, seqs, quals, tag=""):
    with open(path, "wb") as f:
        for i, (s, q) in enumerate(zip(seqs, quals)):
            f.write(b"@r%d%s\n%s\n+\n%s\n" % (i, tag.encode(), s, q))


def run_ref(d, args):
    out = os.path.join(d, "ref")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    cmd = [REF] + args + ["-od", out]
    p = subprocess.run(cmd, cwd=d, stderr=subprocess.PIPE, check=True)
    open(os.path.join(out, "stderr.txt"), "wb").write(p.stderr)
    open(os.path.join(d, "cmd.txt"), "w").write(" ".join(args) + "\n")
    # the -verbose transcript (per-read counts, every (strong, trust) iteration, the trusted
    # bitmap, post-correction counts: ErrorCorrection.cpp:686-689,759-770,856-857,1088-1094,
    # 1590-1597), kept gzip-compressed next to the outputs
    import gzip
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        pv = subprocess.run([REF] + args + ["-od", tmp, "-verbose"], cwd=d, stdout=subprocess.PIPE,
                            stderr=subprocess.DEVNULL, check=True)
    with gzip.GzipFile(os.path.join(d, "verbose.txt.gz"), "wb", mtime=0) as g:
        g.write(pv.stdout)


def dump_of(path, arrays, k, lens=None, order=None):
    keys, cnt = synth.count_kmers(arrays, k, lens)
    synth.write_dump(path, keys, cnt, k, order)
    return len(keys)


def rows(a, lens, n):
    return [a[i, :(a.shape[1] if lens is None else lens[i])].tobytes() for i in range(n)]


def main():
    if not os.path.exists(REF):
        raise SystemExit("build oracle/_ref first (make -C oracle ref)")
    # --- config 1: the reference's own Sample pair -------------------------------------------------
    d = os.path.join(HERE, "fx_sample")
    os.makedirs(d, exist_ok=True)
    for n in ("sample_read1.fq", "sample_read2.fq"):
        shutil.copy(os.path.join(SAMPLE, n), os.path.join(d, n))
    reads = []
    for n in ("sample_read1.fq", "sample_read2.fq"):
        lines = open(os.path.join(d, n), "rb").read().split(b"\n")
        reads += [lines[i] for i in range(1, len(lines), 4)]
    L = max(len(r) for r in reads)
    arr = np.full((len(reads), L), ord('N'), dtype=np.uint8)
    lens = np.array([len(r) for r in reads])
    for i, r in enumerate(reads):
        arr[i, :len(r)] = np.frombuffer(r, dtype=np.uint8)
    dump_of(os.path.join(d, "dump.jf"), [arr], 23, [lens])
    run_ref(d, ["-p", "sample_read1.fq", "sample_read2.fq", "-k", "23", "-c", "dump.jf"])

    # --- synthetic sets ---------------------------------------------------------------------------
    def synth_set(name, k, n_keep, extra, paired=False, interleaved=False, shuffle=False, **kw):
        d = os.path.join(HERE, name)
        os.makedirs(d, exist_ok=True)
        seed, n, length = kw.pop("seed"), kw.pop("n"), kw.pop("length")
        s1, q1, s2, q2, lens = synth.make_reads(seed, n, length, paired=paired, **kw)
        nk = len(synth.count_kmers([s1, s2], k, [lens, lens] if lens is not None else None)[0])
        order = np.random.Generator(np.random.PCG64(seed + 7)).permutation(nk) if shuffle else None
        dump_of(os.path.join(d, "dump.jf"), [s1, s2], k, [lens, lens] if lens is not None else None, order)
        r1, qq1 = rows(s1, lens, n_keep), rows(q1, lens, n_keep)
        if paired and interleaved:
            r2, qq2 = rows(s2, lens, n_keep), rows(q2, lens, n_keep)
            with open(os.path.join(d, "reads_il.fq"), "wb") as f:
                for i in range(n_keep):
                    f.write(b"@r%d/1\n%s\n+\n%s\n@r%d/2\n%s\n+\n%s\n" % (i, r1[i], qq1[i], i, r2[i], qq2[i]))
            args = ["-i", "reads_il.fq"]
        elif paired:
            fq(os.path.join(d, "reads_1.fq"), r1, qq1, "/1")
            fq(os.path.join(d, "reads_2.fq"), rows(s2, lens, n_keep), rows(q2, lens, n_keep), "/2")
            args = ["-p", "reads_1.fq", "reads_2.fq"]
        else:
            fq(os.path.join(d, "reads.fq"), r1, qq1)
            args = ["-r", "reads.fq"]
        run_ref(d, args + ["-k", str(k), "-c", "dump.jf"] + extra)

    synth_set("fx_se_k23", 23, 400, [], seed=201, n=400, length=100, e=0.01, n_tx=20)
    synth_set("fx_pe_k23", 23, 300, [], paired=True, seed=202, n=300, length=150, e=0.005, n_tx=20)
    synth_set("fx_il_k23", 23, 300, [], paired=True, interleaved=True, seed=202, n=300, length=150, e=0.005, n_tx=20)
    synth_set("fx_k31_mc8", 31, 300, ["-maxcorK", "8"], seed=204, n=300, length=150, e=0.05, n_tx=10)
    synth_set("fx_skew", 23, 400, [], seed=205, n=400, length=150, e=0.005, alpha=1.5, bias3=True, n_tx=20)
    synth_set("fx_highcov", 23, 300, ["-wk", "0.9"], shuffle=True, seed=206, n=20000, length=100, e=0.002,
              n_tx=1, l_tx=300, alpha=0.0)
    synth_set("fx_k32", 32, 300, [], seed=207, n=300, length=150, e=0.01, n_tx=10)
    synth_set("fx_k15", 15, 300, [], seed=208, n=300, length=75, e=0.01, n_tx=10)
    synth_set("fx_varlen_n", 23, 400, [], seed=209, n=400, length=100, e=0.02, var_len=True, p_n=0.01, n_tx=20)

    # --- adversarial edge reads (short, N-rich, IUPAC, poly-A/T, ...) -------------------------------
    d = os.path.join(HERE, "fx_edge")
    os.makedirs(d, exist_ok=True)
    r, q = datasets.adversarial_reads()
    s1, _, _, _, _ = synth.make_reads(7, 600, 100, e=0.01)
    dump_of(os.path.join(d, "dump.jf"), [s1], 23)
    fq(os.path.join(d, "reads.fq"), r, q)
    run_ref(d, ["-r", "reads.fq", "-k", "23", "-c", "dump.jf"])
    # --- FASTA input (Reads.h:108-162).  The reference's -t 1 loop hands ErrorCorrection a NULL quality
    # pointer for FASTA records and crashes in the pairwise veto (main.cpp:376-379, ErrorCorrection.cpp:1316);
    # its batch path (-t > 1) passes a buffer whose first byte is 0 (Reads.h:241) and works: that is the
    # behaviour pinned here.  No -verbose transcript (threads interleave their prints).  The directory
    # is not named fx_*: the fixture-wide tests (which add -verbose / assume -t 1) do not apply.
    d = os.path.join(HERE, "fa_se_k23")
    os.makedirs(os.path.join(d, "ref"), exist_ok=True)
    lines = open(os.path.join(HERE, "fx_se_k23", "reads.fq"), "rb").read().split(b"\n")
    with open(os.path.join(d, "reads.fa"), "wb") as f:
        for i in range(0, len(lines) - 1, 4):
            f.write(b">" + lines[i][1:] + b"\n" + lines[i + 1] + b"\n")
    shutil.copy(os.path.join(HERE, "fx_se_k23", "dump.jf"), os.path.join(d, "dump.jf"))
    args = ["-r", "reads.fa", "-k", "23", "-c", "dump.jf", "-t", "2"]
    p = subprocess.run([REF] + args + ["-od", os.path.join(d, "ref")], cwd=d, stderr=subprocess.PIPE, check=True)
    open(os.path.join(d, "ref", "stderr.txt"), "wb").write(p.stderr)
    open(os.path.join(d, "cmd.txt"), "w").write(" ".join(args) + "\n")
    sz = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(HERE) for f in fs)
    print("golden fixtures: %.1f MB" % (sz / 1e6))


if __name__ == "__main__":
    main()
