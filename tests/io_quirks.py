"""Random FASTQ files with input quirks -- empty / one-base / very short sequences, quality lines
longer than the sequence, '+' lines with text, long headers with blanks, a
file without its final newline -- plus a matching k-mer dump.  Shared by the CPU test (oracle CLI ==
the unmodified reference on them) and the GPU test (rcorrector == oracle CLI).

Left out on purpose, because the reference's own behaviour is undefined there: characters outside
'A'-'Z' such as lower-case bases or the '\r' of CR-LF files (they index past its 26-entry
nucleotide table, KmerCode.cpp / main.cpp:17-22) and records cut short by the end of the file (the
missing lines keep whatever the reused record buffer held -- a different record for every -t), and
for the same reason quality lines SHORTER than their sequence (the vetoes then read the bytes an
earlier record left behind the short line)."""
import os

import numpy as np

import synth


def _quirky_records(rng, seqs, quals, k, tag):
    p_quirk = float(rng.choice([0.02, 0.1, 0.4]))
    nl = b"\n"
    out = []
    q = b""
    for i in range(len(seqs)):
        r, q = seqs[i].tobytes(), quals[i].tobytes()
        h = b"@r%d%s" % (i, tag)
        plus = b"+"
        if rng.random() < p_quirk:
            u = int(rng.integers(0, 10))
            if u == 0:
                r = q = b""
            elif u == 1:
                r, q = r[:1], q[:1]
            elif u == 2:
                r = r[:k]        # exactly one k-mer
                q = q[:k]
            elif u == 3:
                q = q + b"#" * int(rng.integers(1, 4))
            elif u == 4:
                q = q + b"I" * int(rng.integers(1, 9))
            elif u == 5:
                r, q = r[:k - 1], q[:k - 1]   # one base short of a k-mer
            elif u == 6:
                plus = b"+r%d some text" % i
            elif u == 7:
                h = b"@r%d%s " % (i, tag) + b"x" * int(rng.integers(100, 900)) + b"\tend of header"
            elif u == 8:
                r = b""          # empty sequence, quality kept
            elif u == 9:
                ln = int(rng.integers(1, k + 3))
                r, q = r[:ln], q[:ln]
        out.append(h + nl + r + nl + plus + nl + q + nl)
    return out, len(q)


def _write(path, records, strip_final_newline, gz):
    data = b"".join(records)
    if strip_final_newline:
        data = data[:-1]                             # no newline at the end of the file
    if gz:
        import gzip
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


def make_case(seed, d, n=120, length=70, k=23, modes=(0,)):
    """Writes the input file(s) and the dump into d; returns the command-line arguments.
    modes: which layouts may be drawn -- 0 single-end, 1 paired files, 2 interleaved; inputs are
    gzip-compressed in a fifth of the cases when more than single-end is allowed."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mode = int(rng.choice(list(modes)))
    gz = len(modes) > 1 and rng.random() < 0.2
    ext = ".fq.gz" if gz else ".fq"
    s1, q1, s2, q2, _ = synth.make_reads(seed, n, length, n_tx=3, l_tx=300, e=0.01, paired=mode != 0, frag_len=2 * length)
    keys, cnt = synth.count_kmers([s1, s2], k)
    synth.write_dump(os.path.join(d, "d.jf"), keys, cnt, k)
    rec1, lastq1 = _quirky_records(rng, s1, q1, k, b"/1" if mode else b"")
    if mode == 0:
        _write(os.path.join(d, "a" + ext), rec1, rng.random() < 0.3 and lastq1 > 0, gz)
        args = ["-r", "a" + ext]
    else:
        rec2, lastq2 = _quirky_records(rng, s2, q2, k, b"/2")
        if mode == 1:
            _write(os.path.join(d, "a_1" + ext), rec1, rng.random() < 0.3 and lastq1 > 0, gz)
            _write(os.path.join(d, "a_2" + ext), rec2, rng.random() < 0.3 and lastq2 > 0, gz)
            args = ["-p", "a_1" + ext, "a_2" + ext]
        else:
            both = [x for pair in zip(rec1, rec2) for x in pair]
            _write(os.path.join(d, "a_il" + ext), both, rng.random() < 0.3 and lastq2 > 0, gz)
            args = ["-i", "a_il" + ext]
    if len(modes) > 1 and rng.random() < 0.3:
        # a second, single-end input in the same run (files are processed one after the other; the
        # bad-quality scan covers the primary files in order, main.cpp:88-128)
        s3, q3, _, _, _ = synth.make_reads(seed + 7919, n // 2, length, n_tx=3, l_tx=300, e=0.01)
        rec3, lastq3 = _quirky_records(rng, s3, q3, k, b"")
        _write(os.path.join(d, "b" + ext), rec3, rng.random() < 0.3 and lastq3 > 0, gz)
        args = (["-r", "b" + ext] + args) if rng.random() < 0.5 else (args + ["-r", "b" + ext])
    extra = []
    if len(modes) > 1 and rng.random() < 0.2:
        extra.append("-stdout")
    if len(modes) > 1 and rng.random() < 0.2:
        extra += ["-maxcorK", str(int(rng.integers(1, 7)))]
    return args + ["-k", str(k), "-c", "d.jf"] + extra


def make_quirky_dump(seed, path, k=23, n=4000, mid_n=True):
    """A k-mer dump in the layout `jellyfish dump` writes with departures from it sprinkled in --
    what main.cpp:295-307 (two fscanf("%s") per entry, atoi on the first, every letter of the second
    pushed through KmerCode) accepts: blanks and tabs between tokens, empty lines, signs and trailing
    junk in the count, counts of 0 / 1 (dropped) and beyond int range, k-mers that are too long or too
    short or hold N, repeated k-mers (the later count wins), no final newline.
    mid_n=False leaves out the k-mers with an N before their last base: the first of them ends the
    ERROR_RATE scan (the KmerCode object stays invalid, main.cpp:323), so the estimate falls back to
    0.01 for most dumps that hold one."""
    rng = np.random.Generator(np.random.PCG64(seed))
    codes = rng.integers(0, 4, size=(n, k), dtype=np.uint8)
    kmers = [synth.NUC[c].tobytes() for c in codes]
    out = []
    for i, km in enumerate(kmers):
        cnt = int(rng.choice([0, 1, 2, 3, 7, 50, 1200, 5000, 70000]))
        head = b">%d" % cnt
        sep1, sep2 = b"\n", b"\n"
        if rng.random() < 0.15:
            u = int(rng.integers(0, 10))
            if u == 0:
                head = b">+%d" % cnt
            elif u == 1:
                head = b">%dx7" % cnt
            elif u == 2:
                sep1 = b" "
            elif u == 3:
                sep1, sep2 = b"\t", b"\n\n"
            elif u == 4 and mid_n:
                km = km[:10] + b"N" + km[11:]
            elif u == 5:
                km = km + b"ACG"
            elif u == 6:
                km = km[:k - 2]
            elif u == 7:
                km = kmers[int(rng.integers(0, i + 1))]     # an earlier k-mer again
            elif u == 8:
                head = b">99999999999"
            elif u == 9:
                km = km[:k - 1] + b"N"
        out.append(head + sep1 + km + sep2)
        if cnt >= 1000 and len(km) == k and rng.random() < 0.5:
            # a sibling that differs in the last base, with a smaller count: what the ERROR_RATE pass
            # (main.cpp:310-358) looks for
            sib = km[:-1] + (b"A" if km[-1:] != b"A" else b"C")
            out.append(b">%d\n" % (cnt // int(rng.integers(20, 400)) + 2) + sib + b"\n")
    data = b"".join(out)
    if rng.random() < 0.5:
        data = data.rstrip(b"\n")
    with open(path, "wb") as f:
        f.write(data)
