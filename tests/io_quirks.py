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
