"""Random FASTQ files with input quirks -- empty / one-base / very short sequences, quality lines
shorter or longer than the sequence (or empty), '+' lines with text, long headers with blanks, a
file without its final newline -- plus a matching k-mer dump.  Shared by the CPU test (oracle CLI ==
the unmodified reference on them) and the GPU test (rcorrector == oracle CLI).

Left out on purpose, because the reference's own behaviour is undefined there: characters outside
'A'-'Z' such as lower-case bases or the '\r' of CR-LF files (they index past its 26-entry
nucleotide table, KmerCode.cpp / main.cpp:17-22) and records cut short by the end of the file (the
missing lines keep whatever the reused record buffer held -- a different record for every -t)."""
import os

import numpy as np

import synth


def make_case(seed, d, n=120, length=70, k=23):
    rng = np.random.Generator(np.random.PCG64(seed))
    s1, q1, _, _, _ = synth.make_reads(seed, n, length, n_tx=3, l_tx=300, e=0.01)
    keys, cnt = synth.count_kmers([s1], k)
    synth.write_dump(os.path.join(d, "d.jf"), keys, cnt, k)
    p_quirk = float(rng.choice([0.02, 0.1, 0.4]))
    nl = b"\n"
    out = []
    for i in range(n):
        r, q = s1[i].tobytes(), q1[i].tobytes()
        h = b"@r%d" % i
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
                q = q[:int(rng.integers(0, len(q)))]
            elif u == 4:
                q = q + b"I" * int(rng.integers(1, 9))
            elif u == 5:
                q = b""
            elif u == 6:
                plus = b"+r%d some text" % i
            elif u == 7:
                h = b"@r%d " % i + b"x" * int(rng.integers(100, 900)) + b"\tend of header"
            elif u == 8:
                r = b""          # empty sequence, quality kept
            elif u == 9:
                ln = int(rng.integers(1, k + 3))
                r, q = r[:ln], q[:ln]
        out.append(h + nl + r + nl + plus + nl + q + nl)
    data = b"".join(out)
    if rng.random() < 0.3 and len(q) > 0:                # (an empty last line would simply be missing)
        data = data[:-len(nl)]                       # no newline at the end of the file
    with open(os.path.join(d, "a.fq"), "wb") as f:
        f.write(data)
    return ["-r", "a.fq", "-k", str(k), "-c", "d.jf"]
