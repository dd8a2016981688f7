#!/usr/bin/env python3
"""Golden vectors for the accuracy scorer: tests/golden/verify/{raw,cor,indel}.fq (simulated reads
with Mason-style truth headers: uncorrected, corrected by the UNMODIFIED reference binary, and
trimmed / indel variants) and, for every option set, what the UNMODIFIED reference scorer
(oracle/_ref/verify_ref, built from /root/reference/verify.cpp by oracle/Makefile) prints.
Build container only."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
import datasets  # noqa: E402
from test_verify_cpu import OPTION_SETS  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "rcorrector_ref")
REF_VERIFY = os.path.join(ROOT, "oracle", "_ref", "verify_ref")


def main():
    d = os.path.join(HERE, "verify")
    os.makedirs(d, exist_ok=True)
    heads, reads, quals = datasets.mason_style_reads(seed=11)
    datasets.write_mason_fastq(os.path.join(d, "raw.fq"), heads, reads, quals)
    datasets.write_mason_fastq(os.path.join(d, "indel.fq"), *datasets.mason_indel_variants(heads, reads, quals))
    arr = np.frombuffer(b"".join(reads), dtype=np.uint8).reshape(len(reads), -1)
    keys, cnt = synth.count_kmers([arr], 23)
    synth.write_dump(os.path.join(d, "dump.jf"), keys, cnt, 23)
    subprocess.run([REF, "-r", "raw.fq", "-k", "23", "-c", "dump.jf", "-od", d], cwd=d, check=True, stderr=subprocess.DEVNULL)
    os.replace(os.path.join(d, "raw.cor.fq"), os.path.join(d, "cor.fq"))
    os.remove(os.path.join(d, "dump.jf"))
    for f in ("raw", "cor", "indel"):
        for name, opts in OPTION_SETS.items():
            out = subprocess.run([REF_VERIFY, os.path.join(d, f + ".fq")] + opts, stdout=subprocess.PIPE, check=True).stdout
            open(os.path.join(d, "%s.%s.txt" % (f, name)), "wb").write(out)
    print("verify goldens: %d files" % len(os.listdir(d)))


if __name__ == "__main__":
    main()
