#!/usr/bin/env python3
"""One-off extended fuzz: tests/test_gpu_fuzz.py's random cases for an arbitrary seed range
(GPU CLI vs the pinned CPU oracle CLI, every output byte, stderr and -verbose transcript).
  fuzz_more.py LO HI        read-content fuzz
  fuzz_more.py LO HI io     input-format quirks (tests/io_quirks.py)
  fuzz_more.py LO HI long   read-content fuzz with read lengths up to 1023
  fuzz_more.py LO HI nodump [io]   without -c: the GPU CLI counts the k-mers itself in ONE pass over the files (text kept in host
                            memory, bases kept in HBM, rc_submit_resident; every other seed in batches of 10 reads) and writes its
                            table with -write-dump; the oracle CLI given that dump must produce the same bytes and stderr"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_gpu_fuzz as F  # noqa: E402
from oracle import pyoracle  # noqa: E402

def _content(path):
    """file bytes; .gz outputs are compared after decompression (the drop-in binary writes parallel
    gzip members, the reference one stream -- Reads.h:140-147 -- with identical content)"""
    data = open(path, "rb").read()
    if path.endswith(".gz"):
        import gzip
        return gzip.decompress(data)
    return data


pyoracle.build()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
nodump = len(sys.argv) > 3 and sys.argv[3] == "nodump"
if nodump:
    del sys.argv[3]
io_mode = len(sys.argv) > 3 and sys.argv[3] == "io"
long_mode = len(sys.argv) > 3 and sys.argv[3] == "long"   # reads of up to 1023 bases   # tests/io_quirks.py cases instead of the read-content fuzz
if io_mode:
    import io_quirks  # noqa: E402
bad = 0
for seed in range(lo, hi):
    with tempfile.TemporaryDirectory() as d:
        args = io_quirks.make_case(seed, d, modes=(0, 1, 2)) if io_mode else F._random_case(seed, d, max_len=1024 if long_mode else 160)
        outs = {}
        verbose = ["-verbose"] if (io_mode or seed % 3 == 0) and not nodump else []
        gpu_flags = os.environ.get("RC_FUZZ_GPU_FLAGS", "").split()   # e.g. "-gpus 3 -inflight 2" (with RC_SHARED_GPU=1 on a one-GPU box)
        runs = (("gpu", F.CLI, (["-batch", "64"] if seed % 2 else []) + gpu_flags, args, None), ("cpu", pyoracle.CLI_BIN, ["-t", "2"], args, None))
        if nodump:
            a = list(args)
            i = a.index("-c")
            del a[i:i + 2]
            runs = (("gpu", F.CLI, ["-write-dump", os.path.join(d, "own.jf")] + gpu_flags, a, {"RC_RESIDENT": "10" if seed % 2 else "1"}),
                    ("cpu", pyoracle.CLI_BIN, ["-t", "2", "-c", os.path.join(d, "own.jf")], a, None))
        for name, binary, more, argv, env in runs:
            od = os.path.join(d, name)
            os.makedirs(od)
            p = subprocess.run([binary] + argv + ["-od", od] + more + verbose, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               env=dict(os.environ, **env) if env else None)
            outs[name] = (p.returncode, p.stderr, {f: _content(os.path.join(od, f)) for f in sorted(os.listdir(od))}, p.stdout)
        if outs["gpu"] != outs["cpu"]:
            bad += 1
            print("MISMATCH seed", seed, args, outs["gpu"][0], outs["cpu"][0], flush=True)
            for j, what in ((1, "stderr"), (3, "stdout")):
                if outs["gpu"][j] != outs["cpu"][j]:
                    a, b = outs["gpu"][j].split(b"\n"), outs["cpu"][j].split(b"\n")
                    i = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
                    print("   %s line %d: gpu %r | cpu %r | before %r" % (what, i, a[i:i + 1], b[i:i + 1], b[max(0, i - 2):i]), flush=True)
            for f in outs["cpu"][2]:
                if outs["gpu"][2].get(f) != outs["cpu"][2][f]:
                    a, b = outs["gpu"][2].get(f, b"").split(b"\n"), outs["cpu"][2][f].split(b"\n")
                    i = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
                    print("   %s line %d: gpu %r | cpu %r | before %r" % (f, i, a[i:i + 1], b[i:i + 1], b[max(0, i - 3):i]), flush=True)
print("seeds %d..%d: %d mismatches" % (lo, hi, bad))
