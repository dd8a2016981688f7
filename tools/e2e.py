#!/usr/bin/env python3
"""End-to-end timing of the `rcorrector` CLI (FASTQ in -> *.cor.fq out) on synthetic data.
Generates reads on the GPU (bench.synth_reads_gpu), writes FASTQ + a jf_dump-format k-mer dump
(counted on the GPU, exported through the C ABI), then times the CLI.  Dev/measurement tool."""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import rcorrector_amd  # noqa: E402
import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("-k", type=int, default=23)
    ap.add_argument("--n-tx", type=int, default=2000)
    ap.add_argument("--err", type=float, default=0.005)
    ap.add_argument("--dir", default="/tmp/rc_e2e")
    ap.add_argument("--cli-args", default="")
    ap.add_argument("--paired", action="store_true", help="paired-end: --reads counts both mates, files x_1.fq / x_2.fq")
    ap.add_argument("--count", action="store_true", help="also run every variant without -c (k-mers counted by the CLI)")
    ap.add_argument("--json", action="store_true", help="print one JSON line for the first variant (bench.py --e2e)")
    a = ap.parse_args()
    os.makedirs(a.dir, exist_ok=True)
    dev = torch.device("cuda", 0)
    n, L, k = a.reads, a.len, a.k
    seq, qual = bench.synth_reads_gpu(77000, n, L, a.n_tx, 1500, 0.8, a.err, dev, paired=a.paired)
    ctx = rcorrector_amd.Context(k=k)
    ctx.count_begin()   # (an arena handed to the counter stays below 4 GiB)
    step = 20_000_000 * (L + 1)
    for lo in range(0, seq.numel(), step):
        piece = seq[lo:lo + step]
        ctx.count_add_device(piece, piece.numel())
    nk = ctx.count_finish(2)
    codes, counts = ctx.table_export()
    o = np.argsort(codes)
    codes, counts = codes[o], counts[o]
    t0 = time.time()
    txt = synth.decode_kmers(codes, k)
    with open(os.path.join(a.dir, "x.jf"), "wb") as f:
        parts = []
        for i in range(len(codes)):
            parts.append(b">%d\n%s\n" % (counts[i], txt[i].tobytes()))
            if len(parts) >= 1 << 16:
                f.write(b"".join(parts))
                parts = []
        f.write(b"".join(parts))
    # FASTQ with fixed-width ids: one numpy 2-D array per file (paired: first mates, second mates)
    S = seq.view(n, L + 1)[:, :L].cpu().numpy()
    Q = qual.view(n, L + 1)[:, :L].cpu().numpy()

    def write_fq(path, s, q):
        m = len(s)
        ids = np.char.zfill(np.arange(m).astype(str), 9)
        idb = np.frombuffer("".join(ids.tolist()).encode(), dtype=np.uint8).reshape(m, 9)
        rec = np.empty((m, 2 + 9 + 1 + L + 1 + 2 + L + 1), dtype=np.uint8)
        rec[:, 0] = ord('@'); rec[:, 1] = ord('r'); c = 2
        rec[:, c:c + 9] = idb; c += 9
        rec[:, c] = 10; c += 1
        rec[:, c:c + L] = s; c += L
        rec[:, c] = 10; c += 1
        rec[:, c] = ord('+'); rec[:, c + 1] = 10; c += 2
        rec[:, c:c + L] = q; c += L
        rec[:, c] = 10
        rec.tofile(path)

    if a.paired:
        write_fq(os.path.join(a.dir, "x_1.fq"), S[:n // 2], Q[:n // 2])
        write_fq(os.path.join(a.dir, "x_2.fq"), S[n // 2:], Q[n // 2:])
        inputs, first_out = ["-p", "x_1.fq", "x_2.fq"], "x_1.cor.fq"
    else:
        write_fq(os.path.join(a.dir, "x.fq"), S, Q)
        inputs, first_out = ["-r", "x.fq"], "x.cor.fq"
    print("generated %d reads, %d k-mers in %.1f s (fq %.0f MB, dump %.0f MB)" % (
        n, nk, time.time() - t0, sum(os.path.getsize(os.path.join(a.dir, f)) for f in inputs[1:]) / 1e6,
        os.path.getsize(os.path.join(a.dir, "x.jf")) / 1e6), file=sys.stderr)
    del ctx
    cli = os.path.join(ROOT, "rcorrector_amd", "rcorrector")
    env = dict(os.environ, RC_TIMING="1")
    import hashlib
    import json
    first = None
    for variant in a.cli_args.split(";"):
        for dump in ([["-c", "x.jf"], []] if a.count else [["-c", "x.jf"]]):
            import shutil
            shutil.rmtree(a.dir + "/out", ignore_errors=True)   # truncating last run's multi-GB output is not part of the run
            os.sync()
            toks = variant.split()
            venv = dict(env)
            while toks and "=" in toks[0] and not toks[0].startswith("-"):   # leading NAME=VALUE tokens: environment of this variant
                kk, vv = toks.pop(0).split("=", 1)
                venv[kk] = vv
            t0 = time.time()
            venv["RC_T0"] = repr(t0)
            p = subprocess.run([cli] + inputs + ["-k", str(k), "-od", a.dir + "/out"] + dump + toks,
                               cwd=a.dir, env=venv, stderr=subprocess.PIPE)
            dt = time.time() - t0
            sys.stderr.write(p.stderr.decode())
            sys.stderr.write("[e2e] process returned at +%.3f s\n" % dt)
            md5 = hashlib.md5(open(a.dir + "/out/" + first_out, "rb").read()).hexdigest()
            print("CLI [%s %s] wall %.2f s -> %.2f M reads/s end to end, output md5 %s" % (
                " ".join(dump) or "(counting)", variant, dt, n / dt / 1e6, md5), file=sys.stderr if a.json else sys.stdout)
            if first is None:
                timing = [ln for ln in p.stderr.decode().splitlines() if ln.startswith("[rc timing]")]
                first = {"metric": "end-to-end corrected reads/sec (FASTQ files -> .cor.fq, %d bp %s, k=%d, rcorrector binary, 1 GPU)"
                         % (L, "paired-end" if a.paired else "single-end", k),
                         "value": n / dt, "unit": "reads/s", "reads": n, "wall_s": dt, "table_kmers": nk,
                         "includes": "process start, HIP init, dump parse + table build, ERROR_RATE, bad-quality scan, read -> correct -> write",
                         "rc_timing": timing, "output_md5": md5}
    if a.json:
        print(json.dumps(first))


if __name__ == "__main__":
    main()
