#!/usr/bin/env python3
"""Dev / measurement tool: the `rcorrector` binary, files to files, on the headline preset's reads (25 M x 150 bp pairs by
default) under several flag / environment variants, each timed as its parent sees it with the RC_TIMING stamps next to it.
usage: exp_e2e_variants.py [--reads N] [--repeat R] 'label|ENV=1 ENV2=x|-flag value' ...
(the files are written once into --dir; the outputs of a run are removed before the next one)"""
import argparse
import hashlib
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth_int  # noqa: E402


def write_fq(path, S, Q, L):
    m = S.shape[0]
    ids = np.char.zfill(np.arange(m).astype(str), 9)
    idb = np.frombuffer("".join(ids.tolist()).encode(), dtype=np.uint8).reshape(m, 9)
    rec = np.empty((m, 2 + 9 + 1 + L + 1 + 2 + L + 1), dtype=np.uint8)
    rec[:, 0] = ord('@'); rec[:, 1] = ord('r'); c = 2
    rec[:, c:c + 9] = idb; c += 9
    rec[:, c] = 10; c += 1
    rec[:, c:c + L] = S; c += L
    rec[:, c] = 10; c += 1
    rec[:, c] = ord('+'); rec[:, c + 1] = 10; c += 2
    rec[:, c:c + L] = Q; c += L
    rec[:, c] = 10
    rec.tofile(path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=25_000_000)
    ap.add_argument("--preset", type=int, default=2, help="bench.py preset whose reads are written (its k, maxcorK, error rate, pairing)")
    ap.add_argument("--dir", default="/tmp/rc_e2e_var")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("variants", nargs="*", default=["default||"])
    a = ap.parse_args()
    os.makedirs(a.dir, exist_ok=True)
    import bench
    P = bench.PRESETS[a.preset]
    L, n, paired = P["len"], a.reads, P["paired"]
    dev = torch.device("cuda", 0)
    gen = synth_int.Synth(P["seed"], L, 30000, 1500, P["alpha"], P["err"], paired, bias3=P["bias3"], device=dev)
    seq, qual = gen.generate(0, n // 2 if paired else n)
    S = seq.view(n, L + 1)[:, :L].cpu().numpy()
    Q = qual.view(n, L + 1)[:, :L].cpu().numpy()
    if paired:
        write_fq(os.path.join(a.dir, "x_1.fq"), S[:n // 2], Q[:n // 2], L)
        write_fq(os.path.join(a.dir, "x_2.fq"), S[n // 2:], Q[n // 2:], L)
        inputs, first_out = ["-p", "x_1.fq", "x_2.fq"], "x_1.cor.fq"
    else:
        write_fq(os.path.join(a.dir, "x.fq"), S, Q, L)
        inputs, first_out = ["-r", "x.fq"], "x.cor.fq"
    del seq, qual, S, Q, gen
    torch.cuda.empty_cache()
    cli = os.path.join(ROOT, "rcorrector_amd", "rcorrector")
    md5s = set()
    for v in a.variants:
        label, env_s, flags = (v.split("|") + ["", ""])[:3]
        env = dict(os.environ, RC_TIMING="1")
        for kv in env_s.split():
            kk, vv = kv.split("=", 1)
            env[kk] = vv
        for rep in range(a.repeat):
            out = os.path.join(a.dir, "out")
            shutil.rmtree(out, ignore_errors=True)
            os.sync()
            t0 = time.time()
            env["RC_T0"] = repr(t0)
            p = subprocess.run([env.get("RC_CLI_BIN", cli)] + inputs + ["-k", str(P["k"]), "-maxcorK", str(P["maxcork"]), "-od", out] + flags.split(), cwd=a.dir, env=env,
                               stderr=subprocess.PIPE, stdout=subprocess.DEVNULL)
            wall = time.time() - t0
            err = p.stderr.decode()
            stamps = [ln[len("[rc timing] +"):] for ln in err.splitlines() if ln.startswith("[rc timing] +")]
            other = [ln for ln in err.splitlines() if ln.startswith("[rc") and not ln.startswith("[rc timing] +")]
            h = hashlib.md5()
            try:
                with open(os.path.join(out, first_out), "rb") as f:
                    for blk in iter(lambda: f.read(1 << 24), b""):
                        h.update(blk)
                md5s.add(h.hexdigest())
            except OSError:
                pass
            print("== %s (run %d): rc %d, wall %.3f s = %.2f M reads/s; md5 %s" % (label, rep, p.returncode, wall, n / wall / 1e6, h.hexdigest()[:8]))
            print("   " + " | ".join(s.replace(" s ", " ", 1) for s in stamps))
            if rep == a.repeat - 1:
                for ln in other:
                    print("   " + ln)
            sys.stdout.flush()
    print("distinct outputs: %d" % len(md5s))
    shutil.rmtree(a.dir, ignore_errors=True)


if __name__ == "__main__":
    main()
