#!/usr/bin/env python3
"""How many of the probes of one tile of the fused probe kernel repeat a k-mer another probe of the tile asks about?  CPU model on the
bench's own reads at reduced scale (transcripts and reads scaled together: same coverage): units ordered as k_unit_key orders them
(smallest hash over the first mate's canonical 16-mers), tiles of 64 reads (32 pairs, mates adjacent), per tile the distinct
canonical k-mers and how many of them the table holds (count >= 2).  Experiment infrastructure."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth_int  # noqa: E402
from local_model import read_kmers, revcomp, mix32, U  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=0.02)
ap.add_argument("-k", type=int, default=23)
ap.add_argument("--err", type=float, default=0.005)
ap.add_argument("--tile", type=int, default=64)
a = ap.parse_args()
k, L = a.k, 150
n_tx = max(50, int(30000 * a.scale))
n_reads = int(25_000_000 * a.scale) & ~1
g = synth_int.Synth(1002, L, n_tx, 1500, 0.8, a.err, True)
seq, _ = g.generate(0, n_reads // 2)
S = seq.numpy().reshape(n_reads, L + 1)[:, :L]
half = n_reads // 2
print("reads %d x %d (pairs: mate 2 of unit u is read u + %d), %d transcripts, k = %d" % (n_reads, L, half, n_tx, k), flush=True)
f = read_kmers(S, k)
canon = np.minimum(f, revcomp(f.reshape(-1), k).reshape(f.shape))
keys, counts = np.unique(canon.reshape(-1), return_counts=True)
table = keys[counts >= 2]
print("table: %d k-mers; %.1f %% of all probes are of k-mers in it" % (table.size, 100.0 * np.isin(canon.reshape(-1), table).mean()), flush=True)
m16 = read_kmers(S[:half], 16)
c16 = np.minimum(m16, revcomp(m16.reshape(-1), 16).reshape(m16.shape))
ukey = mix32(c16).min(axis=1)
order = np.argsort(ukey, kind="stable")
upt = a.tile // 2
dist, dist_in, probes = [], [], []
for t in range(0, min(half, 4000 * upt), upt):
    u = order[t:t + upt]
    ck = np.concatenate([canon[u].reshape(-1), canon[u + half].reshape(-1)])
    d = np.unique(ck)
    dist.append(d.size); probes.append(ck.size); dist_in.append(np.isin(d, table).sum())
dist, dist_in, probes = map(np.array, (dist, dist_in, probes))
print("tiles of %d reads: %d probes each; distinct k-mers per tile: mean %.0f (%.1f %% of the probes), 10 / 50 / 90 %%: %d / %d / %d; of them in the table: mean %.0f"
      % (a.tile, probes[0], dist.mean(), 100.0 * dist.sum() / probes.sum(), *np.percentile(dist, [10, 50, 90]).astype(int), dist_in.mean()))
print("probes that repeat a k-mer of their tile: %.1f %%; probes of k-mers outside the table: %.1f %% of all (each of them distinct within its tile, as a rule)"
      % (100.0 * (1 - dist.sum() / probes.sum()), 100.0 * (dist - dist_in).sum() / probes.sum()))
