#!/usr/bin/env python3
"""bench.py -- corrected reads/s of the MI355X correction path (BASELINE.json metric).

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 under torch.distributed.run, one
rank per GPU).  A step = one pass of the whole hot path (probe K1 -> threshold K2 -> correct K3,
rc_correct_device) over one batch of synthetic reads already resident in HBM.

Default workload (`--config 2`) = the configuration BASELINE.json's metric ("corrected reads/sec
(150 bp, k=23) at 1/2/4/8 MI355X") is quoted on, configs[2]: 200 M synthetic 150 bp paired-end
reads on 8 GPUs, read-sharded with the table replicated -- i.e. 25 M reads (12.5 M pairs) per GPU,
weak scaling, so `--gpus 8` is configs[2] itself.  `--config 1` = configs[1]: 10 M synthetic
100 bp single-end reads, 1 GPU.  Both: k=23, synth-v1 transcriptome of 30 000 x 1 500 bp, 0.5 %
substitutions, k-mer table counted on the GPU from rank 0's shard (~50 M k-mers), ERROR_RATE
estimated from that table as the reference does from its dump (main.cpp:310-358).  No data-path
collective; `value` = reads all ranks corrected / max-over-ranks wall time.

The JSON line also carries
  roofline     : the hash-probe kernel (K1) timed with HIP events on the library's stream inside
                 the timed region; achieved = algorithmic bytes (64 B per valid
                 k-mer + 1 B per base + 4 B per count) / average launch time, vs 8 TB/s HBM3E.
  cpu_baseline : the CPU oracle (restatement of the reference, oracle/) on a bounded sample of
                 the same reads and the same table, all host cores -- reported, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (before the HIP library: one HIP runtime per process)
import torch.distributed as dist  # noqa: E402

import rcorrector_amd  # noqa: E402
from rcorrector_amd.distributed import reduce_summary  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E nominal (MI355X_MICROARCH.md)


def synth_reads_gpu(seed, n_reads, length, n_tx, l_tx, alpha, err, dev, chunk=1 << 20, paired=False, frag_len=300):
    """synth-v1 on the GPU: returns (seq arena uint8 [n*(L+1)], qual arena) with a NUL
    after every read.  The transcriptome depends only on `seed // 1000` so all ranks share it.
    paired: n_reads/2 fragments of frag_len bases; the arena holds all first mates, then all second
    mates (mate 2 = reverse complement of the fragment's tail), the layout rc_correct_device mode 1 wants."""
    g_tx = torch.Generator(device=dev)
    g_tx.manual_seed(seed // 1000)
    tx = torch.randint(0, 4, (n_tx * l_tx,), dtype=torch.uint8, device=dev, generator=g_tx)
    w = (torch.arange(n_tx, device=dev, dtype=torch.float64) + 1.0) ** (-alpha)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    seq = torch.zeros((n_reads, length + 1), dtype=torch.uint8, device=dev)
    qual = torch.zeros((n_reads, length + 1), dtype=torch.uint8, device=dev)
    span = frag_len if paired else length
    n_units = n_reads // 2 if paired else n_reads
    ar = torch.arange(span, device=dev)

    def emit(codes, lo, m):
        mut = torch.rand((m, length), device=dev, generator=g) < err
        shift = torch.randint(1, 4, (m, length), dtype=torch.uint8, device=dev, generator=g)
        codes = torch.where(mut, (codes + shift) & 3, codes)
        seq[lo:lo + m, :length] = lut[codes.long()]
        q = torch.full((m, length), ord('I'), dtype=torch.uint8, device=dev)
        q[mut] = ord('#')
        qual[lo:lo + m, :length] = q

    for lo in range(0, n_units, chunk):
        m = min(chunk, n_units - lo)
        tid = torch.multinomial(w.float(), m, replacement=True, generator=g)
        start = torch.randint(0, l_tx - span + 1, (m,), device=dev, generator=g)
        idx = (tid * l_tx + start)[:, None] + ar[None, :]
        codes = tx[idx]
        rev = torch.rand(m, device=dev, generator=g) < 0.5
        codes = torch.where(rev[:, None], 3 - codes.flip(1), codes)
        emit(codes[:, :length], lo, m)
        if paired:
            emit((3 - codes.flip(1))[:, :length], n_units + lo, m)
    return seq.reshape(-1), qual.reshape(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=[1, 2],
                    help="BASELINE.json configs[i]: 2 = 150 bp paired-end, 25 M reads per GPU (default); 1 = 10 M x 100 bp single-end")
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU per step (overrides the config's)")
    ap.add_argument("--len", type=int, default=None, help="read length (overrides the config's)")
    ap.add_argument("-k", type=int, default=23)
    ap.add_argument("--n-tx", type=int, default=30000)
    ap.add_argument("--l-tx", type=int, default=1500)
    ap.add_argument("--err", type=float, default=0.005)
    ap.add_argument("--alpha", type=float, default=0.8)
    ap.add_argument("--seed", type=int, default=1001000)
    ap.add_argument("--maxcork", type=int, default=4, help="-maxcorK (MAX_FIX_PER_K)")
    ap.add_argument("--paired", action="store_true", default=None, help="paired-end batch (mode 1): --reads counts both mates")
    ap.add_argument("--single", dest="paired", action="store_false", help="single-end batch (mode 0)")
    ap.add_argument("--fixed-error-rate", type=float, default=None, help="use this ERROR_RATE instead of estimating it from the table")
    ap.add_argument("--host-path", action="store_true",
                    help="also time the host-buffer entry point rc_correct_batch (PCIe inclusive; reported, never `value`)")
    ap.add_argument("--cpu-sample", type=int, default=3000000, help="reads of the CPU-baseline sample (0 = skip)")
    a = ap.parse_args()
    preset = {1: (10_000_000, 100, False), 2: (25_000_000, 150, True)}[a.config]
    if a.reads is None:
        a.reads = preset[0]
    if a.len is None:
        a.len = preset[1]
    if a.paired is None:
        a.paired = preset[2]
    is_preset = (a.reads, a.len, a.paired) == preset and a.k == 23 and a.maxcork == 4

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, a.gpus))
    # RC_BENCH_SHARED_GPU=1 (tests only): every rank uses GPU 0 and the collectives run over gloo, so
    # that the N>1 code path can be exercised on a one-GPU box
    shared_gpu = os.environ.get("RC_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    coll_dev = torch.device("cpu") if shared_gpu else dev
    if world > 1:
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()

    t_prog = time.time()

    def progress(what):
        if rank == 0 and os.environ.get("RC_BENCH_PROGRESS"):
            print("[bench %6.1f s] %s" % (time.time() - t_prog, what), file=sys.stderr, flush=True)

    L, k, n = a.len, a.k, a.reads
    mode = 1 if a.paired else 0
    if a.paired and n % 2:
        raise SystemExit('--paired needs an even --reads')
    ctx = rcorrector_amd.Context(k=k, max_fix_per_k=a.maxcork, device=local)

    # the replicated table: every rank counts the k-mers of the rank-0 shard (same seed => same
    # table everywhere, no communication), then generates its own shard of reads
    t0 = time.time()
    seq0, qual0 = synth_reads_gpu(a.seed, n, L, a.n_tx, a.l_tx, a.alpha, a.err, dev, paired=a.paired)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    t0 = time.time()
    n_kmers = ctx.count_reads_device(seq0, seq0.numel(), 2)
    t_count = time.time() - t0
    progress("synth %.2f s, count+table %.2f s, %d k-mers" % (t_gen, t_count, n_kmers))
    if rank != 0:
        del seq0, qual0
        seq0, qual0 = synth_reads_gpu(a.seed + rank, n, L, a.n_tx, a.l_tx, a.alpha, a.err, dev, paired=a.paired)
    nbytes = seq0.numel()
    off = (torch.arange(n + 1, device=dev, dtype=torch.int64) * (L + 1)).to(torch.int32)  # < 2^31 here
    first_q = qual0[0::(L + 1)][:1000000]
    last_q = qual0[L - 1::(L + 1)][:1000000]
    fh = torch.bincount(first_q.long(), minlength=300)[:300].cpu().numpy().astype(np.int32)
    lh = torch.bincount(last_q.long(), minlength=300)[:300].cpu().numpy().astype(np.int32)
    bad_q = ctx.bad_quality_from_hist(fh, lh, int(first_q.numel()))
    # ERROR_RATE as the reference derives it from its dump (main.cpp:310-358), here from the table
    # counted above (scanned in the library's dump order); every rank gets the same value
    error_rate = a.fixed_error_rate if a.fixed_error_rate is not None else ctx.estimate_error_rate(0.95)
    ctx.set_run_params(error_rate, bad_q)
    progress("ERROR_RATE %.6f, bad quality %r" % (error_rate, bad_q))

    work = seq0.clone()
    ret = torch.zeros(n, dtype=torch.int32, device=dev)
    l_ = torch.zeros_like(ret)
    m_ = torch.zeros_like(ret)
    h_ = torch.zeros_like(ret)

    def step():
        work.copy_(seq0)           # restore the uncorrected reads (the kernel corrects in place)
        torch.cuda.current_stream().synchronize()
        ctx.correct_device(mode, n, nbytes, L, work, qual0, off, ret, l_, m_, h_)
        ctx.sync()

    for _ in range(a.warmup):
        step()
        progress("warm-up step done")
    ctx.profile(True)
    ctx.profile_reset()
    barrier()
    torch.cuda.synchronize()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ctx.sync()
    barrier()
    dt = time.perf_counter() - t0
    progress("timed steps done: %.1f ms per step" % (dt / a.steps * 1e3))
    ctx.profile(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    n_cor_reads = int((ret > 0).sum().item())
    # the one RCCL collective of the whole path: global count statistics (main.cpp:32-36)
    g_reads, g_bases = reduce_summary(n, int(ret.clamp(min=0).sum().item()), device=coll_dev)
    stats = [g_reads, n_cor_reads * world, g_bases]

    if rank == 0:
        total_reads = n * world * a.steps
        value = total_reads / dt
        ms_probe, launches = ctx.profile_get(0)
        ms_thr, _ = ctx.profile_get(1)
        ms_cor, _ = ctx.profile_get(2)
        kcnt = L - k + 1
        alg_bytes = float(n) * (kcnt * 64 + L + 4 * kcnt)   # no N in this workload: every k-mer valid
        avg_probe_s = ms_probe / max(launches, 1) / 1e3
        achieved = alg_bytes / avg_probe_s / 1e9 if avg_probe_s > 0 else 0.0
        ts = ctx.table_stats()

        # the host-buffer entry point (rc_correct_batch: H2D + kernels + D2H from pageable memory),
        # reported for context only -- never `value`
        host_rate = None
        if a.host_path:
            hn = min(n, 1_000_000)
            hseq = seq0[:hn * (L + 1)].cpu().numpy().copy()
            hqual = qual0[:hn * (L + 1)].cpu().numpy().copy()
            hoff = (np.arange(hn + 1, dtype=np.int64) * (L + 1)).astype(np.uint32)
            ctx.correct_batch(0, hseq.copy(), hqual, hoff)
            th = time.perf_counter()
            ctx.correct_batch(0, hseq, hqual, hoff)
            host_rate = hn / (time.perf_counter() - th)

        cpu = None
        if a.cpu_sample > 0 and world == 1:   # rank 0 at N=1 only
            cpu = cpu_baseline(ctx, seq0, qual0, n, L, k, error_rate, bad_q, a.cpu_sample, ret, work, a.maxcork, a.paired)

        out = {
            "metric": "corrected reads/sec (%d bp, k=%d)" % (L, k),
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int32/u64", "data": "synthetic",
            "config": {"workload": "%d synthetic %d bp %s reads per GPU, k=%d, %d-k-mer table%s"
                       % (n, L, "paired-end" if a.paired else "single-end", k, n_kmers,
                          (" (BASELINE.json configs[1])" if a.config == 1 else
                           " (BASELINE.json configs[2]: 200 M reads on 8 GPUs = 25 M per GPU, read-sharded)") if is_preset else ""),
                       "reads_per_gpu": n, "read_len": L, "k": k, "table_kmers": n_kmers,
                       "table_bytes": ts["bytes"], "sub_rate": a.err, "error_rate_param": error_rate,
                       "bad_quality": bad_q.decode("latin1"), "parallelism": "reads sharded x%d, table replicated" % world,
                       "reads_corrected_frac": stats[1] / float(stats[0]), "bases_corrected": stats[2],
                       "host_buffer_entry_reads_per_s_pcie_inclusive": host_rate,
                       "setup_s": {"synth": round(t_gen, 2), "count_and_build_table": round(t_count, 2)},
                       "mode": "paired" if a.paired else "single", "kernel_ms_per_step": {"probe": ms_probe / a.steps, "threshold": ms_thr / a.steps,
                                              "correct": ms_cor / a.steps}},
            "roofline": {"kernel": "k_probe (hash probe, K1)", "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_probe_s * 1e3,
                         "launches": launches, "traffic": None},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(ctx, seq0, qual0, n, L, k, error_rate, bad_q, sample, ret_gpu, work_gpu, maxcork=4, paired=False):
    """The CPU oracle on the first `sample` reads (paired: the first sample/2 pairs) with the same
    table, all host cores; also checks the GPU results of those reads against it (the oracle is
    the checker here, never the product)."""
    from oracle import pyoracle as po
    po.build()
    sample = min(sample, n)
    codes, counts = ctx.table_export()
    T = po.Table(k, len(codes))
    T.put_many(codes, counts)
    P = po.make_params(k, maxcork, error_rate, bad_q)
    cores = os.cpu_count() or 1
    if not paired:
        nb = sample * (L + 1)
        arena = seq0[:nb].cpu().numpy().copy()
        qa = qual0[:nb].cpu().numpy().copy()
        off = (np.arange(sample + 1, dtype=np.int64) * (L + 1)).astype(np.uint32)
        t0 = time.perf_counter()
        r, _, _, _ = po.correct_batch(P, T, 0, arena, qa, off, threads=cores)
        dt = time.perf_counter() - t0
        same = bool(np.array_equal(r, ret_gpu[:sample].cpu().numpy()) and
                    np.array_equal(arena, work_gpu[:nb].cpu().numpy()))
        what = "first %d reads" % sample
    else:
        # device layout (mode 1): all first mates, then all second mates
        half, sp = n // 2, sample // 2
        sample = 2 * sp
        nb = sp * (L + 1)
        b2 = half * (L + 1)
        a1 = seq0[:nb].cpu().numpy().copy()
        q1 = qual0[:nb].cpu().numpy().copy()
        a2 = seq0[b2:b2 + nb].cpu().numpy().copy()
        q2 = qual0[b2:b2 + nb].cpu().numpy().copy()
        off = (np.arange(sp + 1, dtype=np.int64) * (L + 1)).astype(np.uint32)
        t0 = time.perf_counter()
        r, _, _, _ = po.correct_batch(P, T, 1, a1, q1, off, a2, q2, off, threads=cores)
        dt = time.perf_counter() - t0
        rg = ret_gpu.cpu().numpy()
        same = bool(np.array_equal(r[:sp], rg[:sp]) and np.array_equal(r[sp:], rg[half:half + sp]) and
                    np.array_equal(a1, work_gpu[:nb].cpu().numpy()) and
                    np.array_equal(a2, work_gpu[b2:b2 + nb].cpu().numpy()))
        what = "first %d pairs (%d reads)" % (sp, sample)
    return {"value": sample / dt, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "%s of the same batch, same table, %d pthreads, %.1f s" % (what, cores, dt),
            "gpu_matches_oracle_on_sample": same}


if __name__ == "__main__":
    main()
