"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Integer / byte work => bit-exact everywhere."""
import os

import numpy as np
import pytest

import datasets
import rcorrector_amd
import synth

pytestmark = pytest.mark.gpu


def _table(ctx_factory, d):
    ctx = ctx_factory(d["k"], d["mfk"])
    ctx.table_build(d["keys"], d["counts"])
    ctx.set_run_params(d["rate"], b"H")
    return ctx


def _revcomp_codes(codes, k):
    out = np.zeros_like(codes)
    c = codes.copy()
    for _ in range(k):
        out = (out << np.uint64(2)) | (np.uint64(3) - (c & np.uint64(3)))
        c = c >> np.uint64(2)
    return out


@pytest.mark.parametrize("k", [15, 23, 31, 32])
def test_table_lookup_matches_store(gpu_ctx_factory, k):
    rng = np.random.Generator(np.random.PCG64(k))
    n = 200000
    mask = np.uint64((1 << (2 * k)) - 1) if k < 32 else np.uint64(0xFFFFFFFFFFFFFFFF)
    fwd = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    fwd &= mask
    can = np.minimum(fwd, _revcomp_codes(fwd, k))
    can = np.unique(can)
    counts = rng.integers(2, 1 << 30, size=len(can)).astype(np.int32)
    ctx = gpu_ctx_factory(k)
    ctx.table_build(can, counts)
    assert np.array_equal(ctx.lookup(can), counts)                       # canonical form
    assert np.array_equal(ctx.lookup(_revcomp_codes(can, k)), counts)    # the other strand
    absent = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) & mask
    absent_can = np.minimum(absent, _revcomp_codes(absent, k))
    present = np.isin(absent_can, can)
    got = ctx.lookup(absent)
    assert (got[~present] == 0).all()
    st = ctx.table_stats()
    assert st["entries"] == len(can) and st["bytes"] in (st["buckets"] * _bucket_bytes(), st["buckets"] * 2 * _bucket_bytes())


def _bucket_bytes():
    """bytes of a table bucket as the library is built (rc_common.h: RC_BUCKET_DWORDS; PACKED: 8 bytes a slot)"""
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rcorrector_amd", "csrc", "rc_common.h")).read()
    return 4 * int(re.search(r"#ifndef RC_BUCKET_DWORDS\n#define RC_BUCKET_DWORDS (\d+)", src).group(1))


def test_table_later_duplicate_wins(gpu_ctx_factory):
    # Store.h:55: hash[key] = cnt overwrites
    k = 23
    codes = np.array([5, 77, 5, 1234567, 77, 5], dtype=np.uint64)
    counts = np.array([10, 20, 30, 40, 50, 60], dtype=np.int32)
    ctx = gpu_ctx_factory(k)
    ctx.table_build(codes, counts)
    assert ctx.lookup(np.array([5, 77, 1234567, 9], dtype=np.uint64)).tolist() == [60, 50, 40, 0]


def test_table_empty_and_tiny(gpu_ctx_factory):
    ctx = gpu_ctx_factory(23)
    ctx.table_build(np.zeros(0, np.uint64), np.zeros(0, np.int32))
    assert ctx.lookup(np.array([0, 1, 2], dtype=np.uint64)).tolist() == [0, 0, 0]
    ctx.table_build(np.array([0], np.uint64), np.array([7], np.int32))  # poly-A k-mer has code 0
    assert ctx.lookup(np.array([0, 1], dtype=np.uint64)).tolist() == [7, 0]


@pytest.mark.parametrize("name", ["se_k23", "k31_mc8", "nrich", "varlen", "k15", "k32", "edge", "max1023", "k11", "polya_k23"])
def test_probe_kernel_counts(gpu_ctx_factory, oracle, name):
    import torch
    d = datasets.make(name)
    ctx = _table(gpu_ctx_factory, d)
    arena, off = oracle.pack_reads(d["seqs1"])
    d_seq = torch.from_numpy(arena).cuda()
    d_cnt = torch.full((len(arena),), -7, dtype=torch.int32, device="cuda")
    ctx.probe_device(d_seq, len(arena), d_cnt)
    ctx.sync()
    got = d_cnt.cpu().numpy()
    T = oracle.Table(d["k"], len(d["keys"]))
    T.put_many(d["keys"], d["counts"])
    P = oracle.make_params(d["k"], d["mfk"], d["rate"], b"H")
    for i, s in enumerate(d["seqs1"]):
        want = oracle.kmer_counts(P, T, s)
        o = int(off[i])
        assert np.array_equal(got[o:o + len(want)], want), "read %d of %s" % (i, name)
        # positions that start no k-mer of this read are left untouched
        assert (got[o + len(want):int(off[i + 1])] == -7).all()


@pytest.mark.parametrize("name", ["se_k23", "pe_k23", "il_k23", "k31_mc8", "skew", "nrich", "varlen", "k15", "k32", "pe_var", "edge",
                                  "long300", "long600_k31", "max1023", "k11", "polya_k23", "polya_k31", "polya_k15",
                                  "se_151", "pe_151", "pe_160_k15", "tiers_se", "tiers_pe", "tiers_il"])
def test_correct_batch_matches_oracle(gpu_ctx_factory, oracle, name):
    d = datasets.make(name)
    want = datasets.run_oracle(oracle, d)
    ctx = _table(gpu_ctx_factory, d)
    a, off = oracle.pack_reads(d["seqs1"])
    qa, _ = oracle.pack_reads(d["quals1"])
    if d["mode"] == 1:
        a2, off2 = oracle.pack_reads(d["seqs2"])
        qa2, _ = oracle.pack_reads(d["quals2"])
        got = ctx.correct_batch(1, a, qa, off, a2, qa2, off2) + (a, a2)
    else:
        got = ctx.correct_batch(d["mode"], a, qa, off) + (a,)
    for w, g, what in zip(want, got, ["ret", "l", "m", "h", "seq1", "seq2"]):
        bad = np.nonzero(w != g)[0]
        assert len(bad) == 0, "%s differs on %s at %s (want %s got %s)" % (what, name, bad[:5], w[bad[:5]], g[bad[:5]])
    reads, cors = ctx.summary()
    assert reads == len(want[0]) and cors == int(want[0][want[0] > 0].sum())


def test_paired_equals_interleaved_and_batch_split_invariance(gpu_ctx_factory, oracle):
    """Size-independent properties: -p and -i give the same per-read results, and the
    result of a read does not depend on which batch it travels in."""
    d = datasets.make("pe_k23")
    ctx = _table(gpu_ctx_factory, d)
    n = len(d["seqs1"])
    a1, o1 = oracle.pack_reads(d["seqs1"]); q1, _ = oracle.pack_reads(d["quals1"])
    a2, o2 = oracle.pack_reads(d["seqs2"]); q2, _ = oracle.pack_reads(d["quals2"])
    rp = ctx.correct_batch(1, a1, q1, o1, a2, q2, o2)
    il_s = [x for p in zip(d["seqs1"], d["seqs2"]) for x in p]
    il_q = [x for p in zip(d["quals1"], d["quals2"]) for x in p]
    ai, oi = oracle.pack_reads(il_s); qi, _ = oracle.pack_reads(il_q)
    ri = ctx.correct_batch(2, ai, qi, oi)
    for x, y in zip(rp, ri):
        assert np.array_equal(x[:n], y[0::2]) and np.array_equal(x[n:], y[1::2])
    got_il = oracle.unpack_reads(ai, oi)
    assert got_il[0::2] == oracle.unpack_reads(a1, o1) and got_il[1::2] == oracle.unpack_reads(a2, o2)
    # split the interleaved batch in two uneven halves
    cut = 2 * (n // 3)
    parts = []
    for lo, hi in ((0, cut), (cut, 2 * n)):
        a, o = oracle.pack_reads(il_s[lo:hi]); q, _ = oracle.pack_reads(il_q[lo:hi])
        r = ctx.correct_batch(2, a, q, o)
        parts.append((r, oracle.unpack_reads(a, o)))
    for j in range(4):
        assert np.array_equal(np.concatenate([parts[0][0][j], parts[1][0][j]]), ri[j])
    assert parts[0][1] + parts[1][1] == got_il


def test_errors_are_loud(gpu_ctx_factory):
    import rcorrector_amd
    ctx = gpu_ctx_factory(23)
    a, off = rcorrector_amd.pack_reads([b"ACGT" * 10])
    with pytest.raises(rcorrector_amd.RcorrectorError):
        ctx.correct_batch(0, a, a.copy(), off)  # no table, no run parameters
    with pytest.raises(rcorrector_amd.RcorrectorError):
        rcorrector_amd.Context(k=33)


@pytest.mark.parametrize("env", [{}, {"RC_LOCALITY": "force"}, {"RC_LOCALITY": "force", "RC_NO_FUSE": "1"}, {"RC_LOCALITY": "off"}])
def test_no_table_is_an_error_on_every_probe_path(env, monkeypatch):
    """A context without a table refuses to correct before anything is launched, whichever probe kernel
    the batch would have taken (the list-driven ones dereference the table without a check of their own)."""
    import rcorrector_amd
    for kk, v in env.items():
        monkeypatch.setenv(kk, v)
    ctx = rcorrector_amd.Context(k=23, device=0)
    ctx.set_run_params(0.01, b"H")
    a, off = rcorrector_amd.pack_reads([b"ACGT" * 10, b"TTGCA" * 9])
    with pytest.raises(rcorrector_amd.RcorrectorError, match="no k-mer table"):
        ctx.correct_batch(0, a, a.copy(), off)
    ctx.close()


def test_count_reads_device_matches_exact_counts(gpu_ctx_factory, oracle):
    """stages 0-2 replacement: k-mer counting on the GPU == exact canonical counts >= 2 (what
    `jellyfish count -C` + `dump -L 2` yields), incl. reads with N and ragged lengths."""
    import torch
    import synth
    for name in ("varlen", "nrich", "k31_mc8"):
        d = datasets.make(name)
        arena, off = oracle.pack_reads(d["seqs1"])
        ctx = gpu_ctx_factory(d["k"])
        n = ctx.count_reads_device(torch.from_numpy(arena).cuda(), len(arena), 2)
        codes, counts = ctx.table_export()
        o = np.argsort(codes)
        assert n == len(d["keys"]) == len(codes)
        assert np.array_equal(codes[o], d["keys"]) and np.array_equal(counts[o].astype(np.int64), d["counts"])
        # and the table answers like the one built from the dump
        assert np.array_equal(ctx.lookup(d["keys"]), d["counts"].astype(np.int32))


@pytest.mark.parametrize("fx,k,wk", [("fx_highcov", 23, 0.9), ("fx_se_k23", 23, 0.95), ("fx_k32", 32, 0.95)])
def test_jfdump_load_and_error_rate_match_oracle(gpu_ctx_factory, oracle, fx, k, wk):
    """main.cpp:294-358 through the C ABI: 'Stored N kmers' and ERROR_RATE, incl. the data-driven
    (non-0.01) branch and a shuffled dump order."""
    import os
    import golden_util as gu
    dump = os.path.join(gu.GOLDEN, fx, "dump.jf")
    T = oracle.Table(k, 1 << 16)
    stored = T.load_dump(dump)
    want = T.error_rate(dump, wk)
    ctx = gpu_ctx_factory(k)
    assert ctx.load_jfdump(dump) == stored
    got = ctx.estimate_error_rate(wk)
    assert got == want  # the same IEEE doubles, bit for bit
    if fx == "fx_highcov":
        assert got != 0.01
    codes, counts = T.export()
    assert np.array_equal(ctx.lookup(codes), counts)


def test_full_scale_properties(gpu_ctx_factory):
    """At bench scale (1 M reads here, the bench itself checks 3 M against the oracle): results do
    not depend on how the batch is cut, and correcting an already corrected batch with the same
    table only ever touches reads it touched before (ret == 0 reads are fixed points)."""
    import torch
    import bench as B
    dev = torch.device("cuda", 0)
    n, L, k = 1_000_000, 100, 23
    seq, qual = B.synth_reads_gpu(4242000, n, L, 3000, 1500, 0.8, 0.005, dev)
    ctx = gpu_ctx_factory(k)
    ctx.count_reads_device(seq, seq.numel(), 2)
    ctx.set_run_params(0.01, b"H")
    off = (torch.arange(n + 1, device=dev, dtype=torch.int64) * (L + 1)).to(torch.int32)

    def run(s, lo, hi):
        m = hi - lo
        r = [torch.zeros(m, dtype=torch.int32, device=dev) for _ in range(4)]
        o = (off[lo:hi + 1] - off[lo]).contiguous()
        ctx.correct_device(0, m, m * (L + 1), L, s[lo * (L + 1):hi * (L + 1)], qual[lo * (L + 1):hi * (L + 1)], o, *r)
        ctx.sync()
        return r
    a = seq.clone()
    whole = run(a, 0, n)
    b = seq.clone()
    cut = 333_334
    p1, p2 = run(b, 0, cut), run(b, cut, n)
    assert torch.equal(a, b)
    for x, y, z in zip(whole, p1, p2):
        assert torch.equal(x, torch.cat([y, z]))
    assert int((whole[0] > 0).sum()) > 100_000
    # fixed points: reads reported clean (ret == 0) come back unchanged and clean again
    c = a.clone()
    again = run(c, 0, n)
    clean = whole[0] == 0
    assert torch.equal(again[0][clean], torch.zeros_like(again[0][clean]))
    rows_a = a.view(n, L + 1)[clean]
    rows_c = c.view(n, L + 1)[clean]
    assert torch.equal(rows_a, rows_c)


@pytest.mark.parametrize("rate", [0.01, 0.004046, 0.001362, 0.007897, 0.010297, 0.25, 1.0, 1e-9])
def test_get_bound_device_equals_x86(gpu_ctx_factory, oracle, rate):
    """GetBound (ErrorCorrection.cpp:139-142) is the only floating-point arithmetic on the path.  The
    device evaluates it in IEEE double with explicitly rounded mul / sqrt / add; it must give the
    bits the reference's x86-64 SSE2 code gives: the doubles themselves and the cvttsd2si
    conversions (INT_MIN for NaN and for out-of-range values)."""
    import ctypes as C
    rng = np.random.Generator(np.random.PCG64(int(rate * 1e9) % 2**31))
    c = np.concatenate([np.arange(-3, 300000, dtype=np.int64),
                        rng.integers(0, 2**31 - 1, size=300000),
                        np.array([2**31 - 1, 2**31 - 2, 2**30, 1000000000, -1, -2147483648])]).astype(np.int32)
    ctx = gpu_ctx_factory(23)
    gi, gd = ctx.selftest_get_bound(c, rate)
    P = oracle.make_params(23, 4, rate, b"H")
    L = oracle.lib()
    wi = np.array([L.rco_get_bound_int(C.byref(P), int(x)) for x in c[:5000].tolist() + c[-20006::400].tolist()], dtype=np.int32)
    sel = np.concatenate([np.arange(5000), np.arange(len(c))[-20006::400]])
    assert np.array_equal(gi[sel], wi)
    # the doubles: numpy float64 ops are the same IEEE operations (mul, sqrt, add; no FMA)
    cf = c.astype(np.float64)
    with np.errstate(invalid="ignore"):
        ce = cf * rate
        want_d = (ce + 6.0 * np.sqrt(ce)) + 1.0
    ok = c >= 0
    assert np.array_equal(gd[ok].view(np.uint64), want_d[ok].view(np.uint64))
    assert np.isnan(gd[~ok]).all() and (gi[~ok] == -2147483648).all()
    want_i = np.where(want_d[ok] < 2147483648.0, np.trunc(np.minimum(want_d[ok], 2147483647.0)), -2147483648.0).astype(np.int64)
    assert np.array_equal(gi[ok].astype(np.int64), want_i)


@pytest.mark.gpu
@pytest.mark.parametrize("k,length", [(23, 150), (23, 100), (31, 158), (15, 90), (32, 159), (19, 146), (11, 138),
                                      (25, 250), (25, 280), (32, 287), (23, 161), (21, 320), (23, 300),
                                      (23, 151), (23, 160), (17, 160), (16, 160), (15, 160)])
def test_strong_threshold_quarter_wave_equals_wave_per_read_and_oracle(gpu_ctx_factory, oracle, k, length, monkeypatch):
    """GetStrongTrustedThreshold through both threshold kernels -- four reads per wave
    (rc_quarter.h: four register layouts, up to 128 / 144 / 160 k-mers in 160 bases and up to 256 in 320;
    (23, 300) is beyond all of them) and one read per wave
    -- and through the oracle, on reads with ragged lengths (also < k), N runs, poly-A/T tails and
    count spectra with and without a 'drop'."""
    import torch
    rng = np.random.Generator(np.random.PCG64(1000 + k + length))
    s1, _, _, _, _ = synth.make_reads(4100 + k, 6000, length, n_tx=12, l_tx=max(600, length + 50), e=0.01, alpha=1.2)
    reads = []
    for i in range(len(s1)):
        r = bytearray(s1[i].tobytes())
        u = rng.random()
        if u < 0.25:
            r = r[:int(rng.integers(1, length + 1))]
        if u > 0.5 and u < 0.6 and len(r) > 30:
            t = int(rng.integers(10, len(r)))
            r[len(r) - t:] = (b"A" if rng.random() < 0.5 else b"T") * t
        if u > 0.6 and u < 0.7 and len(r) > 10:
            for p in rng.choice(len(r), int(rng.integers(1, 9)), replace=False):
                r[p] = ord("N") if rng.random() < 0.8 else ord("R")
        reads.append(bytes(r))
    keys, cnt = synth.count_kmers([s1], k)
    keep = rng.random(len(keys)) < 0.9
    ctx = gpu_ctx_factory(k=k)
    ctx.table_build(keys[keep], cnt[keep].astype(np.int32))
    ctx.set_run_params(0.01, b"#")
    arena, off = rcorrector_amd.pack_reads(reads)
    d_seq = torch.from_numpy(arena.copy()).cuda()
    d_off = torch.from_numpy(off.astype(np.int32)).cuda()
    out = {}
    for name, env in (("quarter", None), ("wave", "1")):
        if env:
            monkeypatch.setenv("RC_K2_WAVE_PER_READ", env)
        else:
            monkeypatch.delenv("RC_K2_WAVE_PER_READ", raising=False)
        d_strong = torch.full((len(reads),), -7, dtype=torch.int32, device="cuda")
        ctx.strong_threshold_device(d_seq, d_off, len(reads), arena.size, max(len(r) for r in reads), d_strong)
        ctx.sync()
        out[name] = d_strong.cpu().numpy()
    monkeypatch.delenv("RC_K2_WAVE_PER_READ", raising=False)
    T = oracle.Table(k, len(keys))
    T.put_many(keys[keep], cnt[keep].astype(np.int32))
    P = oracle.make_params(k, 4, 0.01, b"#")
    import ctypes
    want = np.array([oracle.lib().rco_strong_trusted_threshold(ctypes.byref(P), T.h, r) for r in reads], dtype=np.int32)
    assert np.array_equal(out["wave"], want)
    assert np.array_equal(out["quarter"], want)
    assert (want == -1).any() and (want > 10).any()


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu(tmp_path):
    """The N>1 path of bench.py as the driver launches it (torch.distributed.run, one rank per
    GPU) -- exercised here with both ranks on GPU 0 and gloo collectives (RC_BENCH_SHARED_GPU=1):
    one JSON line from rank 0, the whole-job value over both ranks, weak scaling."""
    import json
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RC_BENCH_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--reads", "400000", "--n-tx", "2000", "--cpu-sample", "0"]
    p = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["config"]["reads_per_gpu"] == 400000 and "x2" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 400000 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-6
    assert 0.2 < d["config"]["reads_corrected_frac"] < 0.9
    # every rank rebuilt the same replicated table (deterministic generator) and the digests were compared
    assert d["config"]["table_replicas_identical"] is True and len(d["config"]["table_digest"]) == 16


@pytest.mark.gpu
@pytest.mark.parametrize("config", [1, 2, 3, 4])
def test_full_size_presets_match_the_oracle_on_a_sample(config):
    """BASELINE.json configs[1] to [4] at FULL size (configs[2], the headline shard of 25 M x 150 bp pairs, also has the
    exhaustive test below): 10 M x 100 bp,
    50 M x 150 bp with the skewed spectrum, 100 M x 150 bp at k = 31 / maxcorK 8 / 5 % errors over the 871 M-k-mer table
    counted from all of them -- one timed step each through bench.py, then the oracle with the same table on three strata of
    the whole shard (bench.py: parity_strata): the first 100 000 reads, 100 000 drawn across every device arena by a seeded
    stride, and per arena the 1 000 reads with the most gather rounds (with their mates) -- the reads that stress the search
    at scale are in the sample by construction, not by luck: return values, l / m / h and corrected bases must be identical."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--config", str(config), "--steps", "1", "--warmup", "0",
           "--parity-only", "--cpu-sample", "200000"]
    p = subprocess.run(cmd, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["config"]["preset"] == config and d["n_gpus"] == 1
    assert d["config"]["reads_per_gpu"] == {1: 10_000_000, 2: 25_000_000, 3: 50_000_000, 4: 100_000_000}[config]
    cb = d["cpu_baseline"]
    assert cb["gpu_matches_oracle_on_sample"] is True, cb
    st = cb["strata"]
    assert set(st) == {"first", "stride", "heavy"} and all(v["identical"] for v in st.values()), st
    assert st["first"]["reads"] == 100_000 and abs(st["stride"]["reads"] - 100_000) < 200
    assert st["heavy"]["reads"] >= 1000 and st["heavy"]["max_gather_rounds"] == cb["worst_read_gather_rounds"] > st["first"]["max_gather_rounds"] - 1
    assert cb["instrumented_build_same_ret"] is True and "seeded stride" in cb["sample"] and d["config"]["reads_corrected_frac"] > 0.2


@pytest.mark.gpu
def test_every_read_of_the_headline_shard_matches_the_oracle():
    """The workload the bench line is quoted on, exhaustively: all 25 000 000 reads of the headline shard (BASELINE.json
    configs[2]: 150 bp pairs, k = 23) through the kernels the bench times, then the oracle -- all host cores, the same table --
    on every one of them: return values, l / m / h and corrected bases (`bench.py --parity-only --parity-full`)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--config", "2", "--steps", "1", "--warmup", "0",
           "--parity-only", "--parity-full", "--cpu-sample", "200000"]
    p = subprocess.run(cmd, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    d = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][0])
    cb = d["cpu_baseline"]
    assert d["config"]["preset"] == 2 and d["config"]["reads_per_gpu"] == 25_000_000
    assert cb["whole_shard"] is True and cb["strata"]["first"]["reads"] == 25_000_000, cb
    assert cb["gpu_matches_oracle_on_sample"] is True and all(v["identical"] for v in cb["strata"].values()), cb
    assert cb["instrumented_build_same_ret"] is True and cb["strata"]["first"]["corrected_reads"] > 5_000_000


@pytest.mark.gpu
def test_bench_without_a_launcher_runs_the_ranks_it_was_asked_for():
    """`python bench.py --gpus 2` with no launcher in front (round-5 review: it used to measure ONE GPU and print n_gpus 1):
    the process re-launches itself under torch.distributed.run -- here with both ranks on GPU 0 (RC_BENCH_SHARED_GPU=1) --
    and the one line that comes out is the two-rank one."""
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "400000", "--n-tx", "2000", "--cpu-sample", "0"]
    env = {kk: v for kk, v in os.environ.items() if kk not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, cwd=root, env=dict(env, RC_BENCH_SHARED_GPU="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["multi_gpu"]["world_size_seen"] == 2 and len(d["multi_gpu"]["ranks"]) == 2
    assert "x2" in d["config"]["parallelism"] and b"re-running as" in p.stderr


@pytest.mark.gpu
def test_bench_eight_ranks_on_one_gpu():
    """bench.py --gpus 8 as the driver launches it, all eight ranks on GPU 0 with gloo collectives
    (RC_BENCH_SHARED_GPU=1), 100 k reads per rank: so that the first run on eight real GPUs is not the first run of
    the eight-rank code path (digest all-reduce over 8 replicas, max-over-ranks timing, summary reduce)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RC_BENCH_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1",
           "--reads", "100000", "--n-tx", "500", "--cpu-sample", "0"]
    p = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and "x8" in d["config"]["parallelism"]
    assert abs(d["value"] - 8 * 100000 / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-6
    assert d["config"]["table_replicas_identical"] is True and 0.2 < d["config"]["reads_corrected_frac"] < 0.9
    mg = d["multi_gpu"]   # what the driver reads to see that the collectives saw eight ranks
    assert mg["world_size_seen"] == 8 and mg["collective_backend"] == "gloo" and len(mg["ranks"]) == 8
    assert [r["rank"] for r in mg["ranks"]] == list(range(8)) and all(r["device"] == 0 for r in mg["ranks"])
    assert mg["ms_per_step_min"] <= mg["ms_per_step_max"] <= d["ms_per_step"] * 1.001


def _bench_json(args, env=None, timeout=900, launcher=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable] + (launcher or []) + [os.path.join(root, "bench.py")] + args
    p = subprocess.run(cmd, cwd=root, env=dict(os.environ, **(env or {})), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_in_process_contexts_replicate_the_table(tmp_path):
    """`bench.py --in-process --gpus 3`: the `rcorrector -gpus N` shape -- one process, a context and a host thread per
    GPU, the table counted once and replicated with all copies in flight (rc_table_replicate_async), digests compared.
    Here all three contexts sit on GPU 0 (RC_BENCH_SHARED_GPU=1); the same single-GPU run must find the same reads to
    correct in shard 0 (results do not depend on how many contexts share the work)."""
    args = ["--steps", "2", "--warmup", "1", "--reads", "300000", "--n-tx", "1500"]
    d = _bench_json(["--in-process", "--gpus", "3"] + args, env={"RC_BENCH_SHARED_GPU": "1"})
    mg = d["multi_gpu"]
    assert d["n_gpus"] == 3 and mg["world_size_seen"] == 3 and len(mg["ranks"]) == 3 and mg["table_replicas_identical"] is True
    assert len(set(mg["table_digests"])) == 1 and mg["table_replicate_s"] > 0
    assert abs(d["value"] - 3 * 300000 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-6
    one = _bench_json(["--gpus", "1", "--no-extras"] + args)
    assert one["config"]["table_digest"] == mg["table_digests"][0] and one["config"]["table_kmers"] == d["config"]["table_kmers"]
    assert 0.2 < d["config"]["reads_corrected_frac"] < 0.9


@pytest.mark.gpu
def test_two_real_gpus_when_the_box_has_them(tmp_path):
    """Only where torch sees more than one GPU (the driver's 8-GPU node; skipped on the one-GPU boxes): bench.py under
    torch.distributed.run with the nccl (= RCCL) backend on two devices, the in-process mode with a peer-to-peer table copy,
    and `rcorrector -gpus 2` WITHOUT RC_SHARED_GPU -- the code paths the one-GPU suite can only run with every context on
    device 0 (distinct devices, peer access, NUMA binding per GPU)."""
    import socket
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    args = ["--steps", "2", "--warmup", "1", "--reads", "2000000", "--n-tx", "3000", "--cpu-sample", "0", "--no-extras"]
    d = _bench_json(["--gpus", "2"] + args, launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                                                      "127.0.0.1", "--master-port", str(port)])
    mg = d["multi_gpu"]
    assert d["n_gpus"] == 2 and mg["collective_backend"] == "nccl" and mg["world_size_seen"] == 2
    assert sorted(r["device"] for r in mg["ranks"]) == [0, 1] and d["config"]["table_replicas_identical"] is True
    ip = _bench_json(["--in-process", "--gpus", "2"] + args)
    assert sorted(r["device"] for r in ip["multi_gpu"]["ranks"]) == [0, 1] and len(set(ip["multi_gpu"]["table_digests"])) == 1
    assert ip["config"]["table_kmers"] == d["config"]["table_kmers"]
    # the CLI on two devices equals the CLI on one
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fx = os.path.join(root, "tests", "golden", "fx_pe_k23")
    cli_args = open(os.path.join(fx, "cmd.txt")).read().split()
    outs = []
    for g in ("1", "2"):
        od = tmp_path / ("g" + g)
        p = subprocess.run([os.path.join(root, "rcorrector_amd", "rcorrector")] + cli_args + ["-od", str(od), "-gpus", g, "-batch", "64"], cwd=fx,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        outs.append({f: open(os.path.join(od, f), "rb").read() for f in sorted(os.listdir(od))})
    assert outs[0] == outs[1] and len(outs[0]) == 2


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(400, 420)))
def test_jfdump_loader_on_quirky_dumps(gpu_ctx_factory, oracle, seed, tmp_path):
    """The multi-threaded dump parser (fast path for the clean layout, general tokeniser for the
    rest) against the oracle's loader -- itself pinned to the reference on the same dumps
    (tests/test_oracle_vs_ref.py): number stored, every count, ERROR_RATE."""
    import io_quirks
    path = str(tmp_path / "d.jf")
    io_quirks.make_quirky_dump(seed, path, n=60000 if seed % 4 == 0 else 4000, mid_n=seed % 2 == 0)   # some above the parser's threading threshold
    T = oracle.Table(23, 1 << 12)
    stored = T.load_dump(path)
    want_rate = T.error_rate(path, 0.5)
    ctx = gpu_ctx_factory(23)
    assert ctx.load_jfdump(path) == stored
    assert ctx.estimate_error_rate(0.5) == want_rate   # (0.01 fallback for some seeds, estimated for others)
    codes, counts = T.export()
    assert np.array_equal(ctx.lookup(codes), counts)
    got_c, got_n = ctx.table_export()
    o1, o2 = np.argsort(got_c), np.argsort(codes)
    assert np.array_equal(got_c[o1], codes[o2]) and np.array_equal(got_n[o1], counts[o2])


@pytest.mark.gpu
@pytest.mark.parametrize("pinned", [False, True])
def test_submit_wait_slots_in_flight_equal_synchronous(gpu_ctx_factory, oracle, pinned):
    """rc_submit / rc_wait: several batches in flight in one context (pageable buffers staged through
    the slots' pinned memory, or page-locked buffers used directly) give what the oracle gives for
    each batch, in any completion order of the waits, and the device-side summary adds them up."""
    d = datasets.make("pe_k23")
    want = datasets.run_oracle(oracle, d)
    ctx = _table(gpu_ctx_factory, d)
    n = len(d["seqs1"])
    cuts = [0, n // 5, n // 2, n // 2, n]   # four batches, one of them empty
    of1, of2 = oracle.pack_reads(d["seqs1"])[1], oracle.pack_reads(d["seqs2"])[1]
    subs = []
    for s in range(4):
        lo, hi = cuts[s], cuts[s + 1]
        a1, o1 = oracle.pack_reads(d["seqs1"][lo:hi]); q1, _ = oracle.pack_reads(d["quals1"][lo:hi])
        a2, o2 = oracle.pack_reads(d["seqs2"][lo:hi]); q2, _ = oracle.pack_reads(d["quals2"][lo:hi])
        res = None
        if pinned:
            def pin(x):
                y = ctx.host_array(len(x), x.dtype)
                y[:] = x
                return y
            a1, q1, a2, q2 = pin(a1), pin(q1), pin(a2), pin(q2)
            res = [ctx.host_array(2 * (hi - lo), np.int32) for _ in range(4)]
        ctx.submit(s, 1, a1, q1, o1, a2, q2, o2, res=res)
        subs.append((lo, hi, a1, a2))
    with pytest.raises(Exception):
        ctx.submit(0, 1, subs[0][2], subs[0][2], np.zeros(1, np.uint32), subs[0][3], subs[0][3], np.zeros(1, np.uint32))  # slot busy
    for s in (2, 0, 3, 1):
        lo, hi, a1, a2 = subs[s]
        ret, l, m, h = ctx.wait(s)
        m_ = hi - lo
        for got, w in ((ret, want[0]), (l, want[1]), (m, want[2]), (h, want[3])):
            assert np.array_equal(got[:m_], w[lo:hi]) and np.array_equal(got[m_:], w[n + lo:n + hi]), "slot %d" % s
        assert np.array_equal(np.asarray(a1), want[4][of1[lo]:of1[hi]]) and np.array_equal(np.asarray(a2), want[5][of2[lo]:of2[hi]])
    reads, cors = ctx.summary()
    assert reads == 2 * n and cors == int(want[0][want[0] > 0].sum())
    ctx.close()


@pytest.mark.gpu
def test_correct_batch_refuses_wrong_dtype(gpu_ctx_factory, oracle):
    d = datasets.make("se_k23")
    ctx = _table(gpu_ctx_factory, d)
    a, off = oracle.pack_reads(d["seqs1"])
    qa, _ = oracle.pack_reads(d["quals1"])
    with pytest.raises(TypeError):
        ctx.correct_batch(0, a.astype(np.int32), qa, off)
    with pytest.raises(TypeError):
        ctx.correct_batch(0, a[::2], qa, off)
    ret = ctx.correct_batch(0, a, qa, off.astype(np.int64))[0]   # offsets are converted
    assert (ret > 0).sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["plain", "core"])
@pytest.mark.parametrize("k", [15, 23, 31, 32])
def test_table_absence_filter_never_hides_an_entry(k, kind, monkeypatch):
    """The absence filter of large PACKED tables (forced here): a lookup that fails the filter answers 0 without reading a
    bucket, so the filter must hold every entry -- in both of its kinds (the word chosen by the mixed code, or by the first
    k - 1 bases of either orientation: rc_common.h) -- whichever orientation the lookup is made in; absent keys stay absent."""
    rng = np.random.Generator(np.random.PCG64(7 + k))
    mask = np.uint64((1 << (2 * k)) - 1) if k < 32 else np.uint64(0xFFFFFFFFFFFFFFFF)
    fwd = (rng.integers(0, 1 << 63, size=400000, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=400000, dtype=np.uint64)) & mask
    can = np.unique(np.minimum(fwd, _revcomp_codes(fwd, k)))
    counts = rng.integers(2, 60000, size=len(can)).astype(np.int32)
    absent = (rng.integers(0, 1 << 63, size=100000, dtype=np.uint64) * np.uint64(2) + np.uint64(1)) & mask
    absent = absent[~np.isin(np.minimum(absent, _revcomp_codes(absent, k)), can)]
    res = {}
    for filt in ("off", "force"):
        monkeypatch.setenv("RC_TABLE_FILTER", filt)
        monkeypatch.setenv("RC_TABLE_FILTER_KIND", kind)
        ctx = rcorrector_amd.Context(k=k, device=0)
        ctx.table_build(can, counts)
        res[filt] = (ctx.table_layout(), ctx.lookup(can), ctx.lookup(_revcomp_codes(can, k)), ctx.lookup(absent), ctx.table_digest())
        ctx.close()
    if res["force"][0] == 1:   # (PACKED: the filter is there)
        assert np.array_equal(res["force"][1], counts) and np.array_equal(res["force"][2], counts)
        assert not res["force"][3].any()
    for a, b in zip(res["off"], res["force"]):
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("k,n", [(11, 300000), (16, 300000), (17, 300000), (23, 300000), (25, 300000), (28, 300000),
                                 (31, 300000), (31, 17000000), (32, 300000)])
def test_table_packed_and_wide_layouts_hold_the_same_table(k, n, monkeypatch):
    """The two slot layouts (rc_common.h) of the k-mer table answer every lookup alike, export the same
    (code, count) set and have the same content digest.  PACKED is chosen when k, the counts and the
    placement allow it: its remainder takes `ext` = 2k - 32 - log2(buckets) extra bits out of the count
    field when the table is small for its k (up to 8: k = 31 needs 17 M entries), and
    counts that need more than the 27 - ext bits left go to a side array in front of the buckets; WIDE
    takes over when there are too many of those, or when a key would sit too far from its home."""
    rng = np.random.Generator(np.random.PCG64(100 + k))
    mask = np.uint64((1 << (2 * k)) - 1) if k < 32 else np.uint64(0xFFFFFFFFFFFFFFFF)
    fwd = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    fwd &= mask
    can = np.unique(np.minimum(fwd, _revcomp_codes(fwd, k)))
    buckets = int((len(can) + 1000) / (_bucket_bytes() // 8 * 0.4)) + 1   # the build's choice for a small table: entries / (slots per bucket * load 0.4)
    ext = 0
    while 2 * k > 32 and (buckets << ext) < (1 << (2 * k - 32)):
        ext += 1
    limit = 1 << (27 - min(ext, 8))
    counts = rng.integers(2, min(1 << 20, limit) - 1, size=len(can)).astype(np.int32)
    counts[2000:2007] = [2, limit - 1, 3, 100, 2, min(65535, limit - 2), limit >> 1]
    # duplicates: the later Put wins (Store.h:55)
    codes = np.concatenate([can, can[:1000]])
    cnts = np.concatenate([counts, counts[:1000] + 1])
    want = counts.copy()
    want[:1000] += 1
    probes = np.concatenate([can[:2000000], rng.integers(0, 1 << 62, size=50000, dtype=np.uint64) & mask])
    res = {}
    for layout in ("packed", "wide"):
        monkeypatch.setenv("RC_TABLE_LAYOUT", layout)
        ctx = rcorrector_amd.Context(k=k, device=0)
        ctx.table_build(codes, cnts)
        ec, en = ctx.table_export()
        o = np.argsort(ec)
        res[layout] = (ctx.table_layout(), ctx.lookup(probes), ec[o], en[o], ctx.table_digest(), ctx.table_stats()["bytes"])
        ctx.close()
    assert res["wide"][0] == 0
    if ext <= 8:
        assert res["packed"][0] == 1 and res["packed"][5] < res["wide"][5]
    else:
        assert res["packed"][0] == 0
    for a, b in zip(res["packed"][1:5], res["wide"][1:5]):
        assert np.array_equal(a, b)
    assert np.array_equal(res["wide"][1][:min(len(can), 2000000)], want[:2000000])
    assert np.array_equal(res["wide"][2], can) and np.array_equal(res["wide"][3], want)
    # counts that do not fit the count field go to the table's prefix (rc_common.h) and the table stays
    # PACKED -- also where a key was Put twice (the later Put is the table's entry, Store.h:55) ...
    monkeypatch.setenv("RC_TABLE_LAYOUT", "packed")
    if ext <= 8:
        ctx = rcorrector_amd.Context(k=k, device=0)
        cnts2 = cnts.copy()
        big = [limit - 1, limit, 3 * limit + 7, (1 << 31) - 1]
        mid = len(can) // 2
        for j, v in enumerate(big):
            cnts2[mid + j] = v
        cnts2[5], cnts2[len(can) + 5] = 2 * limit, 2 * limit + 1     # both Puts overflow
        cnts2[6], cnts2[len(can) + 6] = 2 * limit, 77                # only the shadowed one does
        cnts2[7], cnts2[len(can) + 7] = 78, 2 * limit + 9            # only the live one does
        ctx.table_build(codes, cnts2)
        assert ctx.table_layout() == 1
        got = ctx.lookup(np.concatenate([can[mid:mid + 4], can[5:8], can[100:110]]))
        assert got.tolist() == big + [2 * limit + 1, 77, 2 * limit + 9] + (counts[100:110] + 1).tolist()
        ec, en = ctx.table_export()
        o = np.argsort(ec)
        w2 = want.copy()
        w2[mid:mid + 4] = big
        w2[5:8] = [2 * limit + 1, 77, 2 * limit + 9]
        assert np.array_equal(ec[o], can) and np.array_equal(en[o], w2)
        ctx.close()
    # ... up to 4000 of them; beyond that the build falls back to WIDE by itself
    ctx = rcorrector_amd.Context(k=k, device=0)
    cnts3 = cnts.copy()
    cnts3[20000:24100] = limit + 5
    ctx.table_build(codes, cnts3)
    assert ctx.table_layout() == 0 and ctx.lookup(can[20000:20001])[0] == limit + 5
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["se_k23", "pe_var", "il_k23", "nrich"])
def test_quality_bits_give_the_same_results_as_quality_bytes(gpu_ctx_factory, oracle, name):
    """rc_set_quality_bits: the vetoes only compare qualities with badQualityThreshold, so a bit per base
    (packed by rc_pack_quality_bits) must reproduce the oracle exactly as the byte arenas do -- here with
    a threshold in the middle of the qualities the data sets use ('#' < 'H' < 'I'), ragged paired arenas
    (the second arena's bits live in a region of their own) and both host entry points."""
    d = datasets.make(name)
    want = datasets.run_oracle(oracle, d)
    ctx = _table(gpu_ctx_factory, d)
    ctx.set_quality_bits(True)
    a, off = oracle.pack_reads(d["seqs1"])
    qa, _ = oracle.pack_reads(d["quals1"])
    qb = ctx.pack_quality_bits(qa, b"H")
    assert qb.size == (qa.size + 7) // 8 and np.array_equal(np.unpackbits(qb, bitorder="little")[:qa.size], (qa.view(np.int8) > ord("H")).astype(np.uint8))
    if d["mode"] == 1:
        a2, off2 = oracle.pack_reads(d["seqs2"])
        qa2, _ = oracle.pack_reads(d["quals2"])
        got = ctx.correct_batch(1, a, qb, off, a2, ctx.pack_quality_bits(qa2, b"H"), off2) + (a, a2)
    else:
        got = ctx.correct_batch(d["mode"], a, qb, off) + (a,)
    for w, g, what in zip(want, got, ["ret", "l", "m", "h", "seq1", "seq2"]):
        assert np.array_equal(w, g), "%s differs on %s in quality-bit mode" % (what, name)
    ctx.set_quality_bits(False)
    a, off = oracle.pack_reads(d["seqs1"])
    if d["mode"] == 1:
        a2, _ = oracle.pack_reads(d["seqs2"])
        got = ctx.correct_batch(1, a, qa, off, a2, qa2, off2)
    else:
        got = ctx.correct_batch(d["mode"], a, qa, off)
    assert np.array_equal(got[0], want[0])


def _packed_inputs(ctx, oracle, d, fasta=False):
    """One arena (mode 1: first mates, then second mates) and its packed form, all in page-locked arrays."""
    a, off = oracle.pack_reads(d["seqs1"])
    qa, _ = oracle.pack_reads(d["quals1"])
    if d["mode"] == 1:
        a2, off2 = oracle.pack_reads(d["seqs2"])
        qa2, _ = oracle.pack_reads(d["quals2"])
        off = np.concatenate([off, (off2[1:].astype(np.int64) + a.size).astype(np.uint32)])
        a, qa = np.concatenate([a, a2]), np.concatenate([qa, qa2])
    arena = ctx.host_array(a.size)
    arena[:] = a
    bases, exc_pos, exc_chr = ctx.pack_bases(arena, bases=ctx.host_array((a.size + 15) // 16, np.uint32))
    qb = None
    if not fasta:
        qb = ctx.host_array((a.size + 7) // 8)
        ctx.pack_quality_bits(qa, b"H", out=qb)
    return arena, off, bases, exc_pos, exc_chr, qb


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["se_k23", "pe_k23", "il_k23", "pe_var", "nrich", "edge", "varlen", "k31_mc8", "tiers_pe", "long600_k31"])
def test_packed_boundary_gives_the_oracles_results(gpu_ctx_factory, oracle, name):
    """rc_submit_packed / rc_wait_packed (packed reads down, fix list up): 2 bits per base + the
    letters outside ACGT as a list + a quality bit per base go down, ret / l / m / h and the substitutions come back.
    ret / l / m / h must be the oracle's, and the caller's arena with the fixes applied must be the oracle's corrected
    arena -- on N-rich and adversarial reads (IUPAC letters, N replaced by A: the one substitution the packed codes
    cannot show), ragged pairs, interleaved pairs, long reads; the fix count equals the bases the oracle changed."""
    d = datasets.make(name)
    want = datasets.run_oracle(oracle, d)
    ctx = _table(gpu_ctx_factory, d)
    arena, off, bases, exc_pos, exc_chr, qb = _packed_inputs(ctx, oracle, d)
    before = arena.copy()
    assert len(exc_pos) == int(((before != 0) & ~np.isin(before, np.frombuffer(b"ACGT", np.uint8))).sum())
    ctx.submit_packed(0, d["mode"], arena.size, off, bases, qb, exc_pos, exc_chr)
    ret, l, m, h, fix_pos, fix_chr = ctx.wait_packed(0)
    for w, g, what in zip(want[:4], (ret, l, m, h), ["ret", "l", "m", "h"]):
        assert np.array_equal(w, g), "%s differs on %s through the packed boundary" % (what, name)
    assert np.array_equal(arena, before)       # the caller's arena is not touched ...
    ctx.apply_fixes(arena, fix_pos, fix_chr)   # ... until the caller applies the fixes
    want_arena = np.concatenate(want[4:])
    assert np.array_equal(arena, want_arena), "corrected bases differ on %s through the packed boundary" % name
    assert len(fix_pos) == int((before != want_arena).sum()) and len(np.unique(fix_pos)) == len(fix_pos)
    assert len(fix_pos) == int(want[0][want[0] > 0].sum())   # ErrorCorrection's return value counts exactly these (:1468-1479)


@pytest.mark.gpu
def test_packed_boundary_slots_fasta_and_errors(gpu_ctx_factory, oracle):
    """Three packed batches in flight give what three synchronous byte batches give; no quality array = FASTA input
    (qual[0] == 0, Reads.h:224-266) = the byte path with zeroed quality arenas; too little room for the fixes and a
    byte-path wait on a packed slot are errors, and the context works on after them."""
    d = datasets.make("pe_var")
    ctx = _table(gpu_ctx_factory, d)
    n1 = len(d["seqs1"])
    parts = []
    for j in range(3):
        lo, hi = j * n1 // 3, (j + 1) * n1 // 3
        sub = dict(d, seqs1=d["seqs1"][lo:hi], quals1=d["quals1"][lo:hi], seqs2=d["seqs2"][lo:hi], quals2=d["quals2"][lo:hi])
        parts.append((sub, _packed_inputs(ctx, oracle, sub)))
    for j, (sub, (arena, off, bases, exc_pos, exc_chr, qb)) in enumerate(parts):
        ctx.submit_packed(j, 1, arena.size, off, bases, qb, exc_pos, exc_chr)
    for j, (sub, (arena, off, bases, exc_pos, exc_chr, qb)) in enumerate(parts):
        ret, l, m, h, fix_pos, fix_chr = ctx.wait_packed(j)
        want = datasets.run_oracle(oracle, sub)
        ctx.apply_fixes(arena, fix_pos, fix_chr)
        assert np.array_equal(ret, want[0]) and np.array_equal(m, want[2]) and np.array_equal(arena, np.concatenate(want[4:]))
    # FASTA: no qualities
    sub, (arena, off, bases, exc_pos, exc_chr, _qb) = parts[0][0], _packed_inputs(ctx, oracle, parts[0][0], fasta=True)
    ctx.submit_packed(1, 1, arena.size, off, bases, None, exc_pos, exc_chr)
    ret, l, m, h, fix_pos, fix_chr = ctx.wait_packed(1)
    a1, o1 = oracle.pack_reads(sub["seqs1"])
    a2, o2 = oracle.pack_reads(sub["seqs2"])
    got = ctx.correct_batch(1, a1, np.zeros_like(a1), o1, a2, np.zeros_like(a2), o2)
    ctx.apply_fixes(arena, fix_pos, fix_chr)
    assert np.array_equal(ret, got[0]) and np.array_equal(h, got[3]) and np.array_equal(arena, np.concatenate([a1, a2]))
    # errors
    arena, off, bases, exc_pos, exc_chr, qb = _packed_inputs(ctx, oracle, sub)
    ctx.submit_packed(2, 1, arena.size, off, bases, qb, exc_pos, exc_chr, fix_cap=1)
    with pytest.raises(rcorrector_amd.RcorrectorError, match="room for 1"):
        ctx.wait_packed(2)
    ctx.submit_packed(0, 1, arena.size, off, bases, qb, exc_pos, exc_chr)
    with pytest.raises(rcorrector_amd.RcorrectorError, match="packed"):
        ctx._ck(ctx._L.rc_wait(ctx._h, 0))
    ret2 = ctx.wait_packed(0)[0]
    assert np.array_equal(ret2, datasets.run_oracle(oracle, sub)[0])
    with pytest.raises(rcorrector_amd.RcorrectorError):
        ctx.submit_packed(0, 1, arena.size + 1, off, bases, qb, exc_pos, exc_chr)   # off[total] is not the arena's size
    # offsets that do not ascend and exceptions outside the arena are refused before anything is launched (the terminator and
    # exception kernels write seq[off[i+1]-1] / seq[exc_pos[i]] unchecked), and a refused submit leaves the slot free
    bad = off.copy()
    bad[3] = bad[2]
    with pytest.raises(rcorrector_amd.RcorrectorError, match="ascend"):
        ctx.submit_packed(0, 1, arena.size, bad, bases, qb, exc_pos, exc_chr)
    bad = off.copy()
    bad[0] = 1
    with pytest.raises(rcorrector_amd.RcorrectorError, match=r"off\[0\]"):
        ctx.submit_packed(0, 1, arena.size, bad, bases, qb, exc_pos, exc_chr)
    with pytest.raises(rcorrector_amd.RcorrectorError, match="outside the arena"):
        ctx.submit_packed(0, 1, arena.size, off, bases, qb, np.array([arena.size], dtype=np.uint32), np.array([ord("N")], dtype=np.uint8))
    with pytest.raises(rcorrector_amd.RcorrectorError, match="holds no packed batch"):
        ctx.wait_packed(0)
    ctx.submit_packed(0, 1, arena.size, off, bases, qb, exc_pos, exc_chr)
    assert np.array_equal(ctx.wait_packed(0)[0], ret2)


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [True, False])
def test_slot_lanes_run_batches_side_by_side_with_the_same_results(gpu_ctx_factory, oracle, lanes):
    """Slots > 0 of the asynchronous entry points are lanes -- contexts of their own that borrow the table, the parameters and
    the kept arenas -- so batches in flight overlap on the GPU and may finish in any order; rc_set_slot_lanes(0) keeps every
    slot in the one context.  Four packed batches in four slots, waited for in reverse order, and a switch between a submit
    and its wait: every batch gives the oracle's results either way, and rc_summary counts them all."""
    d = datasets.make("pe_var")
    ctx = _table(gpu_ctx_factory, d)
    ctx.set_slot_lanes(lanes)
    n1 = len(d["seqs1"])
    parts = []
    for j in range(4):
        lo, hi = j * n1 // 4, (j + 1) * n1 // 4
        sub = dict(d, seqs1=d["seqs1"][lo:hi], quals1=d["quals1"][lo:hi], seqs2=d["seqs2"][lo:hi], quals2=d["quals2"][lo:hi])
        parts.append((sub, _packed_inputs(ctx, oracle, sub)))
    for rnd in range(2):
        for j, (sub, (arena, off, bases, exc_pos, exc_chr, qb)) in enumerate(parts):
            ctx.submit_packed(j, 1, arena.size, off, bases, qb, exc_pos, exc_chr)
        if rnd == 1:
            ctx.set_slot_lanes(not lanes)   # the batches in flight are waited for where they were submitted
        for j in (3, 2, 1, 0):
            sub, (arena, off, bases, exc_pos, exc_chr, qb) = parts[j]
            ret, l, m, h, fix_pos, fix_chr = ctx.wait_packed(j)
            want = datasets.run_oracle(oracle, sub)
            a = arena.copy()
            ctx.apply_fixes(a, fix_pos, fix_chr)
            assert np.array_equal(ret, want[0]) and np.array_equal(l, want[1]) and np.array_equal(m, want[2]) and np.array_equal(h, want[3])
            assert np.array_equal(a, np.concatenate(want[4:]))
    reads, bases_fixed = ctx.summary()
    want_all = datasets.run_oracle(oracle, d)
    assert reads == 2 * 2 * n1 and bases_fixed == 2 * int(want_all[0][want_all[0] > 0].sum())
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pe_k23", "se_k23", "k31_mc8"])
def test_parked_arenas_are_corrected_with_a_table_from_another_context(gpu_ctx_factory, oracle, name):
    """rc_table_count_park: a counting session ended WITHOUT counting leaves its arenas in HBM as kept arenas (no table is
    built, a table that is there stays) -- what a GPU that corrects reads whose k-mers another GPU counts does (`rcorrector -gpus
    N` without -c: one Store, T workers, main.cpp:294-308,451).  The table comes from a second context (rc_table_replicate); the
    parked reads, corrected through rc_submit_resident, give the oracle's results; sequence errors are errors."""
    d = datasets.make(name)
    want = datasets.run_oracle(oracle, d)
    owner = _table(gpu_ctx_factory, d)                      # "GPU 0": holds the table
    ctx = gpu_ctx_factory(d["k"], d["mfk"])                  # the GPU that only keeps and corrects
    with pytest.raises(rcorrector_amd.RcorrectorError, match="count_begin"):
        ctx.count_park()
    a1, off1 = oracle.pack_reads(d["seqs1"])
    q1, _ = oracle.pack_reads(d["quals1"])
    ctx.count_begin()
    ctx.count_add(a1)
    arena, qa, off = a1, q1, off1
    if d["mode"] == 1:
        a2, off2 = oracle.pack_reads(d["seqs2"])
        q2, _ = oracle.pack_reads(d["quals2"])
        ctx.count_add(a2)
        off = np.concatenate([off1, (off2[1:].astype(np.int64) + a1.size).astype(np.uint32)])
        arena, qa = np.concatenate([a1, a2]), np.concatenate([q1, q2])
    ctx.count_park()
    assert list(ctx.count_arenas()) == ([a1.size, a2.size] if d["mode"] == 1 else [a1.size])
    ctx.replicate_table_of(owner)
    assert ctx.table_digest() == owner.table_digest()
    ctx.set_run_params(d["rate"], b"H")
    qb = ctx.host_array((arena.size + 7) // 8)
    ctx.pack_quality_bits(qa, b"H", out=qb)
    args = dict(arena_a=0, begin_a=0, bytes_a=a1.size)
    if d["mode"] == 1:
        args.update(arena_b=1, begin_b=0, bytes_b=a2.size)
    ctx.submit_resident(0, d["mode"], off, qb, **args)
    ret, l, m, h, fix_pos, fix_chr = ctx.wait_resident(0)
    for w, g, what in zip(want[:4], (ret, l, m, h), ["ret", "l", "m", "h"]):
        assert np.array_equal(w, g), "%s differs on %s" % (what, name)
    host = arena.copy()
    ctx.apply_fixes(host, fix_pos, fix_chr)
    assert np.array_equal(host, np.concatenate(want[4:]))
    ctx.count_release()
    assert len(ctx.count_arenas()) == 0
    ctx.close()
    owner.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["se_k23", "pe_k23", "il_k23", "pe_var", "nrich", "edge", "varlen", "k31_mc8", "tiers_pe", "long600_k31"])
def test_resident_boundary_gives_the_oracles_results(gpu_ctx_factory, oracle, name):
    """rc_submit_resident / rc_wait_resident: the reads are the arenas the k-mer counter kept in HBM (rc_table_count_keep) --
    counted once, corrected where they lie; only offsets and quality bits go down.  Same contract as the packed boundary:
    ret / l / m / h are the oracle's, the host arena plus the fix list is the oracle's corrected arena, the kept arenas stay
    as they were (a second submission gives the same list), sub-ranges of an arena are batches of their own."""
    d = datasets.make(name)
    want = datasets.run_oracle(oracle, d)
    ctx = gpu_ctx_factory(d["k"], d["mfk"])
    a1, off1 = oracle.pack_reads(d["seqs1"])
    q1, _ = oracle.pack_reads(d["quals1"])
    ctx.count_keep(True)
    ctx.count_begin()
    ctx.count_add(a1)
    arena, qa, off = a1, q1, off1
    if d["mode"] == 1:
        a2, off2 = oracle.pack_reads(d["seqs2"])
        q2, _ = oracle.pack_reads(d["quals2"])
        ctx.count_add(a2)
        off = np.concatenate([off1, (off2[1:].astype(np.int64) + a1.size).astype(np.uint32)])
        arena, qa = np.concatenate([a1, a2]), np.concatenate([q1, q2])
    ctx.count_finish(2)
    kept = ctx.count_arenas()
    assert list(kept) == ([a1.size, a2.size] if d["mode"] == 1 else [a1.size])
    # (counting built a table of these reads' own k-mers: the data set's table and parameters go back in)
    ctx.table_build(d["keys"], d["counts"])
    ctx.set_run_params(d["rate"], b"H")
    qb = ctx.host_array((arena.size + 7) // 8)
    ctx.pack_quality_bits(qa, b"H", out=qb)
    args = dict(arena_a=0, begin_a=0, bytes_a=a1.size)
    if d["mode"] == 1:
        args.update(arena_b=1, begin_b=0, bytes_b=a2.size)
    lists = []
    for _ in range(2):
        ctx.submit_resident(0, d["mode"], off, qb, **args)
        ret, l, m, h, fix_pos, fix_chr = ctx.wait_resident(0)
        for w, g, what in zip(want[:4], (ret, l, m, h), ["ret", "l", "m", "h"]):
            assert np.array_equal(w, g), "%s differs on %s through the resident boundary" % (what, name)
        lists.append(sorted(zip(fix_pos.tolist(), fix_chr.tolist())))
    assert lists[0] == lists[1]
    host = arena.copy()
    ctx.apply_fixes(host, fix_pos, fix_chr)
    want_arena = np.concatenate(want[4:])
    assert np.array_equal(host, want_arena), "corrected bases differ on %s through the resident boundary" % name
    assert len(fix_pos) == int((arena != want_arena).sum()) == int(want[0][want[0] > 0].sum())
    # the second half of the units as a batch of its own: ranges that start inside the kept arenas (any alignment)
    n1 = len(off1) - 1
    step = 2 if d["mode"] == 2 else 1
    lo = (n1 // 2) // step * step
    if 0 < lo < n1:
        sub_off = [off1[lo:].astype(np.int64) - int(off1[lo])]
        sargs = dict(arena_a=0, begin_a=int(off1[lo]), bytes_a=int(a1.size - off1[lo]))
        sub_q = [q1[off1[lo]:]]
        if d["mode"] == 1:
            sub_off.append(off2[lo + 1:].astype(np.int64) - int(off2[lo]) + sargs["bytes_a"])
            sargs.update(arena_b=1, begin_b=int(off2[lo]), bytes_b=int(a2.size - off2[lo]))
            sub_q.append(q2[off2[lo]:])
        sqb = ctx.host_array((sargs["bytes_a"] + sargs.get("bytes_b", 0) + 7) // 8)
        ctx.pack_quality_bits(np.concatenate(sub_q), b"H", out=sqb)
        ctx.submit_resident(1, d["mode"], np.concatenate(sub_off).astype(np.uint32), sqb, **sargs)
        ret, l, m, h, fix_pos, fix_chr = ctx.wait_resident(1)
        idx = np.arange(lo, n1)
        if d["mode"] == 1:
            idx = np.concatenate([idx, n1 + idx])
        for w, g, what in zip(want[:4], (ret, l, m, h), ["ret", "l", "m", "h"]):
            assert np.array_equal(w[idx], g), "%s differs on the second half of %s through the resident boundary" % (what, name)
        sub = np.concatenate([a1[off1[lo]:]] + ([a2[off2[lo]:]] if d["mode"] == 1 else []))
        ctx.apply_fixes(sub, fix_pos, fix_chr)
        assert np.array_equal(sub, np.concatenate([want[4][off1[lo]:]] + ([want[5][off2[lo]:]] if d["mode"] == 1 else [])))
    # errors: a range beyond the arena, an arena that is not there, released arenas
    with pytest.raises(rcorrector_amd.RcorrectorError, match="kept arena"):
        ctx.submit_resident(0, d["mode"], off, qb, **dict(args, bytes_a=a1.size + 16))
    ctx.count_release()
    assert len(ctx.count_arenas()) == 0
    with pytest.raises(rcorrector_amd.RcorrectorError, match="kept arena"):
        ctx.submit_resident(0, d["mode"], off, qb, **args)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["se_k23", "pe_k23", "il_k23", "pe_var", "varlen", "edge", "k31_mc8", "long600_k31", "polya_k23",
                                  "se_151", "pe_151", "pe_160_k15", "tiers_se", "tiers_pe", "tiers_il"])
def test_locality_order_does_not_change_results(oracle, name, monkeypatch):
    """Large batches are processed in min-hash order (overlapping reads next to each other, rc_table.hip):
    a pure reordering -- forced here on the small parity sets, ragged, paired and interleaved ones included,
    it must leave every result and every corrected base where the caller's order has them."""
    d = datasets.make(name)
    want = datasets.run_oracle(oracle, d)
    monkeypatch.setenv("RC_LOCALITY", "force")
    ctx = rcorrector_amd.Context(k=d["k"], max_fix_per_k=d["mfk"], device=0)
    ctx.table_build(d["keys"], d["counts"])
    ctx.set_run_params(d["rate"], b"H")
    a, off = oracle.pack_reads(d["seqs1"])
    qa, _ = oracle.pack_reads(d["quals1"])
    if d["mode"] == 1:
        a2, off2 = oracle.pack_reads(d["seqs2"])
        qa2, _ = oracle.pack_reads(d["quals2"])
        got = ctx.correct_batch(1, a, qa, off, a2, qa2, off2) + (a, a2)
    else:
        got = ctx.correct_batch(d["mode"], a, qa, off) + (a,)
    for w, g, what in zip(want, got, ["ret", "l", "m", "h", "seq1", "seq2"]):
        assert np.array_equal(w, g), "%s differs on %s in locality order" % (what, name)
    ctx.close()


def _oracle_on_pairs(oracle, ctx, k, mfk, rate, bad_q, seq, qual, n_reads, L, pairs):
    """The oracle on the first `pairs` pairs of a mode-1 device batch (all first mates, then all second
    mates), with the table exported from the context.  Returns ((ret, l, m, h), arena1, arena2)."""
    codes, counts = ctx.table_export()
    T = oracle.Table(k, len(codes))
    T.put_many(codes, counts)
    P = oracle.make_params(k, mfk, rate, bad_q)
    half = n_reads // 2
    nb, b2 = pairs * (L + 1), half * (L + 1)
    a1, q1 = seq[:nb].cpu().numpy().copy(), qual[:nb].cpu().numpy().copy()
    a2, q2 = seq[b2:b2 + nb].cpu().numpy().copy(), qual[b2:b2 + nb].cpu().numpy().copy()
    off = (np.arange(pairs + 1, dtype=np.int64) * (L + 1)).astype(np.uint32)
    r = oracle.correct_batch(P, T, 1, a1, q1, off, a2, q2, off, threads=8)
    return r, a1, a2


@pytest.mark.gpu
def test_default_dispatch_at_a_size_that_selects_the_timed_path(oracle):
    """The kernels bench.py times -- k_unit_key -> radix sort -> k_probe_threshold_list on a PACKED table, the
    work-list driven k_correct compiled for k = 23 -- are what rc_correct_device picks BY ITSELF for >= 2^18
    reads over a table beyond 128 MB (no environment knob here).  400 k paired reads over a 6.6 M-base
    transcriptome: the first 10 000 pairs against the oracle, 100 000 more through the batch-split property."""
    import torch
    import bench as B
    dev = torch.device("cuda", 0)
    n, L, k = 400_000, 150, 23
    cnt_reads = 1_600_000
    seq_all, qual_all = B.synth_reads_gpu(777001, cnt_reads, L, 4400, 1500, 0.8, 0.005, dev, paired=True)
    ctx = rcorrector_amd.Context(k=k, device=0)
    ctx.count_reads_device(seq_all, seq_all.numel(), 2)
    st = ctx.table_stats()
    assert st["bytes"] > (128 << 20) and ctx.table_layout() == 1, st
    rate = ctx.estimate_error_rate(0.95)
    ctx.set_run_params(rate, b"H")
    # mode 1 batch of n reads: first mates [0, n/2), second mates [n/2, n) -- cut out of the generator's halves
    half_all, half = cnt_reads // 2, n // 2
    rows = seq_all.view(cnt_reads, L + 1)
    qrows = qual_all.view(cnt_reads, L + 1)
    seq = torch.cat([rows[:half], rows[half_all:half_all + half]]).reshape(-1).contiguous()
    qual = torch.cat([qrows[:half], qrows[half_all:half_all + half]]).reshape(-1).contiguous()
    off = (torch.arange(n + 1, device=dev, dtype=torch.int64) * (L + 1)).to(torch.int32)
    work = seq.clone()
    res = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4)]
    ctx.profile(True)
    ctx.profile_reset()
    ctx.correct_device(1, n, work.numel(), L, work, qual, off, *res)
    ctx.sync()
    _, launches = ctx.profile_get(0)
    _, thr_launches = ctx.profile_get(1)
    ctx.profile(False)
    assert launches == 1 and thr_launches == 0, "the fused probe + threshold kernel of the locality order did not run"
    pairs = 10_000
    (ret, l, m, h), a1, a2 = _oracle_on_pairs(oracle, ctx, k, 4, rate, b"H", seq, qual, n, L, pairs)
    for got, want in zip(res, (ret, l, m, h)):
        g = got.cpu().numpy()
        assert np.array_equal(g[:pairs], want[:pairs]) and np.array_equal(g[half:half + pairs], want[pairs:])
    nb, b2 = pairs * (L + 1), half * (L + 1)
    assert np.array_equal(work[:nb].cpu().numpy(), a1) and np.array_equal(work[b2:b2 + nb].cpu().numpy(), a2)
    assert (ret > 0).sum() > 2000
    # batch-split property: 100 k other pairs as a batch of their own (small batch: arena-order probe kernel,
    # separate threshold kernel) give what they gave inside the large one
    lo, m2 = 100_000, 100_000
    s2 = torch.cat([rows[lo:lo + m2], rows[half_all + lo:half_all + lo + m2]]).reshape(-1).contiguous()
    q2 = torch.cat([qrows[lo:lo + m2], qrows[half_all + lo:half_all + lo + m2]]).reshape(-1).contiguous()
    o2 = (torch.arange(2 * m2 + 1, device=dev, dtype=torch.int64) * (L + 1)).to(torch.int32)
    r2 = [torch.zeros(2 * m2, dtype=torch.int32, device=dev) for _ in range(4)]
    ctx.correct_device(1, 2 * m2, s2.numel(), L, s2, q2, o2, *r2)
    ctx.sync()
    for a, b in zip(res, r2):
        assert torch.equal(a[lo:lo + m2], b[:m2]) and torch.equal(a[half + lo:half + lo + m2], b[m2:])
    wrows = work.view(n, L + 1)
    srows = s2.view(2 * m2, L + 1)
    assert torch.equal(wrows[lo:lo + m2], srows[:m2]) and torch.equal(wrows[half + lo:half + lo + m2], srows[m2:])
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["len151", "mixed"])
def test_default_dispatch_on_151_base_reads_and_on_a_mixed_length_batch(oracle, shape):
    """The fast path is not a property of the batch's longest read (the reference treats every read of up to 1 023 bases
    alike, utils.h:7).  `len151`: 151-base pairs at k = 23 have 129 k-mers -- the nine-register instances of the fused probe
    + threshold kernel and of k_single.  `mixed`: 90 % 150-base, 9 % 151-base, 1 % 250-base reads, mates drawn
    independently: the short units stay on the fused kernel / k_single / the compiled-for-k k_correct, the units with a
    250-base read take the list-driven probe kernel, the quarter-wave threshold kernel and k_correct<320> (rc_api_batch.hip:
    length tiers).  400 k paired reads over a table beyond 128 MB, no knob; the first 10 000 pairs against the oracle."""
    import torch
    import bench as B
    dev = torch.device("cuda", 0)
    mix = [(151, 1.0)] if shape == "len151" else [(150, 0.9), (151, 0.09), (250, 0.01)]
    n, L, k = 400_000, max(m[0] for m in mix), 23
    cnt_reads = 1_600_000
    seq_u, qual_u = B.synth_reads_gpu(777003, cnt_reads, L, 4400, 1500, 0.8, 0.005, dev, paired=True)
    seq_all, qual_all, off_all, _, lens_all = B.cut_reads(seq_u, qual_u, cnt_reads, L, mix, 777003, 0, dev)
    del seq_u, qual_u
    ctx = rcorrector_amd.Context(k=k, device=0)
    ctx.count_reads_device(seq_all, seq_all.numel(), 2)
    st = ctx.table_stats()
    assert st["bytes"] > (128 << 20) and ctx.table_layout() == 1, st
    rate = ctx.estimate_error_rate(0.95)
    ctx.set_run_params(rate, b"H")
    # mode 1 batch of n reads: first mates [0, n/2), second mates [n/2, n) -- cut out of the generator's halves
    half_all, half = cnt_reads // 2, n // 2
    o = off_all.long()

    def piece(lo, m):
        b0, b1 = int(o[lo].item()), int(o[lo + m].item())
        return seq_all[b0:b1], qual_all[b0:b1], (o[lo:lo + m + 1] - b0)
    s1, q1, o1 = piece(0, half)
    s2, q2, o2 = piece(half_all, half)
    seq = torch.cat([s1, s2]).contiguous()
    qual = torch.cat([q1, q2]).contiguous()
    off = torch.cat([o1, o2[1:] + o1[-1]]).to(torch.int32)
    work = seq.clone()
    res = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4)]
    ctx.profile(True)
    ctx.profile_reset()
    ctx.correct_device(1, n, work.numel(), L, work, qual, off, *res)
    ctx.sync()
    _, launches = ctx.profile_get(0)
    _, thr_launches = ctx.profile_get(1)
    _, single_launches = ctx.profile_get(3)
    ctx.profile(False)
    if shape == "len151":
        assert launches == 1 and thr_launches == 0 and single_launches == 1, "151-base reads fell off the fused probe + threshold kernel"
    else:   # fused kernel + the list-driven probe kernel of the long units; one threshold kernel for the 250-base tier
        assert launches == 2 and thr_launches == 1 and single_launches == 1, (launches, thr_launches, single_launches)
        assert int((lens_all[:half] > 160).sum().item()) > 1000
    pairs = 10_000
    codes, counts = ctx.table_export()
    T = oracle.Table(k, len(codes))
    T.put_many(codes, counts)
    P = oracle.make_params(k, 4, rate, b"H")
    b1, b2 = int(o1[pairs].item()), int(o2[pairs].item())
    a1, qa1 = s1[:b1].cpu().numpy().copy(), q1[:b1].cpu().numpy().copy()
    a2, qa2 = s2[:b2].cpu().numpy().copy(), q2[:b2].cpu().numpy().copy()
    ho1, ho2 = o1[:pairs + 1].cpu().numpy().astype(np.uint32), o2[:pairs + 1].cpu().numpy().astype(np.uint32)
    want = oracle.correct_batch(P, T, 1, a1, qa1, ho1, a2, qa2, ho2, threads=8)
    for got, w in zip(res, want):
        g = got.cpu().numpy()
        assert np.array_equal(g[:pairs], w[:pairs]) and np.array_equal(g[half:half + pairs], w[pairs:])
    base2 = int(o1[-1].item())
    assert np.array_equal(work[:b1].cpu().numpy(), a1) and np.array_equal(work[base2:base2 + b2].cpu().numpy(), a2)
    assert (want[0] > 0).sum() > 2000
    # a pure re-arrangement: the batch's longest read deciding for every read (RC_NO_TIER=1) gives the same
    os.environ["RC_NO_TIER"] = "1"
    try:
        ctx2 = rcorrector_amd.Context(k=k, device=0)
        ctx2.table_build(codes, counts)
        ctx2.set_run_params(rate, b"H")
        work2 = seq.clone()
        res2 = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4)]
        ctx2.correct_device(1, n, work2.numel(), L, work2, qual, off, *res2)
        ctx2.sync()
    finally:
        del os.environ["RC_NO_TIER"]
    for x, y in zip(res, res2):
        assert torch.equal(x, y)
    assert torch.equal(work, work2)
    ctx2.close()
    ctx.close()


@pytest.mark.gpu
def test_k31_maxcork8_on_a_packed_table_with_extension_bits(oracle):
    """BASELINE configs[4]'s code path at test size: k = 31, -maxcorK 8, 5 % substitutions over a table of
    >= 17 M entries, which the build lays out PACKED with remainder extension bits (ext > 0: the k = 31
    instance of k_correct and the EXT probe kernels).  10 000 reads against the oracle."""
    import torch
    import bench as B
    dev = torch.device("cuda", 0)
    n, L, k = 10_000, 150, 31
    seq, qual = B.synth_reads_gpu(777002, 200_000, L, 300, 1500, 0.8, 0.05, dev)
    ctx = rcorrector_amd.Context(k=k, max_fix_per_k=8, device=0)
    ctx.count_reads_device(seq, seq.numel(), 2)
    codes, counts = ctx.table_export()
    # pad the table with k-mers no read holds, so that it has the size (and so the layout) of a real one
    rng = np.random.Generator(np.random.PCG64(31))
    mask = np.uint64((1 << 62) - 1)
    fwd = rng.integers(0, 1 << 62, size=17_500_000, dtype=np.uint64) & mask
    pad = np.unique(np.minimum(fwd, _revcomp_codes(fwd, k)))
    pad = pad[~np.isin(pad, codes)]
    all_codes = np.concatenate([codes, pad])
    all_counts = np.concatenate([counts, rng.integers(2, 200, size=len(pad)).astype(np.int32)])
    ctx.table_build(all_codes, all_counts)
    st = ctx.table_stats()
    assert ctx.table_layout() == 1 and st["entries"] >= 17_000_000
    assert st["buckets"] < (1 << (2 * k - 32)), "the table is large enough to need no extension bits: not the path under test"
    rate = 0.01
    ctx.set_run_params(rate, b"H")
    nb = n * (L + 1)
    off = (torch.arange(n + 1, device=dev, dtype=torch.int64) * (L + 1)).to(torch.int32)
    work = seq[:nb].clone()
    res = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4)]
    ctx.correct_device(0, n, nb, L, work, qual[:nb].contiguous(), off, *res)
    ctx.sync()
    T = oracle.Table(k, len(all_codes))
    T.put_many(all_codes, all_counts)
    P = oracle.make_params(k, 8, rate, b"H")
    arena = seq[:nb].cpu().numpy().copy()
    qa = qual[:nb].cpu().numpy().copy()
    ho = (np.arange(n + 1, dtype=np.int64) * (L + 1)).astype(np.uint32)
    want = oracle.correct_batch(P, T, 0, arena, qa, ho, threads=8)
    for got, w in zip(res, want):
        assert np.array_equal(got.cpu().numpy(), w)
    assert np.array_equal(work.cpu().numpy(), arena)
    assert (want[0] > 0).sum() > 3000 and (want[0] < 0).sum() > 100
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [1000, 40000])
@pytest.mark.parametrize("name", ["se_k23", "pe_k23", "skew"])
def test_counts_in_the_tens_of_thousands_and_millions(oracle, name, scale):
    """The medians of the reads the threshold kernel and k_single finish come from a rank selection over the bits of the
    largest count; counts of 2^14 and more take k_single's sorting network instead (rc_single.h), and the descent of the
    threshold kernel gets longer (rc_quarter.h).  The test sets' counts are in the tens: the same sets with every count
    multiplied by 1 000 (tens of thousands) and by 40 000 (millions) -- different thresholds, same code paths end to end.
    The weak thresholds are then in the tens and hundreds: left searches take the inverse of GetBound from the table in
    device memory (rc_bs_lookup beyond RC_BS_INLINE), in k_single and in the alternative chains of k_correct."""
    d = datasets.make(name)
    counts = (np.asarray(d["counts"], dtype=np.int64) * scale).astype(np.int32)
    d = dict(d, counts=counts)
    want = datasets.run_oracle(oracle, d)
    ctx = rcorrector_amd.Context(k=d["k"], max_fix_per_k=d["mfk"], device=0)
    ctx.table_build(d["keys"], d["counts"])
    ctx.set_run_params(d["rate"], b"H")
    a, off = oracle.pack_reads(d["seqs1"])
    qa, _ = oracle.pack_reads(d["quals1"])
    if d["mode"] == 1:
        a2, off2 = oracle.pack_reads(d["seqs2"])
        qa2, _ = oracle.pack_reads(d["quals2"])
        got = ctx.correct_batch(1, a, qa, off, a2, qa2, off2) + (a, a2)
    else:
        got = ctx.correct_batch(d["mode"], a, qa, off) + (a,)
    for w, g, what in zip(want, got, ["ret", "l", "m", "h", "seq1", "seq2"]):
        assert np.array_equal(w, g), "%s differs on %s with counts x %d" % (what, name, scale)
    assert int(np.max(want[3])) >= (1 << 14)   # h: the largest count of some read is beyond the selection's limit
    ctx.close()


KNOBS = [{"RC_TABLE_LAYOUT": "wide"}, {"RC_TABLE_LOAD": "0.85"}, {"RC_TABLE_LOAD": "0.25"}, {"RC_LOCALITY": "force"},
         {"RC_NO_FUSE": "1", "RC_LOCALITY": "force"}, {"RC_K2_WAVE_PER_READ": "1"}, {"RC_NO_CLASSIFY": "1"},
         {"RC_NO_ALT": "1"}, {"RC_K3_GENERIC": "1"}, {"RC_K3_GENERIC": "1", "RC_NO_ALT": "1", "RC_LOCALITY": "force"},
         {"RC_TABLE_FILTER": "force", "RC_TABLE_FILTER_KIND": "plain"}, {"RC_TABLE_FILTER": "force", "RC_TABLE_FILTER_KIND": "core"},
         {"RC_TABLE_FILTER": "force", "RC_TABLE_FILTER_KIND": "plain", "RC_LOCALITY": "force", "RC_TABLE_LOAD": "0.85"},
         {"RC_TABLE_FILTER": "force", "RC_TABLE_FILTER_KIND": "core", "RC_LOCALITY": "force", "RC_TABLE_LOAD": "0.85"},
         {"RC_TABLE_FILTER": "force", "RC_TABLE_FILTER_KIND": "core", "RC_K3_GENERIC": "1", "RC_NO_ALT": "1"},
         {"RC_TABLE_FILTER": "search"}, {"RC_TABLE_FILTER": "search", "RC_K3_GENERIC": "1"}, {"RC_TABLE_FILTER": "search", "RC_TABLE_LOAD": "0.85", "RC_NO_SINGLE": "1"},
         {"RC_NO_SINGLE": "1"}, {"RC_NO_SINGLE": "1", "RC_NO_ALT": "1"}, {"RC_NO_BS_EXT": "1"}, {"RC_NO_TIER": "1"},
         {"RC_NO_TIER": "1", "RC_LOCALITY": "force"},
         # round 6: the fused probe kernel's switches (the tile's k-mer set, XCD-contiguous tiles, a quad of lanes per bucket) and
         # k_correct's list in locality order -- with the layouts / filters / loads that change what a bucket read finds
         {"RC_LOCALITY": "force", "RC_FUSED_DEDUP": "1"}, {"RC_LOCALITY": "force", "RC_FUSED_DEDUP": "1", "RC_FUSED_XCD": "1", "RC_TABLE_LOAD": "0.85"},
         {"RC_LOCALITY": "force", "RC_FUSED_XCD": "1", "RC_K3_LOCAL": "1"}, {"RC_LOCALITY": "force", "RC_PROBE_QUAD": "1"},
         {"RC_LOCALITY": "force", "RC_PROBE_QUAD": "1", "RC_TABLE_LOAD": "0.85", "RC_TABLE_FILTER": "force", "RC_TABLE_FILTER_KIND": "core"},
         {"RC_LOCALITY": "force", "RC_PROBE_QUAD": "1", "RC_TABLE_LAYOUT": "wide"},
         {"RC_LOCALITY": "force", "RC_FUSED_WAVE_TILES": "1"}, {"RC_LOCALITY": "force", "RC_FUSED_WAVE_TILES": "1", "RC_FUSED_XCD": "1", "RC_TABLE_LOAD": "0.85"},
         {"RC_LOCALITY": "force", "RC_FUSED_DEDUP": "1", "RC_TABLE_FILTER": "force", "RC_TABLE_FILTER_KIND": "plain", "RC_K3_LOCAL": "1"}]


@pytest.mark.gpu
@pytest.mark.parametrize("knobs", KNOBS, ids=lambda d: "+".join("%s=%s" % kv for kv in d.items()))
@pytest.mark.parametrize("name", ["se_k23", "pe_var", "k31_mc8", "tiers_pe"])
def test_every_alternative_code_path_gives_the_oracles_results(oracle, name, knobs, monkeypatch):
    """tools/knob_matrix.sh as a test: the library's alternative code paths (slot layout, load factor, list-driven and
    unfused probe kernels on small batches, wave-per-read threshold kernel, no classification, no alternative chains
    in the search, the any-k instance of k_correct) are each a pure re-arrangement of the same computation."""
    for kk, v in knobs.items():
        monkeypatch.setenv(kk, v)
    d = datasets.make(name)
    want = datasets.run_oracle(oracle, d)
    ctx = rcorrector_amd.Context(k=d["k"], max_fix_per_k=d["mfk"], device=0)
    ctx.table_build(d["keys"], d["counts"])
    ctx.set_run_params(d["rate"], b"H")
    a, off = oracle.pack_reads(d["seqs1"])
    qa, _ = oracle.pack_reads(d["quals1"])
    if d["mode"] == 1:
        a2, off2 = oracle.pack_reads(d["seqs2"])
        qa2, _ = oracle.pack_reads(d["quals2"])
        got = ctx.correct_batch(1, a, qa, off, a2, qa2, off2) + (a, a2)
    else:
        got = ctx.correct_batch(d["mode"], a, qa, off) + (a,)
    for w, g, what in zip(want, got, ["ret", "l", "m", "h", "seq1", "seq2"]):
        assert np.array_equal(w, g), "%s differs on %s under %s" % (what, name, knobs)
    ctx.close()
