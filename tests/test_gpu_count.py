"""GPU: stages 0-2 replacement (run_rcorrector.pl:262-281) -- the streaming k-mer counter, the
jellyfish-dump writer and `rcorrector` without -c.

  * counts over several arenas (host and device, ragged, with N) == exact numpy counts
  * the written dump holds the table's entries in dump order and loads back to the same table
  * ERROR_RATE from a counted table == ERROR_RATE the reference-pinned file-order path gives on
    the written dump (with > 100000 qualifying entries, so the order matters)
  * `rcorrector` without -c reproduces the reference's golden outputs for every fixture whose
    dump.jf is the exact count of its own reads
  * `rcorrector` without -c == the pinned CPU oracle CLI (and the reference binary when built)
    given the dump `-write-dump` wrote, on inputs whose ERROR_RATE is not the 0.01 default
"""
import os
import subprocess

import numpy as np
import pytest

import golden_util as gu
import synth


@pytest.fixture(scope="module")
def rc():
    import rcorrector_amd
    return rcorrector_amd

pytestmark = pytest.mark.gpu
CLI = os.path.join(gu.ROOT, "rcorrector_amd", "rcorrector")
M64 = (1 << 64) - 1


def dump_order_key(z):
    z = np.asarray(z, dtype=np.uint64).copy()   # rc_common.h: rc_dump_order_key (splitmix64 finaliser)
    z ^= z >> np.uint64(30)
    z *= np.uint64(0xbf58476d1ce4e5b9)
    z ^= z >> np.uint64(27)
    z *= np.uint64(0x94d049bb133111eb)
    z ^= z >> np.uint64(31)
    return z


def arena_of(rows):
    return b"".join(bytes(r) + b"\0" for r in rows)


def parse_dump(path, k):
    toks = open(path, "rb").read().split()
    cnt = np.array([int(t[1:]) for t in toks[0::2]], dtype=np.int64)
    kmers = toks[1::2]
    assert all(len(x) == k for x in kmers)
    a = np.frombuffer(b"".join(kmers), dtype=np.uint8).reshape(-1, k) if kmers else np.zeros((0, k), np.uint8)
    code = np.zeros(len(kmers), dtype=np.uint64)
    for j in range(k):
        code = (code << np.uint64(2)) | synth.CODE[a[:, j]].astype(np.uint64)
    return code, cnt


def sorted_pairs(codes, counts):
    o = np.argsort(codes, kind="stable")
    return np.asarray(codes)[o], np.asarray(counts)[o]


@pytest.mark.parametrize("k", [15, 23, 32])
def test_streaming_count_equals_exact_counts(rc, k):
    import torch
    s1, _, s2, _, lens = synth.make_reads(77 + k, 3000, 120, n_tx=8, l_tx=600, e=0.01, p_n=0.003, paired=True, var_len=True)
    rows = [s1[i, :lens[i]] for i in range(len(s1))] + [s2[i, :lens[i]] for i in range(len(s2))]
    want_k, want_c = synth.count_kmers([s1, s2], k, [lens, lens])
    ctx = rc.Context(k=k)
    ctx.count_begin()
    cuts = [0, 1, 1, 700, 2500, 2501, len(rows)]          # uneven pieces, one empty
    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        ar = arena_of(rows[a:b])
        if i % 2:
            t = torch.frombuffer(bytearray(ar), dtype=torch.uint8).cuda() if ar else torch.zeros(0, dtype=torch.uint8, device="cuda")
            ctx.count_add_device(t, len(ar))
            torch.cuda.synchronize()
        else:
            ctx.count_add(ar)
    n = ctx.count_finish(2)
    assert n == len(want_k)
    got_k, got_c = sorted_pairs(*ctx.table_export())
    assert np.array_equal(got_k, want_k) and np.array_equal(got_c, want_c)
    # min_count other than 2, and a second use of the same context
    ctx.count_begin()
    ctx.count_add(arena_of(rows))
    n5 = ctx.count_finish(5)
    keep = want_c >= 5
    got_k, got_c = sorted_pairs(*ctx.table_export())
    assert n5 == keep.sum() and np.array_equal(got_k, want_k[keep]) and np.array_equal(got_c, want_c[keep])


@pytest.mark.parametrize("mem_mb,retain_mb", [(1, None), (4, None), (None, 1), (-1, None)])
def test_count_in_bounded_memory_equals_exact_counts(rc, mem_mb, retain_mb, monkeypatch):
    """The counter works through the key space in passes sized by RC_COUNT_MEM_MB (what `jellyfish bc` is for in
    run_rcorrector.pl:262-273: singletons must not decide how much memory the counter takes).  With an artificial cap
    of 1 / 4 MB, 1.2 M k-mer occurrences at 5 % errors -- nearly all of them singletons -- go through 41 / 11 passes:
    the table must still be the exact counts >= 2.  And reads beyond what the counter may keep in HBM
    (RC_COUNT_RETAIN_MB) are refused with a message, not counted wrong."""
    k = 31
    s1, _, _, _, _ = synth.make_reads(4100, 10000, 150, n_tx=6, l_tx=900, e=0.05)
    want_k, want_c = synth.count_kmers([s1], k)
    if mem_mb == -1:   # the arrays of kept entries are sized from the first pass: here without slack, so every pass regrows them
        mem_mb = 1
        monkeypatch.setenv("RC_COUNT_TIGHT", "1")
    if mem_mb is not None:
        monkeypatch.setenv("RC_COUNT_MEM_MB", str(mem_mb))
    if retain_mb is not None:
        monkeypatch.setenv("RC_COUNT_RETAIN_MB", str(retain_mb))
    ctx = rc.Context(k=k)
    ctx.count_begin()
    rows = [s1[i] for i in range(len(s1))]
    if retain_mb is not None:
        with pytest.raises(rc.RcorrectorError, match="RC_COUNT_RETAIN_MB"):
            for lo in range(0, len(rows), 2500):
                ctx.count_add(arena_of(rows[lo:lo + 2500]))
        return
    for lo in range(0, len(rows), 2500):
        ctx.count_add(arena_of(rows[lo:lo + 2500]))
    n = ctx.count_finish(2)
    got_k, got_c = sorted_pairs(*ctx.table_export())
    assert n == len(want_k) and np.array_equal(got_k, want_k) and np.array_equal(got_c, want_c)
    assert (want_c >= 2).all() and len(want_k) > 5000


@pytest.mark.parametrize("n_ctx,mem_mb,staged", [(2, None, False), (3, 1, False), (3, 4, True), (8, 1, False), (2, -1, False)])
def test_sharded_count_over_several_contexts_equals_one_gpus_count(rc, n_ctx, mem_mb, staged, monkeypatch):
    """rc_table_count_finish_sharded: the reads are spread over n contexts (one per GPU in `rcorrector -gpus N`; here all on
    device 0), each scans its own arenas, the key space's slices are shared out, the owners sort and reduce, the entries are
    put end to end in slice order on the first context.  The table must be the exact counts >= 2 and the same table -- digest,
    ERROR_RATE estimate -- as one context counting all the reads, with one slice (more contexts than slices), with many
    (RC_COUNT_MEM_MB), with arrays that regrow at every slice (RC_COUNT_TIGHT) and with the copies between GPUs staged through
    the host; with rc_table_count_keep every context keeps the arenas it was given."""
    k = 31
    s1, _, _, _, _ = synth.make_reads(4200, 9000, 150, n_tx=6, l_tx=900, e=0.03)
    want_k, want_c = synth.count_kmers([s1], k)
    if mem_mb == -1:
        mem_mb = 1
        monkeypatch.setenv("RC_COUNT_TIGHT", "1")
    if mem_mb is not None:
        monkeypatch.setenv("RC_COUNT_MEM_MB", str(mem_mb))
    if staged:
        monkeypatch.setenv("RC_REPLICATE_STAGED", "1")
    rows = [s1[i] for i in range(len(s1))]
    pieces = [arena_of(rows[lo:lo + 1000]) for lo in range(0, len(rows), 1000)]
    one = rc.Context(k=k)
    one.count_begin()
    for a in pieces:
        one.count_add(a)
    n_one = one.count_finish(2)
    ctxs = [rc.Context(k=k) for _ in range(n_ctx)]
    ctxs[0].count_keep(True)
    with pytest.raises(rc.RcorrectorError, match="count_begin"):
        ctxs[0].count_finish_sharded(ctxs[1:])
    for c in ctxs:
        c.count_begin()
    given = [[] for _ in ctxs]
    for i, a in enumerate(pieces):   # dealt unevenly: the last context gets nothing when there are many
        g = (i * 3) % n_ctx if n_ctx < 8 else i % (n_ctx - 1)
        ctxs[g].count_add(a)
        given[g].append(len(a))
    n = ctxs[0].count_finish_sharded(ctxs[1:])
    got_k, got_c = sorted_pairs(*ctxs[0].table_export())
    assert n == n_one == len(want_k) and np.array_equal(got_k, want_k) and np.array_equal(got_c, want_c)
    assert ctxs[0].table_digest() == one.table_digest()
    assert ctxs[0].estimate_error_rate(0.95) == one.estimate_error_rate(0.95)
    for c, want in zip(ctxs, given):
        assert list(c.count_arenas()) == want
    for c in ctxs + [one]:
        c.close()


def test_count_sequence_errors(rc):
    ctx = rc.Context(k=23)
    with pytest.raises(rc.RcorrectorError):
        ctx.count_add(b"ACGT\0")
    with pytest.raises(rc.RcorrectorError):
        ctx.count_finish(2)
    ctx.count_begin()
    assert ctx.count_finish(2) == 0            # nothing added: an empty table
    assert ctx.lookup(np.array([5], dtype=np.uint64))[0] == 0


def test_write_jfdump_order_and_reload(rc, tmp_path):
    k = 23
    s1, _, _, _, _ = synth.make_reads(5, 4000, 100, n_tx=10, l_tx=800, e=0.01)
    want_k, want_c = synth.count_kmers([s1], k)
    ctx = rc.Context(k=k)
    ctx.count_begin()
    ctx.count_add(arena_of(s1))
    ctx.count_finish(2)
    path = str(tmp_path / "out.jf")
    assert ctx.write_jfdump(path) == len(want_k)
    code, cnt = parse_dump(path, k)
    keys = dump_order_key(code)
    assert np.all(keys[1:] > keys[:-1]), "entries must be in ascending rc_dump_order_key order"
    a, b = sorted_pairs(code, cnt)
    assert np.array_equal(a, want_k) and np.array_equal(b, want_c)
    ctx2 = rc.Context(k=k)
    assert ctx2.load_jfdump(path) == len(want_k)
    a2, b2 = sorted_pairs(*ctx2.table_export())
    assert np.array_equal(a2, want_k) and np.array_equal(b2, want_c)


def test_error_rate_of_counted_table_equals_file_order_path(rc, tmp_path):
    """> 100000 entries qualify (max of the 4 last-base variants >= 1000), so which 100000 are
    sampled depends on the scan order: the in-memory dump-order path must equal the file-order
    path (pinned to the reference by test_gpu_parity/test_golden_*) run on the written dump."""
    k = 25
    rng = np.random.Generator(np.random.PCG64(31))
    n = 160000
    base = rng.integers(0, 1 << 62, size=n, dtype=np.uint64) & np.uint64((1 << (2 * k)) - 1)
    base &= ~np.uint64(3)
    base = np.unique(base)
    codes, counts = [], []
    hi = rng.integers(1000, 6000, size=len(base))
    frac = rng.random(len(base)) * 0.05
    for v in range(4):
        codes.append(base | np.uint64(v))
        counts.append(hi if v == 0 else np.maximum(2, (hi * frac * rng.random(len(base))).astype(np.int64)))
    codes = np.concatenate(codes)
    counts = np.concatenate(counts).astype(np.int32)
    ctx = rc.Context(k=k)
    ctx.table_build(codes, counts)
    r_mem = ctx.estimate_error_rate(0.95)
    path = str(tmp_path / "t.jf")
    ctx.write_jfdump(path)
    ctx2 = rc.Context(k=k)
    ctx2.load_jfdump(path)
    r_file = ctx2.estimate_error_rate(0.95)
    assert r_mem == r_file and 0 < r_mem < 0.05 and r_mem != 0.01
    # and a different order gives a different sample (so the test can tell)
    code, cnt = parse_dump(path, k)
    synth.write_dump(str(tmp_path / "asc.jf"), *sorted_pairs(code, cnt), k)
    ctx3 = rc.Context(k=k)
    ctx3.load_jfdump(str(tmp_path / "asc.jf"))
    assert ctx3.estimate_error_rate(0.95) != r_file


EXACT_DUMP_FIXTURES = ["fx_sample", "fx_se_k23", "fx_pe_k23", "fx_il_k23", "fx_k31_mc8", "fx_skew", "fx_k32", "fx_k15", "fx_varlen_n"]


@pytest.mark.parametrize("resident", ["0", "1", "6"])
@pytest.mark.parametrize("name", EXACT_DUMP_FIXTURES)
def test_cli_without_c_reproduces_reference_outputs(name, resident, tmp_path, monkeypatch):
    """These fixtures' dump.jf is the exact k-mer count (>= 2) of the fixture's own reads, i.e. what
    stages 0-2 produce; counting on the GPU instead must give the reference's bytes -- in two passes over the files
    (RC_RESIDENT=0: count, then read again and correct) and in one (the default for plain files that fit: the text stays
    in host memory, the bases in HBM with the counter, rc_submit_resident; "6" = the same in batches of 6 reads)."""
    monkeypatch.setenv("RC_RESIDENT", resident)
    args = open(os.path.join(gu.GOLDEN, name, "cmd.txt")).read().split()
    i = args.index("-c")
    del args[i:i + 2]
    p = gu.run_fixture(CLI, name, tmp_path, args_override=args)
    gu.assert_same_as_reference(name, tmp_path, p.stderr)


@pytest.mark.parametrize("env", [{"RC_HBM_FREE_MB": "1"}, {"RC_NO_READ_AHEAD": "1"}, {"RC_HBM_FREE_MB": "1", "RC_NO_READ_AHEAD": "1"}, {"RC_NUMA": "0"},
                                 {"RC_NUMA": "0", "RC_HBM_FREE_MB": "1"}])
@pytest.mark.parametrize("name", ["fx_pe_k23", "fx_se_k23", "fx_il_k23", "fx_k31_mc8"])
def test_cli_reader_that_runs_ahead_of_the_gpu_runtime(name, env, tmp_path, monkeypatch):
    """Without -c the reader of a one-pass run starts before the contexts exist (the host's half of the one-pass test has
    passed; HIP is still coming up).  When the device then has too little memory free (RC_HBM_FREE_MB: as if this much were),
    what it read is dropped, the sources are rewound and the run takes two passes; RC_NO_READ_AHEAD keeps the reader
    behind the contexts; RC_NUMA=0: no NUMA node to find first, so the reader always goes ahead.  The reference's bytes and
    messages in every case, and the timing lines say which way the run went."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    args = open(os.path.join(gu.GOLDEN, name, "cmd.txt")).read().split()
    i = args.index("-c")
    del args[i:i + 2]
    p = gu.run_fixture(CLI, name, tmp_path, args_override=args)
    gu.assert_same_as_reference(name, tmp_path, p.stderr)
    monkeypatch.setenv("RC_TIMING", "1")
    again = tmp_path / "again"
    again.mkdir()
    q = gu.run_fixture(CLI, name, again, args_override=args)
    gu.assert_same_as_reference(name, again, b"", check_stderr=False)
    went_ahead, started_over = b"the reader starts before" in q.stderr, b"started over" in q.stderr
    if "RC_NO_READ_AHEAD" in env:
        assert not went_ahead
    if "RC_NUMA" in env and "RC_NO_READ_AHEAD" not in env:
        assert went_ahead
    assert started_over == (went_ahead and "RC_HBM_FREE_MB" in env)
    assert (b"(one pass:" in q.stderr) == ("RC_HBM_FREE_MB" not in env)


@pytest.mark.parametrize("how", ["resident", "packed"])
@pytest.mark.parametrize("name", ["fx_pe_k23", "fx_se_k23", "fx_k31_mc8"])
def test_cli_fix_list_overflow_falls_back_to_the_byte_path(name, how, tmp_path, monkeypatch):
    """A batch with more substitutions than its fix list has room for (RC_FIX_CAP=1 here; in the field a heavily corrected
    tail batch under -maxcorK) comes back from rc_wait_resident / rc_wait_packed with RC_STATUS_NOSPACE and is run again
    through rc_submit: same bytes as ever."""
    monkeypatch.setenv("RC_FIX_CAP", "1")
    args = open(os.path.join(gu.GOLDEN, name, "cmd.txt")).read().split()
    if how == "resident":
        monkeypatch.setenv("RC_RESIDENT", "40")
        i = args.index("-c")
        del args[i:i + 2]
    else:
        args += ["-packed", "-batch", "40"]
    p = gu.run_fixture(CLI, name, tmp_path, args_override=args)
    gu.assert_same_as_reference(name, tmp_path, p.stderr)


@pytest.mark.parametrize("gpus,inflight", [(2, 1), (3, 2), (8, 2)])
@pytest.mark.parametrize("name", ["fx_pe_k23", "fx_se_k23", "fx_il_k23", "fx_k31_mc8"])
def test_cli_without_c_on_several_gpus_reads_the_files_once(name, gpus, inflight, tmp_path, monkeypatch):
    """`-gpus N` without -c in ONE pass: the batches are dealt round-robin as the files are read -- every arena goes to GPU 0,
    which counts all of them, and to the GPU that will correct it, which only keeps it (rc_table_count_park) -- the table is
    replicated, and each batch is corrected by the context that holds its bases (a worker only takes its own GPU's batches).
    The reference's bytes (these fixtures' dumps are the exact counts of their reads).  RC_SHARED_GPU=1: every "GPU" is
    device 0; batches of 40 reads, so every context gets several."""
    monkeypatch.setenv("RC_SHARED_GPU", "1")
    monkeypatch.setenv("RC_RESIDENT", "40")
    if gpus == 3:   # the older way: every arena to GPU 0 as well, which counts alone (the others rc_table_count_park)
        monkeypatch.setenv("RC_COUNT_SHARDED", "0")
    args = open(os.path.join(gu.GOLDEN, name, "cmd.txt")).read().split()
    i = args.index("-c")
    del args[i:i + 2]
    p = gu.run_fixture(CLI, name, tmp_path, args_override=args + ["-gpus", str(gpus), "-inflight", str(inflight)])
    gu.assert_same_as_reference(name, tmp_path, p.stderr)


def test_cli_without_c_several_inputs_of_every_kind_one_pass_equals_two(tmp_path):
    """-r FASTQ, -p FASTQ pair, -i interleaved, -r FASTA and a file with format quirks in ONE run without -c: the k-mers of
    all of them are counted into one table; the one-pass path (batches kept per file in input order, arenas numbered across
    files, quality bits only for FASTQ, a batch with an empty quality line falling back to the bytes) must write what the
    two-pass path writes, whatever the batch size."""
    import shutil
    d = str(tmp_path)
    g = gu.GOLDEN
    names = {"se.fq": ("fx_se_k23", "reads.fq"), "p_1.fq": ("fx_pe_k23", "reads_1.fq"), "p_2.fq": ("fx_pe_k23", "reads_2.fq"),
             "il.fq": ("fx_il_k23", "reads_il.fq"), "fa.fa": ("fa_se_k23", "reads.fa"), "q.fq": ("fx_io_quirks", "reads.fq")}
    for dst, (fx, src) in names.items():
        src_path = os.path.join(g, fx, src)
        if not os.path.exists(src_path):   # (fixture file names differ: take the fixture's first input of that kind)
            cand = sorted(f for f in os.listdir(os.path.join(g, fx)) if f.endswith(os.path.splitext(dst)[1]) and not f.startswith("dump"))
            src_path = os.path.join(g, fx, cand[0])
        shutil.copy(src_path, os.path.join(d, dst))
    args = ["-r", "se.fq", "-p", "p_1.fq", "p_2.fq", "-i", "il.fq", "-r", "fa.fa", "-r", "q.fq", "-k", "23"]
    res = []
    for tag, env in (("two", "0"), ("one", "1"), ("one_small", "14")):
        err, out = _run_env(CLI, args, d, os.path.join(d, tag), {"RC_RESIDENT": env})
        res.append((err, out))
    assert len(res[0][1]) == 6 and all(len(v) > 0 for v in res[0][1].values())
    assert res[1] == res[0] and res[2] == res[0]


def test_cli_without_c_gz_inputs_one_pass_equals_two(tmp_path):
    """.gz inputs without -c: one pass inflates each file once -- whole, in memory, with libdeflate where the system has it
    (RC_LIBDEFLATE=0: zlib streams) -- two passes inflate it twice with zlib; a file of several gzip members (what this
    program writes itself, and what `cat a.gz b.gz` makes) must read as their concatenation, and a BGZF file (bgzip: 64 KB
    members that carry their length, which are inflated in parallel) as what gunzip makes of it.  The outputs are .gz files of
    members deflated in parallel: compared after decompression, as everywhere."""
    import gzip
    import shutil
    d = str(tmp_path)
    g = gu.GOLDEN
    for dst, (fx, src) in {"p_1.fq": ("fx_pe_k23", "reads_1.fq"), "p_2.fq": ("fx_pe_k23", "reads_2.fq"), "se.fq": ("fx_se_k23", "reads.fq")}.items():
        data = open(os.path.join(g, fx, src), "rb").read()
        if dst == "se.fq":   # three members, cut at record boundaries
            recs = data.split(b"\n@")
            cut1, cut2 = len(recs) // 3, 2 * len(recs) // 3
            parts = [b"\n@".join(recs[:cut1]) + b"\n", b"@" + b"\n@".join(recs[cut1:cut2]) + b"\n", b"@" + b"\n@".join(recs[cut2:])]
            assert b"".join(parts) == data
            with open(os.path.join(d, dst + ".gz"), "wb") as f:
                for part in parts:
                    f.write(gzip.compress(part, 1))
        elif dst == "p_1.fq":   # BGZF (bgzip): blocks of at most 64 KB that carry their own length, inflated side by side
            import struct
            import zlib
            with open(os.path.join(d, dst + ".gz"), "wb") as f:
                for lo in list(range(0, len(data), 60000)) + [len(data)]:   # (the last one: the empty end-of-file block)
                    chunk = data[lo:lo + 60000] if lo < len(data) else b""
                    co = zlib.compressobj(6, zlib.DEFLATED, -15)
                    raw = co.compress(chunk) + co.flush()
                    bsize = 18 + len(raw) + 8
                    f.write(b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1))
                    f.write(raw + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
            assert gzip.decompress(open(os.path.join(d, dst + ".gz"), "rb").read()) == data
        else:
            with open(os.path.join(d, dst + ".gz"), "wb") as f:
                f.write(gzip.compress(data, 6))
    args = ["-p", "p_1.fq.gz", "p_2.fq.gz", "-r", "se.fq.gz", "-k", "23"]
    res = []
    for tag, env in (("two", {"RC_RESIDENT": "0"}), ("one", {"RC_RESIDENT": "1"}), ("one_zlib", {"RC_RESIDENT": "1", "RC_LIBDEFLATE": "0"}),
                     ("one_small", {"RC_RESIDENT": "10"})):
        err, out = _run_env(CLI, args, d, os.path.join(d, tag), env)
        res.append((err, {f: gzip.decompress(v) for f, v in out.items()}))
    assert sorted(res[0][1]) == ["p_1.cor.fq.gz", "p_2.cor.fq.gz", "se.cor.fq.gz"] and all(len(v) > 0 for v in res[0][1].values())
    for r in res[1:]:
        assert r == res[0]


def _run_env(binary, args, cwd, od, env):
    os.makedirs(od)
    p = subprocess.run([binary] + args + ["-od", od], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
    assert p.returncode == 0, p.stderr.decode()
    return p.stderr, {f: open(os.path.join(od, f), "rb").read() for f in sorted(os.listdir(od))}


@pytest.mark.parametrize("name,flags", [("fx_se_k23", ["-s", "reads.fq"]), ("fx_pe_k23", ["-1", "reads_1.fq", "-2", "reads_2.fq"]),
                                        ("fx_il_k23", None), ("fx_k31_mc8", None)])
def test_wrapper_with_run_rcorrector_pl_flags(name, flags, tmp_path):
    """tools/run_rcorrector_gpu takes run_rcorrector.pl's command line (-s / -1 / -2 / -i lists, -k, -od, -tmpd, -stage
    ...): a run from stage 0 counts the k-mers on the GPU and corrects -- the reference's bytes for fixtures whose dump
    is the exact count of their own reads -- and leaves nothing behind in -tmpd; `-stage 3` restarts from a dump of
    the name the Perl wrapper derives (tmp_<md5 of the file list>.jf_dump) and removes it afterwards as the Perl
    wrapper does."""
    import hashlib
    import shutil
    import sys
    d = os.path.join(gu.GOLDEN, name)
    args = open(os.path.join(d, "cmd.txt")).read().split()
    if flags is None:   # translate the fixture's -r / -i / -p command line
        flags = []
        i = 0
        while i < len(args):
            if args[i] == "-r":
                flags += ["-s", args[i + 1]]; i += 2
            elif args[i] == "-i":
                flags += ["-i", args[i + 1]]; i += 2
            elif args[i] == "-p":
                flags += ["-1", args[i + 1], "-2", args[i + 2]]; i += 3
            else:
                i += 1
    rest = []
    i = 0
    while i < len(args):   # everything but the inputs and -c
        if args[i] in ("-r", "-i", "-c"):
            i += 2
        elif args[i] == "-p":
            i += 3
        else:
            rest.append(args[i]); i += 1
    wrapper = os.path.join(gu.ROOT, "tools", "run_rcorrector_gpu")
    tmpd = tmp_path / "tmp"
    tmpd.mkdir()
    od = tmp_path / "o0"
    p = subprocess.run([sys.executable, wrapper] + flags + rest + ["-od", str(od), "-tmpd", str(tmpd), "-ek", "1000"], cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    gu.assert_same_as_reference(name, od, b"", check_stderr=False)
    want_err = open(os.path.join(d, "ref", "stderr.txt"), "rb").read()
    assert p.stderr.endswith(want_err)          # the wrapper's own two lines come first, as the Perl wrapper's do
    assert os.listdir(tmpd) == []
    # -stage 3: from the dump the earlier stages would have left
    files = [f for f in flags if not f.startswith("-")]
    names = [x for f in files for x in f.split(",")]
    crc = hashlib.md5("".join(n + " " for n in names).encode()).hexdigest()
    shutil.copy(os.path.join(d, "dump.jf"), tmpd / ("tmp_%s.jf_dump" % crc))
    od3 = tmp_path / "o3"
    p = subprocess.run([sys.executable, wrapper] + flags + rest + ["-od", str(od3), "-tmpd", str(tmpd), "-stage", "3"], cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    gu.assert_same_as_reference(name, od3, b"", check_stderr=False)
    assert p.stderr.endswith(want_err) and os.listdir(tmpd) == []
    # the Perl wrapper's own refusals
    p = subprocess.run([sys.executable, wrapper, "-s", "reads.fq", "-k", "33"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"can not be greater than 32" in p.stderr


def _run(binary, args, cwd, od, more=()):
    os.makedirs(od)
    p = subprocess.run([binary] + args + ["-od", od] + list(more), cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    return p.stderr, {f: open(os.path.join(od, f), "rb").read() for f in sorted(os.listdir(od))}


@pytest.mark.parametrize("paired", [False, True])
def test_cli_without_c_equals_cpu_given_the_written_dump(oracle, paired, tmp_path):
    """High coverage of one short transcript: > 100 entries reach count 1000, so ERROR_RATE is
    estimated rather than defaulted -- the whole chain count -> dump order -> ERROR_RATE ->
    correction must agree with the CPU implementations reading our dump."""
    d = str(tmp_path)
    s1, q1, s2, q2, _ = synth.make_reads(4242, 30000, 100, n_tx=1, l_tx=400, alpha=0.0, e=0.004, paired=paired)
    if paired:
        synth.write_fastq(os.path.join(d, "a_1.fq"), s1, q1, "/1")
        synth.write_fastq(os.path.join(d, "a_2.fq"), s2, q2, "/2")
        inp = ["-p", "a_1.fq", "a_2.fq"]
    else:
        synth.write_fastq(os.path.join(d, "a.fq"), s1, q1)
        inp = ["-r", "a.fq"]
    dump = os.path.join(d, "gpu.jf")
    err_g, out_g = _run(CLI, inp + ["-k", "23", "-write-dump", dump, "-batch", "4096"], d, os.path.join(d, "g"))
    assert b"Weak kmer threshold rate: 0.01000" not in err_g
    code, cnt = parse_dump(dump, 23)
    wk, wc = synth.count_kmers([s1, s2], 23)
    a, b = sorted_pairs(code, cnt)
    assert np.array_equal(a, wk) and np.array_equal(b, wc)
    binaries = [oracle.CLI_BIN] + ([oracle.REF_BIN] if os.path.exists(oracle.REF_BIN) else [])
    for j, binary in enumerate(binaries):
        err_c, out_c = _run(binary, inp + ["-k", "23", "-c", dump, "-t", "8"], d, os.path.join(d, "c%d" % j))
        assert out_c.keys() == out_g.keys() and out_c
        for f in out_c:
            assert out_g[f] == out_c[f], "%s differs from %s" % (f, binary)
        assert err_g == err_c


def test_simulated_reads_end_to_end_with_accuracy_score(tmp_path):
    """Whole pipeline on simulated reads with truth headers: count on the GPU, correct, score with
    `verify` -- the corrected file and the scorer's report must equal what the reference binaries
    produced for the same reads (tests/golden/verify/)."""
    g = os.path.join(gu.GOLDEN, "verify")
    p = subprocess.run([CLI, "-r", os.path.join(g, "raw.fq"), "-k", "23", "-od", str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    got = open(tmp_path / "raw.cor.fq", "rb").read()
    assert got == open(os.path.join(g, "cor.fq"), "rb").read()
    subprocess.run(["make", "-C", os.path.join(gu.ROOT, "rcorrector_amd", "csrc"), "../verify"], check=True, stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(gu.ROOT, "rcorrector_amd", "verify"), str(tmp_path / "raw.cor.fq")], stdout=subprocess.PIPE, check=True).stdout
    assert out == open(os.path.join(g, "cor.plain.txt"), "rb").read()


def test_write_jfdump_of_an_empty_table(rc, tmp_path):
    ctx = rc.Context(k=23)
    ctx.count_begin()
    ctx.count_add(b"ACGTACGT\0")            # shorter than k: no k-mer at all
    assert ctx.count_finish(2) == 0
    path = str(tmp_path / "empty.jf")
    assert ctx.write_jfdump(path) == 0 and os.path.getsize(path) == 0
    ctx2 = rc.Context(k=23)
    assert ctx2.load_jfdump(path) == 0
    assert ctx2.estimate_error_rate(0.95) == 0.01   # the reference's fallback (main.cpp:355-356)
