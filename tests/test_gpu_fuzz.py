"""GPU fuzz: random small inputs (ragged lengths incl. < k, N and IUPAC letters, poly-A/T runs, random
qualities, tiny and empty tables, single/paired/interleaved, several k) through the drop-in
`rcorrector` binary must equal the pinned CPU oracle's CLI byte for byte -- every output file and
the stderr parameter lines."""
import os
import subprocess

import numpy as np
import pytest

import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "rcorrector_amd", "rcorrector")


def _content(path):
    """file bytes; .gz outputs are compared after decompression (the drop-in binary writes parallel
    gzip members, the reference one stream -- Reads.h:140-147 -- with identical content)"""
    data = open(path, "rb").read()
    if path.endswith(".gz"):
        import gzip
        return gzip.decompress(data)
    return data


def _random_case(seed, d, max_len=160):
    rng = np.random.Generator(np.random.PCG64(seed))
    k = int(rng.choice([15, 19, 23, 27, 31, 32]))
    mode = int(rng.integers(0, 3))
    n = int(rng.integers(1, 400))
    L = int(rng.integers(k + 5, max_len))
    e = float(rng.choice([0.0, 0.005, 0.02, 0.06]))
    n_tx = int(rng.integers(1, 12))
    s1, q1, s2, q2, _ = synth.make_reads(seed, n, L, n_tx=n_tx, l_tx=max(400, 2 * L + 10), e=e, paired=mode != 0,
                                         frag_len=min(max(400, 2 * L + 10), 2 * L))
    keys, cnt = synth.count_kmers([s1, s2], k)
    if rng.random() < 0.15:
        keys, cnt = keys[:0], cnt[:0]            # empty table
    elif rng.random() < 0.3:
        keep = rng.random(len(keys)) < 0.5       # half the k-mers missing
        keys, cnt = keys[keep], cnt[keep]
    order = rng.permutation(len(keys)) if rng.random() < 0.5 else None
    synth.write_dump(os.path.join(d, "dump.jf"), keys, cnt, k, order)

    def mangle(s, q):
        reads, quals = [], []
        for i in range(len(s)):
            r = bytearray(s[i].tobytes())
            ln = int(rng.integers(1, L + 1)) if rng.random() < 0.3 else L
            r = r[:ln]
            u = rng.random()
            if u < 0.1 and ln > 8:
                for p in rng.choice(ln, int(rng.integers(1, 8)), replace=False):
                    r[p] = ord('N')
            elif u < 0.15 and ln > 4:
                for p in rng.choice(ln, 2, replace=False):
                    r[p] = int(rng.choice(list(b"RYKMSWBDHV")))
            elif u < 0.22 and ln > 20:
                t = int(rng.integers(5, ln))
                r[ln - t:] = (b"A" if rng.random() < 0.5 else b"T") * t
            qq = bytes(rng.integers(33, 75, size=ln, dtype=np.uint8).tolist())
            reads.append(bytes(r))
            quals.append(qq)
        return reads, quals
    r1, qq1 = mangle(s1, q1)
    if mode == 0:
        with open(os.path.join(d, "a.fq"), "wb") as f:
            for i, (r, q) in enumerate(zip(r1, qq1)):
                f.write(b"@x%d some comment\n%s\n+\n%s\n" % (i, r, q))
        args = ["-r", "a.fq"]
    else:
        r2, qq2 = mangle(s2, q2)
        if mode == 1:
            for name, rr, qq in (("a_1.fq", r1, qq1), ("a_2.fq", r2, qq2)):
                with open(os.path.join(d, name), "wb") as f:
                    for i, (r, q) in enumerate(zip(rr, qq)):
                        f.write(b"@x%d\n%s\n+x%d\n%s\n" % (i, r, i, q))
            args = ["-p", "a_1.fq", "a_2.fq"]
        else:
            with open(os.path.join(d, "a_il.fq"), "wb") as f:
                for i in range(len(r1)):
                    f.write(b"@x%d/1\n%s\n+\n%s\n@x%d/2\n%s\n+\n%s\n" % (i, r1[i], qq1[i], i, r2[i], qq2[i]))
            args = ["-i", "a_il.fq"]
    extra = []
    if rng.random() < 0.3:
        extra += ["-maxcorK", str(int(rng.integers(2, 9)))]
    if rng.random() < 0.3:
        extra += ["-wk", "0.8"]
    return args + ["-k", str(k), "-c", "dump.jf"] + extra


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(900, 940)))
def test_cli_equals_oracle_cli_on_random_inputs(oracle, seed, tmp_path):
    d = str(tmp_path)
    args = _random_case(seed, d)
    outs = {}
    verbose = ["-verbose"] if seed % 3 == 0 else []   # every third case also compares the -verbose transcript
    for name, binary, more in (("gpu", CLI, ["-batch", "64"] if seed % 2 else []), ("cpu", oracle.CLI_BIN, ["-t", "2"])):
        od = os.path.join(d, name)
        os.makedirs(od)
        p = subprocess.run([binary] + args + ["-od", od] + more + verbose, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()
        outs[name] = (p.stderr, {f: _content(os.path.join(od, f)) for f in sorted(os.listdir(od))}, p.stdout)
    assert outs["gpu"][1].keys() == outs["cpu"][1].keys() and outs["gpu"][1]
    for f in outs["cpu"][1]:
        assert outs["gpu"][1][f] == outs["cpu"][1][f], "%s differs (seed %d, args %s)" % (f, seed, args)
    assert outs["gpu"][0] == outs["cpu"][0], "stderr differs (seed %d)" % seed
    assert outs["gpu"][2] == outs["cpu"][2], "-verbose transcript differs (seed %d)" % seed
    # the same through the packed boundary (rc_submit_packed: N-rich reads, IUPAC letters, ragged pairs, empty tables)
    od = os.path.join(d, "packed")
    os.makedirs(od)
    p = subprocess.run([CLI] + args + ["-od", od, "-packed"] + (["-batch", "32"] if seed % 2 else []), cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    assert p.stderr == outs["cpu"][0], "stderr differs through the packed boundary (seed %d)" % seed
    for f in outs["cpu"][1]:
        assert _content(os.path.join(od, f)) == outs["cpu"][1][f], "%s differs through the packed boundary (seed %d, args %s)" % (f, seed, args)


def _one_pass_equals_two(args, d, seed):
    """without -c the k-mers are counted here: one pass over the files (the default for plain inputs: text kept in host
    memory, bases kept in HBM, rc_submit_resident) must write what the two passes write (count, read again, correct)"""
    a = list(args)
    i = a.index("-c")
    del a[i:i + 2]
    res = []
    for tag, env in (("two", {"RC_RESIDENT": "0"}), ("one", {"RC_RESIDENT": "1"}), ("one_small", {"RC_RESIDENT": "1" if seed % 2 else "10"})):
        od = os.path.join(d, "nc_" + tag)
        os.makedirs(od)
        p = subprocess.run([CLI] + a + ["-od", od], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        res.append((p.returncode, p.stderr, {f: _content(os.path.join(od, f)) for f in sorted(os.listdir(od))}))
    assert res[0][0] == 0, res[0][1].decode()
    for r, tag in zip(res[1:], ("one pass", "one pass, small batches")):
        assert r == res[0], "%s differs from two passes (seed %d, args %s)" % (tag, seed, a)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(900, 940, 2)) + list(range(300, 340, 2)))
def test_cli_one_pass_without_c_equals_two_passes(seed, tmp_path):
    import io_quirks
    d = str(tmp_path)
    args = _random_case(seed, d) if seed >= 900 else io_quirks.make_case(seed, d, modes=(0, 1, 2))
    _one_pass_equals_two(args, d, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(300, 340)))
def test_cli_equals_oracle_cli_on_io_quirks(oracle, seed, tmp_path):
    """Input quirks (tests/io_quirks.py: empty / one-base / k-long reads, quality lines of the wrong
    length or empty, '+' lines with text, kilobyte headers, no final newline): the drop-in binary
    against the oracle CLI, which tests/test_oracle_vs_ref.py pins to the reference on the same cases."""
    import io_quirks
    d = str(tmp_path)
    args = io_quirks.make_case(seed, d, modes=(0, 1, 2))
    res = []
    for name, binary, more in (("gpu", CLI, ["-batch", "32"] if seed % 2 else []), ("cpu", oracle.CLI_BIN, [])):
        od = os.path.join(d, name)
        os.makedirs(od)
        p = subprocess.run([binary] + args + ["-od", od, "-verbose"] + more, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        res.append((p.returncode, p.stderr, {f: _content(os.path.join(od, f)) for f in sorted(os.listdir(od))}, p.stdout))
    assert res[0][0] == 0
    for j, what in enumerate(("exit status", "stderr", "output files", "-verbose transcript")):
        assert res[0][j] == res[1][j], "%s differs (seed %d)" % (what, seed)
    # through the packed boundary (a batch with an empty quality line falls back to the bytes: one bit per quality cannot
    # say "no quality string")
    od = os.path.join(d, "packed")
    os.makedirs(od)
    p = subprocess.run([CLI] + args + ["-od", od, "-packed"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert (p.returncode, p.stderr) == (res[1][0], res[1][1])
    assert {f: _content(os.path.join(od, f)) for f in sorted(os.listdir(od))} == res[1][2], "output differs through the packed boundary (seed %d)" % seed
