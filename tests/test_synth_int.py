"""The bench's read generator (tools/synth_int.py) is a pure function of (seed, read index): the same
bytes whatever the chunking, the starting unit or the device -- the property the replicated table of a
multi-GPU bench run rests on."""
import numpy as np
import pytest
import torch

import synth_int


def _gen(dev, **kw):
    return synth_int.Synth(1002, 150, n_tx=300, l_tx=1500, paired=True, device=dev, **kw)


def test_chunking_and_offset_do_not_change_the_bytes():
    s = _gen("cpu")
    a_seq, a_q = s.generate(0, 1000, chunk=1 << 17)
    b_seq, b_q = s.generate(0, 1000, chunk=37)
    assert torch.equal(a_seq, b_seq) and torch.equal(a_q, b_q)
    # units [400, 600) generated on their own = rows 400..599 of both mate blocks
    c_seq, _ = s.generate(400, 200)
    L1 = 151
    a = a_seq.reshape(2000, L1)
    c = c_seq.reshape(400, L1)
    assert torch.equal(c[:200], a[400:600]) and torch.equal(c[200:], a[1400:1600])
    # a second generator object: same bytes
    d_seq, _ = _gen("cpu").generate(0, 1000)
    assert torch.equal(d_seq, a_seq)


def test_statistics_are_what_the_workload_says():
    s = synth_int.Synth(1001, 100, n_tx=300, l_tx=1500, err=0.01, device="cpu")
    seq, q = s.generate(0, 20000)
    seq = seq.reshape(20000, 101).numpy()
    q = q.reshape(20000, 101).numpy()
    assert (seq[:, 100] == 0).all() and set(np.unique(seq[:, :100])) == set(b"ACGT")
    frac = (q[:, :100] == ord('#')).mean()
    assert 0.008 < frac < 0.012
    # mate 2 is the reverse complement of the fragment's tail: with err = 0 a 150-base pair from a
    # 150-base fragment is its own reverse complement
    p = synth_int.Synth(7, 150, n_tx=50, l_tx=1500, err=0.0, paired=True, frag_len=150, device="cpu")
    ps, _ = p.generate(0, 100)
    ps = ps.reshape(200, 151).numpy()[:, :150]
    comp = np.zeros(256, np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    assert np.array_equal(comp[ps[:100, ::-1]], ps[100:])


@pytest.mark.gpu
def test_gpu_bytes_equal_cpu_bytes():
    for kw in (dict(), dict(bias3=True, alpha=1.5)):
        a, aq = _gen("cpu", **kw).generate(5000, 3000)
        b, bq = _gen("cuda:0", **kw).generate(5000, 3000)
        assert torch.equal(a, b.cpu()) and torch.equal(aq, bq.cpu())


def test_committed_counter_passes_are_tied_to_the_kernel_sources(monkeypatch):
    """roofline.traffic / served_by and the k_correct counter figures are measured live by bench.py (rocprofv3 --pmc sub-runs);
    where that cannot run, the committed summary (profiles/r6_traffic.json, written by the same code: tools/measure_r6.sh)
    stands in -- it carries the git blob hashes of the kernel sources it measured: bench.py reports it only while the sources
    are the ones measured (a changed kernel nulls the figure with a note instead of leaving a stale one), and never for a
    workload that is no preset."""
    import json
    import os
    import types
    import bench
    monkeypatch.setenv("RC_BENCH_PMC", "committed")
    if not os.path.exists(bench.TRAFFIC_JSON):
        import pytest
        pytest.skip("no committed counter summary yet for this round (tools/measure_r6.sh writes it)")
    doc = json.load(open(bench.TRAFFIC_JSON))
    assert set(doc["sources"]) == set(bench.KERNEL_SOURCES)
    fresh = doc["sources"] == bench.source_hashes()
    assert doc["configs"]
    for c in sorted(doc["configs"]):
        rec = doc["configs"][c]
        assert rec["k_probe"]["fetch_size_kb"] * 1024 * 2 > 1e9 and rec["k_probe"]["tcc_hit"] > 0 and rec["k_correct"]["insts_valu"] > 1e9
        a = types.SimpleNamespace(reads=None, len=None, k=None, err=None, alpha=None, seed=None, maxcork=None, paired=None, config=int(c))
        got, note = bench.counter_pass(a)
        if fresh:
            assert got == rec and "FETCH_SIZE" in note and "replayed" in note
        else:
            assert got is None and "stale" in note
    a = types.SimpleNamespace(reads=1000, len=None, k=None, err=None, alpha=None, seed=None, maxcork=None, paired=None, config=2)
    assert bench.counter_pass(a)[0] is None
    # a kernel edit makes the committed figures stale
    real = bench.source_hashes()
    monkeypatch.setattr(bench, "source_hashes", lambda: dict(real, **{"rc_device.h": "0" * 40}))
    a = types.SimpleNamespace(reads=None, len=None, k=None, err=None, alpha=None, seed=None, maxcork=None, paired=None, config=2)
    got, note = bench.counter_pass(a)
    assert got is None and "rc_device.h" in note
