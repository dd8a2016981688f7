"""CPU: the C-ABI library loads without a GPU, exports every function include/rcorrector_amd.h
declares, and refuses to run without a device (no CPU fallback)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import rcorrector_amd
    rcorrector_amd.build_library()
    return rcorrector_amd


def header_functions():
    h = open(os.path.join(ROOT, "include", "rcorrector_amd.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(rc_[a-z_0-9]+)\s*\(", h)))


def test_library_exports_every_declared_symbol(built):
    lib = built.load_library()
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "librcorrector_amd.so does not export %s" % n
    assert sorted(built.ABI_SYMBOLS) == names


def test_kernels_are_gfx950_code_objects(built):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          "--input=" + built.library_path()], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    txt = subprocess.run(["strings", "-n", "6", built.library_path()], stdout=subprocess.PIPE).stdout.decode(errors="ignore")
    assert "gfx950" in txt
    for k in ("k_probe", "k_threshold", "k_correct", "k_scatter"):
        assert k in txt


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(built.RcorrectorError) as e:
        built.Context(k=23)
    assert "no HIP device" in str(e.value) or "HIP" in str(e.value)
    cli = os.path.join(ROOT, "rcorrector_amd", "rcorrector")
    d = os.path.join(ROOT, "tests", "golden", "fx_se_k23")
    p = subprocess.run([cli, "-r", "reads.fq", "-k", "23", "-c", "dump.jf", "-od", "/tmp"], cwd=d, stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"no HIP device" in p.stderr


def test_product_sources_do_not_reference_the_oracle():
    # the oracle is test infrastructure: nothing under rcorrector_amd/ or include/ may use it
    bad = []
    for base in ("rcorrector_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                    t = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"pyoracle|liboracle|rc_oracle|rco_|import oracle|from oracle", t):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_bad_quality_from_hist_equals_oracle(built, oracle):
    """GetBadQuality's arithmetic (main.cpp:108-127) is host code of the library: random
    histograms (spiky, flat, empty, all in one bin, totals that do not match the bins) through
    rc_bad_quality_from_hist and through the oracle's restatement."""
    import ctypes as C
    import numpy as np
    L = built.load_library()
    O = oracle.lib()
    rng = np.random.Generator(np.random.PCG64(5))
    for it in range(2000):
        kind = it % 5
        fh = np.zeros(300, dtype=np.int32)
        lh = np.zeros(300, dtype=np.int32)
        if kind == 0:
            fh[rng.integers(33, 75, size=1000)] += 1
            lh[rng.integers(33, 75, size=1000)] += 1
        elif kind == 1:
            fh[int(rng.integers(0, 300))] = int(rng.integers(1, 10 ** 6))
            lh[int(rng.integers(0, 300))] = int(rng.integers(1, 10 ** 6))
        elif kind == 2:
            fh[:] = rng.integers(0, 50, size=300)
            lh[:] = rng.integers(0, 50, size=300)
        elif kind == 3:
            pass
        else:
            np.add.at(fh, rng.integers(0, 300, size=200), 1)
            np.add.at(lh, rng.integers(0, 300, size=200), 1)
        total = int(fh.sum()) if kind != 4 else int(rng.integers(0, 2000))
        a = L.rc_bad_quality_from_hist(fh.ctypes.data, lh.ctypes.data, total)
        b = O.rco_bad_quality_from_hist(fh.ctypes.data, lh.ctypes.data, total)
        assert a == b, (it, a, b)


def test_pack_bases_and_apply_fixes_are_plain_host_code(built):
    """rc_pack_bases (the packed boundary's host half): 2 bits per base at bits 30 - 2 (p & 15) of word p >> 4, NULs and
    the letters outside ACGT as code 0 with the latter listed -- against a numpy restatement on a random arena, whole and
    in ranges that start at multiples of 16 bytes (how several threads pack one arena); rc_apply_fixes is the loop."""
    import ctypes as C
    import numpy as np
    L = built.load_library()
    rng = np.random.default_rng(3)
    n = 100_003
    a = rng.choice(np.frombuffer(b"ACGTACGTACGTACGTNRYKM\0\0", np.uint8), size=n).astype(np.uint8)
    code = np.zeros(256, np.uint32)
    code[ord("C")], code[ord("G")], code[ord("T")] = 1, 2, 3
    pad = np.zeros((n + 15) // 16 * 16, np.uint8)
    pad[:n] = a
    want = (code[pad].reshape(-1, 16) << (30 - 2 * np.arange(16, dtype=np.uint32))).sum(axis=1).astype(np.uint32)
    exc = np.nonzero((a != 0) & ~np.isin(a, np.frombuffer(b"ACGT", np.uint8)))[0]
    bases = np.full((n + 15) // 16, 0xDEADBEEF, np.uint32)
    ep, ec = np.zeros(len(exc) + 4, np.uint32), np.zeros(len(exc) + 4, np.uint8)
    got = L.rc_pack_bases(a.ctypes.data, 0, n, bases.ctypes.data, ep.ctypes.data, ec.ctypes.data, len(ep))
    assert got == len(exc) and np.array_equal(ep[:got], exc) and np.array_equal(ec[:got], a[exc]) and np.array_equal(bases, want)
    assert L.rc_pack_bases(a.ctypes.data, 0, n, bases.ctypes.data, None, None, 0) == len(exc)   # counting only
    bases2 = np.zeros_like(bases)
    cuts = [0, 16 * 100, 16 * 2500, 16 * 2501, n]
    tot = 0
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        e2p, e2c = np.zeros(len(exc), np.uint32), np.zeros(len(exc), np.uint8)
        k = L.rc_pack_bases(a.ctypes.data, lo, hi, bases2.ctypes.data, e2p.ctypes.data, e2c.ctypes.data, len(e2p))
        assert np.array_equal(e2p[:k], exc[(exc >= lo) & (exc < hi)])
        tot += k
    assert tot == len(exc) and np.array_equal(bases2, want)
    b = a.copy()
    fp = rng.choice(n, size=500, replace=False).astype(np.uint32)
    fc = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=500).astype(np.uint8)
    L.rc_apply_fixes(b.ctypes.data, fp.ctypes.data, fc.ctypes.data, 500)
    w = a.copy()
    w[fp] = fc
    assert np.array_equal(b, w)


def test_build_provenance_matches_the_tree(built):
    """profiles/r6_build.json (tools/build_provenance.py) records the sha256 of the shipped binaries and of every source they
    are built from: while the sources in the tree are the recorded ones, the binaries must be the recorded ones too -- i.e. the
    library a GPU box loads is the one the profiles were measured with, checkable without a rebuild."""
    import importlib.util
    import json
    path = os.path.join(ROOT, "profiles", "r6_build.json")
    if not os.path.exists(path):
        pytest.skip("no provenance record yet (written after the last build of a round)")
    rec = json.load(open(path))
    spec = importlib.util.spec_from_file_location("build_provenance", os.path.join(ROOT, "tools", "build_provenance.py"))
    bp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bp)
    src, bins = bp.snapshot()
    assert rec["sources"] and rec["binaries"]
    if src != rec["sources"]:
        pytest.skip("the sources have changed since the record was written")
    assert bins["rcorrector_amd/librcorrector_amd.so"]["sha256"] == rec["binaries"]["rcorrector_amd/librcorrector_amd.so"]["sha256"]
