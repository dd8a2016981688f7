"""CPU: the host code and the CPU restatement under sanitizers ("ASan/UBSan builds of the CPU restatement").

* AddressSanitizer + UndefinedBehaviorSanitizer: the CLI's host units (rc_pool, rc_reader, rc_format: buffers from mmap /
  mremap, SSE loads that run up to 15 bytes past a quality line, the line index) through the 60 host cases and every kind of
  .gz input; the oracle's command line (`oracle/rc_oracle*.c`) on four edge goldens; the lane-serial build of the search
  kernel's control flow (tests/hostsim) on three data sets; the PACKED addressing arithmetic (tests/hostmath).
* ThreadSanitizer: the helper-thread pool, the parallel block reads / newline scans / BGZF inflate of the reader and the
  parallel quality-bit packing, on the same host cases and .gz inputs.

Any sanitizer report fails the test (`-fno-sanitize-recover`, `halt_on_error`); the programs end with _exit(), so leak
checking -- which would list the HIP runtime's process-lifetime allocations -- is off."""
import os
import subprocess

import numpy as np
import pytest

import golden_util as gu
from test_hostmain import build_host_test

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN = ("-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer")
TSAN = ("-O1", "-fsanitize=thread", "-fno-omit-frame-pointer")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
           TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")


def _sanitizer_works(flags, tmp_path):
    """the image's g++ has the sanitizer run-times; a box without them skips instead of failing"""
    src = tmp_path / "probe.cpp"
    src.write_text("int main() { return 0; }\n")
    p = subprocess.run(["g++"] + list(flags) + [str(src), "-o", str(tmp_path / "probe")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return p.returncode == 0 and subprocess.run([str(tmp_path / "probe")], env=ENV).returncode == 0


def _gz_files(d):
    import gzip
    import struct
    import zlib
    rng = np.random.default_rng(5)
    recs = []
    for i in range(12000):
        L = int(rng.integers(20, 160))
        recs.append(b"@r%d\n%s\n+\n%s\n" % (i, rng.choice(np.frombuffer(b"ACGTN", np.uint8), L).tobytes(), rng.integers(33, 74, L).astype(np.uint8).tobytes()))
    data = b"".join(recs)
    files = {"one.gz": gzip.compress(data, 6)}
    third = len(data) // 3
    files["three.gz"] = gzip.compress(data[:third], 1) + gzip.compress(data[third:2 * third], 9) + gzip.compress(data[2 * third:], 6)
    bg = b""
    for lo in list(range(0, len(data), 60000)) + [len(data)]:
        chunk = data[lo:lo + 60000] if lo < len(data) else b""
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        raw = co.compress(chunk) + co.flush()
        bg += b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(raw) + 8 - 1)
        bg += raw + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
    files["bgzf.gz"] = bg
    files["zeros_behind.gz"] = files["one.gz"] + b"\0" * 300
    files["truncated.gz"] = files["one.gz"][:len(files["one.gz"]) * 6 // 10]
    files["plain.gz"] = data[:100000]
    files["empty.gz"] = b""
    paths = []
    for name, content in files.items():
        paths.append(os.path.join(d, name))
        open(paths[-1], "wb").write(content)
    return paths


@pytest.mark.parametrize("kind", ["asan_ubsan", "tsan"])
def test_host_units_under_sanitizers(tmp_path, kind):
    import rcorrector_amd
    rcorrector_amd.build_library()
    flags = ASAN if kind == "asan_ubsan" else TSAN
    if not _sanitizer_works(flags, tmp_path):
        pytest.skip("no %s run-time for g++ on this box" % kind)
    objdir = str(tmp_path / "obj")
    for prog, args, ok in (("hostmain_test", [str(tmp_path)], "ok 60 cases"), ("gz_test", _gz_files(str(tmp_path)), "ok")):
        exe = str(tmp_path / prog)
        build_host_test(os.path.join(ROOT, "tests", "hostmain", prog + ".cpp"), exe, flags=flags, objdir=objdir)
        p = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=ENV)
        out = p.stdout.decode()
        assert p.returncode == 0 and out.rstrip().endswith(ok) and "Sanitizer" not in out and "runtime error" not in out, out[-4000:]


def test_oracle_cli_under_asan_ubsan(tmp_path):
    if not _sanitizer_works(ASAN, tmp_path):
        pytest.skip("no ASan run-time on this box")
    odir = os.path.join(ROOT, "oracle")
    exe = str(tmp_path / "oracle_cli_san")
    subprocess.run(["gcc", "-std=c99", "-g", "-fopenmp"] + list(ASAN) + ["-o", exe, os.path.join(odir, "rc_oracle_cli.c"), os.path.join(odir, "rc_oracle.c"),
                    os.path.join(odir, "rc_oracle_io.c"), "-lm", "-lpthread", "-lz"], check=True)
    for name in ("fx_edge", "fx_varlen_n", "fx_k32", "fx_k31_mc8"):
        out = tmp_path / name
        out.mkdir()
        d = os.path.join(gu.GOLDEN, name)
        args = open(os.path.join(d, "cmd.txt")).read().split()
        p = subprocess.run([exe] + args + ["-od", str(out), "-t", "2"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=ENV, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-4000:]
        gu.assert_same_as_reference(name, out, p.stderr)


def test_hostsim_and_packed_math_under_asan_ubsan(tmp_path, oracle):
    """the search kernel's control flow (rc_correct_core.h, lane-serial) and the PACKED addressing, instrumented"""
    if not _sanitizer_works(ASAN, tmp_path):
        pytest.skip("no ASan run-time on this box")
    csrc = os.path.join(ROOT, "rcorrector_amd", "csrc")
    exe = str(tmp_path / "packed_math")
    subprocess.run(["g++", "-std=c++17", "-g"] + list(ASAN) + ["-I", csrc, os.path.join(ROOT, "tests", "hostmath", "packed_math.cpp"), "-o", exe], check=True)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=ENV)
    assert p.returncode == 0 and p.stdout.decode().startswith("ok "), p.stdout.decode()[-4000:]
    # hostsim is a shared object loaded by Python: an instrumented copy runs in a child interpreter with the ASan run-time
    # preloaded (the interpreter itself is not instrumented)
    import conftest
    so = str(tmp_path / "libhostsim_san.so")
    conftest.build_hostsim(so, flags=("-g",) + ASAN)
    rt = subprocess.run(["g++", "-print-file-name=libasan.so"], stdout=subprocess.PIPE, check=True).stdout.decode().strip()
    child = r"""
import sys, numpy as np
sys.path[:0] = [%r, %r, %r]
import datasets
from oracle import pyoracle as po
po.build()
import conftest
hs = conftest.load_hostsim(%r)
for name in ("pe_k23", "edge", "k31_mc8"):
    d = datasets.make(name)
    want = datasets.run_oracle(po, d)
    got = datasets.run_oracle(po, d, fn=lambda p, t, b: hs.hostsim_correct_batch(p, t, b, None))
    for w, g in zip(want, got):
        assert np.array_equal(w, g), name
print("hostsim ok")
""" % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), ROOT, so)
    p = subprocess.run(["python3", "-c", child], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900,
                       env=dict(ENV, LD_PRELOAD=rt, ASAN_OPTIONS=ENV["ASAN_OPTIONS"] + ":verify_asan_link_order=0"))
    out = p.stdout.decode()
    assert p.returncode == 0 and "hostsim ok" in out and "Sanitizer" not in out and "runtime error" not in out, out[-4000:]
