"""CPU: the arithmetic of the CLI's one-pass path that runs on the host -- quality bits packed from the FASTQ text, fixes
applied to the text's sequence lines (rc_format.cpp) -- against the byte-arena code paths, on ragged, empty and over-long
quality lines, single and paired arenas, any split into pieces (tests/hostmain/hostmain_test.cpp; no GPU involved: the
library is only linked for rc_pack_quality_bits / rc_host_register, which fails harmlessly without a device)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rcorrector_amd", "csrc")
HOST_UNITS = ["rc_pool", "rc_reader", "rc_format", "rc_writer", "rc_dispatch"]


def build_host_test(src, exe, flags=("-O2",), objdir=None):
    """A host test program linked against the CLI's host units (csrc/Makefile: hostobjs; `flags` / `objdir`: a differently
    compiled set, e.g. with a sanitizer, kept apart from the product's objects)."""
    lib = os.path.join(ROOT, "rcorrector_amd")
    if objdir is None:
        subprocess.run(["make", "-C", CSRC, "-j4", "hostobjs"], check=True, stdout=subprocess.DEVNULL)
        objdir = CSRC
    else:
        os.makedirs(objdir, exist_ok=True)
        subprocess.run(["make", "-C", CSRC, "-j4", "hostobjs", "HOSTOBJDIR=" + objdir,
                        "HOSTFLAGS=-std=c++17 -Wall -g " + " ".join(flags)], check=True, stdout=subprocess.DEVNULL)
    objs = [os.path.join(objdir, u + ".host.o") for u in HOST_UNITS]
    subprocess.run(["g++", "-std=c++17", "-Wno-unused-function"] + list(flags) + [src, "-o", exe] + objs +
                   ["-L" + lib, "-lrcorrector_amd", "-lz", "-lpthread", "-ldl", "-Wl,-rpath," + lib], check=True)


def test_quality_bits_and_fixes_on_the_text(tmp_path):
    import rcorrector_amd
    rcorrector_amd.build_library()
    exe = str(tmp_path / "hostmain_test")
    build_host_test(os.path.join(ROOT, "tests", "hostmain", "hostmain_test.cpp"), exe)
    p = subprocess.run([exe, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode()
    assert p.returncode == 0 and "ok 60 cases" in out, out


def test_gz_inputs_read_the_same_through_libdeflate_and_zlib(tmp_path):
    """Source::inflate_whole (libdeflate: whole members, BGZF blocks side by side) against gzread on every kind of .gz file"""
    import gzip
    import struct
    import zlib
    import numpy as np
    import rcorrector_amd
    rcorrector_amd.build_library()
    rng = np.random.default_rng(3)
    recs = []
    for i in range(30000):
        L = int(rng.integers(20, 160))
        recs.append(b"@r%d\n%s\n+\n%s\n" % (i, rng.choice(np.frombuffer(b"ACGTN", np.uint8), L).tobytes(), rng.integers(33, 74, L).astype(np.uint8).tobytes()))
    data = b"".join(recs)
    d = str(tmp_path)
    files = {}
    files["one.gz"] = gzip.compress(data, 6)
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
    files["text_behind.gz"] = files["one.gz"] + b"this is not gzip\n" * 20
    files["truncated.gz"] = files["one.gz"][:len(files["one.gz"]) * 6 // 10]
    files["plain.gz"] = data[:200000]
    files["empty.gz"] = b""
    files["tiny.gz"] = gzip.compress(b"@a\nACGT\n+\nIIII\n", 6)
    for name, content in files.items():
        open(os.path.join(d, name), "wb").write(content)
    exe = str(tmp_path / "gz_test")
    build_host_test(os.path.join(ROOT, "tests", "hostmain", "gz_test.cpp"), exe)
    p = subprocess.run([exe] + [os.path.join(d, n) for n in files], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode()
    assert p.returncode == 0 and out.rstrip().endswith("ok"), out
    if os.path.exists("/usr/lib/x86_64-linux-gnu/libdeflate.so.0"):   # (where the library is there, the usual files do take it)
        for name in ("one.gz", "three.gz", "bgzf.gz"):
            assert ("%s: %d bytes via zlib, %d via libdeflate" % (os.path.join(d, name), len(data), len(data))) in out, out
