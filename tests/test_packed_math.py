"""PACKED table addressing (rcorrector_amd/csrc/rc_common.h) checked on the host: the 2k-bit bijection
and its inverse, and (home, remainder, extra remainder bits) <-> canonical code for every k, several
table sizes and the number of extra bits the build would choose for them."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packed_addressing_round_trips(tmp_path):
    exe = str(tmp_path / "packed_math")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "rcorrector_amd", "csrc"),
                    os.path.join(ROOT, "tests", "hostmath", "packed_math.cpp"), "-o", exe], check=True)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    out = p.stdout.decode()
    assert p.returncode == 0 and out.startswith("ok "), out


def test_bound_steps_table_equals_bisection(tmp_path):
    """rc_bound_steps_build (the integer steps of GetBound the kernels compare counts against) from its closed-form first guess
    against plain bisection, 14 error rates including degenerate ones"""
    exe = str(tmp_path / "bound_steps")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "rcorrector_amd", "csrc"),
                    os.path.join(ROOT, "tests", "hostmath", "bound_steps.cpp"), "-o", exe], check=True)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode()
    assert p.returncode == 0 and out.rstrip().endswith("ok") and " 0 differences" in out, out


def test_pack16m_swar_equals_its_definition(tmp_path):
    """rc_pack16m (the fused probe kernel's staging: 16 arena bytes -> 2-bit codes + A / T / not-ACGT masks, four bytes per
    operation) against the per-byte definition: every byte value in every position in front of seven backgrounds, and
    two million random blocks"""
    exe = str(tmp_path / "pack16m")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "rcorrector_amd", "csrc"),
                    os.path.join(ROOT, "tests", "hostmath", "pack16m.cpp"), "-o", exe], check=True)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    out = p.stdout.decode()
    assert p.returncode == 0 and out.startswith("ok "), out


def test_kmer_walk_reciprocal_is_exact():
    """The fused probe kernel's k-mer walk (rc_correct.hip: tiles whose reads all have K k-mers) maps k-mer v to read v / K by one
    multiply-high with floor(2^32 / K) + 1: exact for every K a read of up to 160 bases can have and every v a tile can hold
    (64 reads x 160 k-mers < 2^14), which is what lets the loop skip the positions that start no k-mer."""
    import numpy as np
    v = np.arange(1 << 14, dtype=np.uint64)
    for K in range(2, 161):
        rcp = np.uint64((1 << 32) // K + 1)
        assert np.array_equal((v * rcp) >> np.uint64(32), v // np.uint64(K)), K
