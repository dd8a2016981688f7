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
