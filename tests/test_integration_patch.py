"""The drop-in boundary, proved on the reference's own main.cpp: tools/apply_integration.py (the
INTEGRATION.md section 2 patch) is applied to a temporary copy, the result compiles against
include/rcorrector_amd.h and links against librcorrector_amd.so (CPU: no run); on the GPU box the
binary built that way in the build container (oracle/_ref/rcorrector_patched, it travels with the
repo) must reproduce the golden fixtures -- the reference's I/O and output code around our hot path."""
import os
import subprocess
import sys

import pytest

import golden_util as gu

REF = "/root/reference"
PATCHED = os.path.join(gu.ROOT, "oracle", "_ref", "rcorrector_patched")


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference sources exist in the build container only")
def test_patch_applies_compiles_and_links(tmp_path):
    lib = os.path.join(gu.ROOT, "rcorrector_amd", "librcorrector_amd.so")
    assert os.path.exists(lib), "build the library first"
    out = tmp_path / "main.cpp"
    subprocess.run([sys.executable, os.path.join(gu.ROOT, "tools", "apply_integration.py"), os.path.join(REF, "main.cpp"), str(out)], check=True)
    text = out.read_text()
    for call in ("rc_create", "rc_table_load_jfdump", "rc_estimate_error_rate", "rc_set_run_params", "rc_correct_batch"):
        assert call in text
    assert "pthread_create(" not in text.replace(" ", "") and "kmers.Put" not in text
    objs = []
    for src, inc in ((str(out), ["-I" + REF, "-I" + os.path.join(gu.ROOT, "include")]),
                     (os.path.join(REF, "KmerCode.cpp"), ["-I" + REF]), (os.path.join(REF, "ErrorCorrection.cpp"), ["-I" + REF])):
        o = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.run(["g++", "-w", "-O1", "-std=c++0x"] + inc + ["-c", src, "-o", o], check=True)
        objs.append(o)
    exe = str(tmp_path / "rcorrector_patched")
    subprocess.run(["g++", "-o", exe] + objs + ["-L" + os.path.dirname(lib), "-lrcorrector_amd", "-lpthread", "-lz"], check=True)
    assert os.path.getsize(exe) > 10000
    # a different main.cpp is refused, not mangled
    bad = tmp_path / "other.cpp"
    bad.write_text("int main() { return 0; }\n")
    assert subprocess.run([sys.executable, os.path.join(gu.ROOT, "tools", "apply_integration.py"), str(bad), str(tmp_path / "x.cpp")]).returncode != 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["fx_sample", "fx_se_k23", "fx_pe_k23", "fx_il_k23", "fx_k31_mc8", "fx_skew", "fx_edge", "fx_varlen_n"])
def test_patched_reference_binary_reproduces_goldens(name, tmp_path):
    if not os.path.exists(PATCHED):
        pytest.skip("oracle/_ref/rcorrector_patched was not built (no /root/reference where build() ran)")
    p = gu.run_fixture(PATCHED, name, tmp_path)
    gu.assert_same_as_reference(name, tmp_path, p.stderr)
