"""CPU: the arithmetic of the CLI's one-pass path that runs on the host -- quality bits packed from the FASTQ text, fixes
applied to the text's sequence lines (rc_main.cpp) -- against the byte-arena code paths, on ragged, empty and over-long
quality lines, single and paired arenas, any split into pieces (tests/hostmain/hostmain_test.cpp; no GPU involved: the
library is only linked for rc_pack_quality_bits / rc_host_register, which fails harmlessly without a device)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_quality_bits_and_fixes_on_the_text(tmp_path):
    import rcorrector_amd
    rcorrector_amd.build_library()
    exe = str(tmp_path / "hostmain_test")
    lib = os.path.join(ROOT, "rcorrector_amd")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wno-unused-function", os.path.join(ROOT, "tests", "hostmain", "hostmain_test.cpp"), "-o", exe,
                    "-L" + lib, "-lrcorrector_amd", "-lz", "-lpthread", "-ldl", "-Wl,-rpath," + lib], check=True)
    p = subprocess.run([exe, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode()
    assert p.returncode == 0 and "ok 60 cases" in out, out
