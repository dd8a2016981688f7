import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/liboracle.so on demand."""
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def oracle_cli(oracle):
    """The oracle's command-line front end (mimics main.cpp); pinned to the reference by test_golden_cpu."""
    assert os.path.exists(oracle.CLI_BIN)
    return oracle.CLI_BIN


def build_hostsim(so, flags=("-O2",)):
    """Lane-serial build of the kernel control flow (tests/hostsim) into `so`; `flags`: e.g. a sanitizer's."""
    d = os.path.join(ROOT, "tests", "hostsim")
    subprocess.run(["g++", "-std=c++17", "-fPIC", "-shared"] + list(flags) + ["-o", so, os.path.join(d, "hostsim.cpp"),
                    "-L" + os.path.join(ROOT, "oracle"), "-l:liboracle.so",
                    "-Wl,-rpath," + os.path.join(ROOT, "oracle")], check=True)


def load_hostsim(so):
    import ctypes as C
    lib = C.CDLL(so)
    lib.hostsim_correct_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


@pytest.fixture(scope="session")
def hostsim(oracle):
    so = os.path.join(ROOT, "tests", "hostsim", "libhostsim.so")
    build_hostsim(so)
    return load_hostsim(so)


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    """Factory of rcorrector_amd.Context objects on cuda:0 (fails loudly without the HIP library)."""
    import rcorrector_amd

    def make(k=23, max_fix_per_k=4):
        return rcorrector_amd.Context(k=k, max_fix_per_k=max_fix_per_k, device=0)
    return make
