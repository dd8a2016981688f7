"""rcorrector_amd -- MI355X-native implementation of Rcorrector's per-read correction path.

The product is the C-ABI shared library ``librcorrector_amd.so`` (HIP kernels for gfx950 plus the
host glue, sources under ``rcorrector_amd/csrc``, interface in ``include/rcorrector_amd.h``) and
the ``rcorrector`` command-line front end built from ``csrc/rc_main.cpp``.  This Python module is
a thin ctypes binding over that ABI used by the test-suite and ``bench.py``; it contains no
algorithm and no CPU fallback -- if the library or a GPU is missing, it raises.
"""
from .binding import (Context, RcorrectorError, build_library, library_path, load_library,  # noqa: F401
                      pack_reads, unpack_reads, runtime_prepare, ABI_SYMBOLS)

__all__ = ["Context", "RcorrectorError", "build_library", "library_path", "load_library",
           "pack_reads", "unpack_reads", "runtime_prepare", "ABI_SYMBOLS"]
