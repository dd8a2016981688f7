"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product package (rcorrector_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_BIN = os.path.join(HERE, "_ref", "rcorrector_ref")
CLI_BIN = os.path.join(HERE, "oracle_cli")


def build(quiet=True):
    """Compile the C restatement (and, where /root/reference exists, the unmodified reference
    into oracle/_ref/)."""
    subprocess.run(["make", "-C", HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


class Params(C.Structure):
    _fields_ = [("k", C.c_int), ("max_fix_per_k", C.c_int), ("error_rate", C.c_double),
                ("bad_qual", C.c_char), ("verbose_fp", C.c_void_p)]


class Batch(C.Structure):
    _fields_ = [("mode", C.c_int), ("n", C.c_size_t),
                ("seq", C.c_void_p), ("qual", C.c_void_p), ("off", C.c_void_p),
                ("seq2", C.c_void_p), ("qual2", C.c_void_p), ("off2", C.c_void_p),
                ("ret", C.c_void_p), ("l", C.c_void_p), ("m", C.c_void_p), ("h", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.rco_table_new.restype = C.c_void_p
        L.rco_table_new.argtypes = [C.c_int, C.c_size_t]
        L.rco_table_free.argtypes = [C.c_void_p]
        L.rco_table_put_canon.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
        L.rco_table_put_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.rco_table_size.restype = C.c_size_t
        L.rco_table_size.argtypes = [C.c_void_p]
        L.rco_table_export.restype = C.c_size_t
        L.rco_table_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.rco_load_dump.restype = C.c_long
        L.rco_load_dump.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
        L.rco_estimate_error_rate.restype = C.c_double
        L.rco_estimate_error_rate.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_double]
        L.rco_bad_quality_from_hist.restype = C.c_char
        L.rco_bad_quality_from_hist.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.rco_correct_batch.argtypes = [C.POINTER(Params), C.c_void_p, C.POINTER(Batch), C.c_int]
        L.rco_kmer_counts.restype = C.c_int
        L.rco_kmer_counts.argtypes = [C.POINTER(Params), C.c_void_p, C.c_char_p, C.c_void_p]
        L.rco_strong_trusted_threshold.restype = C.c_int
        L.rco_strong_trusted_threshold.argtypes = [C.POINTER(Params), C.c_void_p, C.c_char_p]
        L.rco_get_bound_int.restype = C.c_int
        L.rco_get_bound_int.argtypes = [C.POINTER(Params), C.c_int]
        L.rco_get_bound.restype = C.c_double
        L.rco_get_bound.argtypes = [C.POINTER(Params), C.c_int]
        L.rco_set_chunk.argtypes = [C.c_int]
        L.rco_set_interleave.restype = C.c_int
        L.rco_set_interleave.argtypes = [C.c_int]
        _lib = L
    return _lib


def set_chunk(units):
    """units a worker thread takes per visit to the batch's shared counter (1 = the reference's schedule; no effect on results)"""
    lib().rco_set_chunk(int(units))


class Table:
    def __init__(self, k, expected=1 << 16, interleave=False):
        """interleave: spread the table's pages over every memory node of the host (a table one thread fills otherwise lives
        on that thread's node and is probed by the threads of every socket)"""
        self.k = k
        self.interleave = bool(interleave)
        self.interleaved = False
        self.h = lib().rco_table_new(k, expected)

    def __del__(self):
        if getattr(self, "h", None):
            lib().rco_table_free(self.h)
            self.h = None

    def load_dump(self, path):
        n = lib().rco_load_dump(self.h, self.k, os.fsencode(path))
        if n < 0:
            raise IOError("cannot open " + path)
        return n

    def error_rate(self, path, wk=0.95):
        return lib().rco_estimate_error_rate(self.h, self.k, os.fsencode(path), wk)

    def put_many(self, canon_codes, counts):
        cc = np.ascontiguousarray(canon_codes, dtype=np.uint64)
        vv = np.ascontiguousarray(counts, dtype=np.int32)
        # (pages are placed when they are first written, i.e. here: the policy covers the fill)
        ok = self.interleave and lib().rco_set_interleave(1) == 0
        lib().rco_table_put_many(self.h, cc.ctypes.data, vv.ctypes.data, len(cc))
        if ok:
            lib().rco_set_interleave(0)
            self.interleaved = True

    def size(self):
        return lib().rco_table_size(self.h)

    def export(self):
        n = self.size()
        codes = np.empty(n, dtype=np.uint64)
        counts = np.empty(n, dtype=np.int32)
        w = lib().rco_table_export(self.h, codes.ctypes.data, counts.ctypes.data, n)
        return codes[:w], counts[:w]


def pack_reads(seqs):
    """list of bytes -> (arena uint8 with a NUL after every read, offsets uint32[n+1])."""
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    off = np.zeros(len(seqs) + 1, dtype=np.uint32)
    np.cumsum(lens + 1, out=off[1:])
    arena = np.frombuffer(b"\0".join(seqs) + b"\0", dtype=np.uint8).copy() if seqs else np.zeros(0, np.uint8)
    return arena, off


def unpack_reads(arena, off):
    b = arena.tobytes()
    return [b[off[i]:off[i + 1] - 1] for i in range(len(off) - 1)]


def bad_quality(first_chars, last_chars):
    """GetBadQuality (main.cpp:88-128) over first/last quality characters (uint8 arrays)."""
    fh = np.bincount(first_chars, minlength=300)[:300].astype(np.int32)
    lh = np.bincount(last_chars, minlength=300)[:300].astype(np.int32)
    r = lib().rco_bad_quality_from_hist(fh.ctypes.data, lh.ctypes.data, int(len(first_chars)))
    return r


def make_params(k, max_fix_per_k=4, error_rate=0.01, bad_qual=b"!"):
    p = Params()
    p.k = k
    p.max_fix_per_k = max_fix_per_k
    p.error_rate = error_rate
    p.bad_qual = bad_qual
    p.verbose_fp = None
    return p


def correct_batch(params, table, mode, seq, qual, off, seq2=None, qual2=None, off2=None, threads=1,
                  fn=None):
    """Runs ErrorCorrection_Thread semantics over a batch.  seq arenas are modified IN PLACE.
    Returns (ret, l, m, h) int32 arrays of length n (mode 0/2) or 2n (mode 1)."""
    n = len(off) - 1
    total = 2 * n if mode == 1 else n
    ret = np.zeros(total, dtype=np.int32)
    l = np.zeros(total, dtype=np.int32)
    m = np.zeros(total, dtype=np.int32)
    h = np.zeros(total, dtype=np.int32)
    b = Batch()
    b.mode = mode
    b.n = n
    b.seq, b.qual, b.off = seq.ctypes.data, qual.ctypes.data, off.ctypes.data
    if mode == 1:
        b.seq2, b.qual2, b.off2 = seq2.ctypes.data, qual2.ctypes.data, off2.ctypes.data
    b.ret, b.l, b.m, b.h = ret.ctypes.data, l.ctypes.data, m.ctypes.data, h.ctypes.data
    if fn is None:
        lib().rco_correct_batch(C.byref(params), table.h, C.byref(b), threads)
    else:
        fn(C.byref(params), table.h, C.byref(b))
    return ret, l, m, h


def kmer_counts(params, table, seq_bytes):
    buf = np.zeros(1024, dtype=np.int32)
    n = lib().rco_kmer_counts(C.byref(params), table.h, seq_bytes, buf.ctypes.data)
    return buf[:n].copy()
