"""ctypes binding of include/rcorrector_amd.h.  No algorithm lives here."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")

# every function include/rcorrector_amd.h declares (tests check the .so exports all of them)
ABI_SYMBOLS = [
    "rc_create", "rc_destroy", "rc_last_error", "rc_device_numa_node", "rc_device_memory",
    "rc_table_build", "rc_table_build_device", "rc_table_load_jfdump",
    "rc_table_count_begin", "rc_table_count_add", "rc_table_count_add_device", "rc_table_count_finish",
    "rc_table_count_keep", "rc_table_count_arenas", "rc_table_count_release", "rc_table_count_park", "rc_table_count_finish_sharded", "rc_submit_resident", "rc_wait_resident",
    "rc_table_count_reads_device", "rc_table_write_jfdump", "rc_table_share", "rc_table_replicate", "rc_table_replicate_async", "rc_table_lookup", "rc_table_export", "rc_table_digest", "rc_table_layout", "rc_table_stats",
    "rc_estimate_error_rate", "rc_bad_quality_from_hist", "rc_set_run_params", "rc_set_quality_bits", "rc_pack_quality_bits",
    "rc_correct_batch", "rc_set_slot_lanes", "rc_runtime_prepare", "rc_submit", "rc_wait", "rc_host_alloc", "rc_host_free", "rc_host_register", "rc_host_unregister", "rc_correct_batch_traced", "rc_correct_device", "rc_strong_threshold_device", "rc_probe_device", "rc_sync",
    "rc_strong_threshold_read", "rc_correct_read", "rc_kmer_info_read",
    "rc_pack_bases", "rc_submit_packed", "rc_wait_packed", "rc_apply_fixes",
    "rc_profile_enable", "rc_profile_get", "rc_profile_reset", "rc_profile_correct_counters", "rc_profile_read_rounds", "rc_selftest_get_bound", "rc_summary",
]


class RcorrectorError(RuntimeError):
    pass


def library_path():
    # RC_LIB lets a developer A/B a differently built library; the default is the in-tree build
    return os.environ.get("RC_LIB") or os.path.join(HERE, "librcorrector_amd.so")


def build_library(quiet=True):
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", CSRC, "-j4", "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)
    return library_path()


class _Config(C.Structure):
    _fields_ = [("device", C.c_int), ("k", C.c_int), ("max_fix_per_k", C.c_int)]


class _Batch(C.Structure):
    _fields_ = [("mode", C.c_int), ("n", C.c_size_t),
                ("seq", C.c_void_p), ("qual", C.c_void_p), ("off", C.c_void_p),
                ("seq2", C.c_void_p), ("qual2", C.c_void_p), ("off2", C.c_void_p),
                ("ret", C.c_void_p), ("l", C.c_void_p), ("m", C.c_void_p), ("h", C.c_void_p)]


class _PackedBatch(C.Structure):
    _fields_ = [("mode", C.c_int), ("n", C.c_size_t), ("nbytes", C.c_uint64),
                ("off", C.c_void_p), ("bases", C.c_void_p), ("qual_bits", C.c_void_p),
                ("exc_pos", C.c_void_p), ("exc_chr", C.c_void_p), ("n_exc", C.c_size_t),
                ("ret", C.c_void_p), ("l", C.c_void_p), ("m", C.c_void_p), ("h", C.c_void_p),
                ("fix_pos", C.c_void_p), ("fix_chr", C.c_void_p), ("fix_cap", C.c_size_t), ("n_fix", C.c_size_t)]


class _ResidentBatch(C.Structure):
    _fields_ = [("mode", C.c_int), ("n", C.c_size_t), ("arena_a", C.c_int), ("arena_b", C.c_int),
                ("begin_a", C.c_uint64), ("bytes_a", C.c_uint64), ("begin_b", C.c_uint64), ("bytes_b", C.c_uint64),
                ("off", C.c_void_p), ("qual_bits", C.c_void_p),
                ("ret", C.c_void_p), ("l", C.c_void_p), ("m", C.c_void_p), ("h", C.c_void_p),
                ("fix_pos", C.c_void_p), ("fix_chr", C.c_void_p), ("fix_cap", C.c_size_t), ("n_fix", C.c_size_t)]


class _DeviceBatch(C.Structure):
    _fields_ = [("mode", C.c_int), ("n_reads", C.c_uint32), ("nbytes", C.c_uint64),
                ("max_read_len", C.c_int32),
                ("d_seq", C.c_void_p), ("d_qual", C.c_void_p), ("d_off", C.c_void_p),
                ("d_ret", C.c_void_p), ("d_l", C.c_void_p), ("d_m", C.c_void_p), ("d_h", C.c_void_p)]


_lib = None


def load_library():
    """Loads librcorrector_amd.so; raises RcorrectorError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RcorrectorError(
            "%s not found: build the HIP extension first (python -c 'import __graft_entry__ as g; "
            "g.build()' or make -C rcorrector_amd/csrc).  There is no CPU fallback." % path)
    # torch wheels bundle their own copy of the HIP runtime (same soname).  If torch is going to be
    # used in this process (bench.py / tests use it for device tensors and torch.distributed) it
    # has to be loaded FIRST so that exactly one HIP runtime is live; a C/C++ host links ROCm's.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the binding
        pass
    try:
        L = C.CDLL(path)
    except OSError as e:  # pragma: no cover
        raise RcorrectorError("cannot load %s: %s" % (path, e))
    vp, sz = C.c_void_p, C.c_size_t
    L.rc_create.restype = vp
    L.rc_create.argtypes = [C.POINTER(_Config), C.c_char_p, sz]
    L.rc_destroy.argtypes = [vp]
    L.rc_last_error.restype = C.c_char_p
    L.rc_last_error.argtypes = [vp]
    L.rc_table_build.argtypes = [vp, vp, vp, sz]
    L.rc_table_build_device.argtypes = [vp, vp, vp, sz]
    L.rc_table_load_jfdump.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    L.rc_table_count_reads_device.argtypes = [vp, vp, sz, C.c_int, C.POINTER(C.c_int64)]
    L.rc_table_count_begin.argtypes = [vp]
    L.rc_table_count_add.argtypes = [vp, vp, sz]
    L.rc_table_count_add_device.argtypes = [vp, vp, sz]
    L.rc_table_count_finish.argtypes = [vp, C.c_int, C.POINTER(C.c_int64)]
    L.rc_table_count_keep.argtypes = [vp, C.c_int]
    L.rc_table_count_arenas.argtypes = [vp, C.POINTER(C.c_size_t), vp, sz]
    L.rc_table_count_release.argtypes = [vp]
    L.rc_table_count_park.argtypes = [vp]
    L.rc_table_count_finish_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_int64)]
    L.rc_set_slot_lanes.argtypes = [vp, C.c_int]
    if hasattr(L, "rc_runtime_prepare"):   # (RC_LIB may name a library of an earlier round: tools/ab.sh)
        L.rc_runtime_prepare.argtypes = [C.c_int]
    L.rc_submit_resident.argtypes = [vp, C.POINTER(_ResidentBatch), C.c_int]
    L.rc_wait_resident.argtypes = [vp, C.c_int]
    L.rc_table_write_jfdump.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    L.rc_table_share.argtypes = [vp, vp]
    L.rc_table_replicate.argtypes = [vp, vp]
    L.rc_table_replicate_async.argtypes = [vp, vp]
    L.rc_device_numa_node.argtypes = [vp]
    L.rc_table_lookup.argtypes = [vp, vp, sz, vp]
    L.rc_table_export.argtypes = [vp, vp, vp, sz, C.POINTER(C.c_size_t)]
    L.rc_table_digest.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.rc_table_layout.argtypes = [vp]
    L.rc_table_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.rc_estimate_error_rate.argtypes = [vp, C.c_double, C.POINTER(C.c_double)]
    L.rc_bad_quality_from_hist.restype = C.c_char
    L.rc_bad_quality_from_hist.argtypes = [vp, vp, C.c_int32]
    L.rc_set_run_params.argtypes = [vp, C.c_double, C.c_char]
    L.rc_set_quality_bits.argtypes = [vp, C.c_int]
    L.rc_pack_quality_bits.restype = None
    L.rc_pack_quality_bits.argtypes = [vp, sz, C.c_char, vp]
    L.rc_correct_batch.argtypes = [vp, C.POINTER(_Batch)]
    L.rc_submit.argtypes = [vp, C.POINTER(_Batch), C.c_int]
    L.rc_wait.argtypes = [vp, C.c_int]
    L.rc_host_alloc.argtypes = [vp, sz, C.POINTER(C.c_void_p)]
    L.rc_host_free.argtypes = [vp, vp]
    L.rc_correct_device.argtypes = [vp, C.POINTER(_DeviceBatch)]
    L.rc_probe_device.argtypes = [vp, vp, C.c_uint64, vp]
    L.rc_strong_threshold_device.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint64, C.c_int32, vp]
    L.rc_sync.argtypes = [vp]
    L.rc_strong_threshold_read.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int32)]
    L.rc_correct_read.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]
    L.rc_kmer_info_read.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rc_pack_bases.restype = sz
    L.rc_pack_bases.argtypes = [vp, sz, sz, vp, vp, vp, sz]
    L.rc_submit_packed.argtypes = [vp, C.POINTER(_PackedBatch), C.c_int]
    L.rc_wait_packed.argtypes = [vp, C.c_int]
    L.rc_apply_fixes.restype = None
    L.rc_apply_fixes.argtypes = [vp, vp, vp, sz]
    L.rc_profile_enable.argtypes = [vp, C.c_int]
    L.rc_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.rc_profile_reset.argtypes = [vp]
    L.rc_profile_correct_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.rc_profile_read_rounds.argtypes = [vp, vp]
    L.rc_selftest_get_bound.argtypes = [vp, vp, sz, C.c_double, vp, vp]
    L.rc_summary.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    _lib = L
    return L


def runtime_prepare(hw_queues=16):
    """rc_runtime_prepare: ask the HIP runtime for hardware queues for the slot lanes' streams (GPU_MAX_HW_QUEUES unless the process
    has it already).  Before the process first touches HIP -- before torch does; one thread.  1 = set, 0 = was set already."""
    return load_library().rc_runtime_prepare(int(hw_queues))


def pack_reads(seqs):
    """list of bytes -> (uint8 arena with a NUL after every read, uint32 offsets[n+1])."""
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    off = np.zeros(len(seqs) + 1, dtype=np.uint32)
    np.cumsum(lens + 1, out=off[1:])
    if seqs:
        arena = np.frombuffer(b"\0".join(seqs) + b"\0", dtype=np.uint8).copy()
    else:
        arena = np.zeros(0, np.uint8)
    return arena, off


def unpack_reads(arena, off):
    b = arena.tobytes()
    return [b[off[i]:off[i + 1] - 1] for i in range(len(off) - 1)]


def _ptr(x):
    """device/host pointer of a numpy array, a torch tensor or a raw integer address."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        # a torch tensor: whatever produced it ran on torch's stream, the library works on its own --
        # make sure the producer has finished before the address is handed over
        if getattr(x, "is_cuda", False):
            import torch
            torch.cuda.current_stream(x.device).synchronize()
        return x.data_ptr()
    raise TypeError("cannot take a pointer of %r" % type(x))


class Context:
    """One GPU: stream, k-mer table in HBM, scratch.  Mirrors the objects main.cpp sets up
    (KmerCode kcode / Store kmers / globals, main.cpp:17-30,140,270)."""

    def __init__(self, k=23, max_fix_per_k=4, device=0):
        self._L = load_library()
        cfg = _Config(device, k, max_fix_per_k)
        err = C.create_string_buffer(512)
        self._h = self._L.rc_create(C.byref(cfg), err, 512)
        if not self._h:
            raise RcorrectorError(err.value.decode() or "rc_create failed")
        self.k = k

    def close(self):
        if getattr(self, "_h", None):
            for p in getattr(self, "_pinned", []):
                self._L.rc_host_free(self._h, p)
            self._pinned = []
            self._L.rc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise RcorrectorError("rc=%d: %s" % (rc, self._L.rc_last_error(self._h).decode()))

    # ---- table ----
    def table_build(self, codes, counts):
        codes = np.ascontiguousarray(codes, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        self._ck(self._L.rc_table_build(self._h, codes.ctypes.data, counts.ctypes.data, len(codes)))

    def table_build_device(self, d_codes, d_counts, n):
        self._ck(self._L.rc_table_build_device(self._h, _ptr(d_codes), _ptr(d_counts), n))

    def load_jfdump(self, path):
        stored = C.c_int64(0)
        self._ck(self._L.rc_table_load_jfdump(self._h, os.fsencode(path), C.byref(stored)))
        return stored.value

    def count_reads_device(self, d_seq, nbytes, min_count=2):
        n = C.c_int64(0)
        self._ck(self._L.rc_table_count_reads_device(self._h, _ptr(d_seq), nbytes, min_count, C.byref(n)))
        return n.value

    def count_begin(self):
        self._ck(self._L.rc_table_count_begin(self._h))

    def count_add(self, arena):
        """arena: bytes / uint8 array of NUL-separated reads in host memory"""
        a = np.frombuffer(arena, dtype=np.uint8) if isinstance(arena, (bytes, bytearray)) else np.ascontiguousarray(arena, dtype=np.uint8)
        self._ck(self._L.rc_table_count_add(self._h, a.ctypes.data, a.size))

    def count_add_device(self, d_seq, nbytes):
        self._ck(self._L.rc_table_count_add_device(self._h, _ptr(d_seq), nbytes))

    def count_finish(self, min_count=2):
        n = C.c_int64(0)
        self._ck(self._L.rc_table_count_finish(self._h, min_count, C.byref(n)))
        return n.value

    def write_jfdump(self, path):
        n = C.c_int64(0)
        self._ck(self._L.rc_table_write_jfdump(self._h, os.fsencode(path), C.byref(n)))
        return n.value

    def share_table_of(self, other):
        """Use `other`'s table (same device) instead of an own copy; keeps `other` alive."""
        self._ck(self._L.rc_table_share(self._h, other._h))
        self._table_owner = other

    def replicate_table_of(self, other):
        """An own copy of `other`'s table, device to device (rc_table_replicate)."""
        self._ck(self._L.rc_table_replicate(self._h, other._h))

    def replicate_table_of_async(self, other):
        """The same with the copy left in flight on this context's stream (sync() waits for it): queue the copies to all
        GPUs of a node first, they travel at the same time (rc_table_replicate_async)."""
        self._ck(self._L.rc_table_replicate_async(self._h, other._h))

    def numa_node(self):
        """NUMA node of the host this context's GPU hangs off, -1 if the system does not say (rc_device_numa_node)."""
        return int(self._L.rc_device_numa_node(self._h))

    def lookup(self, codes):
        codes = np.ascontiguousarray(codes, dtype=np.uint64)
        out = np.zeros(len(codes), dtype=np.int32)
        self._ck(self._L.rc_table_lookup(self._h, codes.ctypes.data, len(codes), out.ctypes.data))
        return out

    def table_export(self):
        """All stored (canonical code, count) pairs (unspecified order)."""
        cap = int(self.table_stats()["entries"])
        codes = np.zeros(cap, dtype=np.uint64)
        counts = np.zeros(cap, dtype=np.int32)
        n = C.c_size_t(0)
        self._ck(self._L.rc_table_export(self._h, codes.ctypes.data, counts.ctypes.data, cap, C.byref(n)))
        return codes[:n.value], counts[:n.value]

    def table_digest(self):
        """64-bit digest of the table's content (layout independent)."""
        v = C.c_uint64(0)
        self._ck(self._L.rc_table_digest(self._h, C.byref(v)))
        return v.value

    def table_layout(self):
        """0 = WIDE slots, 1 = PACKED slots (rc_table_layout)."""
        r = self._L.rc_table_layout(self._h)
        if r < 0:
            raise RcorrectorError("no table")
        return r

    def table_stats(self):
        b, n, e = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._ck(self._L.rc_table_stats(self._h, C.byref(b), C.byref(n), C.byref(e)))
        return {"bytes": b.value, "buckets": n.value, "entries": e.value}

    # ---- run parameters ----
    def estimate_error_rate(self, wk=0.95):
        r = C.c_double(0)
        self._ck(self._L.rc_estimate_error_rate(self._h, wk, C.byref(r)))
        return r.value

    def bad_quality_from_hist(self, first_hist, last_hist, total):
        fh = np.ascontiguousarray(first_hist, dtype=np.int32)
        lh = np.ascontiguousarray(last_hist, dtype=np.int32)
        return self._L.rc_bad_quality_from_hist(fh.ctypes.data, lh.ctypes.data, int(total))

    def set_run_params(self, error_rate, bad_quality):
        if isinstance(bad_quality, int):
            bad_quality = bytes([bad_quality & 0xFF])
        self._ck(self._L.rc_set_run_params(self._h, error_rate, bad_quality))

    def set_quality_bits(self, on=True):
        """Quality arenas handed to this context are bit arrays (rc_set_quality_bits)."""
        self._ck(self._L.rc_set_quality_bits(self._h, 1 if on else 0))

    def pack_quality_bits(self, qual, bad_quality, out=None):
        """uint8 quality arena -> bit array (bit p = qual[p] > bad_quality), rc_pack_quality_bits."""
        qual = self._arena(qual, "qual")
        if isinstance(bad_quality, int):
            bad_quality = bytes([bad_quality & 0xFF])
        if out is None:
            out = np.zeros((qual.size + 7) // 8, dtype=np.uint8)
        self._L.rc_pack_quality_bits(qual.ctypes.data, qual.size, bad_quality, out.ctypes.data)
        return out

    # ---- correction ----
    @staticmethod
    def _arena(a, what):
        # the library reads and (for seq) rewrites these buffers through raw pointers: a wrong dtype or a
        # strided view would silently be garbage, so refuse instead of converting (seq is corrected in place)
        if not (isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.flags.c_contiguous):
            raise TypeError("%s must be a C-contiguous uint8 numpy array" % what)
        return a

    def _batch(self, mode, seq, qual, off, seq2, qual2, off2, res=None):
        off = np.ascontiguousarray(off, dtype=np.uint32)
        n = len(off) - 1
        total = 2 * n if mode == 1 else n
        if res is None:
            res = [np.zeros(total, dtype=np.int32) for _ in range(4)]
        b = _Batch()
        b.mode, b.n = mode, n
        keep = [off, res]
        b.seq, b.qual, b.off = self._arena(seq, "seq").ctypes.data, self._arena(qual, "qual").ctypes.data, off.ctypes.data
        if mode == 1:
            off2 = np.ascontiguousarray(off2, dtype=np.uint32)
            keep.append(off2)
            b.seq2, b.qual2, b.off2 = self._arena(seq2, "seq2").ctypes.data, self._arena(qual2, "qual2").ctypes.data, off2.ctypes.data
        b.ret, b.l, b.m, b.h = (r.ctypes.data for r in res)
        return b, res, keep

    def correct_batch(self, mode, seq, qual, off, seq2=None, qual2=None, off2=None):
        """Host-buffer batch (rc_correct_batch).  seq arenas are corrected IN PLACE.
        Returns (ret, l, m, h)."""
        b, res, _keep = self._batch(mode, seq, qual, off, seq2, qual2, off2)
        self._ck(self._L.rc_correct_batch(self._h, C.byref(b)))
        return tuple(res)

    def submit(self, slot, mode, seq, qual, off, seq2=None, qual2=None, off2=None, res=None):
        """rc_submit: starts a batch in `slot`; wait(slot) returns (ret, l, m, h).  The arrays must not
        be touched until then.  res: optional list of four int32 arrays to receive the results
        (e.g. pinned ones from host_array)."""
        b, res, keep = self._batch(mode, seq, qual, off, seq2, qual2, off2, res)
        self._ck(self._L.rc_submit(self._h, C.byref(b), slot))
        if not hasattr(self, "_inflight"):
            self._inflight = {}
        self._inflight[slot] = (res, keep, (seq, qual, seq2, qual2))

    def wait(self, slot):
        self._ck(self._L.rc_wait(self._h, slot))
        res, _keep, _arr = self._inflight.pop(slot)
        return tuple(res)

    # ---- the packed boundary (rc_packed_batch) ----
    def pack_bases(self, arena, bases=None, exc_pos=None, exc_chr=None):
        """2-bit codes + the letters outside ACGT of a byte arena (rc_pack_bases).  Returns (bases, exc_pos, exc_chr):
        the given arrays (e.g. page-locked ones) or new ones, the exception arrays cut to their length."""
        arena = self._arena(arena, "arena")
        nw = (arena.size + 15) // 16
        if bases is None:
            bases = np.zeros(nw, dtype=np.uint32)
        cap = 0 if exc_pos is None else len(exc_pos)
        n = self._L.rc_pack_bases(arena.ctypes.data, 0, arena.size, bases.ctypes.data, exc_pos.ctypes.data if cap else None,
                                  exc_chr.ctypes.data if cap else None, cap)
        if n > cap:   # (first call without room, or too little of it)
            exc_pos, exc_chr = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint8)
            self._L.rc_pack_bases(arena.ctypes.data, 0, arena.size, bases.ctypes.data, exc_pos.ctypes.data, exc_chr.ctypes.data, n)
        if exc_pos is None:
            exc_pos, exc_chr = np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.uint8)
        return bases, exc_pos[:n], exc_chr[:n]

    def submit_packed(self, slot, mode, nbytes, off, bases, qual_bits, exc_pos, exc_chr, res=None, fix_pos=None, fix_chr=None, fix_cap=None):
        """rc_submit_packed: off over the whole arena (mode 1: first mates then second mates, 2 n + 1 entries).
        wait_packed(slot) returns (ret, l, m, h, fix_pos, fix_chr)."""
        off = np.ascontiguousarray(off, dtype=np.uint32)
        total = len(off) - 1
        if res is None:
            res = [np.zeros(total, dtype=np.int32) for _ in range(4)]
        if fix_pos is None:
            fix_cap = int(nbytes) if fix_cap is None else fix_cap
            fix_pos, fix_chr = np.zeros(fix_cap, dtype=np.uint32), np.zeros(fix_cap, dtype=np.uint8)
        b = _PackedBatch()
        b.mode, b.n, b.nbytes = mode, (total // 2 if mode == 1 else total), int(nbytes)
        b.off, b.bases = off.ctypes.data, bases.ctypes.data
        b.qual_bits = None if qual_bits is None else qual_bits.ctypes.data
        b.n_exc = len(exc_pos)
        b.exc_pos, b.exc_chr = (exc_pos.ctypes.data, exc_chr.ctypes.data) if len(exc_pos) else (None, None)
        b.ret, b.l, b.m, b.h = (r.ctypes.data for r in res)
        b.fix_pos, b.fix_chr, b.fix_cap = fix_pos.ctypes.data, fix_chr.ctypes.data, len(fix_pos)
        self._ck(self._L.rc_submit_packed(self._h, C.byref(b), slot))
        if not hasattr(self, "_inflight_packed"):
            self._inflight_packed = {}
        self._inflight_packed[slot] = (b, res, fix_pos, fix_chr, (off, bases, qual_bits, exc_pos, exc_chr))

    def wait_packed(self, slot):
        self._ck(self._L.rc_wait_packed(self._h, slot))
        b, res, fix_pos, fix_chr, _keep = self._inflight_packed.pop(slot)
        return tuple(res) + (fix_pos[:b.n_fix], fix_chr[:b.n_fix])

    def set_slot_lanes(self, on=True):
        """slots > 0 of the asynchronous entry points in contexts of their own (kernels of batches in flight overlap) or not"""
        self._ck(self._L.rc_set_slot_lanes(self._h, 1 if on else 0))

    def device_memory(self):
        """(free, total) bytes of the context's GPU memory right now"""
        f, t = C.c_uint64(0), C.c_uint64(0)
        self._ck(self._L.rc_device_memory(self._h, C.byref(f), C.byref(t)))
        return f.value, t.value

    # ---- reads the k-mer counter kept in HBM (rc_resident_batch) ----
    def count_keep(self, on=True):
        self._ck(self._L.rc_table_count_keep(self._h, 1 if on else 0))

    def count_arenas(self):
        """bytes of the arenas the counter kept, in the order they were added"""
        n = C.c_size_t(0)
        self._ck(self._L.rc_table_count_arenas(self._h, C.byref(n), None, 0))
        b = np.zeros(max(1, n.value), dtype=np.uint64)
        self._ck(self._L.rc_table_count_arenas(self._h, C.byref(n), b.ctypes.data, len(b)))
        return b[:n.value]

    def count_park(self):
        """rc_table_count_park: the arenas added since count_begin become kept arenas; nothing is counted, no table built"""
        self._ck(self._L.rc_table_count_park(self._h))

    def count_finish_sharded(self, others, min_count=2):
        """rc_table_count_finish_sharded over [self] + others (contexts with open counting sessions, one per GPU): the
        table is built in this context; returns the number of entries kept"""
        hs = (C.c_void_p * (1 + len(others)))(self._h, *[o._h for o in others])
        n = C.c_int64(0)
        self._ck(self._L.rc_table_count_finish_sharded(hs, 1 + len(others), min_count, C.byref(n)))
        return n.value

    def count_release(self):
        self._ck(self._L.rc_table_count_release(self._h))

    def submit_resident(self, slot, mode, off, qual_bits, arena_a, begin_a, bytes_a, arena_b=0, begin_b=0, bytes_b=0, res=None, fix_pos=None,
                        fix_chr=None, fix_cap=None):
        """rc_submit_resident: the batch is a byte range of kept arena `arena_a` (mode 1: + one of `arena_b`); off over the
        batch's own arena (bytes_a + bytes_b bytes).  wait_resident(slot) returns (ret, l, m, h, fix_pos, fix_chr)."""
        off = np.ascontiguousarray(off, dtype=np.uint32)
        total = len(off) - 1
        if res is None:
            res = [np.zeros(total, dtype=np.int32) for _ in range(4)]
        if fix_pos is None:
            fix_cap = int(bytes_a + bytes_b) if fix_cap is None else fix_cap
            fix_pos, fix_chr = np.zeros(fix_cap, dtype=np.uint32), np.zeros(fix_cap, dtype=np.uint8)
        b = _ResidentBatch()
        b.mode, b.n = mode, (total // 2 if mode == 1 else total)
        b.arena_a, b.begin_a, b.bytes_a = int(arena_a), int(begin_a), int(bytes_a)
        b.arena_b, b.begin_b, b.bytes_b = int(arena_b), int(begin_b), int(bytes_b)
        b.off = off.ctypes.data
        b.qual_bits = None if qual_bits is None else qual_bits.ctypes.data
        b.ret, b.l, b.m, b.h = (r.ctypes.data for r in res)
        b.fix_pos, b.fix_chr, b.fix_cap = fix_pos.ctypes.data, fix_chr.ctypes.data, len(fix_pos)
        self._ck(self._L.rc_submit_resident(self._h, C.byref(b), slot))
        if not hasattr(self, "_inflight_resident"):
            self._inflight_resident = {}
        self._inflight_resident[slot] = (b, res, fix_pos, fix_chr, (off, qual_bits))

    def wait_resident(self, slot):
        self._ck(self._L.rc_wait_resident(self._h, slot))
        b, res, fix_pos, fix_chr, _keep = self._inflight_resident.pop(slot)
        return tuple(res) + (fix_pos[:b.n_fix], fix_chr[:b.n_fix])

    def apply_fixes(self, arena, fix_pos, fix_chr):
        arena = self._arena(arena, "arena")
        fp, fc = np.ascontiguousarray(fix_pos, dtype=np.uint32), np.ascontiguousarray(fix_chr, dtype=np.uint8)
        self._L.rc_apply_fixes(arena.ctypes.data, fp.ctypes.data, fc.ctypes.data, len(fp))

    def host_array(self, n, dtype=np.uint8):
        """A page-locked numpy array (rc_host_alloc): rc_submit DMAs straight from / into it.
        Freed with the context."""
        dt = np.dtype(dtype)
        p = C.c_void_p(0)
        self._ck(self._L.rc_host_alloc(self._h, max(1, n * dt.itemsize), C.byref(p)))
        if not hasattr(self, "_pinned"):
            self._pinned = []
        self._pinned.append(p.value)
        buf = (C.c_uint8 * (n * dt.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=dt, count=n)


    def correct_device(self, mode, n_reads, nbytes, max_read_len, d_seq, d_qual, d_off, d_ret, d_l, d_m, d_h):
        b = _DeviceBatch(mode, n_reads, nbytes, max_read_len, _ptr(d_seq), _ptr(d_qual), _ptr(d_off),
                         _ptr(d_ret), _ptr(d_l), _ptr(d_m), _ptr(d_h))
        self._ck(self._L.rc_correct_device(self._h, C.byref(b)))

    def strong_threshold_device(self, d_seq, d_off, n_reads, nbytes, max_read_len, d_strong):
        self._ck(self._L.rc_strong_threshold_device(self._h, _ptr(d_seq), _ptr(d_off), n_reads, nbytes, max_read_len, _ptr(d_strong)))

    # ---- one read per call (ErrorCorrection.h:26-28) ----
    def strong_threshold_read(self, seq):
        """GetStrongTrustedThreshold of one read (bytes)."""
        out = C.c_int32(0)
        self._ck(self._L.rc_strong_threshold_read(self._h, bytes(seq), C.byref(out)))
        return out.value

    def correct_read(self, seq, qual, pair_strong_threshold=-1):
        """ErrorCorrection of one read: returns (return value, corrected read as bytes)."""
        buf = C.create_string_buffer(bytes(seq))
        out = C.c_int32(0)
        self._ck(self._L.rc_correct_read(self._h, buf, None if qual is None else bytes(qual), int(pair_strong_threshold), C.byref(out)))
        return out.value, buf.value

    def kmer_info_read(self, seq):
        """GetKmerInformation of one read: (l, m, h)."""
        l, m, h = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._ck(self._L.rc_kmer_info_read(self._h, bytes(seq), C.byref(l), C.byref(m), C.byref(h)))
        return l.value, m.value, h.value

    def probe_device(self, d_seq, nbytes, d_counts):
        self._ck(self._L.rc_probe_device(self._h, _ptr(d_seq), nbytes, _ptr(d_counts)))

    def sync(self):
        self._ck(self._L.rc_sync(self._h))

    # ---- measurement ----
    def profile(self, on=True):
        """on: False/0 off, True/1 kernel timers, 2 also the instrumented correction kernel."""
        self._ck(self._L.rc_profile_enable(self._h, int(on)))

    def profile_correct_counters(self):
        """(reads handed to the correction kernel, their gather rounds, table buckets read) since the last reset."""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._ck(self._L.rc_profile_correct_counters(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def profile_read_rounds(self, d_rounds=None):
        """int32 device tensor (one entry per read, zeroed) that the instrumented correction kernel fills with every read's
        gather rounds; None switches it off."""
        self._ck(self._L.rc_profile_read_rounds(self._h, None if d_rounds is None else d_rounds.data_ptr()))

    def profile_reset(self):
        self._ck(self._L.rc_profile_reset(self._h))

    def profile_get(self, kernel):
        ms, n = C.c_double(0), C.c_uint64(0)
        self._ck(self._L.rc_profile_get(self._h, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def selftest_get_bound(self, c, error_rate):
        """GetBound(c) as the kernels evaluate it: (int32 array, float64 array)."""
        c = np.ascontiguousarray(c, dtype=np.int32)
        oi = np.zeros(len(c), dtype=np.int32)
        od = np.zeros(len(c), dtype=np.float64)
        self._ck(self._L.rc_selftest_get_bound(self._h, c.ctypes.data, len(c), error_rate, oi.ctypes.data, od.ctypes.data))
        return oi, od

    def summary(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._ck(self._L.rc_summary(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value
