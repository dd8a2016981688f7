// rc_pool.h -- the host runtime shared by the units of the `rcorrector` CLI: run-wide settings and RC_TIMING counters,
// die(), the persistent helper-thread pool, NUMA binding, huge-page buffers (Buf / PinBuf / OutBuf) and the run-time binding
// to libdeflate.  No reference counterpart: the reference's host side is one thread per file handle (File.h, Reads.h).
#pragma once
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rcorrector_amd.h"

#define MAX_READ_FILE 100    // Reads.h:11
#define MAX_READ_LENGTH 1024 // utils.h:7  (fgets buffer: 1023 characters + NUL)
#define MAX_ID_LENGTH 2048   // utils.h:8

extern bool g_stdout;
extern bool g_verbose;   // -verbose: the reference's per-read transcript on stdout
extern int g_trace_iter; // threshold iterations recorded per read under -verbose
extern bool g_timing;    // RC_TIMING=1: phase timings on stderr (off by default: stderr is part of the contract)
extern int g_threads;
extern size_t g_deflate_threads;  // helper threads that deflate the slices of .gz outputs (0: no such output)
extern bool g_packed;    // -packed / RC_TRANSPORT=packed: batches cross PCIe through rc_submit_packed (2-bit bases, quality bits, fix list)
extern bool g_gz_whole;  // one-pass runs: a .gz input is inflated whole, in memory, by libdeflate (Source::inflate_whole)

extern double g_w_reader, g_w_writer, g_w_worker;  // RC_TIMING: time blocked on the neighbouring stage
extern double g_t_read, g_t_pack, g_t_gpu, g_t_format, g_t_write;  // RC_TIMING stage totals (thread-seconds)
// RC_TIMING: inside take_records (all files): pread, newline scan, line index.  The reader thread and the mate thread of a
// paired input add to them concurrently.
extern double g_t_fill, g_t_nl, g_t_idx;
void timing_add(double &acc, double dt);
double now_s();
// RC_TIMING with RC_T0=<seconds since the epoch at which the caller started this process>: where the process is on the
// caller's clock (process start, HIP initialisation and the exit are outside the phases the other lines time)
void stamp(const char *what);
void die(const char *fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));

// Persistent helper threads for the data-parallel pieces of the host pipeline (block reads, newline scans,
// packing, formatting): creating and joining a few dozen threads per call, dozens of calls per batch, costs more
// than some of the pieces themselves.  run(T, fn) executes fn(0) .. fn(T-1), fn(0) on the calling thread, and
// returns when all are done; any number of threads may call it at once (the helpers serve one queue).
struct Pool {
    struct Call {
        size_t left = 0;  // guarded by m: the caller may destroy the Call as soon as it has seen 0 under the lock
        std::mutex m;
        std::condition_variable c;
    };
    struct Task {
        const std::function<void(size_t)> *fn;
        size_t idx;
        Call *call;
    };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Task> q;
    std::vector<std::thread> th;
    bool stop = false;
    void start(size_t n)
    {
        for (size_t i = th.size(); i < n; ++i)
            th.emplace_back([this]() {
                for (;;) {
                    Task t;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || !q.empty(); });
                        if (q.empty()) return;
                        t = q.front();
                        q.pop_front();
                    }
                    (*t.fn)(t.idx);
                    {   // decrement and notify under the call's mutex: run() cannot return (and free the Call on its
                        // stack) between the two, it needs the mutex to leave its wait
                        std::lock_guard<std::mutex> lk(t.call->m);
                        if (--t.call->left == 0) t.call->c.notify_all();
                    }
                }
            });
    }
    void run(size_t T, const std::function<void(size_t)> &fn)
    {
        if (T <= 1 || th.empty()) {
            for (size_t t = 0; t < T; ++t) fn(t);
            return;
        }
        Call call;
        call.left = T - 1;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t t = 1; t < T; ++t) q.push_back(Task{&fn, t, &call});
        }
        cv.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(call.m);
        call.c.wait(lk, [&] { return call.left == 0; });
    }
    void stop_and_join()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto &x : th) x.join();
        th.clear();
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto &x : th) x.join();
    }
};
extern Pool g_pool;

// binds the calling thread (and the threads it creates from now on) to the CPUs of one NUMA node that it is allowed
// to run on already (taskset / a scheduler's pinning is narrowed, never widened); memory it touches first then comes
// from that node too.  Returns false if the node's CPU list cannot be read or shares no CPU with the current mask.
// (The helper threads of g_pool are shared by all GPUs' workers and stay unbound.)
bool bind_to_numa_node(int node);
// The NUMA node of the first GPU this process may open, from sysfs alone (the KFD topology, the render nodes' permissions, the
// PCI device's numa_node) -- what rc_device_numa_node says about device 0 once HIP is up, known before it is.  -1: the device
// has no node (one-node hosts); -2: not known (several GPUs and a *_VISIBLE_DEVICES variable that may reorder them, no KFD topology, ...).
int first_gpu_numa_node();

template <class F>
inline void parallel_for(size_t n, F fn)
{
    const size_t T = std::min<size_t>((size_t)g_threads, n ? (n + 4095) / 4096 : 1);
    if (T <= 1) {
        fn((size_t)0, n);
        return;
    }
    g_pool.run(T, [&](size_t t) { fn(n * t / T, n * (t + 1) / T); });
}

// Buffers of megabytes come straight from mmap with transparent huge pages asked for (the host's THP mode is "madvise"):
// a run touches tens of GB of fresh memory -- the text of every batch, arenas, output slices -- and with 4 KB pages the
// page faults of the threads that fill them and the unmapping at the end (0.3 s per 10 GB after _exit) are a visible share
// of a run that takes two seconds.
static const size_t BIG = (size_t)4 << 20;
void *big_alloc(size_t n, size_t *cap);

// growable byte buffer without value-initialisation (a std::vector<char> zero-fills on resize,
// which at GB/s rates is a pass over memory of its own); contents survive growth
struct Buf {
    char *p = nullptr;
    size_t cap = 0;
    bool big = false;
    Buf() = default;
    Buf(const Buf &) = delete;
    Buf &operator=(const Buf &) = delete;
    Buf(Buf &&o) noexcept : p(o.p), cap(o.cap), big(o.big)
    {
        o.p = nullptr;
        o.cap = 0;
        o.big = false;
    }
    ~Buf() { release(); }
    void release()
    {
        if (big)
            munmap(p, cap);
        else
            free(p);
        p = nullptr;
        cap = 0;
        big = false;
    }
    void swap(Buf &o)
    {
        std::swap(p, o.p);
        std::swap(cap, o.cap);
        std::swap(big, o.big);
    }
    char *data() { return p; }
    const char *data() const { return p; }
    void need(size_t n)
    {
        if (n <= cap) return;
        const size_t nc = std::max(n, cap + cap / 2);
        if (nc >= BIG) {
            if (big) {  // (moves page tables, not bytes)
                const size_t c = (nc + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
                void *q = mremap(p, cap, c, MREMAP_MAYMOVE);
                if (q == MAP_FAILED) {
                    fprintf(stderr, "rcorrector: out of memory (%zu bytes)\n", c);
                    exit(1);
                }
                (void)madvise(q, c, MADV_HUGEPAGE);
                p = (char *)q;
                cap = c;
                return;
            }
            size_t c = 0;
            char *q = (char *)big_alloc(nc, &c);
            if (cap) memcpy(q, p, cap);
            free(p);
            p = q;
            cap = c;
            big = true;
            return;
        }
        p = (char *)realloc(p, nc);
        if (!p) {
            fprintf(stderr, "rcorrector: out of memory (%zu bytes)\n", nc);
            exit(1);
        }
        cap = nc;
    }
};

// A batch arena the DMA engines read and write directly: ordinary memory, page-locked through
// the library (rc_host_register) whenever it is (re)allocated.  Jobs are recycled through a pool,
// so the registration is paid a handful of times per run.
struct PinBuf {
    char *p = nullptr;
    size_t cap = 0;
    bool pinned = false, big = false;
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
    ~PinBuf() { release(); }
    void release()
    {
        if (pinned) rc_host_unregister(p);
        if (big)
            munmap(p, cap);
        else
            free(p);
        p = nullptr;
        cap = 0;
        pinned = big = false;
    }
    char *data() { return p; }
    const char *data() const { return p; }
    void need(size_t n)
    {
        if (n <= cap) return;
        const size_t want = std::max(n + (n >> 3) + (1u << 16), cap + cap / 2);
        release();  // (the old content is never needed: an arena is packed from scratch)
        if (want >= BIG) {
            p = (char *)big_alloc(want, &cap);
            big = true;
        } else {
            cap = (want + 4095) & ~(size_t)4095;
            p = (char *)aligned_alloc(4096, cap);
            if (!p) {
                fprintf(stderr, "rcorrector: out of memory (%zu bytes)\n", cap);
                exit(1);
            }
        }
        pinned = rc_host_register(p, cap) == 0;  // not pinned: the library stages the copy
    }
};

// the formatted records of a slice of a batch: a Buf with a length (big slices are huge-page mappings that go back to the
// system when the job retires; as std::vector<char> they sat in the malloc heap -- gigabytes of 4 KB pages -- until exit)
struct OutBuf {
    Buf b;
    size_t n = 0;
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    char *data() { return b.p; }
    const char *data() const { return b.p; }
    void clear() { n = 0; }
    void reserve(size_t c) { b.need(c); }
    void resize(size_t c)
    {
        b.need(c);
        n = c;
    }
    void swap(OutBuf &o)
    {
        b.swap(o.b);
        std::swap(n, o.n);
    }
};

// libdeflate, where the system has it (libdeflate.so.0, looked up at run time: the image carries the library without its
// header): whole-buffer inflate and deflate two to three times as fast as zlib's streams.  The bytes of a .gz OUTPUT differ
// from zlib's (and from the reference's single stream) -- their content does not, which is what the format promises and the
// tests compare; a .gz INPUT decompresses to the same bytes or the file is read again with zlib.  RC_LIBDEFLATE=0: zlib only.
struct LibDeflate {
    void *h = nullptr;
    void *(*alloc_d)() = nullptr;
    int (*gunzip_ex)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;
    void (*free_d)(void *) = nullptr;
    void *(*alloc_c)(int) = nullptr;
    size_t (*gzip)(void *, const void *, size_t, void *, size_t) = nullptr;
    size_t (*gzip_bound)(void *, size_t) = nullptr;
    void (*free_c)(void *) = nullptr;
    bool ok = false;
    LibDeflate()
    {
        const char *e = getenv("RC_LIBDEFLATE");
        if (e && !strcmp(e, "0")) return;
        h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc_d = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        gunzip_ex = (int (*)(void *, const void *, size_t, void *, size_t, size_t *, size_t *))dlsym(h, "libdeflate_gzip_decompress_ex");
        free_d = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        alloc_c = (void *(*)(int))dlsym(h, "libdeflate_alloc_compressor");
        gzip = (size_t(*)(void *, const void *, size_t, void *, size_t))dlsym(h, "libdeflate_gzip_compress");
        gzip_bound = (size_t(*)(void *, size_t))dlsym(h, "libdeflate_gzip_compress_bound");
        free_c = (void (*)(void *))dlsym(h, "libdeflate_free_compressor");
        ok = alloc_d && gunzip_ex && free_d && alloc_c && gzip && gzip_bound && free_c;
    }
};
const LibDeflate &libdeflate();
