// rc_main.cpp -- `rcorrector`: stage 3 of run_rcorrector.pl on MI355X.  Same command line, stderr
// lines, output file names and *.cor.fq bytes as the reference's main.cpp (paths below are relative
// to /root/reference), with the per-read work done by librcorrector_amd.so (HIP) through its C ABI.
//
//   flags        main.cpp:50-71,165-268     -r/-p/-i/-c/-k/-od/-t/-maxcor/-maxcorK/-wk/-stdout/-verbose
//   file typing  Reads.h:108-162            first byte '>' FASTA, '@' FASTQ; ".gz" by the last two chars
//   output name  Reads.h:39-75,140-157      <od>/<name minus last extension>.cor.f[aq][.gz]
//   record       Reads.h:224-266,360-421    4 lines in; "<id> l:%d m:%d h:%d[ cor| unfixable_error]" out
//   batching     main.cpp:439-523           batches never span files; mates travel together
//
// Host pipeline: one reader thread cuts the input into batches of whole records
// (block reads, memchr line index, parallel packing into the SoA arenas of the C ABI), one worker
// threads per GPU run rc_submit / rc_wait on page-locked arenas, one writer thread formats in parallel and writes in input order.
//
// Extra flags (not in the reference): -gpus N shards batches over N GPUs (table replicated),
// -batch N sets the reads per batch, -inflight N the batches in flight per GPU (worker threads sharing
// the GPU's context through rc_submit / rc_wait slots: upload, kernels and download of consecutive
// batches overlap, and the GPU stays busy while a batch's slowest reads finish).  -t sets the host threads used
// for packing / formatting.
// -verbose prints the reference's per-read transcript (the -t 1 order) from rc_correct_batch_traced;
// -write-dump FILE keeps the k-mer table as jellyfish-dump text; without -c the k-mers are counted here.
#include <dlfcn.h>
#include <fcntl.h>
#include <malloc.h>
#include <sched.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <emmintrin.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rcorrector_amd.h"

#define MAX_READ_FILE 100    // Reads.h:11
#define MAX_READ_LENGTH 1024 // utils.h:7  (fgets buffer: 1023 characters + NUL)
#define MAX_ID_LENGTH 2048   // utils.h:8

static bool g_stdout = false;
static bool g_verbose = false;   // -verbose: the reference's per-read transcript on stdout
static int g_trace_iter = 64;    // threshold iterations recorded per read under -verbose
static bool g_timing = false;  // RC_TIMING=1: phase timings on stderr (off by default: stderr is part of the contract)
static int g_threads = 8;
static size_t g_deflate_threads = 0;  // helper threads that deflate the slices of .gz outputs (0: no such output)
static bool g_packed = false;  // -packed / RC_TRANSPORT=packed: batches cross PCIe through rc_submit_packed (2-bit bases, quality bits, fix list)

static double g_w_reader = 0, g_w_writer = 0, g_w_worker = 0;  // RC_TIMING: time blocked on the neighbouring stage
static double g_t_read = 0, g_t_pack = 0, g_t_gpu = 0, g_t_format = 0, g_t_write = 0;  // RC_TIMING stage totals (thread-seconds)
// RC_TIMING: inside take_records (all files): pread, newline scan, line index.  The reader thread and the mate thread of a
// paired input add to them concurrently.
static std::mutex g_t_mu;
static double g_t_fill = 0, g_t_nl = 0, g_t_idx = 0;
static void timing_add(double &acc, double dt)
{
    if (!g_timing) return;
    std::lock_guard<std::mutex> lk(g_t_mu);
    acc += dt;
}

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// RC_TIMING with RC_T0=<seconds since the epoch at which the caller started this process>: where the process is on the
// caller's clock (process start, HIP initialisation and the exit are outside the phases the other lines time)
static void stamp(const char *what)
{
    static const char *e = getenv("RC_T0");
    if (!g_timing || !e) return;
    const double t = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    fprintf(stderr, "[rc timing] +%.3f s %s\n", t - atof(e), what);
}

static void die(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fflush(NULL);
    _exit(1);  // (not exit(): it would join the helper threads from whichever thread failed)
}

// Persistent helper threads for the data-parallel pieces of the host pipeline (block reads, newline scans,
// packing, formatting): creating and joining a few dozen threads per call, dozens of calls per batch, costs more
// than some of the pieces themselves.  run(T, fn) executes fn(0) .. fn(T-1), fn(0) on the calling thread, and
// returns when all are done; any number of threads may call it at once (the helpers serve one queue).
#include <atomic>
#include <functional>
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
static Pool g_pool;

// binds the calling thread (and the threads it creates from now on) to the CPUs of one NUMA node that it is allowed
// to run on already (taskset / a scheduler's pinning is narrowed, never widened); memory it touches first then comes
// from that node too.  Returns false if the node's CPU list cannot be read or shares no CPU with the current mask.
// (The helper threads of g_pool are shared by all GPUs' workers and stay unbound.)
static bool bind_to_numa_node(int node)
{
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *fp = fopen(path, "r");
    if (!fp) return false;
    char buf[4096];
    const bool ok = fgets(buf, sizeof buf, fp) != nullptr;
    fclose(fp);
    if (!ok) return false;
    cpu_set_t set, cur;
    CPU_ZERO(&set);
    CPU_ZERO(&cur);
    const bool have_cur = sched_getaffinity(0, sizeof cur, &cur) == 0;
    int n_cpu = 0;
    for (char *p = buf; *p;) {  // "0-63,128-191"
        char *e;
        long a = strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        if (*e == '-') b = strtol(e + 1, &e, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) {
            if (have_cur && !CPU_ISSET((int)c, &cur)) continue;
            CPU_SET((int)c, &set);
            ++n_cpu;
        }
        p = *e == ',' ? e + 1 : e;
        if (*e != ',') break;
    }
    return n_cpu > 0 && sched_setaffinity(0, sizeof set, &set) == 0;
}

template <class F>
static void parallel_for(size_t n, F fn)
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
static void *big_alloc(size_t n, size_t *cap)
{
    const size_t c = (n + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    void *p = mmap(nullptr, c, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) {
        fprintf(stderr, "rcorrector: out of memory (%zu bytes)\n", c);
        exit(1);
    }
    (void)madvise(p, c, MADV_HUGEPAGE);
    *cap = c;
    return p;
}

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
static const LibDeflate &libdeflate()
{
    static LibDeflate L;
    return L;
}

static bool g_gz_whole = false;  // one-pass runs: a .gz input is inflated whole, in memory, by libdeflate (Source::inflate_whole)

// ---- input: a stream of bytes cut into blocks of whole records --------------------------------
struct Source {
    std::string path;
    bool is_gz = false, seekable = false;
    int fd = -1;
    gzFile gz = nullptr;
    Buf left;  // bytes read from the file but not handed out yet (the tail behind the last block)
    size_t left_len = 0;
    off_t pos = 0;  // file offset of the next unread byte (seekable files)
    bool eof = false;
    double per_line = 0;  // bytes per line of the last block: sizes the next block's buffer in one go
    // .gz, one-pass runs with libdeflate: the file's whole text, inflated at the first large request (the text of such a run
    // stays in memory anyway); `served` = bytes handed out so far, by zlib before that (the peek at the head of the file)
    size_t served = 0;
    bool whole_tried = false, whole = false;
    Buf dec;
    size_t dec_len = 0;

    bool inflate_bgzf(const LibDeflate &LD, const unsigned char *c, size_t csize)
    {
        struct Blk {
            size_t in, out;
            uint32_t bsize, isize;
        };
        std::vector<Blk> blk;
        size_t pos = 0, out = 0;
        while (pos < csize) {
            if (csize - pos < 26 || c[pos] != 0x1f || c[pos + 1] != 0x8b || c[pos + 2] != 8 || !(c[pos + 3] & 4)) return false;
            const size_t xlen = (size_t)c[pos + 10] | ((size_t)c[pos + 11] << 8);
            if (pos + 12 + xlen > csize) return false;
            size_t bsize = 0;
            for (size_t x = pos + 12; x + 4 <= pos + 12 + xlen;) {
                const size_t slen = (size_t)c[x + 2] | ((size_t)c[x + 3] << 8);
                if (c[x] == 'B' && c[x + 1] == 'C' && slen == 2 && x + 6 <= pos + 12 + xlen) bsize = ((size_t)c[x + 4] | ((size_t)c[x + 5] << 8)) + 1;
                x += 4 + slen;
            }
            if (bsize < 12 + xlen + 8 || pos + bsize > csize) return false;
            const uint32_t isize = (uint32_t)c[pos + bsize - 4] | ((uint32_t)c[pos + bsize - 3] << 8) | ((uint32_t)c[pos + bsize - 2] << 16) |
                                   ((uint32_t)c[pos + bsize - 1] << 24);
            if (isize > (1u << 16)) return false;
            blk.push_back(Blk{pos, out, (uint32_t)bsize, isize});
            pos += bsize;
            out += isize;
        }
        if (blk.empty()) return false;
        dec.need(out + 64);
        const size_t T = std::max<size_t>(1, std::min<size_t>((size_t)g_threads * 2, blk.size() / 64 + 1));
        std::vector<char> good(T, 1);
        g_pool.run(T, [&](size_t t) {
            void *d = LD.alloc_d();
            if (!d) {
                good[t] = 0;
                return;
            }
            for (size_t b = blk.size() * t / T; b < blk.size() * (t + 1) / T; ++b) {
                size_t ain = 0, aout = 0;
                char dummy;
                const int res = LD.gunzip_ex(d, c + blk[b].in, blk[b].bsize, blk[b].isize ? dec.p + blk[b].out : &dummy, blk[b].isize, &ain, &aout);
                if (res != 0 || ain != blk[b].bsize || aout != blk[b].isize) {
                    good[t] = 0;
                    break;
                }
            }
            LD.free_d(d);
        });
        for (char g : good)
            if (!g) return false;
        dec_len = out;
        return true;
    }

    // The whole file through libdeflate: every member, into `dec`.  The size of the text is not known in advance: the last four
    // bytes of a gzip file hold it modulo 2^32 (exactly, for the usual single-member file), so the room is the smallest
    // size with that remainder that is at least three times the compressed size, 4 GiB more whenever that was too little.
    // Anything libdeflate does not like -- not gzip at all (zlib reads such a file as it is), a truncated file, bad data --
    // returns false, and the file is read on through zlib, which owns the reference's behaviour for those.
    bool inflate_whole()
    {
        const LibDeflate &LD = libdeflate();
        if (!LD.ok) return false;
        const int fd2 = ::open(path.c_str(), O_RDONLY);
        if (fd2 < 0) return false;
        struct stat st;
        if (fstat(fd2, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 18) {
            ::close(fd2);
            return false;
        }
        const size_t csize = (size_t)st.st_size;
        Buf comp;
        comp.need(csize + 64);
        {
            const size_t SL = (size_t)8 << 20;
            const size_t T = std::max<size_t>(1, std::min<size_t>((size_t)g_threads, (csize + SL - 1) / SL));
            std::vector<char> good(T, 1);
            g_pool.run(T, [&](size_t t) {
                size_t at = csize * t / T;
                const size_t hi = csize * (t + 1) / T;
                while (at < hi) {
                    const ssize_t n = ::pread(fd2, comp.p + at, hi - at, (off_t)at);
                    if (n <= 0) {
                        good[t] = 0;
                        break;
                    }
                    at += (size_t)n;
                }
            });
            ::close(fd2);
            for (char g : good)
                if (!g) return false;
        }
        const unsigned char *c = (const unsigned char *)comp.p;
        if (c[0] != 0x1f || c[1] != 0x8b) return false;
        const uint64_t isize = (uint64_t)c[csize - 4] | ((uint64_t)c[csize - 3] << 8) | ((uint64_t)c[csize - 2] << 16) | ((uint64_t)c[csize - 1] << 24);
        // BGZF (bgzip, htslib): members of at most 64 KB whose header says how long they are, so the members are found
        // without inflating anything and inflated side by side -- the one kind of .gz several threads can share
        if (inflate_bgzf(LD, c, csize)) return dec_len >= served;
        void *d = LD.alloc_d();
        if (!d) return false;
        size_t in_pos = 0, out_pos = 0;
        bool okay = true;
        while (in_pos < csize) {
            if (csize - in_pos < 18 || c[in_pos] != 0x1f || c[in_pos + 1] != 0x8b) break;  // (what follows the last member is ignored, as gzread does)
            uint64_t room = isize;
            while (room < 3 * (uint64_t)(csize - in_pos)) room += (uint64_t)1 << 32;
            int res = 3;
            size_t ain = 0, aout = 0;
            for (int attempt = 0; attempt < 16 && res == 3; ++attempt, room += (uint64_t)1 << 32) {
                dec.need(out_pos + (size_t)room + 64);
                res = LD.gunzip_ex(d, c + in_pos, csize - in_pos, dec.p + out_pos, (size_t)room, &ain, &aout);  // 3 = not enough room
            }
            if (res != 0 || ain == 0) {
                okay = false;
                break;
            }
            in_pos += ain;
            out_pos += aout;
        }
        LD.free_d(d);
        if (!okay || out_pos < served) return false;
        dec_len = out_pos;
        return true;
    }

    void open(const std::string &p)
    {
        path = p;
        size_t len = p.size();
        is_gz = len >= 2 && p[len - 2] == 'g' && p[len - 1] == 'z';  // File.h:51-55
        if (is_gz) {
            gz = gzopen(p.c_str(), "r");
            if (!gz) die("ERROR: Could not access file %s\n", p.c_str());
            gzbuffer(gz, 1 << 20);
        } else {
            fd = ::open(p.c_str(), O_RDONLY);
            if (fd < 0) die("ERROR: Could not access file %s\n", p.c_str());
            struct stat st;
            seekable = fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
        }
        left_len = 0;
        pos = 0;
        eof = false;
    }
    void close()
    {
        if (gz) gzclose(gz);
        if (fd >= 0) ::close(fd);
        gz = nullptr;
        fd = -1;
    }
    // appends up to `want` bytes of the file at dst; sets eof when the file ends first.  Regular
    // files are read by several threads at once (pread into disjoint slices: the copy out of the
    // page cache is what limits a single reader), streams and .gz by this thread alone.
    size_t fill(char *dst, size_t want)
    {
        size_t got = 0;
        if (is_gz) {
            if (g_gz_whole && !whole_tried && want > ((size_t)1 << 16)) {
                whole_tried = true;
                whole = inflate_whole();
            }
            if (whole) {  // (copied out by several threads, like the block reads of a plain file)
                got = std::min(want, dec_len - served);
                const size_t SL = (size_t)8 << 20;
                const size_t T = std::max<size_t>(1, std::min<size_t>((size_t)g_threads, (got + SL - 1) / SL));
                const char *src = dec.p + served;
                g_pool.run(T, [&](size_t t) { memcpy(dst + got * t / T, src + got * t / T, got * (t + 1) / T - got * t / T); });
                served += got;
                if (served == dec_len) {
                    eof = true;
                    dec.release();
                }
                return got;
            }
            while (got < want) {
                const unsigned ch = (unsigned)std::min<size_t>(want - got, (size_t)1 << 30);
                const int n = gzread(gz, dst + got, ch);
                if (n <= 0) {
                    eof = true;
                    break;
                }
                got += (size_t)n;
            }
            served += got;
            return got;
        }
        if (!seekable) {
            while (got < want) {
                const ssize_t n = ::read(fd, dst + got, want - got);
                if (n <= 0) {
                    eof = true;
                    break;
                }
                got += (size_t)n;
            }
            return got;
        }
        const size_t SL = (size_t)8 << 20;
        const size_t T = std::max<size_t>(1, std::min<size_t>((size_t)g_threads, (want + SL - 1) / SL));
        std::vector<size_t> done(T, 0);
        auto rd = [&](size_t t) {
            const size_t lo = want * t / T, hi = want * (t + 1) / T;
            size_t at = lo;
            while (at < hi) {
                const ssize_t n = ::pread(fd, dst + at, hi - at, pos + (off_t)at);
                if (n <= 0) break;
                at += (size_t)n;
            }
            done[t] = at - lo;
        };
        g_pool.run(T, rd);
        for (size_t t = 0; t < T; ++t) {
            got += done[t];
            if (done[t] < want * (t + 1) / T - want * t / T) {  // the file ended inside this slice
                eof = true;
                break;
            }
        }
        pos += (off_t)got;
        return got;
    }
};

// a batch of raw records: the text plus the start of every line (lines_per_record per record)
struct Block {
    Buf text;
    std::vector<uint32_t> line;  // line i = text[line[i] .. line[i+1]-1), without its '\n'
    size_t records = 0;
    bool unterminated_last = false;  // the file ended without a newline: the last line got one here
    void swap(Block &o)
    {
        text.swap(o.text);
        line.swap(o.line);
        std::swap(records, o.records);
        std::swap(unterminated_last, o.unterminated_last);
    }
};

// positions of the '\n' bytes of p[lo, hi), appended to nl in ascending order
static void find_newlines(const char *p, size_t lo, size_t hi, std::vector<uint32_t> &nl)
{
    const size_t T = std::max<size_t>(1, std::min<size_t>((size_t)g_threads, (hi - lo) >> 22));
    if (T == 1) {
        for (size_t at = lo; at < hi;) {
            const char *q = (const char *)memchr(p + at, '\n', hi - at);
            if (!q) break;
            nl.push_back((uint32_t)(q - p));
            at = (size_t)(q - p) + 1;
        }
        return;
    }
    std::vector<std::vector<uint32_t>> part(T);
    g_pool.run(T, [&](size_t t) {
        const size_t a = lo + (hi - lo) * t / T, b = lo + (hi - lo) * (t + 1) / T;
        part[t].reserve((b - a) / 32 + 16);
        for (size_t at = a; at < b;) {
            const char *q = (const char *)memchr(p + at, '\n', b - at);
            if (!q) break;
            part[t].push_back((uint32_t)(q - p));
            at = (size_t)(q - p) + 1;
        }
    });
    size_t total = nl.size();
    for (auto &v : part) total += v.size();
    nl.reserve(total);
    for (auto &v : part) nl.insert(nl.end(), v.begin(), v.end());
}

// up to max_records whole records from the source (fewer only at end of file), read straight into
// the block's own buffer
static void take_records(Source &s, size_t max_records, int lines_per_record, Block &b)
{
    b.line.clear();
    b.records = 0;
    b.unterminated_last = false;
    const size_t want_lines = max_records * (size_t)lines_per_record;
    std::vector<uint32_t> nl;  // newline positions found so far
    nl.reserve(std::min<size_t>(want_lines, (size_t)1 << 23) + 8);
    size_t have = s.left_len, scanned = 0;
    b.text.need(have + 64);
    if (have) memcpy(b.text.p, s.left.p, have);
    s.left_len = 0;
    for (;;) {
        const double tn0 = now_s();
        find_newlines(b.text.p, scanned, have, nl);
        timing_add(g_t_nl, now_s() - tn0);
        scanned = have;
        if (nl.size() >= want_lines) break;
        if (s.eof) break;
        if (have >= (1ull << 31)) die("ERROR: %s: a batch exceeds 2 GiB of text; lower -batch\n", s.path.c_str());
        size_t want = (size_t)32 << 20;
        if (nl.size() < 64 && s.per_line > 0 && want_lines > nl.size()) {
            want = (size_t)(s.per_line * (double)(want_lines - nl.size()) * 1.01 + 65536.0);
        }
        if (nl.size() >= 64) {  // bytes per line so far -> what the missing lines should need, plus 2 %
            const double per_line = (double)have / (double)nl.size();
            want = (size_t)(per_line * (double)(want_lines - nl.size()) * 1.02) + (1u << 16);
        }
        want = std::min<size_t>(want, ((size_t)1 << 31) - have + 1);
        b.text.need(have + want + 64);
        const double tf0 = now_s();
        have += s.fill(b.text.p + have, want);
        timing_add(g_t_fill, now_s() - tf0);
    }
    size_t n_lines = std::min(nl.size(), want_lines), end;
    if (nl.size() >= want_lines) {
        end = (size_t)nl[want_lines - 1] + 1;
    } else {  // end of file
        end = nl.empty() ? 0 : (size_t)nl.back() + 1;
        if (end < have) {  // last line without '\n' (fgets hands it over as it is): add the newline
            b.text.p[have] = '\n';
            nl.push_back((uint32_t)have);
            ++have;
            end = have;
            ++n_lines;
            b.unterminated_last = true;
        }
        // a record cut short by the end of the file: its missing lines read as empty (fgets leaves "")
        while (n_lines % (size_t)lines_per_record) {
            b.text.need(have + 64);
            b.text.p[have] = '\n';
            nl.push_back((uint32_t)have);
            ++have;
            end = have;
            ++n_lines;
        }
    }
    b.records = n_lines / (size_t)lines_per_record;
    if (n_lines >= 64) s.per_line = (double)end / (double)n_lines;
    if (have > end) {  // the tail behind the block waits in the source for the next call
        s.left.need(have - end);
        memcpy(s.left.p, b.text.p + end, have - end);
    }
    s.left_len = have - end;
    if (b.records == 0) return;
    const double ti0 = now_s();
    b.line.resize(n_lines + 1);
    b.line[0] = 0;
    parallel_for(n_lines, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) b.line[i + 1] = nl[i] + 1;
    });
    timing_add(g_t_idx, now_s() - ti0);
}

struct ReadFile {
    std::string path;
    bool paired = false, interleaved = false, fastq = true, out_gz = false;
    Source src;
    FILE *out = nullptr;  // only its descriptor is used, with pwrite (stdout: fwrite)
    off_t out_off = 0;
    bool wrote = false;
    bool preallocated = false;  // the output's blocks were reserved beyond its final size (open_file)
};

// Reads.h:39-75
static std::string base_name(const std::string &path)
{
    std::string in = path;
    int len = (int)in.size(), i, j;
    for (i = len; i >= 0 && in.c_str()[i] != '.' && in.c_str()[i] != '/'; --i)
        ;
    if (i >= 0 && !strcmp(in.c_str() + i, ".gz")) {
        int tmp = i;
        for (i = i - 1; i >= 0 && in[i] != '.' && in[i] != '/'; --i)
            ;
        std::string stem = in.substr(0, tmp);
        const char *e = i >= 0 ? stem.c_str() + i : "";
        if (!(i >= 0 && (!strcmp(e, ".fastq") || !strcmp(e, ".fasta") || !strcmp(e, ".fq") || !strcmp(e, ".fa")))) i = tmp;
    }
    for (j = len; j >= 0 && in.c_str()[j] != '/'; --j)
        ;
    if (i >= 0 && in.c_str()[i] == '.') return in.substr(j + 1, i - (j + 1));
    return in.substr(j + 1);
}

// Reads.h:108-162: type the file by its first byte, open (truncate) the output
static void open_file(ReadFile &f, const char *path, bool paired, bool interleaved, const std::string &od)
{
    f.path = path;
    f.paired = paired;
    f.interleaved = interleaved;
    f.src.open(path);
    f.src.left.need(4096);  // peek at the head of the file; the bytes stay queued for the first block
    f.src.left_len = f.src.fill(f.src.left.p, 4096);
    const char first = f.src.left_len ? f.src.left.p[0] : 0;
    if (first == '>')
        f.fastq = false;
    else if (first == '@')
        f.fastq = true;
    else {
        std::string l(f.src.left.p, std::min<size_t>(f.src.left_len, 200));
        const size_t nl = l.find('\n');
        if (nl != std::string::npos) l = l.substr(0, nl + 1);
        die("\"%s\"'s format is wrong: %s\n", path, l.c_str());
    }
    size_t len = strlen(path);
    f.out_gz = len >= 2 && path[len - 2] == 'g' && path[len - 1] == 'z';
    std::string outp = od + "/" + base_name(path) + (f.fastq ? ".cor.fq" : ".cor.fa") + (f.out_gz ? ".gz" : "");
    if (g_stdout) {
        f.out = stdout;
        f.out_gz = false;
    } else if (f.out_gz) {
        // compressLevel 1 (Reads.h:84, File.h:62-66).  The formatted slices of a batch are deflated
        // in parallel, each into its own gzip member; a .gz file is a concatenation of members, so
        // gunzip / gzopen read back exactly the bytes the reference's single-stream file holds.
        f.out = fopen(outp.c_str(), "wb");
        if (!f.out) die("ERROR: Could not access file %s\n", outp.c_str());
    } else {
        f.out = fopen(outp.c_str(), "w");
        if (!f.out) die("ERROR: Could not access file %s\n", outp.c_str());
        // the output of a plain input is the input plus a few bytes per record: its blocks are reserved up front
        // (buffered writes into preallocated space: 11.8 GB/s against 10.1 on the GPU box's host, tools/mb/iob2.cpp)
        // -- FALLOC_FL_KEEP_SIZE: the file's length stays what has been written, so a run that ends abnormally leaves a
        // valid prefix and not gigabytes of NUL bytes; a file system without fallocate fails fast (glibc's posix_fallocate
        // would write into every block instead) and the output is simply not preallocated.  The ftruncate at close
        // releases the blocks that were not needed.
        struct stat st;
        if (f.src.seekable && fstat(f.src.fd, &st) == 0 && st.st_size > ((off_t)64 << 20) &&
            fallocate(fileno(f.out), FALLOC_FL_KEEP_SIZE, 0, st.st_size + st.st_size / 8) == 0)
            f.preallocated = true;
    }
}

// the slices of a batch, in order.  One thread per output file: buffered writes to one file are serialised by the
// kernel (the inode's lock), so more writers only add contention -- measured on the GPU box's host: 9.1 GB/s from
// one thread, 8.4 from 32 (tools/mb/iob.cpp); two files written side by side get 14.5 GB/s
static void emit_slices(ReadFile &f, const std::vector<OutBuf> &sl)
{
    size_t total = 0;
    for (const auto &v : sl) total += v.size();
    if (total == 0) return;
    f.wrote = true;
    if (f.out == stdout) {
        for (const auto &v : sl)
            if (!v.empty()) fwrite(v.data(), 1, v.size(), stdout);
        return;
    }
    const int fd = fileno(f.out);
    for (const auto &v : sl) {
        size_t done = 0;
        while (done < v.size()) {
            const ssize_t n = ::pwrite(fd, v.data() + done, v.size() - done, f.out_off + (off_t)done);
            if (n <= 0) die("ERROR: write failed on %s\n", f.path.c_str());
            done += (size_t)n;
        }
        f.out_off += (off_t)v.size();
    }
}

// one gzip member (RFC 1952) holding `in`, deflate level 1
static void gzip_member(const OutBuf &in, OutBuf &out)
{
    const LibDeflate &LD = libdeflate();
    if (LD.ok) {
        static thread_local void *c = nullptr;  // (a compressor per thread: they are not shareable, and cost a few hundred KB)
        if (!c) c = LD.alloc_c(1);
        if (c) {
            out.resize(LD.gzip_bound(c, in.size()) + 64);
            const size_t n = LD.gzip(c, in.data(), in.size(), out.data(), out.size());
            if (n) {
                out.resize(n);
                return;
            }
        }
    }
    z_stream z;
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, 1, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) die("ERROR: zlib deflateInit2 failed\n");
    out.resize(deflateBound(&z, (uLong)in.size()) + 64);
    z.next_in = (Bytef *)in.data();
    z.avail_in = (uInt)in.size();
    z.next_out = (Bytef *)out.data();
    z.avail_out = (uInt)out.size();
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) die("ERROR: zlib deflate failed\n");
    out.resize(z.total_out);
    deflateEnd(&z);
}

// ---- one batch travelling through the pipeline -------------------------------------------------
struct Arena {  // one file's share of a batch
    Block blk;
    int lpr = 4;  // lines per record
    PinBuf seq, qual;
    std::vector<uint32_t> off;
    // resident batches (the reads are in HBM since they were counted): there is no byte arena here, the fixes are applied
    // to the sequence lines of the text itself
    bool seq_in_text = false;
    size_t n() const { return blk.records; }
    const char *sequence(size_t r) const { return seq_in_text ? blk.text.data() + blk.line[r * (size_t)lpr + 1] : seq.data() + off[r]; }
    const char *line(size_t rec, int which, uint32_t *len) const
    {
        const size_t li = rec * (size_t)lpr + (size_t)which;
        *len = blk.line[li + 1] - blk.line[li] - 1;
        return blk.text.data() + blk.line[li];
    }
};

struct Job {
    int file = 0;
    int mode = 0;
    bool fastq = true;
    Arena a, b;
    std::vector<int32_t> ret, l, m, h;
    std::vector<int32_t> tr_before, tr_after, tr_flags, tr_niter, tr_iter;  // -verbose only
    bool resident = false;        // the batch's reads are arenas the k-mer counter kept in HBM (rc_submit_resident)
    int arena_a = 0, arena_b = 0;
    // -packed: the batch as rc_packed_batch wants it (one offset array over both arenas, 2-bit codes, quality bits, the
    // letters outside ACGT) and the room for the fix list
    PinBuf pk_off, pk_bases, pk_qbits, pk_exc_pos, pk_exc_chr, pk_fix_pos, pk_fix_chr;
    std::vector<OutBuf> o1, o2;  // the formatted (and, for .gz, deflated) output records, in slices
    bool done = false;
    int rc = 0;
    std::string err;
};

// Reads.h:224-266 for a whole block: sequence -> NUL-terminated arena, quality cut / padded to the
// sequence length for the kernels (the output prints the quality line verbatim, see put_record)
static uint64_t index_arena(Arena &A, const std::string &path)
{
    const size_t n = A.n();
    A.off.resize(n + 1);
    A.off[0] = 0;
    uint64_t total = 0;
    for (size_t r = 0; r < n; ++r) {
        uint32_t sl, il;
        A.line(r, 1, &sl);
        A.line(r, 0, &il);
        if (sl > MAX_READ_LENGTH - 1)
            die("ERROR: %s: a read of %u bases exceeds the limit of %d (utils.h:7)\n", path.c_str(), sl, MAX_READ_LENGTH - 1);
        if (il > MAX_ID_LENGTH - 1) die("ERROR: %s: a header line longer than %d characters\n", path.c_str(), MAX_ID_LENGTH - 1);
        total += sl + 1;
        A.off[r + 1] = (uint32_t)total;
    }
    if (total >= (1ull << 32)) die("ERROR: batch too large; lower -batch\n");
    return total;
}

// the sequences alone, NUL-terminated, at dst (A.off must be set): what the k-mer counter is given
static void pack_sequences(const Arena &A, char *dst)
{
    parallel_for(A.n(), [&](size_t lo, size_t hi) {
        for (size_t r = lo; r < hi; ++r) {
            uint32_t sl;
            const char *s = A.line(r, 1, &sl);
            char *d = dst + A.off[r];
            memcpy(d, s, sl);
            d[sl] = 0;
        }
    });
}

static void pack_arena(Arena &A, const std::string &path)
{
    const size_t n = A.n();
    const uint64_t total = index_arena(A, path);
    A.seq_in_text = false;
    A.seq.need(total);
    A.qual.need(total);
    parallel_for(n, [&](size_t lo, size_t hi) {
        for (size_t r = lo; r < hi; ++r) {
            uint32_t sl, ql = 0;
            const char *s = A.line(r, 1, &sl);
            char *d = A.seq.data() + A.off[r];
            memcpy(d, s, sl);
            d[sl] = 0;
            char *dq = A.qual.data() + A.off[r];
            uint32_t qc = 0;
            if (A.lpr == 4) {
                const char *q = A.line(r, 3, &ql);
                qc = std::min(ql, sl);
                memcpy(dq, q, qc);
            }
            memset(dq + qc, 0, sl + 1 - qc);
        }
    });
}

// Quality bits of a resident batch straight from the text: bit p of the batch's arena (arena 1's bytes, then arena 2's)
// = the quality character of that base is above the threshold; positions without one (the NUL behind a read, a
// quality line shorter than its sequence: pack_arena pads with 0) compare as 0.  [lo, hi) is a range of arena positions
// that starts and ends at multiples of 8 (or at the arena's end): ranges are packed side by side by different threads.
// Returns false if a read with bases has no first quality character (qual[0] == 0 asks for the byte path,
// ErrorCorrection.cpp:1316).
struct QualView {
    const Arena *A[2];
    size_t bytes1;
    size_t nbytes;
};
static bool pack_quality_bits_from_text(const QualView &V, char bad_q, size_t lo, size_t hi, uint8_t *bits)
{
    bool ok = true;
    uint64_t acc = 0;
    int nacc = 0;
    uint8_t *out = bits + (lo >> 3);
    auto put = [&](uint64_t v, int nb) {  // nb <= 16 bits at a time
        acc |= v << nacc;
        nacc += nb;
        while (nacc >= 8) {
            *out++ = (uint8_t)acc;
            acc >>= 8;
            nacc -= 8;
        }
    };
    const __m128i thr = _mm_set1_epi8(bad_q);
    const bool zero_above = (signed char)0 > (signed char)bad_q;  // (a negative threshold: the padding compares as "good")
    size_t pos = lo;
    while (pos < hi) {
        const int sd = pos >= V.bytes1 ? 1 : 0;
        const Arena &A = *V.A[sd];
        const size_t base = sd ? V.bytes1 : 0, p = pos - base;
        const size_t r = (size_t)(std::upper_bound(A.off.begin(), A.off.begin() + (ptrdiff_t)A.n() + 1, (uint32_t)p) - A.off.begin()) - 1;
        const size_t end_side = std::min(hi, sd ? V.nbytes : V.bytes1);
        for (size_t rr = r; rr < A.n() && base + A.off[rr] < end_side; ++rr) {
            const uint32_t sl = A.off[rr + 1] - A.off[rr] - 1;
            uint32_t ql = 0;
            const char *q = A.line(rr, 3, &ql);
            const uint32_t qc = std::min(ql, sl);
            if (sl && (qc == 0 || q[0] == 0)) ok = false;
            // this read's positions inside [lo, hi): characters j0 .. j1 - 1 of its sl + 1 bytes
            const size_t r0 = base + A.off[rr];
            const uint32_t j0 = r0 < pos ? (uint32_t)(pos - r0) : 0;
            const uint32_t j1 = (uint32_t)std::min<size_t>(sl + 1, end_side - r0);
            uint32_t j = j0;
            while (j < j1) {
                const uint32_t nb = std::min<uint32_t>(16, j1 - j);
                uint32_t m = 0;
                if (j < qc) {  // (the text buffer carries 64 bytes of slack behind its last line)
                    m = (uint32_t)_mm_movemask_epi8(_mm_cmpgt_epi8(_mm_loadu_si128((const __m128i *)(q + j)), thr));
                    if (qc - j < 16) {
                        const uint32_t keep = (1u << (qc - j)) - 1u;
                        m = (m & keep) | (zero_above ? (0xffffu & ~keep) : 0u);
                    }
                } else if (zero_above) {
                    m = 0xffffu;
                }
                put(m & ((1u << nb) - 1u), (int)nb);
                j += nb;
            }
            pos = r0 + j1;
        }
        if (pos < end_side) pos = end_side;  // (cannot happen: the reads tile the arena)
    }
    if (nacc) *out = (uint8_t)acc;
    return ok;
}

// the substitutions of a resident batch, applied to the sequence lines of the text
static void apply_fixes_to_text(Arena &A1, Arena *A2, size_t bytes1, const uint32_t *fix_pos, const uint8_t *fix_chr, size_t lo, size_t hi)
{
    for (size_t q = lo; q < hi; ++q) {
        size_t p = fix_pos[q];
        Arena &A = (A2 && p >= bytes1) ? *A2 : A1;
        if (&A == A2) p -= bytes1;
        const size_t n = A.n();
        size_t r;
        const uint32_t stride = A.off[1];
        if ((uint64_t)stride * n == A.off[n] && A.off[p / stride] == (p / stride) * (size_t)stride && A.off[p / stride + 1] == (p / stride + 1) * (size_t)stride)
            r = p / stride;  // reads of one length
        else
            r = (size_t)(std::upper_bound(A.off.begin(), A.off.begin() + (ptrdiff_t)n + 1, (uint32_t)p) - A.off.begin()) - 1;
        A.blk.text.p[A.blk.line[r * (size_t)A.lpr + 1] + (p - A.off[r])] = (char)fix_chr[q];
    }
}

static inline char *put_int(char *p, int v)
{
    char tmp[16];
    int n = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do {
        tmp[n++] = (char)('0' + u % 10);
        u /= 10;
    } while (u);
    if (v < 0) *p++ = '-';
    while (n) *p++ = tmp[--n];
    return p;
}

// Reads.h:360-421: one record.  The quality line is printed as fgets left it in the reference:
// stripped of its newline only when it is exactly as long as the sequence line (Reads.h:255-262).
template <class B>
static inline void put_record(B &out, const Arena &A, size_t r, bool fastq, int cor, int l, int m, int h)
{
    uint32_t il, ql = 0;
    const char *id = A.line(r, 0, &il);
    const char *seq = A.sequence(r);
    const uint32_t sl = A.off[r + 1] - A.off[r] - 1;
    const char *q = fastq ? A.line(r, 3, &ql) : nullptr;
    const size_t need = (size_t)il + sl + ql + 96;
    const size_t at = out.size();
    out.resize(at + need);
    char *p = out.data() + at;
    memcpy(p, id, il);
    p += il;
    memcpy(p, " l:", 3);
    p = put_int(p + 3, l);
    memcpy(p, " m:", 3);
    p = put_int(p + 3, m);
    memcpy(p, " h:", 3);
    p = put_int(p + 3, h);
    if (cor == -1) {
        memcpy(p, " unfixable_error", 16);
        p += 16;
    } else if (cor > 0) {
        memcpy(p, " cor", 4);
        p += 4;
    }
    *p++ = '\n';
    memcpy(p, seq, sl);
    p += sl;
    *p++ = '\n';
    if (fastq) {
        *p++ = '+';
        *p++ = '\n';
        memcpy(p, q, ql);
        p += ql;
        // fgets kept the quality line's own newline (Reads.h strips it only at index strlen(seq)) --
        // unless this is the last line of a file that does not end with one
        if (ql != sl && !(A.blk.unterminated_last && r + 1 == A.n())) *p++ = '\n';
        *p++ = '\n';
    }
    out.resize((size_t)(p - out.data()));
}

// what the reference prints to stdout for one read under -verbose (ErrorCorrection.cpp:686-689,
// 759-770,856-857,1088-1094 and GetKmerInformation :1590-1597), from the data
// rc_correct_batch_traced returns.  gi = the read's index in ret/l/m/h order, ab = offset of its
// arena in the batch's device arena (0, or the size of arena 1 for second mates)
static void put_transcript(std::vector<char> &out, const Job &J, const Arena &A, size_t r, size_t gi, size_t ab, int k)
{
    uint32_t il, ol;
    const char *id = A.line(r, 0, &il);
    const char *orig = A.line(r, 1, &ol);
    const char *seq = A.seq.data() + A.off[r];
    const int len = (int)(A.off[r + 1] - A.off[r] - 1);
    const int kcnt = len >= k ? len - k + 1 : 0;
    const size_t a0 = ab + A.off[r];
    auto put = [&](const char *p, size_t n) { out.insert(out.end(), p, p + n); };
    auto puts_ = [&](const char *p) { put(p, strlen(p)); };
    auto puti = [&](int v) {
        char tmp[16];
        char *e = put_int(tmp, v);
        put(tmp, (size_t)(e - tmp));
    };
    put(id, il);
    puts_("\n");
    if (J.tr_flags[gi] & 1) {
        puts_("Before correction:\n");
        put(orig, (size_t)len);
        puts_("\n");
        for (int i = 0; i < kcnt; ++i) {
            const int c = J.tr_before[a0 + (size_t)i];
            puti(c != 0 ? c : 1);
            puts_(" ");
        }
        puts_("\n");
        const int n_it = J.tr_niter[gi];
        if (n_it > g_trace_iter)
            die("rcorrector: -verbose: read %.*s went through %d threshold iterations, more than the %d recorded; raise -verbose-iter\n",
                (int)il, id, n_it, g_trace_iter);
        for (int it = 0; it < n_it; ++it) {
            const int32_t *e = J.tr_iter.data() + (gi * (size_t)g_trace_iter + (size_t)it) * RC_TRACE_ITER_WORDS;
            puts_("strong trust threshold=");
            puti(e[0]);
            puts_(" threshold=");
            puti(e[1]);
            puts_("\n");
            if (e[2]) {
                puts_("Is corresponding base strong trusted?\n");
                for (int b = 0; b < len; ++b) out.push_back((char)('0' + (((uint32_t)e[4 + (b >> 5)] >> (b & 31)) & 1u)));
                puts_("\n");
            }
        }
    }
    // GetKmerInformation: the counts of the k-mers without a non-ACGT letter, 0 shown as 1
    int bad = 0, n_valid = 0;
    std::vector<int> cnt;
    for (int i = 0; i < len; ++i) {
        const char ch = seq[i];
        const bool ok = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
        bad = ok ? (bad > 0 ? bad - 1 : 0) : k;  // windows ending at i are invalid while bad > 0
        if (i >= k - 1 && bad == 0) {
            const int c = J.tr_after[a0 + (size_t)(i - k + 1)];
            cnt.push_back(c != 0 ? c : 1);
            ++n_valid;
        }
    }
    if (n_valid > 0) {
        puts_("After coorrection:\n");
        put(seq, (size_t)len);
        puts_("\n");
        for (int c : cnt) {
            puti(c);
            puts_(" ");
        }
        puts_("\n");
    }
}

// ---- k-mer counting pass (only without -c); one pass over the input where it fits: what the counting pass read stays ----
// The reference's pipeline reads every file twice -- jellyfish counts the k-mers (run_rcorrector.pl:262-281), stage 3
// corrects -- and so does the counting pass above followed by the correction loop.  When the inputs are plain files that
// fit (text in host memory, bases in HBM), the counting pass cuts them into the correction loop's batches right away:
// the text and its line index stay here, the sequence arenas stay in HBM with the counter (rc_table_count_keep), and the
// loop corrects them where they lie (rc_submit_resident): files are read, parsed and uploaded once.
struct Retained {
    int file = 0, mode = 0;
    bool fastq = true;
    int lpr_a = 4, lpr_b = 4;
    Block a, b;
    std::vector<uint32_t> off_a, off_b;
    int arena_a = 0, arena_b = 0;
};

// keep = false: the counting pass of a run in two passes (.gz inputs, inputs beyond the memory test, several GPUs): the same
// reader -- both mates' files side by side, parallel block reads, page-locked staging -- over sources of its own; the blocks
// are recycled instead of kept, and the counter releases the arenas when it has counted them.
static void ingest_resident(rc_ctx *ctx, std::vector<ReadFile> &files, std::vector<ReadFile> &mates, size_t batch_reads,
                            std::vector<std::unique_ptr<Retained>> &kept, int64_t *stored, bool keep)
{
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::unique_ptr<Retained>> q;
    std::vector<std::unique_ptr<Retained>> spare;  // keep = false: blocks to fill again
    bool done = false;
    std::thread reader([&]() {
        for (size_t fi = 0; fi < files.size(); ++fi) {
            ReadFile &f = files[fi];
            Source own_a, own_b;
            if (!keep) {
                own_a.open(f.path);
                if (f.paired) own_b.open(mates[fi].path);
            }
            Source &src_a = keep ? f.src : own_a, &src_b = keep ? mates[fi].src : own_b;
            for (;;) {
                std::unique_ptr<Retained> R;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (!spare.empty()) {
                        R = std::move(spare.back());
                        spare.pop_back();
                    }
                }
                if (!R) R.reset(new Retained);
                R->file = (int)fi;
                R->mode = f.paired ? 1 : (f.interleaved ? 2 : 0);
                R->fastq = f.fastq;
                R->lpr_a = f.fastq ? 4 : 2;
                R->lpr_b = f.paired ? (mates[fi].fastq ? 4 : 2) : R->lpr_a;
                const double tr0 = now_s();
                R->b.records = 0;
                if (f.paired) {
                    std::thread mate([&]() { take_records(src_b, batch_reads, R->lpr_b, R->b); });
                    take_records(src_a, batch_reads, R->lpr_a, R->a);
                    mate.join();
                    // (two passes: files that are not paired are the correction loop's to refuse, with the reference's message
                    // in the reference's place on stderr; the counter takes whatever reads there are)
                    if (keep && R->b.records != R->a.records) die("ERROR: The files are not paired!\n");
                } else {
                    take_records(src_a, batch_reads, R->lpr_a, R->a);
                }
                if (R->a.records == 0 && R->b.records == 0) break;
                if (keep && R->mode == 2 && (R->a.records & 1)) die("ERROR: interleaved file %s holds an odd number of reads\n", f.path.c_str());
                g_t_read += now_s() - tr0;
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return q.size() < 3; });
                q.emplace_back(std::move(R));
                cv.notify_all();
            }
            if (!keep) {
                own_a.close();
                if (f.paired) own_b.close();
            }
        }
        std::lock_guard<std::mutex> lk(mu);
        done = true;
        cv.notify_all();
    });
    if (rc_table_count_keep(ctx, keep ? 1 : 0) || rc_table_count_begin(ctx)) die("rcorrector: %s\n", rc_last_error(ctx));
    PinBuf stage;  // the sequences of one file's share of a batch on their way to HBM
    int next_arena = 0;
    for (;;) {
        std::unique_ptr<Retained> R;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return done || !q.empty(); });
            if (q.empty()) break;
            R = std::move(q.front());
            q.pop_front();
            cv.notify_all();
        }
        const double tp0 = now_s();
        for (int sd = 0; sd < (R->mode == 1 ? 2 : 1); ++sd) {
            if ((sd ? R->b : R->a).records == 0) continue;  // (keep = false: one mate's file ended before the other's)
            Arena A;  // (a view for index_arena / pack_sequences: the block is swapped in and out)
            A.lpr = sd ? R->lpr_b : R->lpr_a;
            A.blk.swap(sd ? R->b : R->a);
            A.off.swap(sd ? R->off_b : R->off_a);  // (its capacity, when the block is a recycled one)
            const uint64_t total = index_arena(A, sd ? mates[(size_t)R->file].path : files[(size_t)R->file].path);
            stage.need(total + 64);
            pack_sequences(A, stage.data());
            // (an arena without a byte is not kept: cannot happen, every record has at least its NUL)
            if (rc_table_count_add(ctx, stage.data(), total)) die("rcorrector: %s\n", rc_last_error(ctx));
            (sd ? R->arena_b : R->arena_a) = next_arena++;
            (sd ? R->off_b : R->off_a).swap(A.off);
            A.blk.swap(sd ? R->b : R->a);
        }
        g_t_pack += now_s() - tp0;
        if (keep) {
            kept.emplace_back(std::move(R));
        } else {
            std::lock_guard<std::mutex> lk(mu);
            spare.emplace_back(std::move(R));
        }
    }
    reader.join();
    stamp(keep ? "inputs read, indexed and uploaded" : "inputs read and uploaded for the k-mer count");
    if (rc_table_count_finish(ctx, 2, stored)) die("rcorrector: %s\n", rc_last_error(ctx));
    stamp("k-mers counted, table built");
}

// GetBadQuality's two histograms over the records of one block (main.cpp:88-128), at most `room` of them
static void quality_histograms(const Block &b, int lpr, size_t room, std::vector<int32_t> &fh, std::vector<int32_t> &lh, int *total)
{
    static char qbuf[MAX_READ_LENGTH];  // Reads::qual, reused from record to record
    for (size_t r = 0; r < b.records && r < room; ++r) {
        const uint32_t *L = b.line.data() + r * (size_t)lpr;
        const uint32_t sl = L[2] - L[1] - 1, ql = lpr == 4 ? L[4] - L[3] - 1 : 0;
        ++*total;
        if (lpr != 4) continue;
        const char *q = b.text.data() + L[3];
        // qual[strlen(seq)-1] and qual[0] as GetBadQuality sees them: Reads::Next reads every
        // quality line into ONE reused buffer (bytes behind a short line keep what earlier
        // records left there) and strips a newline only at index strlen(seq)
        // (Reads.h:204-219); qual[-1], for an empty sequence, is the last byte of the
        // sequence buffer in front of it, 0.
        const uint32_t qn = std::min<uint32_t>(ql, MAX_READ_LENGTH - 1);
        memcpy(qbuf, q, qn);
        if (qn + 1 < MAX_READ_LENGTH) {
            qbuf[qn] = '\n';
            qbuf[qn + 1] = 0;
        } else {
            qbuf[qn] = 0;
        }
        if (sl < MAX_READ_LENGTH && qbuf[sl] == '\n') qbuf[sl] = 0;
        const unsigned char lastq = sl ? (unsigned char)qbuf[sl - 1] : 0;
        const unsigned char firstq = (unsigned char)qbuf[0];
        ++lh[lastq];
        ++fh[firstq];
    }
}

static void print_help()
{
    fprintf(stderr,
            "Usage: ./rcorrector [OPTIONS]\n"
            "OPTIONS:\n"
            "Required parameters:\n"
            "\t-r seq_file: seq_file is the path to the sequence file. Can use multiple -r to specifiy multiple sequence files\n"
            "\t-p seq_file_left seq_file_right: the paths to the paired-end data set. Can use multiple -p to specifiy multiple sequence files\n"
            "\t-i seq_file: seq_file is the path to the interleaved mate-pair sequence file. Can use multiple -i\n"
            "\t-c jf_dump: the kmer counts dumped by JellyFish (without -c the k-mers of the input files are counted on the GPU)\n"
            "\t-k kmer_length\n"
            "Other parameters:\n"
            "\t-od output_file_directory (default: ./)\n"
            "\t-t number of threads to use (default: 1)\n"
            "\t-maxcor INT: the maximum number of correction every 100bp (default: 8)\n"
            "\t-maxcorK INT: the maximum number of correction within k-bp window (default: 4)\n"
            "\t-wk FLOAT: the proportion of kmers that are used to estimate weak kmer count threshold (default: 0.95)\n"
            "\t-stdout: output the corrected sequences to stdout (default: not used)\n"
            "\t-verbose: output some correction information to stdout (default: not used)\n"
            "MI355X build only:\n"
            "\t(-t: host threads that read, pack, format and write around the GPU; without it, or with -t 1: 16)\n"
            "\t-gpus INT: number of GPUs to shard the reads over, k-mer table replicated (default: 1)\n"
            "\twithout -c the k-mers are counted here (exact counts >= 2); ERROR_RATE is then estimated over this program's own\n"
            "\t\tdump order, not Jellyfish's: self-consistent, not byte-comparable with a jellyfish + reference run on large inputs\n"
            "\t-batch INT: reads per GPU batch (default: 1048576)\n"
            "\t-inflight INT: batches in flight per GPU, 1-4 (default: 2)\n"
            "\t-write-dump STRING: also write the k-mer table as a jellyfish-dump text file\n"
            "\t-verbose-iter INT: threshold iterations recorded per read for -verbose (default: 64)\n");
}

int main(int argc, char **argv)
{
    int k = 23, max_fix_per_k = 4, gpus = 1, inflight = 2, i;
    double wk = 0.95;
    const char *dump = nullptr, *write_dump = nullptr;
    std::string od = "./";
    size_t batch_reads = 1 << 20;
    bool verbose = false;
    int t_flag = 0;
    if (argc == 1) {
        print_help();
        return 0;
    }
    g_timing = getenv("RC_TIMING") != nullptr;
    stamp("main() entered");
    for (i = 1; i < argc; ++i) {  // main.cpp:165-247
        if (!strcmp("-r", argv[i]) || !strcmp("-i", argv[i]))
            ++i;
        else if (!strcmp("-p", argv[i]))
            i += 2;
        else if (!strcmp("-od", argv[i])) {
            mkdir(argv[i + 1], 0700);
            od = argv[++i];
        } else if (!strcmp("-c", argv[i])) {
            dump = argv[++i];
            FILE *fp = fopen(dump, "r");
            if (!fp) die("Could not open file %s\n", dump);
            fclose(fp);
        } else if (!strcmp("-k", argv[i]))
            k = atoi(argv[++i]);
        else if (!strcmp("-t", argv[i]))
            t_flag = atoi(argv[++i]);
        else if (!strcmp("-maxcor", argv[i]))
            ++i;
        else if (!strcmp("-maxcorK", argv[i]))
            max_fix_per_k = atoi(argv[++i]);
        else if (!strcmp("-wk", argv[i]))
            wk = atof(argv[++i]);
        else if (!strcmp("-stdout", argv[i]))
            g_stdout = true;
        else if (!strcmp("-verbose", argv[i]))
            verbose = true;
        else if (!strcmp("-verbose-iter", argv[i]))
            g_trace_iter = atoi(argv[++i]);
        else if (!strcmp("-gpus", argv[i]))
            gpus = atoi(argv[++i]);
        else if (!strcmp("-batch", argv[i]))
            batch_reads = (size_t)atol(argv[++i]);
        else if (!strcmp("-inflight", argv[i]))
            inflight = atoi(argv[++i]);
        else if (!strcmp("-write-dump", argv[i]))
            write_dump = argv[++i];
        else if (!strcmp("-packed", argv[i]))
            g_packed = true;
        else if (!strcmp("-h", argv[i])) {
            print_help();
            return 0;
        } else {
            fprintf(stderr, "Unknown argument: %s\n", argv[i]);
            return 0;
        }
    }
    g_verbose = verbose;
    if (g_trace_iter < 1) g_trace_iter = 1;
    // -verbose carries RC_TRACE_ITER_WORDS x trace-iter words per read through host and device
    // (9 KB per read at the default 64 iterations): small batches, or a real data set needs tens of GB
    if (verbose && batch_reads > (1u << 16)) batch_reads = 1u << 16;
    if (gpus < 1) gpus = 1;
    if (inflight < 1) inflight = 1;
    if (inflight > 8) inflight = 8;
    if (batch_reads < 2) batch_reads = 2;
    batch_reads &= ~(size_t)1;
    {
        unsigned hc = std::thread::hardware_concurrency();
        // (default: 16 -- the loops these threads share are memory copies and page-cache calls, and beyond that
        // they get in each other's way: 16 M x 150 bp pairs, files to files, 0.53 / 0.50 / 0.49 / 0.51 / 0.52 / 0.62 s
        // at 8 / 12 / 16 / 20 / 24 / 32 threads on a 2 x 64-core host)
        g_threads = t_flag > 1 ? t_flag : (int)std::min<unsigned>(hc ? hc : 8, 16);
        if (g_threads < 1) g_threads = 1;
    }
    g_timing = getenv("RC_TIMING") != nullptr;
    if (const char *e = getenv("RC_TRANSPORT")) g_packed = g_packed || !strcmp(e, "packed");
    if (verbose) g_packed = false;  // (the transcript needs the traced entry point)
    // batches recycle buffers of hundreds of MB: keep freed memory in the heap instead of handing it
    // back to the kernel and faulting it in again page by page (with dozens of threads every
    // mmap/munmap/page fault also serialises on the process's memory-map lock)
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, -1);

    std::vector<ReadFile> files(0), mates(0);
    files.reserve(MAX_READ_FILE);
    mates.reserve(MAX_READ_FILE);
    for (i = 1; i < argc; ++i) {  // main.cpp:250-268
        const bool is_in = !strcmp("-r", argv[i]) || !strcmp("-p", argv[i]) || !strcmp("-i", argv[i]);
        if (!is_in) continue;
        if (files.size() >= MAX_READ_FILE) die("The number of read files exceeds the limit %d.\n", MAX_READ_FILE);
        files.emplace_back();
        mates.emplace_back();
        if (!strcmp("-r", argv[i])) {
            open_file(files.back(), argv[i + 1], false, false, od);
            ++i;
        } else if (!strcmp("-p", argv[i])) {
            open_file(files.back(), argv[i + 1], true, false, od);
            open_file(mates.back(), argv[i + 2], true, false, od);
            if (files.back().fastq != mates.back().fastq)
                die("rcorrector: %s and %s are a FASTQ and a FASTA file: the mates of a pair must have the same format\n", argv[i + 1], argv[i + 2]);
            i += 2;
        } else {
            open_file(files.back(), argv[i + 1], false, true, od);
            ++i;
        }
    }

    // one context per GPU (the table is replicated across GPUs); `inflight` worker threads per GPU keep
    // that many batches in flight in it through rc_submit / rc_wait, one slot each
    if (inflight > RC_MAX_SLOTS) inflight = RC_MAX_SLOTS;
    const int nctx = gpus, nworkers = gpus * inflight;
    std::vector<std::mutex> submit_mu((size_t)gpus);  // rc_submit calls on one context are serialised
    std::vector<rc_ctx *> ctx((size_t)nctx, nullptr);
    char err[512];
    // RC_SHARED_GPU=1 (tests): every "GPU" is device 0, so that the -gpus N path -- one table replica
    // per GPU, batches dealt to whichever context is free -- runs on a one-GPU box
    const bool shared_gpu = getenv("RC_SHARED_GPU") != nullptr;
    for (int c = 0; c < nctx; ++c) {
        rc_config cfg = {shared_gpu ? 0 : c, k, max_fix_per_k};
        ctx[c] = rc_create(&cfg, err, sizeof err);
        if (!ctx[c]) die("rcorrector: %s\n", err);
    }
    // One GPU: the whole host pipeline -- reader, packers, formatters, writers and their buffers -- lives on the NUMA
    // node that GPU hangs off (every byte of a read crosses host memory a dozen times on its way through; across the
    // socket link each crossing costs more).  Several GPUs: each GPU's worker threads bind themselves (below).
    const bool numa_on = !(getenv("RC_NUMA") && !strcmp(getenv("RC_NUMA"), "0"));
    if (numa_on && gpus == 1) {
        const int node = rc_device_numa_node(ctx[0]);
        if (node >= 0 && bind_to_numa_node(node) && g_timing) fprintf(stderr, "[rc timing] host threads bound to NUMA node %d\n", node);
    }
    for (size_t fi = 0; fi < files.size(); ++fi)
        if ((files[fi].out_gz || (files[fi].paired && mates[fi].out_gz)) && !g_stdout) {
            const unsigned hc = std::thread::hardware_concurrency();
            g_deflate_threads = t_flag > 1 ? (size_t)t_flag : std::min<size_t>(hc ? hc / 2 : 8, 96);
        }
    g_pool.start(std::max<size_t>((size_t)g_threads * 2, g_deflate_threads));  // (the reader, the mate's reader and the workers call it side by side)
    stamp("contexts created (HIP initialised, scratch allocated)");
    const double t_start = now_s();
    // While the table loads: the batch buffers of the pipeline -- text blocks, page-locked arenas, output slices --
    // are allocated, sized from the head of the first input, touched and registered with the GPU runtime here, so
    // that the first batches do not pay for a few GB of page faults and hipHostRegister calls one after the other
    const size_t max_in_flight = (size_t)(nworkers + 2);
    // One pass (see ingest_resident): no dump, one GPU, regular files (plain or .gz: one inflate pass instead of two) whose text
    // fits a third of the memory that is available and whose bases fit the counter's share of HBM.  RC_RESIDENT=0 keeps the two passes, =1 skips the size test.
    bool resident = false;
    std::vector<std::unique_ptr<Retained>> kept;
    if (!dump && !verbose && gpus == 1 && !files.empty()) {
        uint64_t text_bytes = 0;
        bool plain = true;
        for (size_t fi = 0; fi < files.size(); ++fi)
            for (const ReadFile *f : {(const ReadFile *)&files[fi], files[fi].paired ? (const ReadFile *)&mates[fi] : (const ReadFile *)nullptr}) {
                if (!f) continue;
                struct stat st;
                if (f->src.is_gz) {  // (its text is taken as eight times the file: FASTQ deflates to a fifth or a quarter)
                    if (stat(f->path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) {
                        plain = false;
                        continue;
                    }
                    text_bytes += (uint64_t)st.st_size * 8;
                    continue;
                }
                if (!f->src.seekable || fstat(f->src.fd, &st) != 0) {
                    plain = false;
                    continue;
                }
                text_bytes += (uint64_t)st.st_size;
            }
        uint64_t avail = 0;
        if (FILE *mi = fopen("/proc/meminfo", "r")) {
            char ln[256];
            while (fgets(ln, sizeof ln, mi))
                if (!strncmp(ln, "MemAvailable:", 13)) avail = (uint64_t)atoll(ln + 13) << 10;
            fclose(mi);
        }
        for (const char *lim : {"/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"})  // a container's own limit
            if (FILE *cg = fopen(lim, "r")) {
                char ln[64];
                if (fgets(ln, sizeof ln, cg) && ln[0] >= '0' && ln[0] <= '9') avail = std::min<uint64_t>(avail, strtoull(ln, nullptr, 10));
                fclose(cg);
            }
        const char *e = getenv("RC_RESIDENT");
        resident = plain && (e ? atoi(e) != 0 : (text_bytes <= avail / 3 && text_bytes / 2 <= ((uint64_t)96 << 30)));
        if (e && atoi(e) > 1) batch_reads = std::max<size_t>(2, (size_t)atoi(e)) & ~(size_t)1;  // (tests: RC_RESIDENT=<batch size>)
        g_gz_whole = resident;
    }
    std::vector<std::shared_ptr<Job>> warm_jobs;
    // (the head of the first file is looked at here, not in the thread: the one-pass reader takes it out of the source)
    size_t head_nl = 0, head_last = 0, head_seq_len = 0;
    if (!files.empty() && !verbose && !files[0].src.is_gz && files[0].src.seekable) {
        const ReadFile &f = files[0];
        const int lpr = f.fastq ? 4 : 2;
        const char *h = f.src.left.p;
        size_t l1 = 0;
        for (size_t i = 0; i < f.src.left_len; ++i)
            if (h[i] == '\n') {
                ++head_nl;
                if (head_nl == 1) l1 = i;
                if (head_nl == 2) head_seq_len = i - l1 - 1;
                if (head_nl % (size_t)lpr == 0) head_last = i + 1;
            }
    }
    std::thread warm([&]() {
        if (files.empty() || verbose || files[0].src.is_gz || !files[0].src.seekable) return;
        const ReadFile &f = files[0];
        const int lpr = f.fastq ? 4 : 2;
        const size_t nl = head_nl, last = head_last, seq_len = head_seq_len;
        if (last == 0 || seq_len == 0) return;
        const double rec_bytes = (double)last / (double)(nl / (size_t)lpr);
        struct stat st;
        if (stat(f.path.c_str(), &st) != 0) return;
        const double file_recs = (double)st.st_size / rec_bytes;
        size_t recs = batch_reads;
        if (f.interleaved) recs = batch_reads;  // (a batch of an interleaved file holds batch_reads records as well)
        if ((double)recs > file_recs * 1.02 + 16) recs = (size_t)(file_recs * 1.02) + 16;
        size_t njobs = (size_t)(file_recs / (double)recs) + 1;
        if (njobs > max_in_flight) njobs = max_in_flight;
        const size_t text_bytes = (size_t)((double)recs * rec_bytes * 1.04) + ((size_t)1 << 20);
        const size_t arena_bytes = (size_t)((double)recs * (double)(seq_len + 1) * 1.02) + 4096;
        const size_t S = std::max<size_t>(1, std::min<size_t>((size_t)g_threads, (recs + 8191) / 8192));
        const size_t out_slice = (size_t)(((double)recs / (double)S + 1.0) * (rec_bytes + 48.0));
        for (size_t jn = 0; jn < njobs; ++jn) {
            auto j = std::make_shared<Job>();
            const int sides = f.paired ? 2 : 1;
            for (int sd = 0; sd < sides; ++sd) {
                Arena &A = sd ? j->b : j->a;
                if (!resident) {
                    A.blk.text.need(text_bytes);
                    A.blk.line.reserve(recs * (size_t)lpr + 8);
                    A.off.reserve(recs + 1);
                    A.seq.need(arena_bytes);   // (page-locked here: rc_host_register)
                    A.qual.need(arena_bytes);
                }
                std::vector<OutBuf> &o = sd ? j->o2 : j->o1;
                o.resize(S);
                for (auto &v : o) v.reserve(out_slice);
            }
            // touch what malloc handed out untouched (the arenas were touched by the registration)
            g_pool.run(16, [&](size_t t) {
                for (int sd = 0; sd < sides; ++sd) {
                    Arena &A = sd ? j->b : j->a;
                    const size_t lo = text_bytes * t / 16, hi = text_bytes * (t + 1) / 16;
                    if (!resident) memset(A.blk.text.p + lo, 0, hi - lo);
                    std::vector<OutBuf> &o = sd ? j->o2 : j->o1;
                    for (size_t s2 = t; s2 < S; s2 += 16) {
                        o[s2].resize(out_slice);
                        memset(o[s2].data(), 0, out_slice);
                        o[s2].clear();
                    }
                }
            });
            const size_t total = (size_t)sides * recs;
            if (resident) {  // what a resident batch sends and receives (page-locked)
                const size_t nb = (size_t)sides * arena_bytes;
                j->pk_off.need((total + 1) * 4);
                j->pk_qbits.need((nb + 7) / 8 + 64);
                j->pk_fix_pos.need((nb / 4 + 64) * 4);
                j->pk_fix_chr.need(nb / 4 + 64);
            }
            j->ret.reserve(total);
            j->l.reserve(total);
            j->m.reserve(total);
            j->h.reserve(total);
            warm_jobs.push_back(j);
        }
    });
    int64_t stored = 0;
    if (dump) {  // main.cpp:294-308: ONE Store, loaded once
        if (rc_table_load_jfdump(ctx[0], dump, &stored)) die("rcorrector: %s\n", rc_last_error(ctx[0]));
    } else {
        // no -c: stages 0-2 of run_rcorrector.pl:262-281 on the GPU -- count the canonical k-mers of
        // every input file (mates included), keep count >= 2, build the table
        std::vector<std::pair<std::string, bool>> inputs;  // path, fastq
        for (size_t fi = 0; fi < files.size(); ++fi) {
            inputs.emplace_back(files[fi].path, files[fi].fastq);
            if (files[fi].paired) inputs.emplace_back(mates[fi].path, mates[fi].fastq);
        }
        ingest_resident(ctx[0], files, mates, resident ? batch_reads : std::max<size_t>(batch_reads, (size_t)1 << 20), kept, &stored, resident);
        if (g_timing)
            fprintf(stderr, "[rc timing] k-mer counting pass over %zu file(s): %.2f s%s\n", inputs.size(), now_s() - t_start,
                    resident ? " (one pass: the text stays in host memory, the bases in HBM)" : "");
    }
    if (gpus > 1) {  // replicate the bucket array device to device (xGMI) and make sure the replicas agree
        uint64_t d0 = 0;
        if (rc_table_digest(ctx[0], &d0)) die("rcorrector: %s\n", rc_last_error(ctx[0]));
        for (int g = 1; g < gpus; ++g)  // every copy is queued before the first one is waited for: one per xGMI link
            if (rc_table_replicate_async(ctx[g], ctx[0])) die("rcorrector: %s\n", rc_last_error(ctx[g]));
        for (int g = 1; g < gpus; ++g) {
            uint64_t dg = 0;
            if (rc_sync(ctx[g]) || rc_table_digest(ctx[g], &dg)) die("rcorrector: %s\n", rc_last_error(ctx[g]));
            if (dg != d0) die("rcorrector: the k-mer table replica on GPU %d differs from the original\n", g);
        }
    }
    if (write_dump && rc_table_write_jfdump(ctx[0], write_dump, nullptr)) die("rcorrector: %s\n", rc_last_error(ctx[0]));
    fprintf(stderr, "Stored %d kmers\n", (int)stored);
    double rate = 0.01;
    if (rc_estimate_error_rate(ctx[0], wk, &rate)) die("rcorrector: %s\n", rc_last_error(ctx[0]));
    fprintf(stderr, "Weak kmer threshold rate: %lf (estimated from %.3lf/1 of the chosen kmers)\n", rate, wk);
    stamp("ERROR_RATE known");

    // GetBadQuality, main.cpp:88-128: first <= 1M records of the primary files, in order
    char bad_q = 0;
    if (!files.empty() && files[0].fastq) {
        std::vector<int32_t> fh(300, 0), lh(300, 0);
        int total = 0;
        if (resident) {  // the same records, from the blocks the counting pass kept (primary files, in order)
            for (const auto &R : kept) {
                if (total >= 1000000) break;
                quality_histograms(R->a, R->lpr_a, (size_t)(1000000 - total), fh, lh, &total);
            }
        } else {
            for (size_t fi = 0; fi < files.size() && total < 1000000; ++fi) {
                Source s;
                s.open(files[fi].path);
                Block b;
                const int lpr = files[fi].fastq ? 4 : 2;
                while (total < 1000000) {
                    take_records(s, std::min<size_t>((size_t)(1000000 - total), (size_t)1 << 18), lpr, b);
                    if (b.records == 0) break;
                    quality_histograms(b, lpr, b.records, fh, lh, &total);
                }
                s.close();
            }
        }
        bad_q = rc_bad_quality_from_hist(fh.data(), lh.data(), total);
    }
    fprintf(stderr, "Bad quality threshold is '%c'\n", bad_q);
    stamp("ERROR_RATE and bad quality known");
    for (int c = 0; c < nctx; ++c)
        if (rc_set_run_params(ctx[c], rate, bad_q)) die("rcorrector: %s\n", rc_last_error(ctx[c]));
    const double t_setup = now_s();
    stamp("start-up done");

    // pipeline: reader (this thread) -> one worker per GPU -> writer thread (input order)
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<Job>> order;  // submission order, for the writer
    std::vector<std::shared_ptr<Job>> pool;  // finished jobs: their buffers are reused (no fresh page faults)
    std::deque<std::shared_ptr<Job>> q;  // one queue for all workers: whichever context is free takes the next batch
    bool closing = false, reader_done = false;
    warm.join();
    stamp("batch buffers ready");
    pool = warm_jobs;

    // the output records of a finished batch, formatted (and deflated for .gz outputs) in slices by
    // the worker that ran it; the writer thread only writes
    auto format_job = [&](Job &J) {
        const Job *j = &J;
        const size_t n = j->a.n();
        ReadFile &f = files[(size_t)j->file];
        const bool alternate = j->mode == 1 && g_stdout;  // main.cpp:487-495
        // compression is a property of each output file (Reads::AddReadFile picks it per input name):
        // `-p a.fq.gz b.fq` writes a gzip stream for the first mates and plain text for the second
        const bool gz1 = f.out_gz && !g_stdout, gz2 = j->mode == 1 && mates[(size_t)j->file].out_gz && !g_stdout;
        // (slices of plain output are copied by at most g_threads threads -- memory-bound, more get in each other's way --
        // slices that are deflated by as many as the pool has: that is arithmetic)
        const size_t width = (gz1 || gz2) ? std::max<size_t>((size_t)g_threads, g_deflate_threads) : (size_t)g_threads;
        const size_t S = std::max<size_t>(1, std::min<size_t>(width, (n + 8191) / 8192));
        std::vector<OutBuf> &o1 = J.o1, &o2 = J.o2;
        o1.resize(S);
        o2.resize(S);
        for (auto &v : o1) v.clear();
        for (auto &v : o2) v.clear();
        auto fmt = [&](size_t lo, size_t hi) {
            for (size_t s = lo; s < hi; ++s) {
                const size_t r0 = n * s / S, r1 = n * (s + 1) / S;
                o1[s].reserve((r1 - r0) * 300);
                for (size_t r = r0; r < r1; ++r) {
                    put_record(o1[s], j->a, r, j->fastq, j->ret[r], j->l[r], j->m[r], j->h[r]);
                    if (alternate) put_record(o1[s], j->b, r, j->fastq, j->ret[n + r], j->l[n + r], j->m[n + r], j->h[n + r]);
                }
                if (j->mode == 1 && !alternate) {
                    o2[s].reserve((r1 - r0) * 300);
                    for (size_t r = r0; r < r1; ++r)
                        put_record(o2[s], j->b, r, j->fastq, j->ret[n + r], j->l[n + r], j->m[n + r], j->h[n + r]);
                }
            }
        };
        g_pool.run(S, [&](size_t s) { fmt(s, s + 1); });
        if (gz1 || gz2) {  // deflate every slice into its own gzip member, in parallel
            std::vector<OutBuf> z1(S), z2(S);
            g_pool.run(S, [&](size_t s) {
                if (gz1 && !o1[s].empty()) gzip_member(o1[s], z1[s]);
                if (gz2 && !o2[s].empty()) gzip_member(o2[s], z2[s]);
            });
            if (gz1) o1.swap(z1);
            if (gz2) o2.swap(z2);
        }
    };

    std::vector<std::thread> workers;
    for (int wk = 0; wk < nworkers; ++wk) {
        workers.emplace_back([&, wk]() {
            const int g = wk % gpus, slot = wk / gpus;
            if (numa_on && gpus > 1 && !shared_gpu) {
                const int node = rc_device_numa_node(ctx[g]);
                if (node >= 0) bind_to_numa_node(node);
            }
            for (;;) {
                std::shared_ptr<Job> j;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    const double tw = now_s();
                    cv.wait(lk, [&] { return closing || !q.empty(); });
                    g_w_worker += now_s() - tw;
                    if (q.empty()) return;
                    j = q.front();
                    q.pop_front();
                }
                const double tp0 = now_s();
                const size_t n = j->a.n();
                const size_t total = j->mode == 1 ? 2 * n : n;
                j->ret.assign(total, 0);
                j->l.assign(total, 0);
                j->m.assign(total, 0);
                j->h.assign(total, 0);
                bool resident_done = false;
                int rrc = 0;
                double tq1 = tp0;
                if (j->resident) {
                    // the reads are in HBM since they were counted: offsets and quality bits go down, the results and the
                    // substitutions come back and are applied to the sequence lines of the text
                    Job &J = *j;
                    const size_t bytes1 = J.a.off[n], bytes2 = J.mode == 1 ? J.b.off[n] : 0, nbytes = bytes1 + bytes2;
                    const size_t cap = nbytes / 4 + 64;
                    J.pk_off.need((total + 1) * 4);
                    J.pk_qbits.need((nbytes + 7) / 8 + 64);
                    J.pk_fix_pos.need(cap * 4);
                    J.pk_fix_chr.need(cap);
                    uint32_t *off = (uint32_t *)J.pk_off.data();
                    memcpy(off, J.a.off.data(), (n + 1) * 4);
                    if (J.mode == 1)
                        for (size_t r = 0; r <= n; ++r) off[n + r] = (uint32_t)bytes1 + J.b.off[r];
                    bool bits_ok = true;
                    if (J.fastq) {
                        QualView V{{&J.a, J.mode == 1 ? &J.b : &J.a}, J.mode == 1 ? bytes1 : nbytes, nbytes};
                        const size_t Q = std::max<size_t>(1, std::min<size_t>((size_t)g_threads, nbytes / 65536 + 1));
                        std::vector<char> okv(Q, 1);
                        g_pool.run(Q, [&](size_t t) {
                            const size_t lo = (nbytes * t / Q) & ~(size_t)7, hi = t + 1 == Q ? nbytes : ((nbytes * (t + 1) / Q) & ~(size_t)7);
                            if (lo < hi) okv[t] = pack_quality_bits_from_text(V, bad_q, lo, hi, (uint8_t *)J.pk_qbits.data()) ? 1 : 0;
                        });
                        for (char c : okv) bits_ok = bits_ok && c;
                    }
                    tq1 = now_s();
                    if (bits_ok) {
                        rc_resident_batch rb;
                        memset(&rb, 0, sizeof rb);
                        rb.mode = J.mode;
                        rb.n = n;
                        rb.arena_a = J.arena_a;
                        rb.bytes_a = bytes1;
                        rb.arena_b = J.arena_b;
                        rb.bytes_b = bytes2;
                        rb.off = off;
                        rb.qual_bits = J.fastq ? (const uint8_t *)J.pk_qbits.data() : nullptr;
                        rb.ret = J.ret.data();
                        rb.l = J.l.data();
                        rb.m = J.m.data();
                        rb.h = J.h.data();
                        rb.fix_pos = (uint32_t *)J.pk_fix_pos.data();
                        rb.fix_chr = (uint8_t *)J.pk_fix_chr.data();
                        rb.fix_cap = cap;
                        {
                            std::lock_guard<std::mutex> lk(submit_mu[(size_t)g]);
                            rrc = rc_submit_resident(ctx[g], &rb, slot);
                        }
                        if (!rrc) rrc = rc_wait_resident(ctx[g], slot);
                        if (!rrc && rb.n_fix) {  // positions are distinct: any number of threads
                            const size_t F = std::max<size_t>(1, std::min<size_t>((size_t)g_threads, rb.n_fix / 16384 + 1));
                            g_pool.run(F, [&](size_t t) {
                                apply_fixes_to_text(J.a, J.mode == 1 ? &J.b : nullptr, bytes1, rb.fix_pos, rb.fix_chr, rb.n_fix * t / F, rb.n_fix * (t + 1) / F);
                            });
                        }
                        resident_done = true;
                    }
                }
                if (!resident_done) {
                    pack_arena(j->a, files[(size_t)j->file].path);
                    if (j->mode == 1) pack_arena(j->b, mates[(size_t)j->file].path);
                }
                const double tp1 = resident_done ? tq1 : now_s();
                rc_batch rb;
                memset(&rb, 0, sizeof rb);
                rb.mode = j->mode;
                rb.n = n;
                rb.seq = j->a.seq.data();
                rb.qual = j->a.qual.data();
                rb.off = j->a.off.data();
                if (j->mode == 1) {
                    rb.seq2 = j->b.seq.data();
                    rb.qual2 = j->b.qual.data();
                    rb.off2 = j->b.off.data();
                }
                rb.ret = j->ret.data();
                rb.l = j->l.data();
                rb.m = j->m.data();
                rb.h = j->h.data();
                int rc;
                const double tg0 = resident_done ? tq1 : now_s();
                if (resident_done) {
                    rc = rrc;
                } else if (g_verbose) {
                    const size_t nbytes = (size_t)j->a.off[n] + (j->mode == 1 ? (size_t)j->b.off[n] : 0);
                    j->tr_before.assign(nbytes, 0);
                    j->tr_after.assign(nbytes, 0);
                    j->tr_flags.assign(total, 0);
                    j->tr_niter.assign(total, 0);
                    j->tr_iter.assign(total * (size_t)g_trace_iter * RC_TRACE_ITER_WORDS, 0);
                    rc_trace tr;
                    tr.max_iter = g_trace_iter;
                    tr.counts_before = j->tr_before.data();
                    tr.counts_after = j->tr_after.data();
                    tr.flags = j->tr_flags.data();
                    tr.n_iter = j->tr_niter.data();
                    tr.iter = j->tr_iter.data();
                    std::lock_guard<std::mutex> lk(submit_mu[(size_t)g]);
                    rc = rc_correct_batch_traced(ctx[g], &rb, &tr);
                } else if (g_packed && [&]() {
                               // One bit per quality cannot say "this read has no quality string" (qual[0] == 0: an empty
                               // quality line in a FASTQ file; ErrorCorrection.cpp:1316 asks): such a batch takes the bytes
                               if (!j->fastq) return true;
                               for (int sd = 0; sd < (j->mode == 1 ? 2 : 1); ++sd) {
                                   const Arena &A = sd ? j->b : j->a;
                                   for (size_t r = 0; r < A.n(); ++r)
                                       if (A.off[r + 1] - A.off[r] > 1 && A.qual.data()[A.off[r]] == 0) return false;
                               }
                               return true;
                           }()) {
                    // the packed boundary: the arenas stay here; 2-bit codes, quality bits and the letters outside ACGT go
                    // down, the substitutions come back as a list and are applied to the arenas in front of the formatter
                    Job &J = *j;
                    const size_t bytes1 = J.a.off[n], bytes2 = J.mode == 1 ? J.b.off[n] : 0, nbytes = bytes1 + bytes2;
                    const size_t n_words = (nbytes + 15) / 16, cap = nbytes / 4 + 64;
                    J.pk_off.need((total + 1) * 4);
                    J.pk_bases.need(n_words * 4 + 64);
                    J.pk_qbits.need((nbytes + 7) / 8 + 64);
                    J.pk_fix_pos.need(cap * 4);
                    J.pk_fix_chr.need(cap);
                    uint32_t *off = (uint32_t *)J.pk_off.data();
                    memcpy(off, J.a.off.data(), (n + 1) * 4);
                    if (J.mode == 1)
                        for (size_t r = 0; r <= n; ++r) off[n + r] = (uint32_t)bytes1 + J.b.off[r];
                    // bases: 16-byte-aligned pieces of the combined arena side by side, the exceptions of each piece after it
                    const size_t P = std::max<size_t>(1, std::min<size_t>((size_t)g_threads, n_words / 4096 + 1));
                    std::vector<std::vector<uint32_t>> ep(P);
                    std::vector<std::vector<uint8_t>> ec(P);
                    auto piece = [&](size_t t) {
                        const size_t w0 = n_words * t / P, w1 = n_words * (t + 1) / P;
                        size_t lo = w0 * 16, hi = std::min(w1 * 16, nbytes);
                        uint32_t *bases = (uint32_t *)J.pk_bases.data();
                        for (int pass = 0; pass < 2; ++pass) {  // (first pass counts the exceptions, second stores them)
                            size_t cnt = 0;
                            uint32_t *pp = pass ? ep[t].data() : nullptr;
                            uint8_t *pc = pass ? ec[t].data() : nullptr;
                            const size_t room = pass ? ep[t].size() : 0;
                            size_t got = 0;
                            if (lo < bytes1) cnt += (got = rc_pack_bases(J.a.seq.data(), lo, std::min(hi, bytes1), bases, pp, pc, room));
                            if (hi > bytes1) {
                                const size_t b0 = std::max(lo, bytes1);
                                cnt += rc_pack_bases(J.b.seq.data() - bytes1, b0, hi, bases, pp ? pp + std::min(got, room) : nullptr,
                                                     pc ? pc + std::min(got, room) : nullptr, room > got ? room - got : 0);
                            }
                            if (pass == 0) {
                                if (cnt == 0) break;
                                ep[t].resize(cnt);
                                ec[t].resize(cnt);
                            }
                        }
                    };
                    g_pool.run(P, piece);
                    size_t n_exc = 0;
                    for (size_t t = 0; t < P; ++t) n_exc += ep[t].size();
                    J.pk_exc_pos.need(n_exc * 4 + 64);
                    J.pk_exc_chr.need(n_exc + 64);
                    {
                        size_t at = 0;
                        for (size_t t = 0; t < P; ++t) {
                            if (ep[t].empty()) continue;
                            memcpy(J.pk_exc_pos.data() + at * 4, ep[t].data(), ep[t].size() * 4);
                            memcpy(J.pk_exc_chr.data() + at, ec[t].data(), ec[t].size());
                            at += ep[t].size();
                        }
                    }
                    // quality bits (FASTQ) over the combined arena; byte-aligned pieces
                    const bool fq = J.fastq;
                    if (fq) {
                        uint8_t *qb = (uint8_t *)J.pk_qbits.data();
                        const size_t Q = std::max<size_t>(1, std::min<size_t>((size_t)g_threads, nbytes / 65536 + 1));
                        // (arena 2's bits start at bit bytes1 of the same array: pack the two arenas' bytes through one view)
                        g_pool.run(Q, [&](size_t t) {
                            const size_t lo = (nbytes * t / Q) & ~(size_t)7, hi = t + 1 == Q ? nbytes : ((nbytes * (t + 1) / Q) & ~(size_t)7);
                            for (size_t p8 = lo; p8 < hi; p8 += 8) {
                                unsigned v = 0;
                                for (size_t q = p8; q < std::min(p8 + 8, hi); ++q) {
                                    const signed char c = q < bytes1 ? (signed char)J.a.qual.data()[q] : (signed char)J.b.qual.data()[q - bytes1];
                                    v |= (unsigned)(c > (signed char)bad_q) << (q - p8);
                                }
                                qb[p8 >> 3] = (uint8_t)v;
                            }
                        });
                    }
                    rc_packed_batch pb;
                    memset(&pb, 0, sizeof pb);
                    pb.mode = J.mode;
                    pb.n = n;
                    pb.nbytes = nbytes;
                    pb.off = off;
                    pb.bases = (const uint32_t *)J.pk_bases.data();
                    pb.qual_bits = fq ? (const uint8_t *)J.pk_qbits.data() : nullptr;
                    pb.exc_pos = n_exc ? (const uint32_t *)J.pk_exc_pos.data() : nullptr;
                    pb.exc_chr = n_exc ? (const uint8_t *)J.pk_exc_chr.data() : nullptr;
                    pb.n_exc = n_exc;
                    pb.ret = J.ret.data();
                    pb.l = J.l.data();
                    pb.m = J.m.data();
                    pb.h = J.h.data();
                    pb.fix_pos = (uint32_t *)J.pk_fix_pos.data();
                    pb.fix_chr = (uint8_t *)J.pk_fix_chr.data();
                    pb.fix_cap = cap;
                    {
                        std::lock_guard<std::mutex> lk(submit_mu[(size_t)g]);
                        rc = rc_submit_packed(ctx[g], &pb, slot);
                    }
                    if (!rc) rc = rc_wait_packed(ctx[g], slot);
                    if (!rc && pb.n_fix) {  // positions are distinct: any number of threads
                        const size_t F = std::max<size_t>(1, std::min<size_t>((size_t)g_threads, pb.n_fix / 16384 + 1));
                        g_pool.run(F, [&](size_t t) {
                            for (size_t q = pb.n_fix * t / F; q < pb.n_fix * (t + 1) / F; ++q) {
                                const size_t pos = pb.fix_pos[q];
                                if (pos < bytes1)
                                    J.a.seq.data()[pos] = (char)pb.fix_chr[q];
                                else
                                    J.b.seq.data()[pos - bytes1] = (char)pb.fix_chr[q];
                            }
                        });
                    }
                } else {
                    {   // upload + kernels + download are queued here; the wait below overlaps with the other
                        // workers' packing, submitting and formatting
                        std::lock_guard<std::mutex> lk(submit_mu[(size_t)g]);
                        rc = rc_submit(ctx[g], &rb, slot);
                    }
                    if (!rc) rc = rc_wait(ctx[g], slot);
                }
                const double tf0 = now_s();
                if (!rc) format_job(*j);
                const double tf1 = now_s();
                {
                    std::lock_guard<std::mutex> lk(mu);
                    g_t_gpu += tf0 - tg0;
                    g_t_format += tf1 - tf0;
                    g_t_pack += tp1 - tp0;
                    j->rc = rc;
                    if (rc) j->err = rc_last_error(ctx[g]);
                    j->done = true;
                }
                cv.notify_all();
            }
        });
    }

    uint64_t total_reads = 0, total_cor = 0;
    std::thread writer([&]() {
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                const double tw = now_s();
                cv.wait(lk, [&] { return (!order.empty() && order.front()->done) || (reader_done && order.empty()); });
                g_w_writer += now_s() - tw;
                if (order.empty()) return;
                j = order.front();
            }
            if (j->rc) die("rcorrector: %s\n", j->err.c_str());
            const size_t n = j->a.n();
            ReadFile &f = files[(size_t)j->file], &g2 = mates[(size_t)j->file];
            const bool alternate = j->mode == 1 && g_stdout;  // main.cpp:487-495
            if (g_verbose) {
                // the transcript in the order of the reference's -t 1 loop (main.cpp:368-438): per
                // unit, mate 1's trace [and record, under -stdout], then mate 2's
                std::vector<char> vt;
                const size_t bytes1 = j->a.off[n];
                auto flush = [&]() {
                    fwrite(vt.data(), 1, vt.size(), stdout);
                    vt.clear();
                };
                for (size_t r = 0; r < n; ++r) {
                    put_transcript(vt, *j, j->a, r, r, 0, k);
                    if (g_stdout) put_record(vt, j->a, r, j->fastq, j->ret[r], j->l[r], j->m[r], j->h[r]);
                    if (j->mode == 1) {
                        put_transcript(vt, *j, j->b, r, n + r, bytes1, k);
                        if (g_stdout) put_record(vt, j->b, r, j->fastq, j->ret[n + r], j->l[n + r], j->m[n + r], j->h[n + r]);
                    }
                    if (vt.size() > (1u << 20)) flush();
                }
                flush();
                fflush(stdout);
            }
            const double tw0 = now_s();
            if (j->mode == 1 && !alternate && !g_stdout) {  // two output files: written side by side
                std::thread second([&]() { emit_slices(g2, j->o2); });
                emit_slices(f, j->o1);
                second.join();
            } else {
                if (!(g_verbose && g_stdout)) emit_slices(f, j->o1);
                if (j->mode == 1 && !alternate) emit_slices(g2, j->o2);
            }
            g_t_write += now_s() - tw0;
            for (size_t r = 0; r < j->ret.size(); ++r) {  // UpdateSummary, main.cpp:73-79
                ++total_reads;
                if (j->ret[r] > 0) total_cor += (uint64_t)j->ret[r];
            }
            bool retire = false;
            {
                std::lock_guard<std::mutex> lk(mu);
                order.pop_front();
                j->done = false;
                j->rc = 0;
                // once the reader has handed out its last batch no job is needed again: its buffers -- a GB of text, arenas
                // and output slices each -- are unmapped now, beside the batches still in flight, instead of after _exit
                // where the parent waits for it (0.25 s of a 2 s run)
                if (reader_done)
                    retire = true;
                else
                    pool.push_back(j);
            }
            cv.notify_all();
            if (retire) {
                Job *raw = new Job;  // (the job's buffers move to an object of the helper thread's own)
                raw->a.blk.swap(j->a.blk);
                raw->b.blk.swap(j->b.blk);
                raw->o1.swap(j->o1);
                raw->o2.swap(j->o2);
                std::thread([raw]() { delete raw; }).detach();
            }
        }
    });

    // reader
    if (resident) {  // the batches are here already: a pooled job takes over the next one's text, line index and offsets
        for (auto &R : kept) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                const double tw = now_s();
                cv.wait(lk, [&] { return order.size() < max_in_flight; });
                g_w_reader += now_s() - tw;
                if (!pool.empty()) {
                    j = pool.back();
                    pool.pop_back();
                }
            }
            if (!j) j = std::make_shared<Job>();
            j->file = R->file;
            j->mode = R->mode;
            j->fastq = R->fastq;
            j->resident = true;
            j->arena_a = R->arena_a;
            j->arena_b = R->arena_b;
            j->a.lpr = R->lpr_a;
            j->b.lpr = R->lpr_b;
            j->a.blk.swap(R->a);
            j->a.off.swap(R->off_a);
            j->a.seq_in_text = true;
            if (R->mode == 1) {
                j->b.blk.swap(R->b);
                j->b.off.swap(R->off_b);
                j->b.seq_in_text = true;
            }
            R.reset();  // (the text of the batch this job carried before: written, no longer needed)
            {
                std::lock_guard<std::mutex> lk(mu);
                order.push_back(j);
                q.push_back(j);
            }
            cv.notify_all();
        }
    } else {
        int ramp = 0;
        for (size_t fi = 0; fi < files.size(); ++fi) {
            ReadFile &f = files[fi];
            const int lpr = f.fastq ? 4 : 2;
            for (;;) {
                std::shared_ptr<Job> j;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (!pool.empty()) {
                        j = pool.back();
                        pool.pop_back();
                    }
                }
                if (!j) j = std::make_shared<Job>();
                j->file = (int)fi;
                j->mode = f.paired ? 1 : (f.interleaved ? 2 : 0);
                j->fastq = f.fastq;
                j->a.lpr = lpr;
                j->b.lpr = f.paired ? (mates[fi].fastq ? 4 : 2) : lpr;
                const double tr0 = now_s();
                // the first batches of a run are small, so that the stages behind the reader start early: an eighth,
                // a quarter, a half of -batch (whole pairs; a read's result does not depend on its batch)
                size_t want_reads = batch_reads;
                if (ramp < 3 && batch_reads >= ((size_t)1 << 19)) want_reads = (batch_reads >> (3 - ramp)) & ~(size_t)1;
                ++ramp;
                if (f.paired) {  // both mates' files at once (two inflate streams run side by side for .gz pairs)
                    std::thread mate([&]() { take_records(mates[fi].src, want_reads, j->b.lpr, j->b.blk); });
                    take_records(f.src, want_reads, lpr, j->a.blk);
                    mate.join();
                    if (j->b.blk.records != j->a.blk.records) die("ERROR: The files are not paired!\n");
                } else {
                    take_records(f.src, want_reads, lpr, j->a.blk);
                }
                if (j->a.blk.records == 0) break;
                if (j->mode == 2 && (j->a.blk.records & 1)) die("ERROR: interleaved file %s holds an odd number of reads\n", f.path.c_str());
                g_t_read += now_s() - tr0;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    const double tw = now_s();
                    cv.wait(lk, [&] { return order.size() < max_in_flight; });
                    g_w_reader += now_s() - tw;
                    order.push_back(j);
                    q.push_back(j);
                }
                cv.notify_all();
            }
        }
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        reader_done = true;
    }
    cv.notify_all();
    writer.join();
    const double t_loop_end = now_s();
    stamp("last batch written");
    {
        std::lock_guard<std::mutex> lk(mu);
        closing = true;
    }
    cv.notify_all();
    for (auto &t : workers) t.join();

    for (size_t fi = 0; fi < files.size(); ++fi) {
        for (ReadFile *f : {&files[fi], &mates[fi]}) {
            if (f->out && f->out_gz && !f->wrote) {  // an empty .gz is still one (empty) gzip member
                OutBuf none, z;
                gzip_member(none, z);
                std::vector<OutBuf> one(1);
                one[0].swap(z);
                emit_slices(*f, one);
            }
            if (f->out && f->out != stdout && f->preallocated && ftruncate(fileno(f->out), f->out_off) != 0)
                die("ERROR: could not set the length of the output of %s\n", f->path.c_str());
            if (f->out && f->out != stdout) fclose(f->out);
            f->src.close();
        }
    }
    // (the contexts are not torn down: the process ends here, and releasing gigabytes of device and
    // host memory piece by piece costs more than everything else after the last write)
    if (g_timing)
        fprintf(stderr, "[rc timing] start-up (dump load, table build, ERROR_RATE, bad quality) %.2f s; correction loop (read, correct, write) %.2f s; %d host threads\n",
                t_setup - t_start, t_loop_end - t_setup, g_threads);
    if (g_timing)
        fprintf(stderr, "[rc timing] stage totals: read+index %.2f s (reader thread); pack %.2f s + correct_batch %.2f s + format %.2f s (sum over %d worker threads); write %.2f s (writer thread)\n",
                g_t_read, g_t_pack, g_t_gpu, g_t_format, nworkers, g_t_write);
    if (g_timing) fprintf(stderr, "[rc timing] inside read+index (all files, thread-seconds): pread %.2f s, newline scan %.2f s, line index %.2f s\n", g_t_fill, g_t_nl, g_t_idx);
    if (g_timing)
        fprintf(stderr, "[rc timing] blocked: reader %.2f s (no free slot), workers %.2f s (no batch), writer %.2f s (next batch not done)\n", g_w_reader, g_w_worker, g_w_writer);
    fprintf(stderr, "Processed %llu reads\n\tCorrected %llu bases.\n", (unsigned long long)total_reads, (unsigned long long)total_cor);
    stamp("outputs closed, leaving");
    if (getenv("RC_TEARDOWN")) {  // dev: where the time between _exit and the parent's wait goes
        pool.clear();
        warm_jobs.clear();
        order.clear();
        q.clear();
        stamp("teardown: job buffers unregistered and freed");
        for (rc_ctx *c : ctx) rc_destroy(c);
        stamp("teardown: contexts destroyed");

    }
    fflush(NULL);
    _exit(0);  // every output is closed: skip unmapping gigabytes of buffers one by one
}
