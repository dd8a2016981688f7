// rc_api.hip -- the C ABI of librcorrector_amd.so (include/rcorrector_amd.h): context, table
// load, run-parameter estimation and the batch entry points.  Host code only drives HIP; every
// per-read computation happens in the kernels of rc_table.hip / rc_correct.hip.  There is no CPU
// fallback anywhere in this library.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <new>
#include <unistd.h>
#include <thread>
#include <vector>

#include "../../include/rcorrector_amd.h"
#include "rc_internal.h"

// parsed dump kept between rc_table_load_jfdump() and rc_estimate_error_rate()
struct rc_dump_cache {
    // forward code of every entry (file order), main.cpp:326-328, and a flag "holds a non-ACGT
    // letter before its last base" -- kept in the chunks the parser threads produced (concatenating
    // a few hundred MB on one thread cost more than parsing them on thirty-two)
    std::vector<std::vector<uint64_t>> codes;
    std::vector<std::vector<int8_t>> inv_mid;
    size_t n = 0;
    int load_state_invalid = 0;   // validity of the KmerCode object the load pass leaves behind
    bool valid = false;
};

struct rc_ctx_full : rc_ctx {
    rc_dump_cache dump;
};

// pinned host buffer, grow-only
struct rc_hbuf {
    void *p = nullptr;
    size_t bytes = 0;
};

// one batch in flight on the asynchronous host-buffer path
struct rc_slot {
    bool busy = false;
    rc_batch b;                      // the caller's descriptor (its buffers stay valid until rc_wait)
    size_t total_reads = 0, bytes1 = 0, bytes2 = 0;
    bool seq_pinned = false, res_pinned = false;  // the caller's buffers are page-locked: DMA straight from / to them
    rc_hbuf p_seq, p_qual, p_off, p_res;           // pinned staging (seq/qual only when the caller's are pageable)
    rc_dbuf d_seq, d_qual, d_off, d_res;
    hipEvent_t e_h2d = nullptr, e_k = nullptr, e_done = nullptr;
    // the packed boundary (rc_submit_packed): the caller's descriptor, the packed arena / exceptions / fix list in HBM,
    // pinned staging for descriptor arrays that are not page-locked, and the fix count's landing place
    rc_packed_batch *pb = nullptr;
    rc_resident_batch *rb = nullptr;  // rc_submit_resident: same slot state, the arena copied from the counter's kept arenas
    rc_dbuf d_packed, d_exc, d_fix;
    rc_hbuf p_in, p_fix, p_nfix;
    uint32_t fix_room = 0;
    bool fix_pinned = false;
};

static thread_local char g_create_err[512];

void rc_set_error(rc_ctx *ctx, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx ? ctx->err : g_create_err, 512, fmt, ap);
    va_end(ap);
}

int rc_dbuf_reserve(rc_ctx *ctx, rc_dbuf *b, size_t bytes)
{
    if (bytes <= b->bytes) return RC_OK;
    if (b->p) {
        RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(b->p);
        b->p = nullptr;
        b->bytes = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    RC_CHECK_HIP(ctx, hipMalloc(&b->p, want));
    b->bytes = want;
    return RC_OK;
}

void rc_table_release(rc_ctx *ctx)
{
    if (ctx->d_buckets && !ctx->buckets_borrowed) (void)hipFree(reinterpret_cast<char *>(ctx->d_buckets) - RC_TABLE_PREFIX_BYTES);
    ctx->d_buckets = nullptr;
    ctx->buckets_borrowed = false;
    ctx->filter_words = 0;
    ctx->table_bytes = 0;
    ctx->n_entries = 0;
}

rc_table_view rc_view(const rc_ctx *ctx)
{
    rc_table_view v;
    v.buckets = ctx->d_buckets;
    v.nb_home = ctx->nb_home;
    v.nbuckets_alloc = ctx->nb_alloc;
    v.layout = ctx->layout;
    v.ext = ctx->ext;
    v.k = ctx->k;
    v.filter_words = ctx->filter_words;
    v.filter = ctx->filter_words ? ctx->d_buckets + ctx->table_bytes / 4 : nullptr;
    v.filter_kind = ctx->filter_kind;
    return v;
}

void rc_timer_begin(rc_ctx *ctx)
{
    if (ctx->profile) (void)hipEventRecord(ctx->ev0, ctx->stream);
}

void rc_timer_end(rc_ctx *ctx, int which)
{
    if (!ctx->profile) return;
    (void)hipEventRecord(ctx->ev1, ctx->stream);
    (void)hipEventSynchronize(ctx->ev1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->timers[which].ms += ms;
    ctx->timers[which].launches += 1;
}

extern "C" {

rc_ctx *rc_create(const rc_config *cfg, char *errbuf, size_t errbuf_len)
{
    auto fail = [&](const char *msg) -> rc_ctx * {
        if (errbuf && errbuf_len) snprintf(errbuf, errbuf_len, "%s", msg);
        return nullptr;
    };
    if (!cfg) return fail("rc_create: null config");
    if (cfg->k < 1 || cfg->k > 32) return fail("rc_create: k must be in 1..32 (run_rcorrector.pl:225-228)");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail("rc_create: no HIP device available (this library has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail("rc_create: device ordinal out of range");
    if (hipSetDevice(cfg->device) != hipSuccess) return fail("rc_create: hipSetDevice failed");
    rc_ctx_full *ctx = new (std::nothrow) rc_ctx_full();
    if (!ctx) return fail("rc_create: out of memory");
    ctx->err[0] = 0;
    ctx->device = cfg->device;
    ctx->k = cfg->k;
    ctx->P.k = cfg->k;
    ctx->P.max_fix_per_k = cfg->max_fix_per_k > 0 ? cfg->max_fix_per_k : 4;
    ctx->P.error_rate = 0.01;
    ctx->P.bad_qual = 0;
    memset(ctx->P.bs, 0, sizeof ctx->P.bs);
    ctx->P.bs_ext = nullptr;
    ctx->P.flags = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipMalloc(&ctx->work.p, RC_WORK_BYTES) != hipSuccess) {
        delete ctx;
        return fail("rc_create: could not create stream/events");
    }
    ctx->work.bytes = RC_WORK_BYTES;
    (void)hipMemset(ctx->work.p, 0, RC_WORK_BYTES);
    if (const char *e = getenv("RC_PHASE_PROF")) ctx->phase_prof = ctx->phase_prof_print = atoi(e) != 0;
    if (const char *e = getenv("RC_TABLE_LOAD")) {
        ctx->table_load = ctx->table_load_packed = atof(e);
        ctx->table_load_set = true;
    }  // tuning knob
    if (const char *e = getenv("RC_TABLE_LAYOUT")) ctx->layout_pref = strcmp(e, "wide") != 0;         // dev: A/B the slot layouts
    ctx->env_k2_wave_per_read = getenv("RC_K2_WAVE_PER_READ") != nullptr;  // dev: force the wave-per-read threshold kernel
    ctx->env_no_classify = getenv("RC_NO_CLASSIFY") != nullptr;            // dev: every read goes through k_correct
    ctx->env_timing = getenv("RC_TIMING") != nullptr;
    ctx->env_no_fuse = getenv("RC_NO_FUSE") != nullptr;
    ctx->env_no_tier = getenv("RC_NO_TIER") != nullptr;
    if (const char *e = getenv("RC_FORCE_EC")) {
        const int v = atoi(e);
        if (v == 9 || v == 10) ctx->env_force_ec = v;
    }
    ctx->env_k3_generic = getenv("RC_K3_GENERIC") != nullptr;
    ctx->env_no_single = getenv("RC_NO_SINGLE") != nullptr;
    ctx->env_no_alt = getenv("RC_NO_ALT") != nullptr;  // dev / tests: no alternative chains in the search's speculation rounds
    if (const char *e = getenv("RC_LOCALITY")) ctx->locality_mode = !strcmp(e, "force") ? 1 : (!strcmp(e, "off") ? -1 : 0);  // tests / A-B
    if (const char *e = getenv("RC_K3_GRID_WAVES")) ctx->env_k3_grid_waves = atoi(e);
    return ctx;
}

void rc_destroy(rc_ctx *c)
{
    if (!c) return;
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    rc_dbuf *bufs[] = {&ctx->counts, &ctx->strong, &ctx->info, &ctx->stack, &ctx->work,
                       &ctx->h_seq, &ctx->h_qual, &ctx->h_off, &ctx->h_res, &ctx->trace, &ctx->cls, &ctx->worklist, &ctx->sel_tmp,
                       &ctx->loc_a, &ctx->loc_list, &ctx->tier_flag, &ctx->tier_list, &ctx->cand, &ctx->single_list, &ctx->runs, &ctx->bs_dev};
    for (rc_dbuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    if (ctx->slots) {
        for (int i = 0; i < RC_MAX_SLOTS; ++i) {
            rc_slot &sl = ctx->slots[i];
            if (sl.e_done) (void)hipEventSynchronize(sl.e_done);
            rc_hbuf *hb[] = {&sl.p_seq, &sl.p_qual, &sl.p_off, &sl.p_res, &sl.p_in, &sl.p_fix, &sl.p_nfix};
            for (rc_hbuf *h : hb)
                if (h->p) (void)hipHostFree(h->p);
            rc_dbuf *db[] = {&sl.d_seq, &sl.d_qual, &sl.d_off, &sl.d_res, &sl.d_packed, &sl.d_exc, &sl.d_fix};
            for (rc_dbuf *d : db)
                if (d->p) (void)hipFree(d->p);
            hipEvent_t ev[] = {sl.e_h2d, sl.e_k, sl.e_done};
            for (hipEvent_t e : ev)
                if (e) (void)hipEventDestroy(e);
        }
        delete[] ctx->slots;
    }
    if (ctx->s_h2d) (void)hipStreamDestroy(ctx->s_h2d);
    if (ctx->s_d2h) (void)hipStreamDestroy(ctx->s_d2h);
    for (auto &a : ctx->cnt_chunks)
        if (a.p) (void)hipFree(a.p);
    rc_kept_release(ctx);
    rc_table_release(ctx);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *rc_last_error(const rc_ctx *ctx) { return ctx ? ctx->err : g_create_err; }

int rc_device_numa_node(const rc_ctx *ctx)
{
    if (!ctx) return -1;
    char bus[64];
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, ctx->device) != hipSuccess) return -1;
    for (char *p = bus; *p; ++p)
        if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');  // sysfs spells the address in lower case
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *fp = fopen(path, "r");
    if (!fp) return -1;
    int node = -1;
    if (fscanf(fp, "%d", &node) != 1) node = -1;
    fclose(fp);
    return node;
}

int rc_device_memory(rc_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes)
{
    if (!ctx) return RC_ERR_ARG;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    size_t f = 0, t = 0;
    RC_CHECK_HIP(ctx, hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (uint64_t)f;
    if (total_bytes) *total_bytes = (uint64_t)t;
    return RC_OK;
}

// ---- table ---------------------------------------------------------------------------------
int rc_table_build_device(rc_ctx *ctx, uint64_t *d_codes, const int32_t *d_counts, size_t n)
{
    if (!ctx) return RC_ERR_ARG;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    static_cast<rc_ctx_full *>(ctx)->dump.valid = false;
    int rc = rc_launch_canonicalize(ctx, d_codes, n);
    if (rc) return rc;
    return rc_build_table_from_device_pairs(ctx, d_codes, d_counts, n);
}

int rc_table_build(rc_ctx *ctx, const uint64_t *codes, const int32_t *counts, size_t n)
{
    if (!ctx || (n && (!codes || !counts))) return RC_ERR_ARG;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    rc_dev_tmp b_codes, b_counts;
    if (n) {
        RC_CHECK_HIP(ctx, b_codes.alloc(n * 8));
        RC_CHECK_HIP(ctx, b_counts.alloc(n * 4));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b_codes.p, codes, n * 8, hipMemcpyHostToDevice, ctx->stream));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b_counts.p, counts, n * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    return rc_table_build_device(ctx, b_codes.as<uint64_t>(), b_counts.as<int32_t>(), n);
}

// main.cpp:294-308.  Tokens are whitespace separated (fscanf "%s"); the first of a pair is
// ">COUNT" (atoi of the text after the first character), the second the k-mer, pushed through
// KmerCode::Append character by character (only the last k characters survive the mask).
int rc_table_load_jfdump(rc_ctx *c, const char *path, int64_t *stored)
{
    if (!c || !path) return RC_ERR_ARG;
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    FILE *fp = fopen(path, "rb");
    if (!fp) {
        rc_set_error(ctx, "Could not open file %s", path);
        return RC_ERR_IO;
    }
    fseek(fp, 0, SEEK_END);
    long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    // the whole text, uninitialised and read by several threads at once (pread into disjoint slices:
    // a single reader is bound by the copy out of the page cache)
    struct text_buf {
        char *p = nullptr;
        ~text_buf() { free(p); }
        char *data() { return p; }
        char &operator[](size_t i) { return p[i]; }
    } buf;
    buf.p = (char *)malloc((size_t)sz + 1);
    if (!buf.p) {
        fclose(fp);
        rc_set_error(ctx, "out of memory reading %s (%ld bytes)", path, sz);
        return RC_ERR_NOMEM;
    }
    {
        const int fd = fileno(fp);
        unsigned RT = std::thread::hardware_concurrency();
        if (RT == 0) RT = 4;
        if (RT > 32) RT = 32;
        if ((size_t)sz < ((size_t)8 << 20)) RT = 1;
        std::vector<char> ok(RT, 1);
        auto rd = [&](unsigned t) {
            size_t at = (size_t)sz * t / RT;
            const size_t hi = (size_t)sz * (t + 1) / RT;
            while (at < hi) {
                const ssize_t n = pread(fd, buf.p + at, hi - at, (off_t)at);
                if (n <= 0) {
                    ok[t] = 0;
                    return;
                }
                at += (size_t)n;
            }
        };
        if (RT == 1) {
            rd(0);
        } else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < RT; ++t) th.emplace_back(rd, t);
            for (auto &x : th) x.join();
        }
        fclose(fp);
        for (unsigned t = 0; t < RT; ++t)
            if (!ok[t]) {
                rc_set_error(ctx, "short read on %s", path);
                return RC_ERR_IO;
            }
    }
    buf[(size_t)sz] = 0;
    const bool tm = ctx->env_timing;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_read = now();

    const int k = ctx->k;
    const uint64_t mask = rc_kmer_mask(k);
    rc_dump_cache &D = ctx->dump;
    D.codes.clear();
    D.inv_mid.clear();
    D.n = 0;
    D.load_state_invalid = 0;
    auto is_ws = [](char ch) { return ch == ' ' || ch == '\n' || ch == '\t' || ch == '\r' || ch == '\f' || ch == '\v'; };

    // the text is cut at entry starts ('>' right after white space) and the pieces are parsed by
    // several host threads; the per-piece results are concatenated in file order
    struct piece {
        std::vector<uint64_t> codes, put_codes;
        std::vector<int8_t> inv_mid;
        std::vector<int32_t> put_counts;
        int64_t accepted = 0;
        int last_state_invalid = -1;  // -1: no accepted entry in this piece
    };
    unsigned T = std::thread::hardware_concurrency();
    if (T == 0) T = 4;
    if (T > 32) T = 32;
    if ((size_t)sz < (1u << 20)) T = 1;
    std::vector<size_t> cut(T + 1, (size_t)sz);
    cut[0] = 0;
    for (unsigned t = 1; t < T; ++t) {
        size_t pos = (size_t)sz * t / T;
        if (pos < cut[t - 1]) pos = cut[t - 1];
        while (pos < (size_t)sz && !(buf[pos] == '>' && (pos == 0 || is_ws(buf[pos - 1])))) ++pos;
        cut[t] = pos;
    }
    std::vector<piece> pieces(T);
    int8_t base_code[256];
    memset(base_code, -1, sizeof base_code);
    base_code[(unsigned char)'A'] = 0;
    base_code[(unsigned char)'C'] = 1;
    base_code[(unsigned char)'G'] = 2;
    base_code[(unsigned char)'T'] = 3;
    // atoi() as glibc implements it, (int)strtol(): the long value saturates at LONG_MAX / LONG_MIN and
    // the conversion to int keeps its low 32 bits (main.cpp:297 applies it to the count token)
    auto atoi_of = [](unsigned long long v, bool ovf, bool neg) -> int {
        long long lv;
        if (ovf)
            lv = neg ? (long long)0x8000000000000000ULL : 0x7fffffffffffffffLL;
        else
            lv = neg ? -(long long)v : (long long)v;
        return (int)(uint32_t)(uint64_t)lv;
    };
    auto parse = [&](unsigned t) {
        piece &P = pieces[t];
        const char *p = buf.data() + cut[t], *end = buf.data() + cut[t + 1];
        const size_t guess = (size_t)(end - p) / (size_t)(k + 4) + 16;
        P.codes.reserve(guess);
        P.inv_mid.reserve(guess);
        P.put_codes.reserve(guess);
        P.put_counts.reserve(guess);
        while (true) {
            // fast path for the layout `jellyfish dump` writes: ">DIGITS\nKMER\n" with exactly k
            // letters out of ACGT -- everything else goes through the general tokeniser below,
            // which is what defines the result
            if (p < end && *p == '>' && p + 1 < end && (unsigned)(p[1] - '0') <= 9u) {
                const char *q = p + 1;
                unsigned long long v = 0;
                bool ovf = false;
                while (q < end && (unsigned)(*q - '0') <= 9u) {
                    const unsigned d = (unsigned)(*q - '0');
                    if (v > (0x7fffffffffffffffULL - d) / 10) ovf = true;
                    if (!ovf) v = v * 10 + d;
                    ++q;
                }
                const int cnt = atoi_of(v, ovf, false);
                if (q < end && *q == '\n' && q + 1 + k < end && q[1 + k] == '\n') {
                    const unsigned char *s2 = reinterpret_cast<const unsigned char *>(q + 1);
                    uint64_t code = 0;
                    int bad = 0;
                    for (int i = 0; i < k; ++i) {
                        const int b = base_code[s2[i]];
                        bad |= b;
                        code = (code << 2) | (uint64_t)(b & 3);
                    }
                    if (bad >= 0) {  // all four codes are non-negative: no other letter in the k-mer
                        p = q + 2 + k;
                        P.codes.push_back(code);
                        P.inv_mid.push_back(0);
                        if (cnt <= 1) continue;
                        P.last_state_invalid = 0;
                        ++P.accepted;
                        P.put_codes.push_back(code);
                        P.put_counts.push_back((int32_t)cnt);
                        continue;
                    }
                }
            }
            while (p < end && is_ws(*p)) ++p;
            if (p >= end) break;
            const char *t0 = p;
            while (p < end && !is_ws(*p)) ++p;
            int cnt = 0;  // atoi(&token[1])
            {
                const char *q = t0 + 1;
                bool neg = false, ovf = false;
                unsigned long long v = 0;
                if (q < p && (*q == '-' || *q == '+')) {
                    neg = *q == '-';
                    ++q;
                }
                while (q < p && *q >= '0' && *q <= '9') {
                    const unsigned d = (unsigned)(*q - '0');
                    if (v > (0x7fffffffffffffffULL - d) / 10) ovf = true;
                    if (!ovf) v = v * 10 + d;
                    ++q;
                }
                cnt = atoi_of(v, ovf, neg);
            }
            while (p < end && is_ws(*p)) ++p;
            const char *k0 = p;
            while (p < end && !is_ws(*p)) ++p;
            uint64_t code = 0;
            int inv = -1;
            for (const char *q = k0; q < p; ++q) {
                int b;
                switch (*q) {
                case 'A': b = 0; break;
                case 'C': b = 1; break;
                case 'G': b = 2; break;
                case 'T': b = 3; break;
                default: b = -1;
                }
                if (inv != -1) ++inv;
                code = ((code << 2) & mask) | (uint64_t)(b & 3);
                if (b == -1) inv = 0;
                if (inv >= k) inv = -1;
            }
            P.codes.push_back(code);
            P.inv_mid.push_back(inv > 0 ? 1 : 0);
            if (cnt <= 1) continue;
            P.last_state_invalid = (inv != -1);
            ++P.accepted;
            if (inv == -1) {  // Store::Put ignores invalid k-mers, Store.h:53-54
                P.put_codes.push_back(code);
                P.put_counts.push_back((int32_t)cnt);
            }
        }
    };
    if (T == 1) {
        parse(0);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t) th.emplace_back(parse, t);
        for (auto &x : th) x.join();
    }
    int64_t accepted = 0;
    size_t n_put = 0;
    for (auto &P : pieces) {
        D.n += P.codes.size();
        n_put += P.put_codes.size();
        accepted += P.accepted;
        if (P.last_state_invalid >= 0) D.load_state_invalid = P.last_state_invalid;
    }
    const double t_parse = now();
    if (stored) *stored = accepted;
    // n x Store::Put in file order: the pieces go to the device one after the other, no host copy
    int rc = RC_OK;
    {
        rc_dev_tmp b_codes, b_counts;
        RC_CHECK_HIP(ctx, b_codes.alloc(n_put * 8));
        RC_CHECK_HIP(ctx, b_counts.alloc(n_put * 4));
        size_t at = 0;
        for (auto &P : pieces) {
            const size_t m = P.put_codes.size();
            if (m) {
                RC_CHECK_HIP(ctx, hipMemcpyAsync(b_codes.as<uint64_t>() + at, P.put_codes.data(), m * 8, hipMemcpyHostToDevice, ctx->stream));
                RC_CHECK_HIP(ctx, hipMemcpyAsync(b_counts.as<int32_t>() + at, P.put_counts.data(), m * 4, hipMemcpyHostToDevice, ctx->stream));
            }
            at += m;
        }
        rc = rc_table_build_device(ctx, b_codes.as<uint64_t>(), b_counts.as<int32_t>(), n_put);
        RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the pieces' host arrays are released below
    }
    for (auto &P : pieces) {
        D.codes.emplace_back(std::move(P.codes));
        D.inv_mid.emplace_back(std::move(P.inv_mid));
    }
    if (tm) fprintf(stderr, "[rc timing] dump: parse %.2f s, table build %.2f s\n", t_parse - t_read, now() - t_parse);
    D.valid = rc == RC_OK;  // (rc_table_build drops the cache of an earlier dump)
    return rc;
}

int rc_table_share(rc_ctx *dst, const rc_ctx *src)
{
    if (!dst || !src || dst == src) return RC_ERR_ARG;
    if (dst->device != src->device || dst->k != src->k) {
        rc_set_error(dst, "table_share: contexts must be on the same device with the same k");
        return RC_ERR_ARG;
    }
    if (!src->d_buckets) {
        rc_set_error(dst, "table_share: the source context has no table");
        return RC_ERR_STATE;
    }
    RC_CHECK_HIP(dst, hipSetDevice(dst->device));
    rc_table_release(dst);
    dst->d_buckets = src->d_buckets;
    dst->buckets_borrowed = true;
    dst->nb_home = src->nb_home;
    dst->layout = src->layout;
    dst->ext = src->ext;
    dst->nb_alloc = src->nb_alloc;
    dst->n_entries = src->n_entries;
    dst->table_bytes = src->table_bytes;
    dst->filter_words = src->filter_words;
    dst->filter_kind = src->filter_kind;
    return RC_OK;
}

// the copy itself is queued on dst's stream (rc_sync(dst) waits for it), so that a host replicating to several GPUs
// has all its copies in flight at once: the GPUs of a node are linked pairwise (xGMI), one copy per link
int rc_table_replicate_async(rc_ctx *dst, const rc_ctx *src)
{
    if (!dst || !src || dst == src) return RC_ERR_ARG;
    if (dst->k != src->k) {
        rc_set_error(dst, "table_replicate: contexts must have the same k");
        return RC_ERR_ARG;
    }
    if (!src->d_buckets) {
        rc_set_error(dst, "table_replicate: the source context has no table");
        return RC_ERR_STATE;
    }
    RC_CHECK_HIP(dst, hipSetDevice(src->device));
    RC_CHECK_HIP(dst, hipStreamSynchronize(src->stream));
    RC_CHECK_HIP(dst, hipSetDevice(dst->device));
    rc_table_release(dst);
    static_cast<rc_ctx_full *>(dst)->dump.valid = false;
    // the new allocation belongs to a guard until every copy is queued: an error on the way leaves dst without a table
    // (d_buckets == nullptr), not with a pointer whose geometry still describes the previous one
    rc_dev_tmp guard;
    const size_t bytes = src->table_bytes + RC_TABLE_PREFIX_BYTES + (size_t)src->filter_words * 4;  // prefix, buckets, filter
    RC_CHECK_HIP(dst, guard.alloc(bytes));
    char *base = guard.as<char>();
    const char *from = reinterpret_cast<const char *>(src->d_buckets) - RC_TABLE_PREFIX_BYTES;
    // the bucket array (and its prefix) is the table
    bool staged = getenv("RC_REPLICATE_STAGED") != nullptr;  // tests: the path of GPUs without peer access
    if (src->device == dst->device) {
        if (!staged) RC_CHECK_HIP(dst, hipMemcpyAsync(base, from, bytes, hipMemcpyDeviceToDevice, dst->stream));
    } else if (!staged) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dst->device, src->device) != hipSuccess) can = 0;
        if (can) {
            const hipError_t e = hipDeviceEnablePeerAccess(src->device, 0);  // (dst is the current device)
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
            (void)hipGetLastError();
        }
        if (can)
            RC_CHECK_HIP(dst, hipMemcpyPeerAsync(base, dst->device, from, src->device, bytes, dst->stream));
        else
            staged = true;
    }
    if (staged) {  // no direct path between the two GPUs: through page-locked host memory, two pieces in flight
        const size_t CH = (size_t)64 << 20;
        char *h[2] = {nullptr, nullptr};
        hipEvent_t up[2] = {nullptr, nullptr};
        int rc = RC_OK;
        auto fail = [&](hipError_t e, const char *what) {
            rc_set_error(dst, "table_replicate: %s failed: %s", what, hipGetErrorString(e));
            rc = RC_ERR_HIP;
        };
        for (int i = 0; i < 2 && rc == RC_OK; ++i) {
            hipError_t e = hipHostMalloc((void **)&h[i], CH, hipHostMallocPortable);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&up[i], hipEventDisableTiming);
            if (e != hipSuccess) fail(e, "hipHostMalloc");
        }
        size_t piece = 0;
        for (size_t at = 0; at < bytes && rc == RC_OK; at += CH, ++piece) {
            const size_t n = std::min(CH, bytes - at);
            const int b = (int)(piece & 1);
            hipError_t e = piece >= 2 ? hipEventSynchronize(up[b]) : hipSuccess;  // the upload that last used this buffer
            if (e == hipSuccess) e = hipSetDevice(src->device);
            if (e == hipSuccess) e = hipMemcpy(h[b], from + at, n, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipSetDevice(dst->device);
            if (e == hipSuccess) e = hipMemcpyAsync(base + at, h[b], n, hipMemcpyHostToDevice, dst->stream);
            if (e == hipSuccess) e = hipEventRecord(up[b], dst->stream);
            if (e != hipSuccess) fail(e, "staged copy");
        }
        (void)hipSetDevice(dst->device);
        (void)hipStreamSynchronize(dst->stream);
        for (int i = 0; i < 2; ++i) {
            if (h[i]) (void)hipHostFree(h[i]);
            if (up[i]) (void)hipEventDestroy(up[i]);
        }
        if (rc) return rc;
    }
    dst->d_buckets = reinterpret_cast<uint32_t *>(base + RC_TABLE_PREFIX_BYTES);
    guard.p = nullptr;  // (dst owns it now)
    dst->nb_home = src->nb_home;
    dst->layout = src->layout;
    dst->ext = src->ext;
    dst->nb_alloc = src->nb_alloc;
    dst->n_entries = src->n_entries;
    dst->table_bytes = src->table_bytes;
    dst->filter_words = src->filter_words;
    dst->filter_kind = src->filter_kind;
    return RC_OK;
}

int rc_table_replicate(rc_ctx *dst, const rc_ctx *src)
{
    int rc = rc_table_replicate_async(dst, src);
    if (rc) return rc;
    RC_CHECK_HIP(dst, hipStreamSynchronize(dst->stream));
    return RC_OK;
}

int rc_table_count_reads_device(rc_ctx *ctx, const uint8_t *d_seq, size_t nbytes, int min_count, int64_t *n_kmers)
{
    if (!ctx || !d_seq) return RC_ERR_ARG;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    static_cast<rc_ctx_full *>(ctx)->dump.valid = false;
    return rc_count_reads(ctx, d_seq, nbytes, min_count, n_kmers);
}

int rc_table_count_begin(rc_ctx *ctx)
{
    if (!ctx) return RC_ERR_ARG;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    return rc_count_begin(ctx);
}

int rc_table_count_keep(rc_ctx *ctx, int on)
{
    if (!ctx) return RC_ERR_ARG;
    ctx->cnt_keep = on != 0;
    return RC_OK;
}

int rc_table_count_arenas(const rc_ctx *ctx, size_t *n_arenas, uint64_t *bytes, size_t cap)
{
    if (!ctx || !n_arenas) return RC_ERR_ARG;
    *n_arenas = ctx->kept_arenas.size();
    for (size_t i = 0; bytes && i < cap && i < ctx->kept_arenas.size(); ++i) bytes[i] = ctx->kept_arenas[i].bytes;
    return RC_OK;
}

int rc_table_count_release(rc_ctx *ctx)
{
    if (!ctx) return RC_ERR_ARG;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (a batch still reading them)
    rc_kept_release(ctx);
    return RC_OK;
}

int rc_table_count_add_device(rc_ctx *ctx, const uint8_t *d_seq, size_t nbytes)
{
    if (!ctx || (nbytes && !d_seq)) return RC_ERR_ARG;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    return rc_count_add(ctx, d_seq, nbytes, true);
}

int rc_table_count_add(rc_ctx *ctx, const char *seq, size_t nbytes)
{
    if (!ctx || (nbytes && !seq)) return RC_ERR_ARG;
    if (nbytes == 0) return RC_OK;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    return rc_count_add(ctx, reinterpret_cast<const uint8_t *>(seq), nbytes, false);
}

int rc_table_count_finish(rc_ctx *ctx, int min_count, int64_t *n_kmers)
{
    if (!ctx) return RC_ERR_ARG;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    static_cast<rc_ctx_full *>(ctx)->dump.valid = false;
    return rc_count_finish(ctx, min_count, n_kmers);
}

int rc_table_count_park(rc_ctx *ctx)
{
    if (!ctx) return RC_ERR_ARG;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    return rc_count_park(ctx);
}

// the table as `jellyfish dump` text (">COUNT\nKMER\n" per entry, canonical k-mer), in dump order
int rc_table_write_jfdump(rc_ctx *ctx, const char *path, int64_t *n_written)
{
    if (!ctx || !path) return RC_ERR_ARG;
    if (!ctx->d_buckets) {
        rc_set_error(ctx, "write_jfdump: no k-mer table loaded");
        return RC_ERR_STATE;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<uint64_t> codes;
    std::vector<int32_t> counts;
    int rc = rc_table_entries_in_dump_order(ctx, &codes, &counts);
    if (rc) return rc;
    FILE *fp = fopen(path, "wb");
    if (!fp) {
        rc_set_error(ctx, "could not open %s for writing", path);
        return RC_ERR_IO;
    }
    const int k = ctx->k;
    const size_t n = codes.size();
    unsigned T = std::thread::hardware_concurrency();
    if (T == 0) T = 4;
    if (T > 32) T = 32;
    const size_t CH = 1u << 20;  // entries formatted per round and thread
    std::vector<std::vector<char>> out(T);
    bool ok = true;
    for (size_t base = 0; base < n && ok; base += CH * T) {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t) {
            const size_t lo = std::min(n, base + (size_t)t * CH), hi = std::min(n, lo + CH);
            out[t].clear();
            if (lo >= hi) continue;
            th.emplace_back([&, t, lo, hi]() {
                std::vector<char> &o = out[t];
                o.resize((hi - lo) * (size_t)(k + 14));
                char *w = o.data();
                for (size_t i = lo; i < hi; ++i) {
                    *w++ = '>';
                    char tmp[12];
                    int nd = 0;
                    uint32_t v = (uint32_t)counts[i];
                    do {
                        tmp[nd++] = (char)('0' + v % 10);
                        v /= 10;
                    } while (v);
                    while (nd) *w++ = tmp[--nd];
                    *w++ = '\n';
                    for (int j = k - 1; j >= 0; --j) *w++ = "ACGT"[(codes[i] >> (2 * j)) & 3];
                    *w++ = '\n';
                }
                o.resize((size_t)(w - o.data()));
            });
        }
        for (auto &x : th) x.join();
        for (unsigned t = 0; t < T && ok; ++t)
            if (!out[t].empty() && fwrite(out[t].data(), 1, out[t].size(), fp) != out[t].size()) ok = false;
    }
    if (fclose(fp) != 0) ok = false;
    if (!ok) {
        rc_set_error(ctx, "short write on %s", path);
        return RC_ERR_IO;
    }
    if (n_written) *n_written = (int64_t)n;
    return RC_OK;
}

int rc_table_lookup(rc_ctx *ctx, const uint64_t *codes, size_t n, int32_t *out)
{
    if (!ctx || (n && (!codes || !out))) return RC_ERR_ARG;
    if (!ctx->d_buckets) {
        rc_set_error(ctx, "lookup: no k-mer table loaded");
        return RC_ERR_STATE;
    }
    if (n == 0) return RC_OK;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    rc_dev_tmp b_codes, b_out;
    RC_CHECK_HIP(ctx, b_codes.alloc(n * 8));
    RC_CHECK_HIP(ctx, b_out.alloc(n * 4));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(b_codes.p, codes, n * 8, hipMemcpyHostToDevice, ctx->stream));
    int rc = rc_launch_lookup(ctx, b_codes.as<uint64_t>(), n, b_out.as<int32_t>());
    if (rc) return rc;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(out, b_out.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

int rc_table_export(rc_ctx *ctx, uint64_t *codes, int32_t *counts, size_t cap, size_t *n_out)
{
    if (!ctx || !n_out || (cap && (!codes || !counts))) return RC_ERR_ARG;
    if (!ctx->d_buckets) {
        rc_set_error(ctx, "export: no k-mer table loaded");
        return RC_ERR_STATE;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    rc_dev_tmp b_codes, b_counts, b_n;
    unsigned long long n = 0;
    RC_CHECK_HIP(ctx, b_codes.alloc((cap + 1) * 8));
    RC_CHECK_HIP(ctx, b_counts.alloc((cap + 1) * 4));
    RC_CHECK_HIP(ctx, b_n.alloc(8));
    RC_CHECK_HIP(ctx, hipMemsetAsync(b_n.p, 0, 8, ctx->stream));
    int rc = rc_launch_export(ctx, b_codes.as<uint64_t>(), b_counts.as<int32_t>(), b_n.as<unsigned long long>(), cap);
    if (rc) return rc;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(&n, b_n.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t m = n < cap ? (size_t)n : cap;
    if (m) {
        RC_CHECK_HIP(ctx, hipMemcpy(codes, b_codes.p, m * 8, hipMemcpyDeviceToHost));
        RC_CHECK_HIP(ctx, hipMemcpy(counts, b_counts.p, m * 4, hipMemcpyDeviceToHost));
    }
    *n_out = (size_t)n;
    return RC_OK;
}

int rc_table_digest(rc_ctx *ctx, uint64_t *digest)
{
    if (!ctx || !digest) return RC_ERR_ARG;
    if (!ctx->d_buckets) {
        rc_set_error(ctx, "digest: no k-mer table loaded");
        return RC_ERR_STATE;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    rc_dev_tmp b;
    unsigned long long v = 0;
    RC_CHECK_HIP(ctx, b.alloc(8));
    RC_CHECK_HIP(ctx, hipMemsetAsync(b.p, 0, 8, ctx->stream));
    int rc = rc_launch_digest(ctx, b.as<unsigned long long>());
    if (rc) return rc;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(&v, b.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *digest = v;
    return RC_OK;
}

int rc_table_layout(const rc_ctx *ctx)
{
    if (!ctx) return RC_ERR_ARG;
    if (!ctx->d_buckets) return RC_ERR_STATE;
    return ctx->layout;
}

int rc_table_stats(const rc_ctx *ctx, uint64_t *bytes, uint64_t *buckets, uint64_t *entries)
{
    if (!ctx) return RC_ERR_ARG;
    if (bytes) *bytes = ctx->table_bytes;
    if (buckets) *buckets = ctx->nb_alloc;
    if (entries) *entries = ctx->n_entries;
    return RC_OK;
}

// ---- run parameters --------------------------------------------------------------------------
static int cmp_double(const void *a, const void *b)
{
    double d = *(const double *)a - *(const double *)b;  // CompDouble, main.cpp:39-48
    return d > 0 ? 1 : (d < 0 ? -1 : 0);
}

// main.cpp:310-358
int rc_estimate_error_rate(rc_ctx *c, double wk, double *rate_out)
{
    if (!c || !rate_out) return RC_ERR_ARG;
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    if (!ctx->d_buckets) {
        rc_set_error(ctx, "estimate_error_rate: no k-mer table loaded");
        return RC_ERR_STATE;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    // The scan keeps an entry when the largest count among its four last-base variants reaches 1000 and stops after
    // 100 000 of them (main.cpp:329-347): the probes, the test and the selection run on the device over the whole dump
    // (k_error_rate_candidates), the first 100 000 kept entries in dump order come back -- a few hundred KB instead of
    // 16 bytes per entry of the dump each way.
    const int rate_size = 100000;
    std::vector<uint64_t> vals;
    if (!ctx->dump.valid) {
        // no dump file was read (the table was counted here or handed over as arrays): the entries in the order
        // rc_table_write_jfdump would write them -- what the reference would see if it were given that dump
        uint64_t *d_codes = nullptr;
        size_t n = 0;
        int rc = rc_table_codes_device(ctx, &d_codes, &n);
        if (rc) return rc;
        rc = rc_error_rate_candidates(ctx, d_codes, n, true, (size_t)rate_size, &vals);
        (void)hipFree(d_codes);
        if (rc) return rc;
    } else {
        const rc_dump_cache &D = ctx->dump;
        // an entry that leaves an invalid KmerCode behind ends the scan (the IsValid() test at main.cpp:323)
        size_t n = 0;
        bool cut = D.load_state_invalid != 0;
        for (size_t c = 0; c < D.inv_mid.size() && !cut; ++c) {
            const std::vector<int8_t> &inv = D.inv_mid[c];
            const void *hit = inv.empty() ? nullptr : memchr(inv.data(), 1, inv.size());
            if (hit) {
                n += (size_t)((const int8_t *)hit - inv.data());
                cut = true;
            } else {
                n += inv.size();
            }
        }
        if (n) {
            rc_dev_tmp b_codes;
            RC_CHECK_HIP(ctx, b_codes.alloc(n * 8));
            size_t at = 0;
            for (const auto &ch : D.codes) {
                if (at >= n) break;
                const size_t take = std::min(ch.size(), n - at);
                if (take) RC_CHECK_HIP(ctx, hipMemcpyAsync(b_codes.as<uint64_t>() + at, ch.data(), take * 8, hipMemcpyHostToDevice, ctx->stream));
                at += take;
            }
            int rc = rc_error_rate_candidates(ctx, b_codes.as<uint64_t>(), n, false, (size_t)rate_size, &vals);
            if (rc) return rc;
        }
    }
    std::vector<double> store((size_t)rate_size + 2, 0.0);
    double *r = store.data() + 1;  // r[-1] readable, as in the reference when k == 0
    int cnt = 0;
    for (size_t i = 0; i < vals.size() && cnt < rate_size; ++i) {
        const int mx = (int)(uint32_t)(vals[i] >> 32), second = (int)(uint32_t)vals[i];
        r[cnt++] = (double)second / (double)mx;
    }
    qsort(r, (size_t)cnt, sizeof(double), cmp_double);
    r[cnt] = r[cnt - 1];
    double rate = r[(int)(cnt * wk)];
    if (rate == 0 || cnt < 100) rate = 0.01;
    *rate_out = rate;
    return RC_OK;
}

char rc_bad_quality_from_hist(const int32_t first_hist[300], const int32_t last_hist[300], int32_t total)
{
    int i, cnt = 0, t1, t2;  // main.cpp:108-127
    for (i = 0; i < 300; ++i) {
        cnt += first_hist[i];
        if (cnt > total * 0.05) break;
    }
    t1 = i - 1;
    cnt = 0;
    for (i = 0; i < 300; ++i) {
        cnt += last_hist[i];
        if (cnt > total * 0.05) break;
    }
    t2 = i;
    return (char)(t2 < t1 ? t2 : t1);
}

int rc_set_run_params(rc_ctx *ctx, double error_rate, char bad_quality)
{
    if (!ctx) return RC_ERR_ARG;
    ctx->P.error_rate = error_rate;
    ctx->P.bad_qual = (int)(signed char)bad_quality;
    // the first integer steps of GetBound at this rate (rc_common.h), computed here with the host's -- the
    // reference's -- arithmetic; they travel to the correction kernel with its arguments
    std::vector<uint32_t> steps_v(RC_BOUND_STEPS);
    uint32_t *steps = steps_v.data();
    rc_bound_steps_build(error_rate, steps);
    for (int v = 0; v < RC_BS_INLINE; ++v) ctx->P.bs[v] = steps[v];
    // ... and the whole table stays in device memory for the thresholds beyond those
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t steps_bytes = (size_t)RC_BOUND_STEPS * sizeof(uint32_t);
    int rc = rc_dbuf_reserve(ctx, &ctx->bs_dev, steps_bytes);
    if (rc) return rc;
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (a batch in flight may still read the old table)
    RC_CHECK_HIP(ctx, hipMemcpy(ctx->bs_dev.p, steps, steps_bytes, hipMemcpyHostToDevice));
    ctx->P.bs_ext = getenv("RC_NO_BS_EXT") ? nullptr : (const uint32_t *)ctx->bs_dev.p;
    ctx->P.flags = ctx->env_no_alt ? RC_PF_NO_ALT : 0;
    ctx->params_set = true;
    return RC_OK;
}

int rc_set_quality_bits(rc_ctx *ctx, int on)
{
    if (!ctx) return RC_ERR_ARG;
    ctx->qual_bits = on != 0;
    return RC_OK;
}

void rc_pack_quality_bits(const char *qual, size_t nbytes, char bad_quality, uint8_t *bits)
{
    const signed char bq = (signed char)bad_quality;
    size_t p = 0;
    for (; p + 8 <= nbytes; p += 8) {
        unsigned v = 0;
        for (int j = 0; j < 8; ++j) v |= (unsigned)((signed char)qual[p + j] > bq) << j;
        bits[p >> 3] = (uint8_t)v;
    }
    if (p < nbytes) {
        unsigned v = 0;
        for (int j = 0; p + j < nbytes; ++j) v |= (unsigned)((signed char)qual[p + j] > bq) << j;
        bits[p >> 3] = (uint8_t)v;
    }
}

// ---- correction ------------------------------------------------------------------------------
int rc_probe_device(rc_ctx *ctx, const uint8_t *d_seq, uint64_t nbytes, int32_t *d_counts)
{
    if (!ctx || !d_seq || !d_counts) return RC_ERR_ARG;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    return rc_launch_probe(ctx, d_seq, (size_t)nbytes, d_counts);
}

static int correct_device_impl(rc_ctx *ctx, const rc_device_batch *b, uint32_t qual_split, uint32_t qual_base2, int qual_bits = -1);

int rc_correct_device(rc_ctx *ctx, const rc_device_batch *b) { return correct_device_impl(ctx, b, 0xFFFFFFFFu, 0); }

// qual_split / qual_base2 (quality-bit mode only): arena bytes from qual_split on have their bits at
// byte qual_base2 of d_qual -- the second arena of a paired host batch, whose bit array is separate
// qual_bits: -1 = as rc_set_quality_bits says, 0 / 1 = this batch's quality arena holds bytes / bits (the packed boundary)
static int correct_device_impl(rc_ctx *ctx, const rc_device_batch *b, uint32_t qual_split, uint32_t qual_base2, int qual_bits)
{
    if (!ctx || !b) return RC_ERR_ARG;
    if (b->n_reads == 0) return RC_OK;
    if (b->mode < 0 || b->mode > 2 || !b->d_seq || !b->d_qual || !b->d_off || !b->d_ret || !b->d_l || !b->d_m || !b->d_h) {
        rc_set_error(ctx, "correct_device: bad batch descriptor");
        return RC_ERR_ARG;
    }
    if (b->nbytes >= (1ull << 32)) {
        rc_set_error(ctx, "correct_device: arena of %llu bytes exceeds the 4 GiB batch limit", (unsigned long long)b->nbytes);
        return RC_ERR_ARG;
    }
    if (!ctx->d_buckets) {  // (before anything is launched: every probe kernel dereferences the table)
        rc_set_error(ctx, "correct: no k-mer table loaded");
        return RC_ERR_STATE;
    }
    // mates travel together (main.cpp:441, :459-468): an odd read count in a paired or interleaved batch has a
    // read without a mate -- refused before the locality order or the pair exchange of the threshold kernel see it
    if (b->mode != 0 && (b->n_reads & 1u)) {
        rc_set_error(ctx, "correct: %s mode needs an even number of reads (got %u)", b->mode == 1 ? "paired" : "interleaved", b->n_reads);
        return RC_ERR_ARG;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    bool fused = false;  // probe and threshold kernels ran as one
    if ((rc = rc_dbuf_reserve(ctx, &ctx->counts, (size_t)b->nbytes * 4 + 256))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->strong, (size_t)b->n_reads * 4 + 256))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->info, (size_t)b->n_reads * 4 + 256))) return rc;
    rc_device_batch_args a;
    a.mode = b->mode;
    a.n = b->n_reads;
    a.seq = b->d_seq;
    a.qual = b->d_qual;
    a.qual_bits = qual_bits >= 0 ? qual_bits : (ctx->qual_bits ? 1 : 0);
    a.qual_split = qual_split;
    a.qual_base2 = qual_base2;
    a.off = b->d_off;
    a.ret = b->d_ret;
    a.l = b->d_l;
    a.m = b->d_m;
    a.h = b->d_h;
    a.max_len = b->max_read_len;
    // the reads the threshold kernel could not finish, as a work list; isolated substitutions are finished four reads to
    // a wave first (rc_single.h: it clears their cls), what is left is k_correct's list
    auto single_and_compact = [&](const rc_device_batch_args &at) -> int {
        if (!ctx->cls_ready) return RC_OK;
        ctx->work_stride = ((size_t)at.n + 63) & ~(size_t)63;
        int e;
        if ((e = rc_dbuf_reserve(ctx, &ctx->worklist, ctx->work_stride * RC_WORK_CLASSES * 4 + 256))) return e;
        bool ran = false;
        if ((e = rc_launch_single(ctx, at, &ran))) return e;
        return rc_launch_compact(ctx, (const uint8_t *)ctx->cls.p, at.n, (uint32_t *)ctx->worklist.p, ctx->work_stride,
                                 (uint32_t *)((char *)ctx->work.p + RC_WORK_NWORK_OFF));
    };
    // large batches over a table that does not fit the caches are probed in min-hash order (rc_table.hip),
    // so that overlapping reads meet in the L2 / Infinity Cache
    const bool locality = ctx->locality_mode >= 0 && (ctx->locality_mode > 0 || (a.n >= (1u << 18) && ctx->table_bytes > ((size_t)128 << 20))) &&
                          a.max_len + 8 <= 4000;
    // Length tiers.  The reference treats every read of up to 1 023 bases alike (utils.h:7, ErrorCorrection.cpp:682-1480);
    // here the fast kernels -- the fused probe + threshold kernel, k_single, the compiled-for-k k_correct -- hold reads
    // of up to 160 bases, the quarter-wave threshold kernel 320, and the longest read of a batch used to decide for all
    // of them.  A batch with longer reads is now processed in up to three passes over the same arena, one per tier a
    // unit's longer read falls into: S (<= 160 bases), M (the quarter-wave layout: <= 320 bases / 256 k-mers), L (the
    // rest); each pass = threshold kernel -> (k_single) -> compaction -> k_correct with the tier's capacity class, and the
    // threshold kernel of a pass marks the other tiers' reads cls = 0.  Same results (a read's result depends on its unit
    // alone), the short reads of a mixed batch keep their kernels.  Needs the classification (work lists).
    const int S_HI = 160;
    const int m_hi = RC_Q_MAX_KCNT - 1 + ctx->k < RC_Q_MAX_LEN ? RC_Q_MAX_KCNT - 1 + ctx->k : RC_Q_MAX_LEN;
    const bool tiered = a.max_len > S_HI && !ctx->env_no_tier && !ctx->env_no_classify && !ctx->env_k2_wave_per_read && ctx->trace_cap == 0;
    if (tiered) {
        rc_device_batch_args at = a;
        at.tier_lo = -1;
        at.tier_hi = S_HI;
        at.max_len = S_HI;
        bool lists = false;  // the middle / long tier's reads as lists in locality order (rc_launch_tier_lists)
        if (locality) {
            if ((rc = rc_launch_locality_order(ctx, a, (size_t)b->nbytes))) return rc;
            if ((rc = rc_launch_probe_threshold_list(ctx, at, (size_t)b->nbytes, &fused))) return rc;
            if (fused) {
                // the other tiers' reads are a few per cent of a typical mixed batch: probed (and, the middle tier,
                // thresholded) through compact lists -- walking the whole batch for them cost 3.7 + 3.1 ms of a 25 M-read step
                rc_device_batch_args al = a;
                if ((rc = rc_launch_tier_lists(ctx, a, S_HI, m_hi))) return rc;
                al.max_len = a.max_len < m_hi ? a.max_len : m_hi;
                if ((rc = rc_launch_probe_tier(ctx, al, (size_t)b->nbytes, (int32_t *)ctx->counts.p, 0))) return rc;
                al.max_len = a.max_len;
                if (a.max_len > m_hi && (rc = rc_launch_probe_tier(ctx, al, (size_t)b->nbytes, (int32_t *)ctx->counts.p, 1))) return rc;
                lists = true;
            } else if ((rc = rc_launch_probe_list(ctx, a, (size_t)b->nbytes, (int32_t *)ctx->counts.p, -1)))
                return rc;
        } else if ((rc = rc_launch_probe(ctx, b->d_seq, (size_t)b->nbytes, (int32_t *)ctx->counts.p)))
            return rc;
        ctx->thr_ready = true;  // every pass runs a threshold kernel: k_correct never computes a threshold itself
        for (int tier = 0; tier < 3; ++tier) {
            if (tier == 1) {
                at.tier_lo = S_HI;
                at.tier_hi = m_hi;
                at.max_len = a.max_len < m_hi ? a.max_len : m_hi;
                if (lists) {
                    at.tier_list = (const uint32_t *)ctx->tier_list.p;
                    at.tier_n = (const uint32_t *)((char *)ctx->work.p + RC_WORK_NTIER_OFF);
                }
            } else if (tier == 2) {
                if (a.max_len <= m_hi) break;
                at.tier_lo = m_hi;
                at.tier_hi = RC_TIER_ALL;
                at.max_len = a.max_len;
                at.tier_list = at.tier_n = nullptr;  // (the wave-per-read threshold kernel walks the batch)
            }
            if (!(tier == 0 && fused) && (rc = rc_launch_threshold(ctx, at, true))) return rc;
            if (!ctx->cls_ready) {
                rc_set_error(ctx, "correct: internal: a length tier ran without classification");
                return RC_ERR_STATE;
            }
            if ((rc = single_and_compact(at))) return rc;
            if ((rc = rc_launch_correct(ctx, at))) return rc;
        }
        return rc_launch_summary(ctx, a.ret, a.n);
    }
    if (locality) {
        if ((rc = rc_launch_locality_order(ctx, a, (size_t)b->nbytes))) return rc;
        // probe + threshold + classification in one kernel where the reads fit it
        if ((rc = rc_launch_probe_threshold_list(ctx, a, (size_t)b->nbytes, &fused))) return rc;
        if (!fused && (rc = rc_launch_probe_list(ctx, a, (size_t)b->nbytes, (int32_t *)ctx->counts.p))) return rc;
    } else if ((rc = rc_launch_probe(ctx, b->d_seq, (size_t)b->nbytes, (int32_t *)ctx->counts.p)))
        return rc;
    // thresholds: mates need each other's before either can be corrected, so paired / interleaved
    // batches always run the threshold kernel first; single-end batches do too when every read fits
    // the four-reads-per-wave kernel (cheaper there than inside k_correct), else k_correct computes them
    ctx->thr_ready = fused;
    if (!fused) ctx->cls_ready = false;
    const bool quarter_ok = a.max_len <= 320 && a.max_len - ctx->k + 1 <= 256 && !ctx->env_k2_wave_per_read;
    if (!fused && (a.mode != 0 || quarter_ok)) {
        if ((rc = rc_launch_threshold(ctx, a, true))) return rc;
        ctx->thr_ready = true;
    }
    if ((rc = single_and_compact(a))) return rc;
    if ((rc = rc_launch_correct(ctx, a))) return rc;
    // UpdateSummary (main.cpp:73-79), on the device: the counters live in HBM until rc_summary() asks
    return rc_launch_summary(ctx, a.ret, a.n);
}

// GetStrongTrustedThreshold (ErrorCorrection.h:26, ErrorCorrection.cpp:1482-1565) for every read of
// an arena in HBM: probe kernel + threshold kernel, the per-read values copied to d_strong
int rc_strong_threshold_device(rc_ctx *ctx, const uint8_t *d_seq, const uint32_t *d_off, uint32_t n_reads, uint64_t nbytes,
                               int32_t max_read_len, int32_t *d_strong)
{
    if (!ctx || !d_seq || !d_off || !d_strong) return RC_ERR_ARG;
    if (n_reads == 0) return RC_OK;
    if (nbytes >= (1ull << 32)) {
        rc_set_error(ctx, "strong_threshold_device: arena of %llu bytes exceeds the 4 GiB batch limit", (unsigned long long)nbytes);
        return RC_ERR_ARG;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->counts, (size_t)nbytes * 4 + 256))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->strong, (size_t)n_reads * 4 + 256))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->info, (size_t)n_reads * 4 + 256))) return rc;
    rc_device_batch_args a = rc_device_batch_args();  // (value-initialised: zeros, and the members with defaults -- no tiers)
    a.mode = 0;
    a.n = n_reads;
    a.seq = const_cast<uint8_t *>(d_seq);
    a.off = d_off;
    a.max_len = max_read_len;
    if ((rc = rc_launch_probe(ctx, d_seq, (size_t)nbytes, (int32_t *)ctx->counts.p))) return rc;
    if ((rc = rc_launch_threshold(ctx, a, false))) return rc;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(d_strong, ctx->strong.p, (size_t)n_reads * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return RC_OK;
}

// ---- per-read entry points: the granularity of ErrorCorrection.h:26-28, each a batch of one through the kernels above
// (a launch and two copies per call -- for bindings that work read by read and for spot checks, not for throughput)
static int one_read_upload(rc_ctx *ctx, const char *seq, const char *qual, rc_device_batch_args &a, size_t *len1)
{
    if (!seq) return RC_ERR_ARG;
    const size_t n1 = strlen(seq) + 1;
    if (n1 > RC_MAX_READ_LENGTH) {
        rc_set_error(ctx, "read of %zu bases exceeds the %d-base limit (utils.h:7)", n1 - 1, RC_MAX_READ_LENGTH - 1);
        return RC_ERR_ARG;
    }
    if (ctx->qual_bits) {
        rc_set_error(ctx, "the per-read entry points take quality bytes (rc_set_quality_bits is on)");
        return RC_ERR_STATE;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->h_seq, n1 + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->h_qual, n1 + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->h_off, 2 * 4))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->h_res, 4 * 4))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->counts, n1 * 4 + 256))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->strong, 4 + 256))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->info, 4 + 256))) return rc;
    const uint32_t off[2] = {0u, (uint32_t)n1};
    RC_CHECK_HIP(ctx, hipMemcpyAsync(ctx->h_seq.p, seq, n1, hipMemcpyHostToDevice, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemsetAsync(ctx->h_qual.p, 0, n1, ctx->stream));  // (no qualities: the FASTA marker qual[0] == 0)
    if (qual) {
        const size_t q1 = strnlen(qual, n1 - 1);
        RC_CHECK_HIP(ctx, hipMemcpyAsync(ctx->h_qual.p, qual, q1, hipMemcpyHostToDevice, ctx->stream));
    }
    RC_CHECK_HIP(ctx, hipMemcpyAsync(ctx->h_off.p, off, sizeof off, hipMemcpyHostToDevice, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (off and possibly seq are on the caller's stack)
    int32_t *d_res = (int32_t *)ctx->h_res.p;
    a = rc_device_batch_args();
    a.mode = 0;
    a.n = 1;
    a.seq = (uint8_t *)ctx->h_seq.p;
    a.qual = (const uint8_t *)ctx->h_qual.p;
    a.off = (const uint32_t *)ctx->h_off.p;
    a.ret = d_res;
    a.l = d_res + 1;
    a.m = d_res + 2;
    a.h = d_res + 3;
    a.max_len = (int)n1 - 1;
    *len1 = n1;
    return RC_OK;
}

int rc_strong_threshold_read(rc_ctx *ctx, const char *seq, int32_t *strong)
{
    if (!ctx || !seq || !strong) return RC_ERR_ARG;
    rc_device_batch_args a;
    size_t n1;
    int rc = one_read_upload(ctx, seq, nullptr, a, &n1);
    if (rc) return rc;
    if ((rc = rc_launch_probe(ctx, a.seq, n1, (int32_t *)ctx->counts.p))) return rc;
    if ((rc = rc_launch_threshold(ctx, a, false))) return rc;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(strong, ctx->strong.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

int rc_correct_read(rc_ctx *ctx, char *seq, const char *qual, int32_t pair_strong_threshold, int32_t *ret)
{
    if (!ctx || !seq || !ret) return RC_ERR_ARG;
    rc_device_batch_args a;
    size_t n1;
    int rc = one_read_upload(ctx, seq, qual, a, &n1);
    if (rc) return rc;
    a.pair_override = pair_strong_threshold;
    if ((rc = rc_launch_probe(ctx, a.seq, n1, (int32_t *)ctx->counts.p))) return rc;
    // no threshold kernel, no classification: k_correct computes the read's own threshold (its single-end front end) and
    // takes the pair's from the argument, exactly the reference's call
    ctx->thr_ready = false;
    ctx->cls_ready = false;
    ctx->cand_ready = false;
    if ((rc = rc_launch_correct(ctx, a))) return rc;
    if ((rc = rc_launch_summary(ctx, a.ret, 1))) return rc;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(seq, a.seq, n1 - 1, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(ret, a.ret, 4, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

int rc_kmer_info_read(rc_ctx *ctx, const char *seq, int32_t *l, int32_t *m, int32_t *h)
{
    if (!ctx || !seq || !l || !m || !h) return RC_ERR_ARG;
    rc_device_batch_args a;
    size_t n1;
    int rc = one_read_upload(ctx, seq, nullptr, a, &n1);
    if (rc) return rc;
    if ((rc = rc_launch_probe(ctx, a.seq, n1, (int32_t *)ctx->counts.p))) return rc;
    if ((rc = rc_launch_kmer_info(ctx, a))) return rc;
    int32_t out[3];
    RC_CHECK_HIP(ctx, hipMemcpyAsync(out, a.l, 12, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *l = out[0];
    *m = out[1];
    *h = out[2];
    return RC_OK;
}

int rc_sync(rc_ctx *ctx)
{
    if (!ctx) return RC_ERR_ARG;
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

static int correct_batch_impl(rc_ctx *c, rc_batch *b, rc_trace *t);

int rc_submit(rc_ctx *c, const rc_batch *b, int slot);
int rc_wait(rc_ctx *c, int slot);

int rc_correct_batch(rc_ctx *c, rc_batch *b)
{
    int rc = rc_submit(c, b, 0);
    if (rc) return rc;
    return rc_wait(c, 0);
}

// rc_correct_batch + what the reference prints under -verbose (VERBOSE, ErrorCorrection.cpp:15):
// the counts before (:759-770) and after (:1590-1597) come from two extra runs of the probe
// kernel, the per-iteration thresholds and bitmaps (:856-857, :1088-1094) from the TRACE build of
// k_correct
int rc_correct_batch_traced(rc_ctx *c, rc_batch *b, rc_trace *t)
{
    if (!c || !b || !t) return RC_ERR_ARG;
    if (t->max_iter < 1 || !t->counts_before || !t->counts_after || !t->flags || !t->n_iter || !t->iter) {
        rc_set_error(c, "correct_batch_traced: bad trace descriptor");
        return RC_ERR_ARG;
    }
    if (c->qual_bits) {
        rc_set_error(c, "correct_batch_traced: not available in quality-bit mode");
        return RC_ERR_STATE;
    }
    c->trace_cap = t->max_iter;
    int rc = correct_batch_impl(c, b, t);
    c->trace_cap = 0;
    return rc;
}

static int correct_batch_impl(rc_ctx *c, rc_batch *b, rc_trace *t)
{
    if (!c || !b) return RC_ERR_ARG;
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    if (b->n == 0) return RC_OK;
    if (b->mode < 0 || b->mode > 2 || !b->seq || !b->qual || !b->off || !b->ret || !b->l || !b->m || !b->h ||
        (b->mode == 1 && (!b->seq2 || !b->qual2 || !b->off2))) {
        rc_set_error(ctx, "correct_batch: bad batch descriptor");
        return RC_ERR_ARG;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n1 = b->n;
    const size_t bytes1 = b->off[n1], bytes2 = b->mode == 1 ? b->off2[n1] : 0;
    const size_t total_reads = b->mode == 1 ? 2 * n1 : n1;
    const size_t nbytes = bytes1 + bytes2;
    if (nbytes >= (1ull << 32) || total_reads >= (1ull << 32)) {
        rc_set_error(ctx, "correct_batch: batch too large (split it)");
        return RC_ERR_ARG;
    }
    std::vector<uint32_t> off(total_reads + 1);
    int max_len = 0;
    for (size_t i = 0; i <= n1; ++i) off[i] = b->off[i];
    for (size_t i = 0; i < n1; ++i) max_len = std::max(max_len, (int)(b->off[i + 1] - b->off[i]) - 1);
    if (b->mode == 1) {
        for (size_t i = 0; i <= n1; ++i) off[n1 + i] = (uint32_t)bytes1 + b->off2[i];
        for (size_t i = 0; i < n1; ++i) max_len = std::max(max_len, (int)(b->off2[i + 1] - b->off2[i]) - 1);
    }
    int rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->h_seq, nbytes + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->h_qual, nbytes + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->h_off, (total_reads + 1) * 4))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->h_res, total_reads * 16))) return rc;
    uint8_t *d_seq = (uint8_t *)ctx->h_seq.p, *d_qual = (uint8_t *)ctx->h_qual.p;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(d_seq, b->seq, bytes1, hipMemcpyHostToDevice, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(d_qual, b->qual, bytes1, hipMemcpyHostToDevice, ctx->stream));
    if (b->mode == 1) {
        RC_CHECK_HIP(ctx, hipMemcpyAsync(d_seq + bytes1, b->seq2, bytes2, hipMemcpyHostToDevice, ctx->stream));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(d_qual + bytes1, b->qual2, bytes2, hipMemcpyHostToDevice, ctx->stream));
    }
    RC_CHECK_HIP(ctx, hipMemcpyAsync(ctx->h_off.p, off.data(), (total_reads + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    int32_t *d_res = (int32_t *)ctx->h_res.p;
    rc_device_batch db;
    db.mode = b->mode;
    db.n_reads = (uint32_t)total_reads;
    db.nbytes = nbytes;
    db.max_read_len = max_len;
    db.d_seq = d_seq;
    db.d_qual = d_qual;
    db.d_off = (const uint32_t *)ctx->h_off.p;
    db.d_ret = d_res;
    db.d_l = d_res + total_reads;
    db.d_m = d_res + 2 * total_reads;
    db.d_h = d_res + 3 * total_reads;
    if ((rc = rc_correct_device(ctx, &db))) return rc;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(b->seq, d_seq, bytes1, hipMemcpyDeviceToHost, ctx->stream));
    if (b->mode == 1) RC_CHECK_HIP(ctx, hipMemcpyAsync(b->seq2, d_seq + bytes1, bytes2, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(b->ret, db.d_ret, total_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(b->l, db.d_l, total_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(b->m, db.d_m, total_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(b->h, db.d_h, total_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (t) {
        // before: K1's output is still in ctx->counts; after: probe the corrected arena once more
        RC_CHECK_HIP(ctx, hipMemcpyAsync(t->counts_before, ctx->counts.p, nbytes * 4, hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = rc_launch_probe(ctx, d_seq, nbytes, (int32_t *)ctx->counts.p))) return rc;
        RC_CHECK_HIP(ctx, hipMemcpyAsync(t->counts_after, ctx->counts.p, nbytes * 4, hipMemcpyDeviceToHost, ctx->stream));
        const size_t rec = 2 + (size_t)t->max_iter * RC_TRACE_WORDS;
        std::vector<int32_t> raw(total_reads * rec);
        RC_CHECK_HIP(ctx, hipMemcpyAsync(raw.data(), ctx->trace.p, raw.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < total_reads; ++i) {
            const int32_t *r = raw.data() + i * rec;
            t->flags[i] = r[0];
            t->n_iter[i] = r[1];
            memcpy(t->iter + i * (size_t)t->max_iter * RC_TRACE_WORDS, r + 2, (size_t)t->max_iter * RC_TRACE_WORDS * 4);
        }
    }
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

// ---- asynchronous host-buffer path ---------------------------------------------------------------
// The reference overlaps the I/O of batch N+1 with the correction of batch N by handing batches to
// worker threads (main.cpp:479-516).  Here one context keeps up to RC_MAX_SLOTS batches in flight on
// three streams: H2D(N+1) || kernels(N) || D2H(N-1).  Scratch memory of the kernels is shared --
// they serialise on the compute stream -- only the arenas and result arrays exist per slot.
static int hbuf_reserve(rc_ctx *ctx, rc_hbuf *h, size_t bytes)
{
    if (bytes <= h->bytes) return RC_OK;
    if (h->p) (void)hipHostFree(h->p);
    h->p = nullptr;
    h->bytes = 0;
    const size_t want = bytes + bytes / 8 + 4096;
    RC_CHECK_HIP(ctx, hipHostMalloc(&h->p, want, hipHostMallocDefault));
    h->bytes = want;
    return RC_OK;
}

static bool is_pinned_at(const void *p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // pageable memory the runtime has never seen
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// the whole range [p, p + bytes) is page-locked: its first and last byte are (a registration or a
// hipHostMalloc block is one contiguous range, so a buffer that starts and ends inside pinned memory and was
// handed over as one array lies in it -- unless it straddles two separate registrations, which then both
// cover their part)
static bool is_pinned(const void *p, size_t bytes)
{
    if (!p) return false;
    if (!is_pinned_at(p)) return false;
    return bytes <= 1 || is_pinned_at(static_cast<const char *>(p) + bytes - 1);
}

int rc_host_alloc(rc_ctx *ctx, size_t bytes, void **out)
{
    if (!ctx || !out) return RC_ERR_ARG;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    RC_CHECK_HIP(ctx, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return RC_OK;
}

int rc_host_free(rc_ctx *ctx, void *p)
{
    if (!ctx) return RC_ERR_ARG;
    if (p) RC_CHECK_HIP(ctx, hipHostFree(p));
    return RC_OK;
}

// page-locks caller memory (any allocation, whole pages) so that rc_submit can DMA straight from / to it
int rc_host_register(void *p, size_t bytes)
{
    if (!p || !bytes) return RC_ERR_ARG;
    return hipHostRegister(p, bytes, hipHostRegisterPortable) == hipSuccess ? RC_OK : RC_ERR_HIP;
}

int rc_host_unregister(void *p)
{
    if (!p) return RC_ERR_ARG;
    return hipHostUnregister(p) == hipSuccess ? RC_OK : RC_ERR_HIP;
}

static int slots_init(rc_ctx *ctx)
{
    if (ctx->slots) return RC_OK;
    RC_CHECK_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_h2d, hipStreamNonBlocking));
    RC_CHECK_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_d2h, hipStreamNonBlocking));
    ctx->slots = new (std::nothrow) rc_slot[RC_MAX_SLOTS];
    if (!ctx->slots) return RC_ERR_NOMEM;
    for (int i = 0; i < RC_MAX_SLOTS; ++i) {
        rc_slot &sl = ctx->slots[i];
        RC_CHECK_HIP(ctx, hipEventCreateWithFlags(&sl.e_h2d, hipEventDisableTiming));
        RC_CHECK_HIP(ctx, hipEventCreateWithFlags(&sl.e_k, hipEventDisableTiming));
        RC_CHECK_HIP(ctx, hipEventCreateWithFlags(&sl.e_done, hipEventDisableTiming));
    }
    return RC_OK;
}

int rc_submit(rc_ctx *c, const rc_batch *b, int slot)
{
    if (!c || !b || slot < 0 || slot >= RC_MAX_SLOTS) return RC_ERR_ARG;
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    if (b->mode < 0 || b->mode > 2 || (b->n && (!b->seq || !b->qual || !b->off || !b->ret || !b->l || !b->m || !b->h)) ||
        (b->n && b->mode == 1 && (!b->seq2 || !b->qual2 || !b->off2))) {
        rc_set_error(ctx, "submit: bad batch descriptor");
        return RC_ERR_ARG;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = slots_init(ctx);
    if (rc) return rc;
    rc_slot &sl = ctx->slots[slot];
    if (sl.busy) {
        rc_set_error(ctx, "submit: slot %d still holds a batch (rc_wait it first)", slot);
        return RC_ERR_STATE;
    }
    sl.b = *b;
    sl.pb = nullptr;
    sl.rb = nullptr;
    const size_t n1 = b->n;
    sl.total_reads = b->mode == 1 ? 2 * n1 : n1;
    sl.bytes1 = n1 ? b->off[n1] : 0;
    sl.bytes2 = (n1 && b->mode == 1) ? b->off2[n1] : 0;
    if (n1 == 0) {
        sl.busy = true;
        return RC_OK;
    }
    const size_t nbytes = sl.bytes1 + sl.bytes2, total = sl.total_reads;
    if (nbytes >= (1ull << 32) || total >= (1ull << 32)) {
        rc_set_error(ctx, "submit: batch too large (split it)");
        return RC_ERR_ARG;
    }
    if (b->mode == 2 && (n1 & 1)) {  // (before any copy is queued)
        rc_set_error(ctx, "submit: interleaved mode needs an even number of reads (got %zu)", n1);
        return RC_ERR_ARG;
    }
    if (!ctx->d_buckets) {
        rc_set_error(ctx, "correct: no k-mer table loaded");
        return RC_ERR_STATE;
    }
    // offsets of the device arena (arena 1 then arena 2) and the longest read, into pinned memory
    if ((rc = hbuf_reserve(ctx, &sl.p_off, (total + 1) * 4))) return rc;
    uint32_t *off = (uint32_t *)sl.p_off.p;
    int max_len = 0;
    memcpy(off, b->off, (n1 + 1) * 4);
    for (size_t i = 0; i < n1; ++i) max_len = std::max(max_len, (int)(b->off[i + 1] - b->off[i]) - 1);
    if (b->mode == 1) {
        for (size_t i = 0; i <= n1; ++i) off[n1 + i] = (uint32_t)sl.bytes1 + b->off2[i];
        for (size_t i = 0; i < n1; ++i) max_len = std::max(max_len, (int)(b->off2[i + 1] - b->off2[i]) - 1);
    }
    // quality arenas: a byte per base, or (rc_set_quality_bits) a bit per arena byte, arena 2's bits in
    // a region of their own
    const bool qbits = ctx->qual_bits;
    const size_t q1 = qbits ? (sl.bytes1 + 7) / 8 : sl.bytes1, q2 = qbits ? (sl.bytes2 + 7) / 8 : sl.bytes2;
    const size_t qbase2 = qbits ? ((q1 + 15) & ~(size_t)15) : sl.bytes1;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_seq, nbytes + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_qual, qbase2 + q2 + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_off, (total + 1) * 4))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_res, total * 16))) return rc;
    sl.seq_pinned = is_pinned(b->seq, sl.bytes1) && is_pinned(b->qual, q1) &&
                    (b->mode != 1 || (is_pinned(b->seq2, sl.bytes2) && is_pinned(b->qual2, q2)));
    sl.res_pinned = is_pinned(b->ret, total * 4) && is_pinned(b->l, total * 4) && is_pinned(b->m, total * 4) && is_pinned(b->h, total * 4);
    const char *h_seq1 = b->seq, *h_qual1 = b->qual, *h_seq2 = b->seq2, *h_qual2 = b->qual2;
    if (!sl.seq_pinned) {  // pageable buffers: through the slot's pinned staging
        if ((rc = hbuf_reserve(ctx, &sl.p_seq, nbytes))) return rc;
        if ((rc = hbuf_reserve(ctx, &sl.p_qual, qbase2 + q2))) return rc;
        memcpy(sl.p_seq.p, b->seq, sl.bytes1);
        memcpy(sl.p_qual.p, b->qual, q1);
        if (b->mode == 1) {
            memcpy((char *)sl.p_seq.p + sl.bytes1, b->seq2, sl.bytes2);
            memcpy((char *)sl.p_qual.p + qbase2, b->qual2, q2);
        }
        h_seq1 = (const char *)sl.p_seq.p;
        h_qual1 = (const char *)sl.p_qual.p;
        h_seq2 = h_seq1 + sl.bytes1;
        h_qual2 = h_qual1 + qbase2;
    }
    if (!sl.res_pinned && (rc = hbuf_reserve(ctx, &sl.p_res, total * 16))) return rc;
    uint8_t *d_seq = (uint8_t *)sl.d_seq.p, *d_qual = (uint8_t *)sl.d_qual.p;
    // one upload stream: bases and qualities on two streams measured 21 GB/s against 26.6 GB/s on one
    // (the link, not a DMA engine, is the bound)
    hipStream_t sq = ctx->s_h2d;
    // from here on copies are in flight from the caller's buffers (or the slot's staging): an error must not
    // return before they have drained, or the caller could free / the next submit could overwrite memory the
    // DMA engines still read
    struct drain_on_error {
        rc_ctx *c;
        bool armed = true;
        ~drain_on_error()
        {
            if (!armed) return;
            (void)hipStreamSynchronize(c->s_h2d);
            (void)hipStreamSynchronize(c->stream);
            (void)hipStreamSynchronize(c->s_d2h);
        }
    } guard{ctx};
    RC_CHECK_HIP(ctx, hipMemcpyAsync(d_seq, h_seq1, sl.bytes1, hipMemcpyHostToDevice, ctx->s_h2d));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(d_qual, h_qual1, q1, hipMemcpyHostToDevice, sq));
    if (b->mode == 1) {
        RC_CHECK_HIP(ctx, hipMemcpyAsync(d_seq + sl.bytes1, h_seq2, sl.bytes2, hipMemcpyHostToDevice, ctx->s_h2d));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(d_qual + qbase2, h_qual2, q2, hipMemcpyHostToDevice, sq));
    }
    RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.d_off.p, off, (total + 1) * 4, hipMemcpyHostToDevice, ctx->s_h2d));
    RC_CHECK_HIP(ctx, hipEventRecord(sl.e_h2d, ctx->s_h2d));
    // kernels
    RC_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->stream, sl.e_h2d, 0));
    int32_t *d_res = (int32_t *)sl.d_res.p;
    rc_device_batch db;
    db.mode = b->mode;
    db.n_reads = (uint32_t)total;
    db.nbytes = nbytes;
    db.max_read_len = max_len;
    db.d_seq = d_seq;
    db.d_qual = d_qual;
    db.d_off = (const uint32_t *)sl.d_off.p;
    db.d_ret = d_res;
    db.d_l = d_res + total;
    db.d_m = d_res + 2 * total;
    db.d_h = d_res + 3 * total;
    if ((rc = correct_device_impl(ctx, &db, qbits && b->mode == 1 ? (uint32_t)sl.bytes1 : 0xFFFFFFFFu, (uint32_t)qbase2))) return rc;
    RC_CHECK_HIP(ctx, hipEventRecord(sl.e_k, ctx->stream));
    // results
    RC_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->s_d2h, sl.e_k, 0));
    char *o_seq1 = sl.seq_pinned ? b->seq : (char *)sl.p_seq.p;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(o_seq1, d_seq, sl.bytes1, hipMemcpyDeviceToHost, ctx->s_d2h));
    if (b->mode == 1) {
        char *o_seq2 = sl.seq_pinned ? b->seq2 : (char *)sl.p_seq.p + sl.bytes1;
        RC_CHECK_HIP(ctx, hipMemcpyAsync(o_seq2, d_seq + sl.bytes1, sl.bytes2, hipMemcpyDeviceToHost, ctx->s_d2h));
    }
    if (sl.res_pinned) {
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->ret, db.d_ret, total * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->l, db.d_l, total * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->m, db.d_m, total * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->h, db.d_h, total * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
    } else {
        RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.p_res.p, d_res, total * 16, hipMemcpyDeviceToHost, ctx->s_d2h));
    }
    RC_CHECK_HIP(ctx, hipEventRecord(sl.e_done, ctx->s_d2h));
    guard.armed = false;
    sl.busy = true;
    return RC_OK;
}

int rc_wait(rc_ctx *c, int slot)
{
    if (!c || slot < 0 || slot >= RC_MAX_SLOTS) return RC_ERR_ARG;
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    if (!ctx->slots || !ctx->slots[slot].busy) {
        rc_set_error(ctx, "wait: slot %d holds no batch", slot);
        return RC_ERR_STATE;
    }
    rc_slot &sl = ctx->slots[slot];
    if (sl.pb || sl.rb) {
        rc_set_error(ctx, "wait: slot %d holds a packed batch (rc_wait_packed / rc_wait_resident)", slot);
        return RC_ERR_STATE;
    }
    sl.busy = false;
    if (sl.b.n == 0) return RC_OK;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    RC_CHECK_HIP(ctx, hipEventSynchronize(sl.e_done));
    const size_t total = sl.total_reads;
    if (!sl.seq_pinned) {
        memcpy(sl.b.seq, sl.p_seq.p, sl.bytes1);
        if (sl.b.mode == 1) memcpy(sl.b.seq2, (char *)sl.p_seq.p + sl.bytes1, sl.bytes2);
    }
    if (!sl.res_pinned) {
        const int32_t *r = (const int32_t *)sl.p_res.p;
        memcpy(sl.b.ret, r, total * 4);
        memcpy(sl.b.l, r + total, total * 4);
        memcpy(sl.b.m, r + 2 * total, total * 4);
        memcpy(sl.b.h, r + 3 * total, total * 4);
    }
    return RC_OK;
}

// ---- the packed boundary (include/rcorrector_amd.h: rc_packed_batch; device side in rc_transport.hip) ------------
size_t rc_pack_bases(const char *seq, size_t begin, size_t end, uint32_t *bases, uint32_t *exc_pos, uint8_t *exc_chr, size_t exc_cap)
{
    // letter -> code: A0 C1 G2 T3, 4 = NUL, 5 = anything else
    static const struct lut {
        uint8_t v[256];
        lut()
        {
            for (int i = 0; i < 256; ++i) v[i] = 5;
            v[0] = 4;
            v[(int)'A'] = 0;
            v[(int)'C'] = 1;
            v[(int)'G'] = 2;
            v[(int)'T'] = 3;
        }
    } L;
    size_t n_exc = 0;
    const unsigned char *s = reinterpret_cast<const unsigned char *>(seq);
    for (size_t w = begin >> 4; (w << 4) < end; ++w) {
        const size_t p0 = w << 4, lo = p0 < begin ? begin : p0, hi = p0 + 16 > end ? end : p0 + 16;
        uint32_t word = 0;
        for (size_t p = lo; p < hi; ++p) {
            const uint8_t c = L.v[s[p]];
            if (c < 4) {
                word |= (uint32_t)c << (30 - 2 * (p & 15));
            } else if (c == 5) {
                if (n_exc < exc_cap) {
                    exc_pos[n_exc] = (uint32_t)p;
                    exc_chr[n_exc] = s[p];
                }
                ++n_exc;
            }
        }
        // a range that starts inside a word keeps the bits of the positions in front of it (the caller packed them first)
        if (lo > p0) word |= bases[w] & ~(0xFFFFFFFFu >> (2 * (lo - p0)));
        bases[w] = word;
    }
    return n_exc;
}

void rc_apply_fixes(char *seq, const uint32_t *fix_pos, const uint8_t *fix_chr, size_t n_fix)
{
    for (size_t j = 0; j < n_fix; ++j) seq[fix_pos[j]] = (char)fix_chr[j];
}

int rc_submit_packed(rc_ctx *c, rc_packed_batch *b, int slot)
{
    if (!c || !b || slot < 0 || slot >= RC_MAX_SLOTS) return RC_ERR_ARG;
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    if (b->mode < 0 || b->mode > 2 || (b->n && (!b->off || !b->bases || !b->ret || !b->l || !b->m || !b->h)) ||
        (b->n_exc && (!b->exc_pos || !b->exc_chr)) || (b->fix_cap && (!b->fix_pos || !b->fix_chr))) {
        rc_set_error(ctx, "submit_packed: bad batch descriptor");
        return RC_ERR_ARG;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = slots_init(ctx);
    if (rc) return rc;
    rc_slot &sl = ctx->slots[slot];
    if (sl.busy) {
        rc_set_error(ctx, "submit_packed: slot %d still holds a batch (rc_wait_packed it first)", slot);
        return RC_ERR_STATE;
    }
    const size_t total = b->mode == 1 ? 2 * b->n : b->n, nbytes = (size_t)b->nbytes;
    b->n_fix = 0;
    if (total == 0) {
        sl.pb = b;
        sl.rb = nullptr;
        sl.b.n = b->n;
        sl.total_reads = total;
        sl.busy = true;
        return RC_OK;
    }
    if (nbytes >= (1ull << 32) || total >= (1ull << 32) || b->n_exc >= (1ull << 32) || b->fix_cap >= (1ull << 32)) {
        rc_set_error(ctx, "submit_packed: batch too large (split it)");
        return RC_ERR_ARG;
    }
    if (b->mode != 0 && (total & 1)) {
        rc_set_error(ctx, "submit_packed: %s mode needs an even number of reads", b->mode == 1 ? "paired" : "interleaved");
        return RC_ERR_ARG;
    }
    if (b->off[0] != 0 || b->off[total] != nbytes) {
        rc_set_error(ctx, "submit_packed: off[0] = %u, off[%zu] = %u do not describe the arena's %zu bytes", b->off[0], total, b->off[total], nbytes);
        return RC_ERR_ARG;
    }
    if (!ctx->d_buckets) {
        rc_set_error(ctx, "correct: no k-mer table loaded");
        return RC_ERR_STATE;
    }
    // every read ends with its NUL: strictly ascending offsets (the terminator / exception kernels write seq[off[i+1]-1] and
    // seq[exc_pos[i]] unchecked)
    int max_len = 0;
    for (size_t i = 0; i < total; ++i) {
        if (b->off[i + 1] <= b->off[i]) {
            rc_set_error(ctx, "submit_packed: off[%zu] = %u, off[%zu] = %u: offsets must ascend (a read is its bases and a NUL)", i, b->off[i], i + 1, b->off[i + 1]);
            return RC_ERR_ARG;
        }
        max_len = std::max(max_len, (int)(b->off[i + 1] - b->off[i]) - 1);
    }
    for (size_t i = 0; i < b->n_exc; ++i)
        if (b->exc_pos[i] >= nbytes) {
            rc_set_error(ctx, "submit_packed: exc_pos[%zu] = %u lies outside the arena's %zu bytes", i, b->exc_pos[i], nbytes);
            return RC_ERR_ARG;
        }
    sl.pb = b;
    sl.rb = nullptr;
    sl.b.n = b->n;
    sl.total_reads = total;
    const size_t n_words = (nbytes + 15) / 16, qb = (nbytes + 7) / 8, n_exc = b->n_exc;
    const uint32_t cap = (uint32_t)b->fix_cap;
    // device memory: the packed arena, the byte arena it expands into, qualities, offsets, results, exceptions, fixes
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_packed, n_words * 4 + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_seq, n_words * 16 + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_qual, (b->qual_bits ? qb : nbytes) + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_off, (total + 1) * 4))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_res, total * 16))) return rc;
    const size_t exc_chr_off = ((size_t)n_exc * 4 + 15) & ~(size_t)15;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_exc, exc_chr_off + n_exc + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_fix, 64))) return rc;  // (the count; the list itself goes to host memory)
    if ((rc = hbuf_reserve(ctx, &sl.p_nfix, 64))) return rc;
    // inputs that are not page-locked go through one staging block of the slot
    const bool in_pinned = is_pinned(b->off, (total + 1) * 4) && is_pinned(b->bases, n_words * 4) && (!b->qual_bits || is_pinned(b->qual_bits, qb)) &&
                           (!n_exc || (is_pinned(b->exc_pos, n_exc * 4) && is_pinned(b->exc_chr, n_exc)));
    const uint32_t *h_off = b->off, *h_bases = b->bases, *h_exc_pos = b->exc_pos;
    const uint8_t *h_qb = b->qual_bits, *h_exc_chr = b->exc_chr;
    if (!in_pinned) {
        const size_t o_bases = ((total + 1) * 4 + 63) & ~(size_t)63, o_qb = (o_bases + n_words * 4 + 63) & ~(size_t)63,
                     o_ep = (o_qb + qb + 63) & ~(size_t)63, o_ec = o_ep + n_exc * 4;
        if ((rc = hbuf_reserve(ctx, &sl.p_in, o_ec + n_exc + 64))) return rc;
        char *s = (char *)sl.p_in.p;
        memcpy(s, b->off, (total + 1) * 4);
        memcpy(s + o_bases, b->bases, n_words * 4);
        if (b->qual_bits) memcpy(s + o_qb, b->qual_bits, qb);
        if (n_exc) {
            memcpy(s + o_ep, b->exc_pos, n_exc * 4);
            memcpy(s + o_ec, b->exc_chr, n_exc);
        }
        h_off = (const uint32_t *)s;
        h_bases = (const uint32_t *)(s + o_bases);
        h_qb = b->qual_bits ? (const uint8_t *)(s + o_qb) : nullptr;
        h_exc_pos = (const uint32_t *)(s + o_ep);
        h_exc_chr = (const uint8_t *)(s + o_ec);
    }
    sl.res_pinned = is_pinned(b->ret, total * 4) && is_pinned(b->l, total * 4) && is_pinned(b->m, total * 4) && is_pinned(b->h, total * 4);
    sl.fix_pinned = !cap || (is_pinned(b->fix_pos, (size_t)cap * 4) && is_pinned(b->fix_chr, cap));
    sl.fix_room = cap;
    if (!sl.res_pinned && (rc = hbuf_reserve(ctx, &sl.p_res, total * 16))) return rc;
    if (!sl.fix_pinned && (rc = hbuf_reserve(ctx, &sl.p_fix, (size_t)cap * 5 + 64))) return rc;
    struct drain_on_error {
        rc_ctx *c;
        bool armed = true;
        ~drain_on_error()
        {
            if (!armed) return;
            (void)hipStreamSynchronize(c->s_h2d);
            (void)hipStreamSynchronize(c->stream);
            (void)hipStreamSynchronize(c->s_d2h);
        }
    } guard{ctx};
    uint32_t *d_exc_pos = (uint32_t *)sl.d_exc.p;
    uint8_t *d_exc_chr = (uint8_t *)sl.d_exc.p + exc_chr_off;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.d_packed.p, h_bases, n_words * 4, hipMemcpyHostToDevice, ctx->s_h2d));
    if (h_qb) RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.d_qual.p, h_qb, qb, hipMemcpyHostToDevice, ctx->s_h2d));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.d_off.p, h_off, (total + 1) * 4, hipMemcpyHostToDevice, ctx->s_h2d));
    if (n_exc) {
        RC_CHECK_HIP(ctx, hipMemcpyAsync(d_exc_pos, h_exc_pos, n_exc * 4, hipMemcpyHostToDevice, ctx->s_h2d));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(d_exc_chr, h_exc_chr, n_exc, hipMemcpyHostToDevice, ctx->s_h2d));
    }
    RC_CHECK_HIP(ctx, hipEventRecord(sl.e_h2d, ctx->s_h2d));
    // kernels: expand, correct, list the substitutions
    RC_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->stream, sl.e_h2d, 0));
    uint8_t *d_seq = (uint8_t *)sl.d_seq.p;
    if ((rc = rc_launch_unpack(ctx, (const uint32_t *)sl.d_packed.p, nbytes, (const uint32_t *)sl.d_off.p, (uint32_t)total, d_exc_pos, d_exc_chr,
                               (uint32_t)n_exc, d_seq)))
        return rc;
    if (!h_qb) RC_CHECK_HIP(ctx, hipMemsetAsync(sl.d_qual.p, 0, nbytes, ctx->stream));  // FASTA: qual[0] == 0 (Reads.h:224-266)
    int32_t *d_res = (int32_t *)sl.d_res.p;
    rc_device_batch db;
    db.mode = b->mode;
    db.n_reads = (uint32_t)total;
    db.nbytes = nbytes;
    db.max_read_len = max_len;
    db.d_seq = d_seq;
    db.d_qual = (const uint8_t *)sl.d_qual.p;
    db.d_off = (const uint32_t *)sl.d_off.p;
    db.d_ret = d_res;
    db.d_l = d_res + total;
    db.d_m = d_res + 2 * total;
    db.d_h = d_res + 3 * total;
    if ((rc = correct_device_impl(ctx, &db, 0xFFFFFFFFu, 0, h_qb ? 1 : 0))) return rc;
    // The fix list is written by the kernel straight into page-locked host memory (the caller's arrays, or the slot's
    // staging where those are pageable): a few bytes per read, consecutive entries from consecutive lanes.  A copy after
    // the kernels would have to wait for the count first -- a second round trip per batch on a stream of its own, which
    // on this runtime shares a hardware queue with one of the other four and stalls behind it.
    void *dp = nullptr, *dc = nullptr;
    if (cap) {
        RC_CHECK_HIP(ctx, hipHostGetDevicePointer(&dp, sl.fix_pinned ? (void *)b->fix_pos : sl.p_fix.p, 0));
        RC_CHECK_HIP(ctx, hipHostGetDevicePointer(&dc, sl.fix_pinned ? (void *)b->fix_chr : (void *)((char *)sl.p_fix.p + (size_t)cap * 4), 0));
    }
    uint32_t *d_fix_pos = (uint32_t *)dp;
    uint8_t *d_fix_chr = (uint8_t *)dc;
    uint32_t *d_nfix = (uint32_t *)sl.d_fix.p;
    if ((rc = rc_launch_fix_list(ctx, (const uint32_t *)sl.d_packed.p, nbytes, d_seq, d_exc_pos, (uint32_t)n_exc, d_nfix, cap, d_fix_pos, d_fix_chr))) return rc;
    RC_CHECK_HIP(ctx, hipEventRecord(sl.e_k, ctx->stream));
    // results; the fix list follows in rc_wait_packed, once its length is known
    RC_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->s_d2h, sl.e_k, 0));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.p_nfix.p, d_nfix, 4, hipMemcpyDeviceToHost, ctx->s_d2h));
    if (sl.res_pinned) {
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->ret, db.d_ret, total * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->l, db.d_l, total * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->m, db.d_m, total * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->h, db.d_h, total * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
    } else {
        RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.p_res.p, d_res, total * 16, hipMemcpyDeviceToHost, ctx->s_d2h));
    }
    RC_CHECK_HIP(ctx, hipEventRecord(sl.e_done, ctx->s_d2h));
    guard.armed = false;
    sl.busy = true;
    return RC_OK;
}

int rc_wait_packed(rc_ctx *c, int slot)
{
    if (!c || slot < 0 || slot >= RC_MAX_SLOTS) return RC_ERR_ARG;
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    if (!ctx->slots || !ctx->slots[slot].busy || !ctx->slots[slot].pb) {
        rc_set_error(ctx, "wait_packed: slot %d holds no packed batch", slot);
        return RC_ERR_STATE;
    }
    rc_slot &sl = ctx->slots[slot];
    rc_packed_batch *b = sl.pb;
    sl.busy = false;
    sl.pb = nullptr;
    const size_t total = sl.total_reads;
    if (total == 0) return RC_OK;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    RC_CHECK_HIP(ctx, hipEventSynchronize(sl.e_done));  // the results and the fix count have landed; the list was written by the kernel
    const uint32_t n_fix = *(const volatile uint32_t *)sl.p_nfix.p, cap = sl.fix_room;
    if (n_fix > cap) {  // (the kernel stopped writing at cap; the results are complete, the list is not)
        b->n_fix = n_fix;
        rc_set_error(ctx, "wait_packed: %u substitutions, room for %u (fix_cap)", n_fix, cap);
        return RC_ERR_NOSPACE;
    }
    const uint32_t *o_pos = (const uint32_t *)sl.p_fix.p;
    const uint8_t *o_chr = (const uint8_t *)sl.p_fix.p + (size_t)cap * 4;
    if (!sl.res_pinned) {
        const int32_t *r = (const int32_t *)sl.p_res.p;
        memcpy(b->ret, r, total * 4);
        memcpy(b->l, r + total, total * 4);
        memcpy(b->m, r + 2 * total, total * 4);
        memcpy(b->h, r + 3 * total, total * 4);
    }
    if (!sl.fix_pinned && n_fix) {
        memcpy(b->fix_pos, o_pos, (size_t)n_fix * 4);
        memcpy(b->fix_chr, o_chr, n_fix);
    }
    b->n_fix = n_fix;
    return RC_OK;
}

int rc_submit_resident(rc_ctx *c, rc_resident_batch *b, int slot)
{
    if (!c || !b || slot < 0 || slot >= RC_MAX_SLOTS) return RC_ERR_ARG;
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    if (b->mode < 0 || b->mode > 2 || (b->n && (!b->off || !b->ret || !b->l || !b->m || !b->h)) || (b->fix_cap && (!b->fix_pos || !b->fix_chr))) {
        rc_set_error(ctx, "submit_resident: bad batch descriptor");
        return RC_ERR_ARG;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = slots_init(ctx);
    if (rc) return rc;
    rc_slot &sl = ctx->slots[slot];
    if (sl.busy) {
        rc_set_error(ctx, "submit_resident: slot %d still holds a batch (wait for it first)", slot);
        return RC_ERR_STATE;
    }
    const size_t total = b->mode == 1 ? 2 * b->n : b->n;
    const uint64_t bytes_b = b->mode == 1 ? b->bytes_b : 0;
    const size_t nbytes = (size_t)(b->bytes_a + bytes_b);
    b->n_fix = 0;
    if (total == 0) {
        sl.pb = nullptr;
        sl.rb = b;
        sl.b.n = b->n;
        sl.total_reads = total;
        sl.busy = true;
        return RC_OK;
    }
    if (b->bytes_a + bytes_b >= (1ull << 32) || total >= (1ull << 32) || b->fix_cap >= (1ull << 32)) {
        rc_set_error(ctx, "submit_resident: batch too large (split it)");
        return RC_ERR_ARG;
    }
    if (b->mode != 0 && (total & 1)) {
        rc_set_error(ctx, "submit_resident: %s mode needs an even number of reads", b->mode == 1 ? "paired" : "interleaved");
        return RC_ERR_ARG;
    }
    const size_t n_kept = ctx->kept_arenas.size();
    auto in_range = [&](int idx, uint64_t begin, uint64_t bytes) {
        return idx >= 0 && (size_t)idx < n_kept && begin <= ctx->kept_arenas[(size_t)idx].bytes && bytes <= ctx->kept_arenas[(size_t)idx].bytes - begin;
    };
    if (!in_range(b->arena_a, b->begin_a, b->bytes_a) || (b->mode == 1 && !in_range(b->arena_b, b->begin_b, b->bytes_b))) {
        rc_set_error(ctx, "submit_resident: no such range of a kept arena (%zu kept; rc_table_count_keep before counting)", n_kept);
        return RC_ERR_ARG;
    }
    if (b->off[0] != 0 || b->off[total] != nbytes || (b->mode == 1 && b->off[b->n] != b->bytes_a)) {
        rc_set_error(ctx, "submit_resident: the offsets do not describe the ranges (off[0] = %u, off[%zu] = %u, %zu bytes)", b->off[0], total, b->off[total], nbytes);
        return RC_ERR_ARG;
    }
    if (!ctx->d_buckets) {
        rc_set_error(ctx, "correct: no k-mer table loaded");
        return RC_ERR_STATE;
    }
    int max_len = 0;
    for (size_t i = 0; i < total; ++i) {
        if (b->off[i + 1] <= b->off[i]) {
            rc_set_error(ctx, "submit_resident: off[%zu] = %u, off[%zu] = %u: offsets must ascend (a read is its bases and a NUL)", i, b->off[i], i + 1, b->off[i + 1]);
            return RC_ERR_ARG;
        }
        max_len = std::max(max_len, (int)(b->off[i + 1] - b->off[i]) - 1);
    }
    sl.pb = nullptr;
    sl.rb = b;
    sl.b.n = b->n;
    sl.total_reads = total;
    const size_t qb = (nbytes + 7) / 8;
    const uint32_t cap = (uint32_t)b->fix_cap;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_seq, ((nbytes + 15) & ~(size_t)15) + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_qual, (b->qual_bits ? qb : nbytes) + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_off, (total + 1) * 4))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_res, total * 16))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_fix, 64))) return rc;
    if ((rc = hbuf_reserve(ctx, &sl.p_nfix, 64))) return rc;
    const bool in_pinned = is_pinned(b->off, (total + 1) * 4) && (!b->qual_bits || is_pinned(b->qual_bits, qb));
    const uint32_t *h_off = b->off;
    const uint8_t *h_qb = b->qual_bits;
    if (!in_pinned) {
        const size_t o_qb = ((total + 1) * 4 + 63) & ~(size_t)63;
        if ((rc = hbuf_reserve(ctx, &sl.p_in, o_qb + qb + 64))) return rc;
        char *s = (char *)sl.p_in.p;
        memcpy(s, b->off, (total + 1) * 4);
        if (b->qual_bits) memcpy(s + o_qb, b->qual_bits, qb);
        h_off = (const uint32_t *)s;
        h_qb = b->qual_bits ? (const uint8_t *)(s + o_qb) : nullptr;
    }
    sl.res_pinned = is_pinned(b->ret, total * 4) && is_pinned(b->l, total * 4) && is_pinned(b->m, total * 4) && is_pinned(b->h, total * 4);
    sl.fix_pinned = !cap || (is_pinned(b->fix_pos, (size_t)cap * 4) && is_pinned(b->fix_chr, cap));
    sl.fix_room = cap;
    if (!sl.res_pinned && (rc = hbuf_reserve(ctx, &sl.p_res, total * 16))) return rc;
    if (!sl.fix_pinned && (rc = hbuf_reserve(ctx, &sl.p_fix, (size_t)cap * 5 + 64))) return rc;
    struct drain_on_error {
        rc_ctx *c;
        bool armed = true;
        ~drain_on_error()
        {
            if (!armed) return;
            (void)hipStreamSynchronize(c->s_h2d);
            (void)hipStreamSynchronize(c->stream);
            (void)hipStreamSynchronize(c->s_d2h);
        }
    } guard{ctx};
    if (h_qb) RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.d_qual.p, h_qb, qb, hipMemcpyHostToDevice, ctx->s_h2d));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.d_off.p, h_off, (total + 1) * 4, hipMemcpyHostToDevice, ctx->s_h2d));
    RC_CHECK_HIP(ctx, hipEventRecord(sl.e_h2d, ctx->s_h2d));
    RC_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->stream, sl.e_h2d, 0));
    // the batch's own arena: its ranges of the kept arenas, side by side
    uint8_t *d_seq = (uint8_t *)sl.d_seq.p;
    const uint8_t *orig_a = (const uint8_t *)ctx->kept_arenas[(size_t)b->arena_a].p + b->begin_a;
    const uint8_t *orig_b = bytes_b ? (const uint8_t *)ctx->kept_arenas[(size_t)b->arena_b].p + b->begin_b : nullptr;
    if (b->bytes_a) RC_CHECK_HIP(ctx, hipMemcpyAsync(d_seq, orig_a, b->bytes_a, hipMemcpyDeviceToDevice, ctx->stream));
    if (bytes_b) RC_CHECK_HIP(ctx, hipMemcpyAsync(d_seq + b->bytes_a, orig_b, bytes_b, hipMemcpyDeviceToDevice, ctx->stream));
    if (!h_qb) RC_CHECK_HIP(ctx, hipMemsetAsync(sl.d_qual.p, 0, nbytes, ctx->stream));  // FASTA: qual[0] == 0 (Reads.h:224-266)
    int32_t *d_res = (int32_t *)sl.d_res.p;
    rc_device_batch db;
    db.mode = b->mode;
    db.n_reads = (uint32_t)total;
    db.nbytes = nbytes;
    db.max_read_len = max_len;
    db.d_seq = d_seq;
    db.d_qual = (const uint8_t *)sl.d_qual.p;
    db.d_off = (const uint32_t *)sl.d_off.p;
    db.d_ret = d_res;
    db.d_l = d_res + total;
    db.d_m = d_res + 2 * total;
    db.d_h = d_res + 3 * total;
    if ((rc = correct_device_impl(ctx, &db, 0xFFFFFFFFu, 0, h_qb ? 1 : 0))) return rc;
    void *dp = nullptr, *dc = nullptr;  // (the fix list goes straight into page-locked host memory, as in rc_submit_packed)
    if (cap) {
        RC_CHECK_HIP(ctx, hipHostGetDevicePointer(&dp, sl.fix_pinned ? (void *)b->fix_pos : sl.p_fix.p, 0));
        RC_CHECK_HIP(ctx, hipHostGetDevicePointer(&dc, sl.fix_pinned ? (void *)b->fix_chr : (void *)((char *)sl.p_fix.p + (size_t)cap * 4), 0));
    }
    uint32_t *d_nfix = (uint32_t *)sl.d_fix.p;
    if ((rc = rc_launch_fix_list_bytes(ctx, orig_a, (size_t)b->bytes_a, orig_b, (size_t)bytes_b, d_seq, d_nfix, cap, (uint32_t *)dp, (uint8_t *)dc))) return rc;
    RC_CHECK_HIP(ctx, hipEventRecord(sl.e_k, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->s_d2h, sl.e_k, 0));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.p_nfix.p, d_nfix, 4, hipMemcpyDeviceToHost, ctx->s_d2h));
    if (sl.res_pinned) {
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->ret, db.d_ret, total * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->l, db.d_l, total * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->m, db.d_m, total * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->h, db.d_h, total * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
    } else {
        RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.p_res.p, d_res, total * 16, hipMemcpyDeviceToHost, ctx->s_d2h));
    }
    RC_CHECK_HIP(ctx, hipEventRecord(sl.e_done, ctx->s_d2h));
    guard.armed = false;
    sl.busy = true;
    return RC_OK;
}

int rc_wait_resident(rc_ctx *c, int slot)
{
    if (!c || slot < 0 || slot >= RC_MAX_SLOTS) return RC_ERR_ARG;
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    if (!ctx->slots || !ctx->slots[slot].busy || !ctx->slots[slot].rb) {
        rc_set_error(ctx, "wait_resident: slot %d holds no resident batch", slot);
        return RC_ERR_STATE;
    }
    rc_slot &sl = ctx->slots[slot];
    rc_resident_batch *b = sl.rb;
    sl.busy = false;
    sl.rb = nullptr;
    const size_t total = sl.total_reads;
    if (total == 0) return RC_OK;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    RC_CHECK_HIP(ctx, hipEventSynchronize(sl.e_done));
    const uint32_t n_fix = *(const volatile uint32_t *)sl.p_nfix.p, cap = sl.fix_room;
    if (n_fix > cap) {  // (the kernel stopped writing at cap; the results are complete, the list is not)
        b->n_fix = n_fix;
        rc_set_error(ctx, "wait_resident: %u substitutions, room for %u (fix_cap)", n_fix, cap);
        return RC_ERR_NOSPACE;
    }
    if (!sl.res_pinned) {
        const int32_t *r = (const int32_t *)sl.p_res.p;
        memcpy(b->ret, r, total * 4);
        memcpy(b->l, r + total, total * 4);
        memcpy(b->m, r + 2 * total, total * 4);
        memcpy(b->h, r + 3 * total, total * 4);
    }
    if (!sl.fix_pinned && n_fix) {
        memcpy(b->fix_pos, sl.p_fix.p, (size_t)n_fix * 4);
        memcpy(b->fix_chr, (const uint8_t *)sl.p_fix.p + (size_t)cap * 4, n_fix);
    }
    b->n_fix = n_fix;
    return RC_OK;
}

// ---- measurement -----------------------------------------------------------------------------
int rc_profile_enable(rc_ctx *ctx, int on)
{
    if (!ctx) return RC_ERR_ARG;
    ctx->profile = on != 0;
    if (!ctx->phase_prof_print) ctx->phase_prof = on == 2;
    return RC_OK;
}

int rc_profile_get(rc_ctx *ctx, int kernel, double *total_ms, uint64_t *launches)
{
    if (!ctx || kernel < 0 || kernel >= RC_T_COUNT) return RC_ERR_ARG;
    if (total_ms) *total_ms = ctx->timers[kernel].ms;
    if (launches) *launches = ctx->timers[kernel].launches;
    return RC_OK;
}

int rc_profile_reset(rc_ctx *ctx)
{
    if (!ctx) return RC_ERR_ARG;
    for (auto &t : ctx->timers) t = rc_kernel_timer();
    ctx->k3_listed = ctx->k3_rounds = ctx->k3_requests = 0;
    return RC_OK;
}

int rc_profile_correct_counters(rc_ctx *ctx, uint64_t *reads_listed, uint64_t *gather_rounds, uint64_t *bucket_requests)
{
    if (!ctx) return RC_ERR_ARG;
    if (reads_listed) *reads_listed = ctx->k3_listed;
    if (gather_rounds) *gather_rounds = ctx->k3_rounds;
    if (bucket_requests) *bucket_requests = ctx->k3_requests;
    return RC_OK;
}

int rc_profile_read_rounds(rc_ctx *ctx, int32_t *d_rounds)
{
    if (!ctx) return RC_ERR_ARG;
    ctx->rounds_out = d_rounds;
    return RC_OK;
}

int rc_selftest_get_bound(rc_ctx *ctx, const int32_t *c, size_t n, double error_rate, int32_t *out_int, double *out_dbl)
{
    if (!ctx || (n && (!c || !out_int || !out_dbl))) return RC_ERR_ARG;
    if (n == 0) return RC_OK;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    rc_dev_tmp b_c, b_i, b_d;
    RC_CHECK_HIP(ctx, b_c.alloc(n * 4));
    RC_CHECK_HIP(ctx, b_i.alloc(n * 4));
    RC_CHECK_HIP(ctx, b_d.alloc(n * 8));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(b_c.p, c, n * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = rc_launch_selftest_bound(ctx, b_c.as<int32_t>(), n, error_rate, b_i.as<int32_t>(), b_d.as<double>());
    if (rc) return rc;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(out_int, b_i.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(out_dbl, b_d.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

int rc_summary(const rc_ctx *c, uint64_t *total_reads, uint64_t *total_corrections)
{
    if (!c) return RC_ERR_ARG;
    rc_ctx *ctx = const_cast<rc_ctx *>(c);  // (reads device memory; the counters themselves do not change)
    unsigned long long v[2] = {0, 0};
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RC_CHECK_HIP(ctx, hipMemcpy(v, (char *)ctx->work.p + RC_WORK_SUMMARY_OFF, sizeof v, hipMemcpyDeviceToHost));
    if (total_reads) *total_reads = v[0];
    if (total_corrections) *total_corrections = v[1];
    return RC_OK;
}

}  // extern "C"
