// rc_api_table.hip -- C ABI, the k-mer table and the run parameters (include/rcorrector_amd.h): dump load, GPU counting,
// sharing / replication, lookup / export / digest, ERROR_RATE, bad quality.  Host code only drives HIP.
#include "rc_api_internal.h"

extern "C" {

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
    dst->filter_all = src->filter_all;
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
    dst->filter_all = src->filter_all;
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
    for (rc_ctx *ln : ctx->lane) {  // the slot lanes hold copies of the descriptors and run on streams of their own
        if (!ln) continue;
        RC_CHECK_HIP(ctx, hipStreamSynchronize(ln->stream));
        ln->kept_arenas.clear();
    }
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

int rc_table_count_finish_sharded(rc_ctx **ctxs, int n, int min_count, int64_t *n_kmers)
{
    if (!ctxs || n < 1 || !ctxs[0]) return RC_ERR_ARG;
    for (int g = 0; g < n; ++g)
        if (!ctxs[g]) return RC_ERR_ARG;
    if (n == 1) return rc_table_count_finish(ctxs[0], min_count, n_kmers);
    RC_CHECK_HIP(ctxs[0], hipSetDevice(ctxs[0]->device));
    static_cast<rc_ctx_full *>(ctxs[0])->dump.valid = false;
    const int rc = rc_count_finish_sharded(ctxs, n, min_count, n_kmers);
    (void)hipSetDevice(ctxs[0]->device);
    return rc;
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
        int rc = RC_OK;
        if (ctx->counted_codes && ctx->counted_n == (size_t)ctx->n_entries) {  // the table was counted here: its codes are still there
            d_codes = (uint64_t *)ctx->counted_codes;
            n = ctx->counted_n;
            ctx->counted_codes = nullptr;
            ctx->counted_n = 0;
        } else {
            rc = rc_table_codes_device(ctx, &d_codes, &n);
        }
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
    // the codes a counted table was built from wait for rc_estimate_error_rate (which takes them); a caller that sets the
    // parameters itself does not need them: 8 bytes per entry of HBM back (rcorrector_amd.h: rc_table_count_finish)
    if (ctx->counted_codes) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(ctx->counted_codes);
        ctx->counted_codes = nullptr;
        ctx->counted_n = 0;
    }
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
    int rc = rc_dbuf_reserve(ctx, &ctx->bs_dev, steps_bytes + RC_BOUND_SMALL);
    if (rc) return rc;
    // ... and behind it the bound itself for small counts (rc_run_params::bound_small), again the host's arithmetic
    uint8_t small[RC_BOUND_SMALL];
    for (int c = 0; c < RC_BOUND_SMALL; ++c) {
        const int v = rc_bound_i(c, error_rate);
        small[c] = v >= 0 && v < 255 ? (uint8_t)v : (uint8_t)255;
    }
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (a batch in flight may still read the old table)
    RC_CHECK_HIP(ctx, hipMemcpy(ctx->bs_dev.p, steps, steps_bytes, hipMemcpyHostToDevice));
    RC_CHECK_HIP(ctx, hipMemcpy((char *)ctx->bs_dev.p + steps_bytes, small, RC_BOUND_SMALL, hipMemcpyHostToDevice));
    ctx->P.bs_ext = getenv("RC_NO_BS_EXT") ? nullptr : (const uint32_t *)ctx->bs_dev.p;
    ctx->P.bound_small = getenv("RC_NO_BS_EXT") ? nullptr : (const uint8_t *)ctx->bs_dev.p + steps_bytes;
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

}  // extern "C"
