// rc_table.hip -- the k-mer count table in HBM (Store.h:17-88 re-designed for the MI355X memory
// system) and the kernels that touch only the table: build (K0), lookup, the per-batch probe
// kernel (K1: reads -> counts[], ErrorCorrection.cpp:716-723), k-mer counting from reads
// (stages 0-2 of run_rcorrector.pl:262-281) and the last-base-variant pass of the ERROR_RATE
// estimation (main.cpp:329-345).
//
// Layout: open addressing over 64-byte buckets (one HBM/L2 sector per probe): 5 slots of
// {key_lo, key_hi, count} + one meta dword.  A key lives in its home bucket hash(key) & mask or,
// if that is full, in the following buckets (no wrap-around: the array carries slack buckets).
// meta bit0 of bucket b says "some key with home <= b was placed after b", so a miss stops at
// the first bucket without that bit -- >96 % of all probes, hit or miss, touch exactly one
// bucket at the load factors used (<= 0.5).
//
// The build is a sort, not a race: entries are stably radix-sorted by home bucket (input order
// reversed so that of two equal keys the LATER one is met first by a probe, which is the
// "later Put overwrites" rule of Store.h:55), slot numbers come from one prefix-max scan
// p_j = max(5*home_j, p_{j-1}+1), and a scatter writes the buckets.  The layout is therefore a
// pure function of the input, independent of thread timing.
#include <cstring>
#include <vector>

#include <algorithm>
#include <vector>
#include <rocprim/rocprim.hpp>

#include <chrono>
#include "rc_internal.h"
#include "rc_device.h"

// ---- K0: build -----------------------------------------------------------------------------
__global__ void k_home_and_index(const uint64_t *__restrict__ canon, uint32_t *__restrict__ home,
                                 uint32_t *__restrict__ idx, size_t n, uint32_t nb_home, int layout, int k, int ext)
{
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    size_t i = n - 1 - j;  // reversed input order (see header)
    uint32_t h, rem, xrem;
    if (layout)
        rc_packed_addr(canon[i], k, nb_home, ext, &h, &rem, &xrem);
    else
        h = rc_home(canon[i], nb_home);
    home[j] = h;
    idx[j] = (uint32_t)i;
}

__global__ void k_slot_seed(const uint32_t *__restrict__ home_sorted, long long *__restrict__ q, size_t n, int slots)
{
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    q[j] = (long long)slots * home_sorted[j] - (long long)j;
}

// what rules the PACKED layout out for a given input: a key pushed more than 14 buckets past its home
// (flags[0] |= 2), or more counts that do not fit the count field than the table's prefix holds
// (flags[1] = how many there are: rc_common.h, RC_PACKED_OVF_MAX).
__global__ void k_packed_feasible(const uint32_t *__restrict__ home_sorted, const long long *__restrict__ qmax,
                                  const int32_t *__restrict__ counts, size_t n, int ext, unsigned *__restrict__ flags)
{
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long long p = qmax[j] + (long long)j;
    if ((uint32_t)counts[j] >= (RC_PACKED_COUNT_MASK >> ext)) atomicAdd(flags + 1, 1u);  // (counts are indexed by input, any order does)
    if ((uint32_t)(p / RC_PACKED_SLOTS) - home_sorted[j] > RC_PACKED_MAX_DISP) atomicOr(flags, 2u);
}

// the entries whose count does not fit: {code, count, input index} in any order (the host sorts the few)
__global__ void k_collect_overflow(const uint64_t *__restrict__ canon, const int32_t *__restrict__ counts, size_t n, uint32_t cmask,
                                   unsigned *__restrict__ n_out, uint4 *__restrict__ out, uint32_t *__restrict__ out_idx)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (uint32_t)counts[i] < cmask) return;
    const unsigned o = atomicAdd(n_out, 1u);
    if (o >= RC_PACKED_OVF_MAX) return;
    out[o] = make_uint4((uint32_t)canon[i], (uint32_t)(canon[i] >> 32), (uint32_t)counts[i], 0u);
    out_idx[o] = (uint32_t)i;
}

__global__ void k_scatter(const uint32_t *__restrict__ home_sorted, const uint32_t *__restrict__ idx_sorted,
                          const long long *__restrict__ qmax, const uint64_t *__restrict__ canon,
                          const int32_t *__restrict__ counts, uint32_t *__restrict__ buckets, size_t n, uint32_t nb_home, int layout, int k, int ext)
{
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long long p = qmax[j] + (long long)j;
    const int S = rc_layout_slots(layout);
    const uint32_t b = (uint32_t)(p / S), s = (uint32_t)(p % S);
    const uint32_t i = idx_sorted[j];
    const uint64_t key = canon[i];
    if (layout) {
        uint32_t h, rem, xrem;
        rc_packed_addr(key, k, nb_home, ext, &h, &rem, &xrem);
        uint32_t *w = buckets + (size_t)b * RC_BUCKET_DWORDS + s * 2;
        w[0] = rem;
        // the last slot's bit 31 is the bucket's continue flag, set (atomically: another thread may own
        // that slot) by whoever lands in slot 0 of the next bucket with an earlier home
        atomicAnd(w + 1, 0x80000000u);  // (the slot starts as RC_PACKED_EMPTY_WORD; the flag bit may already be set)
        const uint32_t cmask = RC_PACKED_COUNT_MASK >> ext, cnt = (uint32_t)counts[i] < cmask ? (uint32_t)counts[i] : cmask;  // all ones: see the prefix
        atomicOr(w + 1, cnt | (xrem << (27 - ext)) | ((b - h) << 27));
        if (s == 0 && h < b) atomicOr(buckets + (size_t)(b - 1) * RC_BUCKET_DWORDS + (RC_BUCKET_DWORDS - 1), 0x80000000u);
    } else {
        uint32_t *w = buckets + (size_t)b * RC_BUCKET_DWORDS + s * 3;
        w[0] = (uint32_t)key;
        w[1] = (uint32_t)(key >> 32);
        w[2] = (uint32_t)counts[i];
        if (s == 0 && home_sorted[j] < b) buckets[(size_t)(b - 1) * RC_BUCKET_DWORDS + (RC_BUCKET_DWORDS - 1)] = 1u;
    }
}

// the absence filter of a PACKED table (rc_common.h: rc_table_view::filter): every entry sets its three bits -- kind 1: under
// both of its orientations
__global__ void k_filter_set(const uint64_t *__restrict__ canon, size_t n, uint32_t nb_home, int k, int ext, uint32_t *__restrict__ filter, uint32_t words,
                             int kind)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (kind) {
        const uint64_t x = canon[i], rv = rc_revcomp(x, k);
        uint32_t w, m;
        rc_filter_core_addr(x, words, &w, &m);
        atomicOr(filter + w, m);
        if (rv != x) {
            rc_filter_core_addr(rv, words, &w, &m);
            atomicOr(filter + w, m);
        }
        return;
    }
    uint32_t h, rem, xrem, top;
    rc_packed_addr(canon[i], k, nb_home, ext, &h, &rem, &xrem, &top);
    atomicOr(filter + rc_mulhi32(top, words), rc_filter_mask(rem));
}

// one attempt at one layout; *ok = false (PACKED only) if a count or a displacement does not fit
static int build_attempt(rc_ctx *ctx, const uint64_t *d_canon, const int32_t *d_counts, size_t n, int layout, int ext, uint32_t nb_home, bool *ok)
{
    *ok = true;
    const int S = rc_layout_slots(layout);
    int bits = 0;
    while ((1ull << bits) < (uint64_t)nb_home) ++bits;
    long long p_last = -1;
    unsigned n_overflow = 0;  // PACKED: counts that go to the table's prefix
    rc_dev_tmp b_home, b_home_s, b_idx, b_idx_s, b_q, b_qm, b_tmp, b_flags;
    const unsigned B = 256;
    const unsigned G = (unsigned)((n + B - 1) / B);
    if (n > 0) {
        size_t tmp_sort = 0, tmp_scan = 0;
        RC_CHECK_HIP(ctx, b_home.alloc(n * 4));
        RC_CHECK_HIP(ctx, b_home_s.alloc(n * 4));
        RC_CHECK_HIP(ctx, b_idx.alloc(n * 4));
        RC_CHECK_HIP(ctx, b_idx_s.alloc(n * 4));
        RC_CHECK_HIP(ctx, b_q.alloc(n * 8));
        RC_CHECK_HIP(ctx, b_qm.alloc(n * 8));
        RC_CHECK_HIP(ctx, b_flags.alloc(8));
        uint32_t *home = b_home.as<uint32_t>(), *home_s = b_home_s.as<uint32_t>();
        uint32_t *idx = b_idx.as<uint32_t>(), *idx_s = b_idx_s.as<uint32_t>();
        long long *q = b_q.as<long long>(), *qm = b_qm.as<long long>();
        hipLaunchKernelGGL(k_home_and_index, dim3(G), dim3(B), 0, ctx->stream, d_canon, home, idx, n, nb_home, layout, ctx->k, ext);
        RC_CHECK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_sort, home, home_s, idx, idx_s, n, 0, bits > 0 ? bits : 1, ctx->stream));
        RC_CHECK_HIP(ctx, rocprim::inclusive_scan(nullptr, tmp_scan, q, qm, n, rocprim::maximum<long long>(), ctx->stream));
        RC_CHECK_HIP(ctx, b_tmp.alloc(tmp_sort > tmp_scan ? tmp_sort : tmp_scan));
        RC_CHECK_HIP(ctx, rocprim::radix_sort_pairs(b_tmp.p, tmp_sort, home, home_s, idx, idx_s, n, 0, bits > 0 ? bits : 1, ctx->stream));
        hipLaunchKernelGGL(k_slot_seed, dim3(G), dim3(B), 0, ctx->stream, home_s, q, n, S);
        RC_CHECK_HIP(ctx, rocprim::inclusive_scan(b_tmp.p, tmp_scan, q, qm, n, rocprim::maximum<long long>(), ctx->stream));
        long long q_last = 0;
        unsigned flags[2] = {0, 0};
        if (layout) {
            RC_CHECK_HIP(ctx, hipMemsetAsync(b_flags.p, 0, 8, ctx->stream));
            hipLaunchKernelGGL(k_packed_feasible, dim3(G), dim3(B), 0, ctx->stream, home_s, qm, d_counts, n, ext, b_flags.as<unsigned>());
            RC_CHECK_HIP(ctx, hipMemcpyAsync(flags, b_flags.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        }
        RC_CHECK_HIP(ctx, hipMemcpyAsync(&q_last, qm + (n - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
        RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (flags[0] || flags[1] > RC_PACKED_OVF_MAX) {
            *ok = false;
            return RC_OK;
        }
        n_overflow = flags[1];
        p_last = q_last + (long long)(n - 1);
    }
    uint64_t need = (uint64_t)(p_last / S) + 2;  // +1 empty bucket after the last used
    uint64_t nb_alloc = need > (uint64_t)nb_home + 1 ? need : (uint64_t)nb_home + 1;
    if (nb_alloc >= (1ull << 32)) {
        rc_set_error(ctx, "table build: bucket count overflow");
        return RC_ERR_ARG;
    }
    ctx->nb_home = nb_home;
    ctx->layout = layout;
    ctx->ext = layout ? ext : 0;
    ctx->nb_alloc = (uint32_t)nb_alloc;
    ctx->table_bytes = (size_t)nb_alloc * RC_BUCKET_BYTES;
    // A bucket array beyond the reach of the TLB (2.7-3.2 GB on the boxes seen) makes every probe a page walk, and most
    // probes of the search -- 61 % at 0.5 % errors, 72 % at 5 % -- are of k-mers that are not in the table at all: a
    // filter of 10 bits per entry behind the buckets (a tenth of their size, inside the TLB's reach up to ~2.5 G
    // entries) answers those without touching the buckets.  PACKED tables beyond 2.5 GiB only: below, a bucket read
    // costs what a filter word costs.  RC_TABLE_FILTER=force / off for tests and A/B runs.
    // RC_TABLE_FILTER_KIND=plain|core: how a k-mer finds its word (rc_common.h: rc_table_view::filter_kind); RC_TABLE_FILTER=search:
    // the filter of a mid-size table (the search's lookups only) on a table of any size (tests).
    ctx->filter_words = 0;
    {
        const char *e = getenv("RC_TABLE_FILTER"), *kd = getenv("RC_TABLE_FILTER_KIND");
        const bool force = e && !strcmp(e, "force"), off = e && !strcmp(e, "off"), search = e && !strcmp(e, "search");
        ctx->filter_kind = kd ? (strcmp(kd, "plain") != 0 ? 1 : 0) : RC_FILTER_KIND_DEFAULT;
        // beyond 2.5 GiB every lookup asks the filter.  RC_TABLE_FILTER=search (dev): a core filter that only the search's
        // lookups ask, on a table of any size -- measured on the tables inside the TLB's reach and not taken: k_correct is
        // bound by instruction issue there, and the filter's hash costs it more than the bucket reads it saves (config 2:
        // 20.1 / 20.3 ms against 19.6 / 19.5; config 3: 53.1 / 51.8 against 51.3 / 49.4; profiles/r5_search_filter_ab.txt)
        const bool big = ctx->table_bytes > ((size_t)5 << 29);
        const bool mid = ctx->filter_kind == 1 && search;
        ctx->filter_all = (force || big) ? 1 : 0;
        if (layout && n > 0 && !off && (force || big || mid)) {
            uint64_t bits = ctx->filter_kind ? 16 : 10;  // per entry
            if (const char *fb = getenv("RC_TABLE_FILTER_BITS")) bits = (uint64_t)atoi(fb) >= 4 ? (uint64_t)atoi(fb) : bits;  // dev
            uint64_t w = ((uint64_t)n * bits + 31) / 32 + 64;
            if (w < (1ull << 32)) ctx->filter_words = (uint32_t)w;
        }
    }
    const size_t filter_bytes = (size_t)ctx->filter_words * 4;
    {   // the bucket array, RC_TABLE_PREFIX_BYTES (zero unless counts overflow, below) in front of it, the filter behind it
        char *base = nullptr;
        RC_CHECK_HIP(ctx, hipMalloc((void **)&base, ctx->table_bytes + RC_TABLE_PREFIX_BYTES + filter_bytes));  // (hipDeviceMallocContiguous: no effect on the TLB cliff, measured)
        ctx->d_buckets = reinterpret_cast<uint32_t *>(base + RC_TABLE_PREFIX_BYTES);
        RC_CHECK_HIP(ctx, hipMemsetAsync(base, 0, RC_TABLE_PREFIX_BYTES, ctx->stream));
        if (filter_bytes) RC_CHECK_HIP(ctx, hipMemsetAsync(base + RC_TABLE_PREFIX_BYTES + ctx->table_bytes, 0, filter_bytes, ctx->stream));
    }
    if (layout && n_overflow) {
        rc_dev_tmp b_o, b_oi, b_on;
        RC_CHECK_HIP(ctx, b_o.alloc(RC_PACKED_OVF_MAX * sizeof(uint4)));
        RC_CHECK_HIP(ctx, b_oi.alloc(RC_PACKED_OVF_MAX * 4));
        RC_CHECK_HIP(ctx, b_on.alloc(4));
        RC_CHECK_HIP(ctx, hipMemsetAsync(b_on.p, 0, 4, ctx->stream));
        hipLaunchKernelGGL(k_collect_overflow, dim3(G), dim3(B), 0, ctx->stream, d_canon, d_counts, n, RC_PACKED_COUNT_MASK >> ext, b_on.as<unsigned>(),
                           b_o.as<uint4>(), b_oi.as<uint32_t>());
        std::vector<uint4> e(n_overflow);
        std::vector<uint32_t> ei(n_overflow);
        RC_CHECK_HIP(ctx, hipMemcpyAsync(e.data(), b_o.p, n_overflow * sizeof(uint4), hipMemcpyDeviceToHost, ctx->stream));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(ei.data(), b_oi.p, n_overflow * 4, hipMemcpyDeviceToHost, ctx->stream));
        RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        // ascending by code; of two Puts of one code the later one is the table's entry (Store.h:55)
        std::vector<uint32_t> o(n_overflow);
        for (uint32_t i = 0; i < n_overflow; ++i) o[i] = i;
        auto code = [&](uint32_t i) { return ((uint64_t)e[i].y << 32) | e[i].x; };
        std::sort(o.begin(), o.end(), [&](uint32_t a, uint32_t b) { return code(a) != code(b) ? code(a) < code(b) : ei[a] < ei[b]; });
        std::vector<uint4> sorted;
        for (uint32_t j = 0; j < n_overflow; ++j)
            if (j + 1 == n_overflow || code(o[j + 1]) != code(o[j])) sorted.push_back(e[o[j]]);
        const uint32_t ns = (uint32_t)sorted.size();
        char *base = reinterpret_cast<char *>(ctx->d_buckets) - RC_TABLE_PREFIX_BYTES;
        RC_CHECK_HIP(ctx, hipMemcpyAsync(base, sorted.data(), ns * sizeof(uint4), hipMemcpyHostToDevice, ctx->stream));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(reinterpret_cast<char *>(ctx->d_buckets) - 64, &ns, 4, hipMemcpyHostToDevice, ctx->stream));
        RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (the vectors go out of scope)
    }
    if (layout)
        RC_CHECK_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->d_buckets, (int)RC_PACKED_EMPTY_WORD, ctx->table_bytes / 4, ctx->stream));
    else
        RC_CHECK_HIP(ctx, hipMemsetAsync(ctx->d_buckets, 0, ctx->table_bytes, ctx->stream));
    if (n > 0) {
        hipLaunchKernelGGL(k_scatter, dim3(G), dim3(B), 0, ctx->stream, b_home_s.as<uint32_t>(), b_idx_s.as<uint32_t>(),
                           b_qm.as<long long>(), d_canon, d_counts, ctx->d_buckets, n, nb_home, layout, ctx->k, ext);
        if (ctx->filter_words)
            hipLaunchKernelGGL(k_filter_set, dim3(G), dim3(B), 0, ctx->stream, d_canon, n, nb_home, ctx->k, ext,
                               ctx->d_buckets + ctx->table_bytes / 4, ctx->filter_words, ctx->filter_kind);
        RC_CHECK_HIP(ctx, hipGetLastError());
    }
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_entries = n;
    return RC_OK;
}

int rc_build_table_from_device_pairs(rc_ctx *ctx, const uint64_t *d_canon, const int32_t *d_counts, size_t n)
{
    if (n >= (1ull << 31)) {
        rc_set_error(ctx, "table build: %zu entries exceed the 2^31 limit", n);
        return RC_ERR_ARG;
    }
    rc_table_release(ctx);
    // home buckets: n / (slots * load).  Random 64-byte gathers on MI355X are request-rate bound
    // (~55 G/s, tools/microbench_gather.hip) and fall off a cliff once the table outgrows the TLB
    // reach (~2 GiB), so a dense table wins: fewer bytes => more MALL/L2 hits per probe.
    auto buckets_for = [&](int slots, double load, double dflt) -> uint64_t {
        if (!(load > 0.05 && load <= 0.95)) load = dflt;
        uint64_t want = (uint64_t)((double)n / (slots * load)) + 1;
        return want < 64 ? 64 : want;
    };
    // PACKED needs nb_home * 2^ext >= 2^(2k-32) for (home, rem, xrem) to identify a code: ext = the
    // smallest such number of extra remainder bits (0 for k <= 28 at ordinary table sizes); it is
    // used when at most RC_PACKED_MAX_EXT are needed and the placement allows (k_packed_feasible), WIDE otherwise
    // load by size: the emptier the table, the fewer probes run on into a second bucket (configs 1-3 at
    // 0.4 / 0.45 / 0.5: 282 / 281 / 275, 201 / 201 / 197 and 212 / 210 / 206 M reads/s; 0.6 / 0.7 / 0.8:
    // config 2 at 187 / 174 / 145) -- until the table leaves the reach of the TLB, 2.7-3.2 GB depending on
    // the box: 201 M entries at 0.5 / 0.6 / 0.65 / 0.7 = 3.2 / 2.7 / 2.5 / 2.3 GB: k_correct 1537 / 1189 / 1227 /
    // 1280 ms.  So: 0.4 up to 2 GiB, 0.5 up to 2.5 GiB, 0.6 beyond.
    double packed_load = ctx->table_load_packed;
    if (!ctx->table_load_set) {
        const double gib = (double)n * 8.0 / 1073741824.0;  // slot bytes of the entries
        packed_load = gib / 0.40 <= 2.0 ? 0.40 : (gib / 0.50 <= 2.5 ? 0.50 : 0.60);
    }
    const uint64_t packed = buckets_for(RC_PACKED_SLOTS, packed_load, 0.50);
    const int kb = 2 * ctx->k;
    int ext = 0;
    while (kb > 32 && kb - 32 - ext > 0 && (packed << ext) < (1ull << (kb - 32))) ++ext;
    if (ctx->layout_pref != 0 && ext <= RC_PACKED_MAX_EXT && packed < (1ull << 32) - 8) {
        bool ok = false;
        int rc = build_attempt(ctx, d_canon, d_counts, n, 1, ext, (uint32_t)packed, &ok);
        if (rc) return rc;
        if (ok) return RC_OK;  // else: too many counts beyond the count field or a chain longer than 15 buckets -- WIDE takes anything
    }
    // WIDE: past the reach of the TLB (a table of 4.9 GB: two L1-TLB misses in three requests) a
    // denser table is worth its longer probe chains -- 201 M entries, 25 M x 150 bp reads at 5 % errors:
    // k_correct 2412 / 2257 / 2024 / 1841 / 1874 / 2551 ms at load 0.5 / 0.6 / 0.65 / 0.7 / 0.75 / 0.8
    double wide_load = ctx->table_load;
    if (!ctx->table_load_set && (double)n / (RC_WIDE_SLOTS * 0.50) * RC_BUCKET_BYTES > 3.0 * 1073741824.0) wide_load = 0.70;
    const uint64_t wide = buckets_for(RC_WIDE_SLOTS, wide_load, 0.50);
    if (wide >= (1ull << 32) - 8) {
        rc_set_error(ctx, "table build: bucket count overflow");
        return RC_ERR_ARG;
    }
    bool ok = false;
    return build_attempt(ctx, d_canon, d_counts, n, 0, 0, (uint32_t)wide, &ok);
}

// forward (or canonical) reference codes -> canonical, in place
__global__ void k_canonicalize(uint64_t *codes, size_t n, int k)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) codes[i] = rc_canonical(codes[i], k);
}

int rc_launch_canonicalize(rc_ctx *ctx, uint64_t *d_codes, size_t n)
{
    if (n == 0) return RC_OK;
    hipLaunchKernelGGL(k_canonicalize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_codes, n, ctx->k);
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}

// Store::GetCount for an array of VALID forward/canonical codes (Store.h:59-66)
__global__ void k_lookup(rc_table_view T, const uint64_t *__restrict__ codes, size_t n, int k, int32_t *__restrict__ out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rc_table_lookup(T, rc_canonical(codes[i], k));
}

int rc_launch_lookup(rc_ctx *ctx, const uint64_t *d_codes, size_t n, int32_t *d_out)
{
    if (n == 0) return RC_OK;
    hipLaunchKernelGGL(k_lookup, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, rc_view(ctx), d_codes, n, ctx->k, d_out);
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}

// main.cpp:329-345: for every dump entry, the counts of its four last-base variants in A,C,G,T
// order, reduced to (max, secondMax) with the reference's tie rule.  out[2*i] = max, [2*i+1] = second.
__global__ void k_last_base_variants(rc_table_view T, const uint64_t *__restrict__ codes, size_t n, int k,
                                     int32_t *__restrict__ out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t base = codes[i] & ~3ull;
    int mx = 0, second = 0;
    for (int c = 0; c < 4; ++c) {
        int cnt = rc_table_lookup_o(T, rc_canonical(base | (uint64_t)c, k), base | (uint64_t)c);  // (the four variants share a filter word)
        if (cnt > mx) {
            second = mx;
            mx = cnt;
        } else if (cnt > second)
            second = cnt;
    }
    out[2 * i] = mx;
    out[2 * i + 1] = second;
}

int rc_launch_last_base_variants(rc_ctx *ctx, const uint64_t *d_codes, size_t n, int32_t *d_max2)
{
    if (n == 0) return RC_OK;
    hipLaunchKernelGGL(k_last_base_variants, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, rc_view(ctx), d_codes, n, ctx->k, d_max2);
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}

// The entries the ERROR_RATE scan keeps (main.cpp:329-347: max >= 1000), picked on the device: of n dump entries only these
// few matter, and only the first 100 000 of them in dump order.  out_key = the entry's place in that order (its index in
// the file, or rc_dump_order_key of its code for a table that was counted here), out_val = max << 32 | second.  n_out
// counts every kept entry, the arrays hold the first `cap` in no particular order (the caller sorts by key).
__global__ __launch_bounds__(256) void k_error_rate_candidates(rc_table_view T, const uint64_t *__restrict__ codes, size_t n, int k, int by_hash,
                                                               uint64_t *__restrict__ out_key, uint64_t *__restrict__ out_val,
                                                               unsigned long long *__restrict__ n_out, size_t cap)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int mx = 0, second = 0;
    uint64_t code = 0;
    if (i < n) {
        code = codes[i];
        const uint64_t base = code & ~3ull;
        for (int c = 0; c < 4; ++c) {
            int cnt = rc_table_lookup_o(T, rc_canonical(base | (uint64_t)c, k), base | (uint64_t)c);  // (the four variants share a filter word)
            if (cnt > mx) {
                second = mx;
                mx = cnt;
            } else if (cnt > second)
                second = cnt;
        }
    }
    const bool keep = i < n && mx >= 1000;
    const unsigned long long m = __ballot(keep);
    if (!m) return;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    unsigned long long base_at = 0;
    if (lane == leader) base_at = atomicAdd(n_out, (unsigned long long)__popcll(m));
    base_at = __shfl(base_at, leader, 64);
    if (keep) {
        const unsigned long long at = base_at + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
        if (at < cap) {
            out_key[at] = by_hash ? rc_dump_order_key(code) : (uint64_t)i;
            out_val[at] = ((uint64_t)(uint32_t)mx << 32) | (uint32_t)second;
        }
    }
}

// the (max, second) pairs of the first `want` kept entries in dump order
int rc_error_rate_candidates(rc_ctx *ctx, const uint64_t *d_codes, size_t n, bool by_hash, size_t want, std::vector<uint64_t> *vals)
{
    vals->clear();
    if (n == 0) return RC_OK;
    rc_dev_tmp b_key, b_val, b_key_s, b_val_s, b_n, b_tmp;
    RC_CHECK_HIP(ctx, b_n.alloc(8));
    size_t cap = std::min<size_t>(n, (size_t)4 << 20);
    unsigned long long found = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        RC_CHECK_HIP(ctx, b_key.alloc(cap * 8));
        RC_CHECK_HIP(ctx, b_val.alloc(cap * 8));
        RC_CHECK_HIP(ctx, hipMemsetAsync(b_n.p, 0, 8, ctx->stream));
        hipLaunchKernelGGL(k_error_rate_candidates, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, rc_view(ctx), d_codes, n, ctx->k, by_hash ? 1 : 0,
                           b_key.as<uint64_t>(), b_val.as<uint64_t>(), b_n.as<unsigned long long>(), cap);
        RC_CHECK_HIP(ctx, hipGetLastError());
        RC_CHECK_HIP(ctx, hipMemcpyAsync(&found, b_n.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (found <= cap) break;
        cap = (size_t)found;  // (a data set with millions of k-mers beyond 1000: once more with room for all of them)
    }
    const size_t m = (size_t)found;
    if (m == 0) return RC_OK;
    RC_CHECK_HIP(ctx, b_key_s.alloc(m * 8));
    RC_CHECK_HIP(ctx, b_val_s.alloc(m * 8));
    size_t t1 = 0;
    RC_CHECK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, t1, b_key.as<uint64_t>(), b_key_s.as<uint64_t>(), b_val.as<uint64_t>(), b_val_s.as<uint64_t>(), m, 0, 64, ctx->stream));
    RC_CHECK_HIP(ctx, b_tmp.alloc(t1));
    RC_CHECK_HIP(ctx, rocprim::radix_sort_pairs(b_tmp.p, t1, b_key.as<uint64_t>(), b_key_s.as<uint64_t>(), b_val.as<uint64_t>(), b_val_s.as<uint64_t>(), m, 0, 64, ctx->stream));
    vals->resize(std::min(m, want));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(vals->data(), b_val_s.p, vals->size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

// the table's canonical codes in HBM, unspecified order: *d_codes is allocated here (hipFree it), *n = their number
int rc_table_codes_device(rc_ctx *ctx, uint64_t **d_codes, size_t *n)
{
    *d_codes = nullptr;
    *n = 0;
    const size_t cap = (size_t)ctx->n_entries;
    if (cap == 0) return RC_OK;
    rc_dev_tmp b_counts, b_n;
    uint64_t *codes = nullptr;
    RC_CHECK_HIP(ctx, hipMalloc((void **)&codes, cap * 8));
    unsigned long long n64 = 0;
    hipError_t e = b_counts.alloc(cap * 4);
    if (e == hipSuccess) e = b_n.alloc(8);
    if (e == hipSuccess) e = hipMemsetAsync(b_n.p, 0, 8, ctx->stream);
    int rc = RC_OK;
    if (e == hipSuccess) rc = rc_launch_export(ctx, codes, b_counts.as<int32_t>(), b_n.as<unsigned long long>(), cap);
    if (e == hipSuccess && rc == RC_OK) e = hipMemcpyAsync(&n64, b_n.p, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rc == RC_OK) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess || rc != RC_OK || n64 > cap) {
        (void)hipFree(codes);
        if (e != hipSuccess) rc_set_error(ctx, "table export: %s", hipGetErrorString(e));
        else if (rc == RC_OK) rc_set_error(ctx, "table export: %llu entries found, at most %zu expected", n64, cap);
        return e != hipSuccess ? RC_ERR_HIP : (rc != RC_OK ? rc : RC_ERR_STATE);
    }
    *d_codes = codes;
    *n = (size_t)n64;
    return RC_OK;
}

// every stored (canonical code, count) pair, in unspecified order (jf_dump writer / test support)
__global__ void k_export(rc_table_view T, size_t nslots, uint64_t *__restrict__ codes,
                         int32_t *__restrict__ counts, unsigned long long *__restrict__ n_out, size_t cap)
{
    size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    const int S = rc_layout_slots(T.layout);
    uint64_t key;
    int32_t cnt;
    if (!rc_table_slot_entry(T, s / S, (int)(s % S), &key, &cnt)) return;
    unsigned long long at = atomicAdd(n_out, 1ull);
    if (at < cap) {
        codes[at] = key;
        counts[at] = cnt;
    }
}

// order-independent 64-bit digest of the table's CONTENT (what Store::GetCount can return): the sum
// over live entries of mix(code, count) -- two tables with the same digest answer every probe alike
// whatever their bucket layout (slot format, load factor, build order)
__global__ void k_digest(rc_table_view T, size_t nslots, unsigned long long *__restrict__ out)
{
    size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    if (s < nslots) {
        const int S = rc_layout_slots(T.layout);
        uint64_t key;
        int32_t cnt;
        if (rc_table_slot_entry(T, s / S, (int)(s % S), &key, &cnt))
            v = rc_dump_order_key(key ^ rc_dump_order_key((uint64_t)(uint32_t)cnt + 0x9E3779B97F4A7C15ull));
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}

int rc_launch_digest(rc_ctx *ctx, unsigned long long *d_out)
{
    const size_t nslots = (size_t)ctx->nb_alloc * rc_layout_slots(ctx->layout);
    hipLaunchKernelGGL(k_digest, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, ctx->stream, rc_view(ctx), nslots, d_out);
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}

int rc_launch_export(rc_ctx *ctx, uint64_t *d_codes, int32_t *d_counts, unsigned long long *d_n, size_t cap)
{
    const size_t nslots = (size_t)ctx->nb_alloc * rc_layout_slots(ctx->layout);
    hipLaunchKernelGGL(k_export, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, ctx->stream, rc_view(ctx), nslots, d_codes, d_counts, d_n, cap);
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}

__global__ void k_dump_order_keys(const uint64_t *__restrict__ codes, size_t n, uint64_t *__restrict__ keys)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = rc_dump_order_key(codes[i]);
}

// every (canonical code, count) of the table in dump order (rc_common.h: rc_dump_order_key), to host
int rc_table_entries_in_dump_order(rc_ctx *ctx, std::vector<uint64_t> *codes, std::vector<int32_t> *counts)
{
    const size_t cap = (size_t)ctx->n_entries;  // upper bound (duplicates were counted in)
    codes->clear();
    if (counts) counts->clear();
    if (cap == 0) return RC_OK;
    rc_dev_tmp b_codes, b_counts, b_n, b_keys, b_keys_s, b_codes_s, b_counts_s, b_tmp;
    unsigned long long n64 = 0;
    RC_CHECK_HIP(ctx, b_codes.alloc(cap * 8));
    RC_CHECK_HIP(ctx, b_counts.alloc(cap * 4));
    RC_CHECK_HIP(ctx, b_n.alloc(8));
    RC_CHECK_HIP(ctx, hipMemsetAsync(b_n.p, 0, 8, ctx->stream));
    int rc = rc_launch_export(ctx, b_codes.as<uint64_t>(), b_counts.as<int32_t>(), b_n.as<unsigned long long>(), cap);
    if (rc) return rc;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(&n64, b_n.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n64 > cap) {
        rc_set_error(ctx, "table export: %llu entries found, at most %zu expected", n64, cap);
        return RC_ERR_STATE;
    }
    const size_t n = (size_t)n64;
    codes->resize(n);
    if (counts) counts->resize(n);
    if (n == 0) return RC_OK;
    RC_CHECK_HIP(ctx, b_keys.alloc(n * 8));
    RC_CHECK_HIP(ctx, b_keys_s.alloc(n * 8));
    RC_CHECK_HIP(ctx, b_codes_s.alloc(n * 8));
    hipLaunchKernelGGL(k_dump_order_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, b_codes.as<uint64_t>(), n, b_keys.as<uint64_t>());
    RC_CHECK_HIP(ctx, hipGetLastError());
    size_t t1 = 0, t2 = 0;
    RC_CHECK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, t1, b_keys.as<uint64_t>(), b_keys_s.as<uint64_t>(), b_codes.as<uint64_t>(), b_codes_s.as<uint64_t>(), n, 0, 64, ctx->stream));
    RC_CHECK_HIP(ctx, b_tmp.alloc(t1));
    RC_CHECK_HIP(ctx, rocprim::radix_sort_pairs(b_tmp.p, t1, b_keys.as<uint64_t>(), b_keys_s.as<uint64_t>(), b_codes.as<uint64_t>(), b_codes_s.as<uint64_t>(), n, 0, 64, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(codes->data(), b_codes_s.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (counts) {
        RC_CHECK_HIP(ctx, b_counts_s.alloc(n * 4));
        RC_CHECK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, t2, b_keys.as<uint64_t>(), b_keys_s.as<uint64_t>(), b_counts.as<int32_t>(), b_counts_s.as<int32_t>(), n, 0, 64, ctx->stream));
        RC_CHECK_HIP(ctx, b_tmp.alloc(t2));
        RC_CHECK_HIP(ctx, rocprim::radix_sort_pairs(b_tmp.p, t2, b_keys.as<uint64_t>(), b_keys_s.as<uint64_t>(), b_counts.as<int32_t>(), b_counts_s.as<int32_t>(), n, 0, 64, ctx->stream));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(counts->data(), b_counts_s.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RC_OK;
}

__global__ void k_selftest_bound(const int32_t *__restrict__ c, size_t n, double e, int32_t *__restrict__ oi, double *__restrict__ od)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    oi[i] = rc_bound_i(c[i], e);
    od[i] = rc_bound_d(c[i], e);
}

int rc_launch_selftest_bound(rc_ctx *ctx, const int32_t *d_c, size_t n, double e, int32_t *d_oi, double *d_od)
{
    if (n == 0) return RC_OK;
    hipLaunchKernelGGL(k_selftest_bound, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_c, n, e, d_oi, d_od);
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}

// Work lists by flag value, in one pass: section c (0 .. NS-1) of d_list, at d_list + c * stride, receives the indices
// i in [0, n) with d_flag[i] == first + c * step; d_count[c] = how many.  A 256-thread workgroup takes 4096 flags (16 per
// thread, one 16-byte load), scans its per-section counts (four 16-bit fields of one 64-bit word) and claims its stretch
// of every section with one atomic per section.  Within a section the indices ascend inside a workgroup's stretch; the
// stretches themselves land in the order the workgroups got there -- nothing downstream depends on the order of a list
// (a read's result depends on its unit alone), only on which section a read is in.  [Rounds 2-3 ran one rocprim::select
// per section: 7 passes of ~0.24 ms over a batch's flags per step, each with its own lookback-scan initialisation.]
// values != nullptr: the lists receive values[i] instead of i (the length-tier lists: i = a position of the locality order,
// values[i] = the read there)
template <int NS>
__global__ __launch_bounds__(256) void k_compact_sections(const uint8_t *__restrict__ flag, uint32_t n, int first, int step,
                                                          uint32_t *__restrict__ list, size_t stride, uint32_t *__restrict__ count,
                                                          const uint32_t *__restrict__ values)
{
    static_assert(NS >= 1 && NS <= 4, "four 16-bit fields");
    __shared__ uint64_t s_wave[4];
    __shared__ uint32_t s_base[NS];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    for (uint32_t b0 = blockIdx.x * 4096u; b0 < n; b0 += gridDim.x * 4096u) {
        const uint32_t i0 = b0 + t * 16u;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i0 < n) v = *reinterpret_cast<const uint4 *>(flag + i0);  // (the flag arrays are reserved with 256 bytes of slack)
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint64_t mine = 0;  // field c = how many of this thread's flags belong to section c
        if (i0 < n && (v.x | v.y | v.z | v.w)) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int f = (int)((w[q >> 2] >> (8 * (q & 3))) & 0xffu);
                if (i0 + q < n && f) {
#pragma unroll
                    for (int c = 0; c < NS; ++c) mine += f == first + c * step ? (1ull << (16 * c)) : 0ull;
                }
            }
        }
        // exclusive scan over the workgroup
        uint64_t inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t y = __shfl_up(inc, o, 64);
            inc += lane >= (uint32_t)o ? y : 0ull;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint64_t before = 0, total = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            before += (uint32_t)q < wave ? s_wave[q] : 0ull;
            total += s_wave[q];
        }
        if (total) {  // (uniform)
            if (t < (uint32_t)NS) {
                const uint32_t c = (uint32_t)((total >> (16 * t)) & 0xffffu);
                s_base[t] = c ? atomicAdd(count + t, c) : 0u;
            }
            __syncthreads();
            if (mine) {
                const uint64_t ex = before + inc - mine;
                uint32_t pos[NS];
#pragma unroll
                for (int c = 0; c < NS; ++c) pos[c] = s_base[c] + (uint32_t)((ex >> (16 * c)) & 0xffffu);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int f = (int)((w[q >> 2] >> (8 * (q & 3))) & 0xffu);
                    if (i0 + q < n && f) {
#pragma unroll
                        for (int c = 0; c < NS; ++c)
                            if (f == first + c * step) list[(size_t)c * stride + pos[c]++] = values ? values[i0 + (uint32_t)q] : i0 + (uint32_t)q;
                    }
                }
            }
        }
        __syncthreads();  // s_wave / s_base are rewritten by the next stretch
    }
}

template <int NS>
static int rc_compact_sections(rc_ctx *ctx, const uint8_t *d_flag, uint32_t n, int first, int step, uint32_t *d_list, size_t stride, uint32_t *d_count,
                               const uint32_t *d_values = nullptr)
{
    RC_CHECK_HIP(ctx, hipMemsetAsync(d_count, 0, NS * sizeof(uint32_t), ctx->stream));
    if (n == 0) return RC_OK;
    unsigned grid = (n + 4095u) / 4096u;
    if (grid > (unsigned)ctx->n_cu * 32u) grid = (unsigned)ctx->n_cu * 32u;
    hipLaunchKernelGGL(k_compact_sections<NS>, dim3(grid), dim3(256), 0, ctx->stream, d_flag, n, first, step, d_list, stride, d_count, d_values);
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}

// section c (0 .. RC_WORK_CLASSES-1) of d_list = the reads with d_cls[i] == RC_WORK_CLASSES - c (k_correct's list: the
// classes expected to be expensive first)
int rc_launch_compact(rc_ctx *ctx, const uint8_t *d_cls, uint32_t n, uint32_t *d_list, size_t stride, uint32_t *d_count)
{
    return rc_compact_sections<RC_WORK_CLASSES>(ctx, d_cls, n, RC_WORK_CLASSES, -1, d_list, stride, d_count);
}

// the same with every section in the batch's locality order (RC_K3_LOCAL=1, dev): the classes are gathered through the order
// (ctx->loc_list: every read once) and the compaction hands out the list's entries instead of positions
__global__ __launch_bounds__(256) void k_gather_u8(const uint8_t *__restrict__ src, const uint32_t *__restrict__ idx, uint32_t n, uint8_t *__restrict__ dst)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
int rc_launch_compact_local(rc_ctx *ctx, const uint8_t *d_cls, uint32_t n, uint32_t *d_list, size_t stride, uint32_t *d_count)
{
    int rc = rc_dbuf_reserve(ctx, &ctx->tier_flag, (size_t)n + 256);  // (the tiers' flag array: free at this point of a pass)
    if (rc) return rc;
    hipLaunchKernelGGL(k_gather_u8, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_cls, (const uint32_t *)ctx->loc_list.p, n, (uint8_t *)ctx->tier_flag.p);
    return rc_compact_sections<RC_WORK_CLASSES>(ctx, (const uint8_t *)ctx->tier_flag.p, n, RC_WORK_CLASSES, -1, d_list, stride, d_count, (const uint32_t *)ctx->loc_list.p);
}

// section v - 1 (v = 1 .. 3) of d_list = the reads with d_flag[i] == v (k_single's list: by number of untrusted stretches)
int rc_launch_compact_flag(rc_ctx *ctx, const uint8_t *d_flag, uint32_t n, uint32_t *d_list, size_t stride, uint32_t *d_count)
{
    return rc_compact_sections<3>(ctx, d_flag, n, 1, 1, d_list, stride, d_count);
}

// ---- locality order of a batch -------------------------------------------------------------------
// Reads arrive in sequencer order, i.e. random with respect to the transcripts they come from, so the
// ~130 probes of a read hit ~130 unrelated buckets and nearly every one of them is an HBM access.
// Reads that overlap share most of their k-mers: probed next to each other, they meet in the L2 /
// Infinity Cache instead.  Large batches are therefore PROBED in "min-hash order": key of a unit (a
// read, or a pair through its first mate) = the smallest hash over its canonical m-mers (m = min(k, 16), see
// k_unit_key), so units that contain the same m-mer as their minimum -- overlapping reads -- become neighbours in the list
// k_probe_list walks.  Nothing is moved: counts land at the reads' own positions, and the threshold
// and correction kernels run as ever.
#define RC_KEY_TILE 40960  // bytes of reads staged per 256-thread workgroup of k_unit_key
__global__ __launch_bounds__(256) void k_unit_key(const uint8_t *__restrict__ seq, size_t nbytes, const uint32_t *__restrict__ off,
                                                  uint32_t n_units, int mode, int k, uint32_t units_per_block,
                                                  uint32_t *__restrict__ keys, uint32_t *__restrict__ idx)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_raw[RC_KEY_TILE + 48];
    const uint32_t u0 = blockIdx.x * units_per_block;
    if (u0 >= n_units) return;
    const uint32_t nu = n_units - u0 < units_per_block ? n_units - u0 : units_per_block;
    const uint32_t r0 = mode == 2 ? 2u * u0 : u0, r1 = mode == 2 ? 2u * (u0 + nu) : u0 + nu;
    const size_t b0 = off[r0], b1 = off[r1], a0 = b0 & ~(size_t)15;
    // the tile's bytes, coalesced (a thread-per-read walk over global memory touches a line per lane per load)
    for (size_t c = threadIdx.x; a0 + 16 * c < b1; c += 256) {
        const size_t g = a0 + 16 * c;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g + 16 <= nbytes) {
            v = *reinterpret_cast<const uint4 *>(seq + g);
        } else {
            uint32_t w[4] = {0, 0, 0, 0};
            for (size_t q = 0; g + q < nbytes; ++q) w[q >> 2] |= (uint32_t)seq[g + q] << (8 * (q & 3));
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        *reinterpret_cast<uint4 *>(s_raw + 16 * c) = v;
    }
    __syncthreads();
    if (threadIdx.x >= nu) return;
    const uint32_t u = u0 + threadIdx.x;
    const uint32_t r = mode == 2 ? 2u * u : u;  // the unit's first mate
    const uint32_t o = (uint32_t)(off[r] - a0);
    const int len = (int)(off[r + 1] - off[r]) - 1;
    // The key only has to make overlapping reads neighbours, so it is the smallest hash over the read's canonical m-mers,
    // m = min(k, 16): two reads that overlap by m bases or more share their minimum whenever it lies in the overlap, as with
    // k-mers -- and an m-mer is one 32-bit word, which takes this loop (a thread per unit, bound by instruction issue) from
    // 64-bit shifts, compares and two more multiplies per base to a handful of 32-bit operations (RC_KEY_KMER=1 at compile
    // time keeps the k-mer version for A/B runs).
#ifndef RC_KEY_KMER
    const int m = k < 16 ? k : 16;
    const uint32_t mask = m == 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);
    const int top = 2 * (m - 1);
    uint32_t fw = 0, rv = 0, best = 0xFFFFFFFFu;
    int valid = 0;
#ifdef RC_KEY_OFFSET
    int best_i = 0;
    bool best_rev = false;
#endif
    for (int i = 0; i < len; ++i) {
        const uint32_t c = s_raw[o + i];
        const uint32_t b = ((c >> 1) ^ (c >> 2)) & 3u;                     // A 0, C 1, G 2, T 3
        const uint32_t d = c - 65u;                                        // 'A' .. 'T': bits 0, 2, 6, 19 of the mask
        const bool acgt = d < 20u && ((0x80045u >> d) & 1u);
        valid = acgt ? valid + 1 : 0;
        fw = ((fw << 2) | b) & mask;
        rv = (rv >> 2) | ((3u - b) << top);
        if (valid >= m) {
            uint32_t h = (fw < rv ? fw : rv) * 0x9E3779B1u;
            h ^= h >> 15;
            h *= 0x2C1B3C6Du;
            h ^= h >> 13;
#ifdef RC_KEY_OFFSET
            if (h < best) {
                best_i = i;
                best_rev = rv < fw;
            }
#endif
            best = h < best ? h : best;
        }
    }
#ifdef RC_KEY_OFFSET
    // Within a group of units that share their minimal m-mer, the order is that of the reads' starts in the frame in which
    // the m-mer is forward (its offset in the read, or in the read's reverse complement): neighbours in the list then
    // overlap in all but a few bases instead of two thirds of them, and a k-mer's second probe follows its first one a
    // read later instead of somewhere in the group -- inside what the L2 still holds.  Seven bits of the key, half a base each.
    if (best != 0xFFFFFFFFu) {
        int p = best_rev ? len - 1 - best_i : best_i - m + 1;
        p = (p >> 1) > 127 ? 127 : (p >> 1);
        best = (best & ~0x7Fu) | (uint32_t)(127 - p);  // (descending offset = ascending start)
    }
#endif
#else
    const uint64_t mask = rc_kmer_mask(k);
    uint64_t fw = 0, rv = 0;
    uint32_t best = 0xFFFFFFFFu;
    int valid = 0;
    for (int i = 0; i < len; ++i) {
        const uint32_t c = s_raw[o + i];
        int b = -1;
        b = c == 'A' ? 0 : b;
        b = c == 'C' ? 1 : b;
        b = c == 'G' ? 2 : b;
        b = c == 'T' ? 3 : b;
        valid = b < 0 ? 0 : valid + 1;
        fw = ((fw << 2) | (uint64_t)(b & 3)) & mask;
        rv = (rv >> 2) | ((uint64_t)(3 - (b & 3)) << (2 * (k - 1)));
        if (valid >= k) {
            const uint32_t h = rc_hash(fw < rv ? fw : rv);
            best = h < best ? h : best;
        }
    }
#endif
    keys[u] = best;
    idx[u] = u;
}

// list[i] = the reads in the order k_probe_list takes them: sorted units, the mates of a pair together
// span[i] (round 6) = where read list[i] lies in the arena -- {first byte, bytes with the NUL}: the fused probe kernel's workgroups
// then find their reads with two coalesced loads side by side instead of a load of the list and, behind it, a gather of the offsets
__global__ void k_probe_order(const uint32_t *__restrict__ unit_sorted, uint32_t n, int mode, uint32_t *__restrict__ list,
                              const uint32_t *__restrict__ off, uint2 *__restrict__ span)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r;
    if (mode == 1)
        r = unit_sorted[i >> 1] + ((i & 1u) ? (n >> 1) : 0u);
    else if (mode == 2)
        r = 2u * unit_sorted[i >> 1] + (i & 1u);
    else
        r = unit_sorted[i];
    list[i] = r;
    const uint32_t g0 = off[r];
    span[i] = make_uint2(g0, off[r + 1] - g0);
}

int rc_launch_locality_order(rc_ctx *ctx, const rc_device_batch_args &a, size_t nbytes)
{
    const uint32_t n = a.n, n_units = a.mode ? n >> 1 : n;
    int rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->loc_a, (size_t)n_units * 16 + 256))) return rc;  // keys, keys', idx, idx'
    if ((rc = rc_dbuf_reserve(ctx, &ctx->loc_list, (size_t)n * 4 + 256))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->loc_span, (size_t)n * 8 + 256))) return rc;
    uint32_t *keys = (uint32_t *)ctx->loc_a.p, *keys2 = keys + n_units, *idx = keys2 + n_units, *idx2 = idx + n_units;
    uint32_t upb = (uint32_t)(RC_KEY_TILE / ((size_t)(a.max_len + 1) * (a.mode == 2 ? 2 : 1)));
    if (upb > 256) upb = 256;
    if (upb < 1) upb = 1;
    hipLaunchKernelGGL(k_unit_key, dim3((n_units + upb - 1) / upb), dim3(256), 0, ctx->stream, a.seq, nbytes, a.off, n_units, a.mode, ctx->k, upb, keys, idx);
    size_t t1 = 0;
    RC_CHECK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, t1, keys, keys2, idx, idx2, (size_t)n_units, 0, 32, ctx->stream));
    if ((rc = rc_dbuf_reserve(ctx, &ctx->sel_tmp, t1))) return rc;
    RC_CHECK_HIP(ctx, rocprim::radix_sort_pairs(ctx->sel_tmp.p, t1, keys, keys2, idx, idx2, (size_t)n_units, 0, 32, ctx->stream));
    hipLaunchKernelGGL(k_probe_order, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, idx2, n, a.mode, (uint32_t *)ctx->loc_list.p, a.off,
                       (uint2 *)ctx->loc_span.p);
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}

// Length tiers of a mixed-length batch (rc_api_batch.hip: rc_correct_device_impl): flag of list position i = 0 if the unit of the read
// there -- the read, or the pair it is a mate of -- belongs to the short tier (its longer read has at most s_hi bases), 1 for
// the middle tier (at most m_hi), 2 for the long one.  Compacted, the positions of a tier give the reads of that tier in
// locality order, mates still adjacent: what the list-driven probe and threshold kernels of the tier's pass walk.
__global__ __launch_bounds__(256) void k_tier_flags(const uint32_t *__restrict__ off, const uint32_t *__restrict__ list, uint32_t n, int mode,
                                                    int s_hi, int m_hi, uint8_t *__restrict__ flag)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = list[i];
    int ml = (int)(off[r + 1] - off[r]) - 1;
    if (mode != 0) {
        const uint32_t half = n >> 1, mr = mode == 1 ? (r < half ? r + half : r - half) : (r ^ 1u);
        const int m1 = (int)(off[mr + 1] - off[mr]) - 1;
        ml = m1 > ml ? m1 : ml;
    }
    flag[i] = ml <= s_hi ? 0 : (ml <= m_hi ? 1 : 2);
}

// ctx->tier_list: two sections of `stride` entries (middle tier, long tier), their lengths at work + RC_WORK_NTIER_OFF
int rc_launch_tier_lists(rc_ctx *ctx, const rc_device_batch_args &a, int s_hi, int m_hi)
{
    int rc;
    const size_t stride = ((size_t)a.n + 63) & ~(size_t)63;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->tier_flag, (size_t)a.n + 256))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->tier_list, stride * 2 * 4 + 256))) return rc;
    ctx->tier_stride = stride;
    hipLaunchKernelGGL(k_tier_flags, dim3((a.n + 255) / 256), dim3(256), 0, ctx->stream, a.off, (const uint32_t *)ctx->loc_list.p, a.n, a.mode, s_hi, m_hi,
                       (uint8_t *)ctx->tier_flag.p);
    RC_CHECK_HIP(ctx, hipGetLastError());
    return rc_compact_sections<2>(ctx, (const uint8_t *)ctx->tier_flag.p, a.n, 1, 1, (uint32_t *)ctx->tier_list.p, stride,
                                  (uint32_t *)((char *)ctx->work.p + RC_WORK_NTIER_OFF), (const uint32_t *)ctx->loc_list.p);
}

// ---- K1: probe kernel ----------------------------------------------------------------------
// counts[a] = GetCount(k-mer starting at arena byte a) for every a whose k-window lies inside one
// read (reads are NUL-terminated inside the arena, so "inside one read" == "no NUL in the
// window").  Windows holding a non-ACGT letter give 0 without touching the table (Store.h:61-62).
// One 256-thread workgroup owns a 4 KiB tile of the arena: it stages the tile (+32 B halo) into
// LDS as 2-bit codes plus two bit masks (non-ACGT, NUL), then every lane extracts its windows
// with funnel shifts, canonicalises with bit-reverse and probes one 64-byte bucket.
template <bool EXT>
__global__ __launch_bounds__(RC_PROBE_THREADS) void k_probe(rc_table_view T, const uint8_t *__restrict__ seq,
                                                            size_t nbytes, int k, int32_t *__restrict__ counts)
{
    __shared__ uint32_t s_code[RC_PROBE_TILE / 16 + 4];
    __shared__ uint16_t s_inv[RC_PROBE_TILE / 16 + 4];
    __shared__ uint16_t s_nul[RC_PROBE_TILE / 16 + 4];
    const size_t tile0 = (size_t)blockIdx.x * RC_PROBE_TILE;
    const int t = threadIdx.x;

    // stage: thread t packs bytes [16t, 16t+16) of the tile; threads 0..1 also pack the halo
    for (int chunk = t; chunk < RC_PROBE_TILE / 16 + 2; chunk += RC_PROBE_THREADS) {
        const size_t g = tile0 + (size_t)chunk * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g + 16 <= nbytes) {
            v = *reinterpret_cast<const uint4 *>(seq + g);
        } else if (g < nbytes) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (size_t j = 0; g + j < nbytes; ++j) w[j >> 2] |= (uint32_t)seq[g + j] << (8 * (j & 3));
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        uint32_t code, inv, nul;
        rc_pack16(v, code, inv, nul);
        s_code[chunk] = code;
        s_inv[chunk ^ 1] = (uint16_t)inv;  // big-endian bit order inside each 32-bit mask word
        s_nul[chunk ^ 1] = (uint16_t)nul;
    }
    if (t < 2) {
        s_code[RC_PROBE_TILE / 16 + 2 + t] = 0xFFFFFFFFu;
    }
    __syncthreads();
    const uint32_t *m_inv = reinterpret_cast<const uint32_t *>(s_inv);
    const uint32_t *m_nul = reinterpret_cast<const uint32_t *>(s_nul);

#pragma unroll 2
    for (int a = t; a < RC_PROBE_TILE; a += RC_PROBE_THREADS) {
        const size_t g = tile0 + (size_t)a;
        if (g + (size_t)k > nbytes) break;
        const int mw = a >> 5, ms = a & 31;
        const uint64_t nulw = (((uint64_t)m_nul[mw] << 32) | m_nul[mw + 1]) << ms;
        if (nulw >> (64 - k)) continue;  // window crosses a read boundary: not a k-mer of any read
        const uint64_t invw = (((uint64_t)m_inv[mw] << 32) | m_inv[mw + 1]) << ms;
        int cnt = 0;
        if (!(invw >> (64 - k))) {
            const int cw = a >> 4, cs = 2 * (a & 15);
            uint64_t x = ((uint64_t)s_code[cw] << 32) | s_code[cw + 1];
            if (cs) x = (x << cs) | ((uint64_t)s_code[cw + 2] >> (32 - cs));
            const uint64_t code = x >> (64 - 2 * k);
            cnt = rc_table_lookup<EXT>(T, rc_canonical(code, k));
        }
        __builtin_nontemporal_store(cnt, &counts[g]);  // streamed once: keep it out of the caches the table lives in
    }
}

// K1 over a list of reads (locality order): the workgroup's reads are copied into a local arena in
// LDS -- each at the byte alignment it has in memory, NULs in between -- packed and probed as in
// k_probe; a count goes to the position of its k-mer in the caller's arena.
// skip_hi >= 0: the reads of units (a read, or a pair: mode as in rc_kernel_args) whose longer read has at most skip_hi
// bases are left out -- the fused probe + threshold kernel of the short tier has their counts (rc_correct.hip).
template <bool EXT>
__global__ __launch_bounds__(RC_PROBE_THREADS) void k_probe_list(rc_table_view T, const uint8_t *__restrict__ seq, size_t nbytes,
                                                                 const uint32_t *__restrict__ off, const uint32_t *__restrict__ list,
                                                                 uint32_t n, uint32_t reads_per_block, int k, int32_t *__restrict__ counts,
                                                                 int mode, int skip_hi, const uint32_t *__restrict__ n_list)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_raw[(RC_PROBE_TILE + 64) / 4];
    __shared__ uint32_t s_code[RC_PROBE_TILE / 16 + 4];
    __shared__ uint16_t s_inv[RC_PROBE_TILE / 16 + 4];
    __shared__ uint16_t s_nul[RC_PROBE_TILE / 16 + 4];
    __shared__ uint32_t s_lpos[RC_PLIST_MAX_READS + 1], s_gpos[RC_PLIST_MAX_READS], s_len1[RC_PLIST_MAX_READS];
    const int t = threadIdx.x;
    // n_list != nullptr: `list` is a tier's list whose length lives on the device (rc_launch_tier_lists); a fixed grid walks it
    const uint32_t n_eff = n_list ? *n_list : n;
    for (uint32_t i0 = blockIdx.x * reads_per_block; i0 < n_eff; i0 += gridDim.x * reads_per_block) {
    const uint32_t nr = n_eff - i0 < reads_per_block ? n_eff - i0 : reads_per_block;
    bool any_live = false;
    if ((uint32_t)t < nr) {
        const uint32_t r = list[i0 + t], g0 = off[r];
        uint32_t len1 = off[r + 1] - g0;  // bases + the NUL
        if (skip_hi >= 0) {
            uint32_t ml1 = len1;
            if (mode != 0) {
                const uint32_t half = n >> 1, mr = mode == 1 ? (r < half ? r + half : r - half) : (r ^ 1u);
                const uint32_t m1 = off[mr + 1] - off[mr];
                ml1 = m1 > ml1 ? m1 : ml1;
            }
            if ((int)ml1 - 1 <= skip_hi) len1 = 0;  // 0 = not a read of this launch
        }
        s_gpos[t] = g0;
        s_len1[t] = len1;
        any_live = len1 != 0;
    }
    // (a mixed-length batch's short reads were probed by the fused kernel: most workgroups of this launch hold none)
    if (!__syncthreads_or(any_live)) continue;
    for (int c = t; c < (RC_PROBE_TILE + 64) / 4; c += RC_PROBE_THREADS) s_raw[c] = 0;
    if (t == 0) {  // local start of each read: same alignment modulo 4 as in memory, a NUL in front
        uint32_t lp = 4;
        for (uint32_t j = 0; j < nr; ++j) {
            lp = ((lp + 3u) & ~3u) + (s_gpos[j] & 3u);
            s_lpos[j] = lp;
            lp += s_len1[j];
        }
        s_lpos[nr] = lp;
    }
    __syncthreads();
    // copy: one 64-lane group per read, aligned dwords, bytes outside the read masked to NUL
    for (uint32_t j = (uint32_t)t >> 6; j < nr; j += RC_PROBE_THREADS / 64) {
        if (!s_len1[j]) continue;
        const uint32_t g0 = s_gpos[j], lp = s_lpos[j], g1 = g0 + s_len1[j] - 1;  // [g0, g1): the bases
        const uint32_t w0 = g0 >> 2, w1 = (g1 + 3) >> 2;
        for (uint32_t w = w0 + ((uint32_t)t & 63u); w < w1; w += 64u) {
            uint32_t v;
            if ((size_t)4 * w + 4 <= nbytes) {
                v = *reinterpret_cast<const uint32_t *>(seq + (size_t)4 * w);
            } else {
                v = 0;
                for (size_t q = 0; (size_t)4 * w + q < nbytes; ++q) v |= (uint32_t)seq[(size_t)4 * w + q] << (8 * q);
            }
            const uint32_t lo = 4 * w < g0 ? g0 - 4 * w : 0, hi = 4 * w + 4 > g1 ? 4 * w + 4 - g1 : 0;  // bytes to drop at either end
            uint32_t m = 0xFFFFFFFFu;
            if (lo) m &= 0xFFFFFFFFu << (8 * lo);
            if (hi) m &= 0xFFFFFFFFu >> (8 * hi);
            s_raw[(lp >> 2) + (w - w0)] = v & m;
        }
    }
    __syncthreads();
    const uint32_t total = s_lpos[nr];
    for (int chunk = t; chunk < RC_PROBE_TILE / 16 + 2; chunk += RC_PROBE_THREADS) {
        const uint4 v = *reinterpret_cast<const uint4 *>(s_raw + 4 * chunk);
        uint32_t code, inv, nul;
        rc_pack16(v, code, inv, nul);
        s_code[chunk] = code;
        s_inv[chunk ^ 1] = (uint16_t)inv;
        s_nul[chunk ^ 1] = (uint16_t)nul;
    }
    if (t < 2) s_code[RC_PROBE_TILE / 16 + 2 + t] = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t *m_inv = reinterpret_cast<const uint32_t *>(s_inv);
    const uint32_t *m_nul = reinterpret_cast<const uint32_t *>(s_nul);
#pragma unroll 2
    for (uint32_t a = 4 + (uint32_t)t; a + (uint32_t)k <= total; a += RC_PROBE_THREADS) {
        const int mw = a >> 5, ms = a & 31;
        const uint64_t nulw = (((uint64_t)m_nul[mw] << 32) | m_nul[mw + 1]) << ms;
        if (nulw >> (64 - k)) continue;  // window crosses a read boundary (or padding)
        const uint64_t invw = (((uint64_t)m_inv[mw] << 32) | m_inv[mw + 1]) << ms;
        int cnt = 0;
        if (!(invw >> (64 - k))) {
            const int cw = a >> 4, cs = 2 * (a & 15);
            uint64_t x = ((uint64_t)s_code[cw] << 32) | s_code[cw + 1];
            if (cs) x = (x << cs) | ((uint64_t)s_code[cw + 2] >> (32 - cs));
            cnt = rc_table_lookup<EXT>(T, rc_canonical(x >> (64 - 2 * k), k));
        }
        // the read this position belongs to: last j with lpos[j] <= a
        uint32_t lo = 0, hi = nr;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_lpos[mid] <= a)
                lo = mid;
            else
                hi = mid;
        }
        __builtin_nontemporal_store(cnt, &counts[s_gpos[lo] + (a - s_lpos[lo])]);
    }
    __syncthreads();  // (the next stretch of the list rewrites the tile)
    }
}

int rc_launch_probe_list(rc_ctx *ctx, const rc_device_batch_args &a, size_t nbytes, int32_t *d_counts, int skip_hi)
{
    if (a.n == 0) return RC_OK;
    if (!ctx->d_buckets) {
        rc_set_error(ctx, "probe: no k-mer table loaded");
        return RC_ERR_STATE;
    }
    uint32_t rpb = (uint32_t)((RC_PROBE_TILE - 8) / (a.max_len + 8));  // a read takes its bases, the NUL and up to 6 bytes of alignment
    if (rpb > RC_PLIST_MAX_READS) rpb = RC_PLIST_MAX_READS;
    if (rpb < 1) rpb = 1;
    rc_timer_begin(ctx);
    if (ctx->ext)
        hipLaunchKernelGGL(k_probe_list<true>, dim3((a.n + rpb - 1) / rpb), dim3(RC_PROBE_THREADS), 0, ctx->stream, rc_view(ctx), a.seq, nbytes, a.off,
                           (const uint32_t *)ctx->loc_list.p, a.n, rpb, ctx->k, d_counts, a.mode, skip_hi, (const uint32_t *)nullptr);
    else
        hipLaunchKernelGGL(k_probe_list<false>, dim3((a.n + rpb - 1) / rpb), dim3(RC_PROBE_THREADS), 0, ctx->stream, rc_view(ctx), a.seq, nbytes, a.off,
                           (const uint32_t *)ctx->loc_list.p, a.n, rpb, ctx->k, d_counts, a.mode, skip_hi, (const uint32_t *)nullptr);
    rc_timer_end(ctx, RC_T_PROBE);
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}

int rc_launch_probe_tier(rc_ctx *ctx, const rc_device_batch_args &a, size_t nbytes, int32_t *d_counts, int section)
{
    if (a.n == 0) return RC_OK;
    uint32_t rpb = (uint32_t)((RC_PROBE_TILE - 8) / (a.max_len + 8));
    if (rpb > RC_PLIST_MAX_READS) rpb = RC_PLIST_MAX_READS;
    if (rpb < 1) rpb = 1;
    const uint32_t *list = (const uint32_t *)ctx->tier_list.p + (size_t)section * ctx->tier_stride;
    const uint32_t *n_list = (const uint32_t *)((char *)ctx->work.p + RC_WORK_NTIER_OFF) + section;
    unsigned grid = (a.n + rpb - 1) / rpb;  // (at most: the tier's share of the batch is not known here)
    if (grid > (unsigned)ctx->n_cu * 16u) grid = (unsigned)ctx->n_cu * 16u;
    rc_timer_begin(ctx);
    if (ctx->ext)
        hipLaunchKernelGGL(k_probe_list<true>, dim3(grid), dim3(RC_PROBE_THREADS), 0, ctx->stream, rc_view(ctx), a.seq, nbytes, a.off, list, a.n, rpb, ctx->k,
                           d_counts, a.mode, -1, n_list);
    else
        hipLaunchKernelGGL(k_probe_list<false>, dim3(grid), dim3(RC_PROBE_THREADS), 0, ctx->stream, rc_view(ctx), a.seq, nbytes, a.off, list, a.n, rpb, ctx->k,
                           d_counts, a.mode, -1, n_list);
    rc_timer_end(ctx, RC_T_PROBE);
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}

int rc_launch_probe(rc_ctx *ctx, const uint8_t *d_seq, size_t nbytes, int32_t *d_counts)
{
    if (nbytes == 0) return RC_OK;
    if (!ctx->d_buckets) {
        rc_set_error(ctx, "probe: no k-mer table loaded");
        return RC_ERR_STATE;
    }
    const unsigned G = (unsigned)((nbytes + RC_PROBE_TILE - 1) / RC_PROBE_TILE);
    rc_timer_begin(ctx);
    if (ctx->ext)
        hipLaunchKernelGGL(k_probe<true>, dim3(G), dim3(RC_PROBE_THREADS), 0, ctx->stream, rc_view(ctx), d_seq, nbytes, ctx->k, d_counts);
    else
        hipLaunchKernelGGL(k_probe<false>, dim3(G), dim3(RC_PROBE_THREADS), 0, ctx->stream, rc_view(ctx), d_seq, nbytes, ctx->k, d_counts);
    rc_timer_end(ctx, RC_T_PROBE);
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}

// ---- exact k-mer counter in bounded memory (stages 0-2 of run_rcorrector.pl:262-281 for reads that are, or
// pass through, HBM).  `jellyfish bc` + `count --bc` exist so that the singletons of a data set -- most of its
// distinct k-mers once reads carry errors -- never occupy the counter (run_rcorrector.pl:262-273).  Here the same
// end is reached by cutting the KEY SPACE instead: the arenas handed over are kept in HBM (one byte per base:
// 100 M x 150 bp are 15 GB of 288), and finish() makes P passes over them; pass p looks only at the k-mers whose
// hash falls into slice p of P -- emit -> radix sort -> run-length encode -> keep count >= min_count -- so that no
// more than 1/P of the k-mer occurrences is ever in flight, whatever share of them are singletons.  A histogram
// pass sizes the slices; P follows from the memory the passes may use (RC_COUNT_MEM_MB, default 24 GiB).  The result
// is what `jellyfish count -C` + `dump -L 2` hands to the reference: every canonical k-mer with its exact count.
__global__ void k_u32_to_i32_clamped(const uint32_t *in, int32_t *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] > 0x7fffffffu ? 0x7fffffff : (int32_t)in[i];
}

__device__ __forceinline__ uint32_t rc_count_slice(uint64_t key, uint32_t P)
{
    return (uint32_t)(((uint64_t)rc_hash(key ^ 0x9E3779B97F4A7C15ull) * P) >> 32);
}

// MODE 0: hist[slice] += valid k-mers of the tile; MODE 1: the canonical codes of slice `p` are appended to out
template <int MODE>
__global__ __launch_bounds__(RC_PROBE_THREADS) void k_count_scan(const uint8_t *__restrict__ seq, size_t nbytes, int k, uint32_t P, uint32_t p,
                                                                 unsigned long long *__restrict__ hist, uint64_t *__restrict__ out,
                                                                 unsigned long long *__restrict__ cursor)
{
    __shared__ uint32_t s_code[RC_PROBE_TILE / 16 + 4];
    __shared__ uint16_t s_inv[RC_PROBE_TILE / 16 + 4];
    __shared__ uint32_t s_hist[64];
    const size_t tile0 = (size_t)blockIdx.x * RC_PROBE_TILE;
    const int t = threadIdx.x;
    if (MODE == 0 && t < 64) s_hist[t] = 0;
    for (int chunk = t; chunk < RC_PROBE_TILE / 16 + 2; chunk += RC_PROBE_THREADS) {
        const size_t g = tile0 + (size_t)chunk * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g + 16 <= nbytes) {
            v = *reinterpret_cast<const uint4 *>(seq + g);
        } else if (g < nbytes) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (size_t j = 0; g + j < nbytes; ++j) w[j >> 2] |= (uint32_t)seq[g + j] << (8 * (j & 3));
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        uint32_t code, inv, nul;
        rc_pack16(v, code, inv, nul);
        s_code[chunk] = code;
        s_inv[chunk ^ 1] = (uint16_t)inv;  // NUL is also "not ACGT"
    }
    if (t < 2) s_code[RC_PROBE_TILE / 16 + 2 + t] = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t *m_inv = reinterpret_cast<const uint32_t *>(s_inv);
    // the canonical code of the window at tile position a, if it is a k-mer of a read, and its slice
    auto window = [&](int a, uint64_t &key, uint32_t &sl) -> bool {
        const size_t g = tile0 + (size_t)a;
        if (g + (size_t)k > nbytes) return false;
        const int mw = a >> 5, ms = a & 31;
        const uint64_t invw = (((uint64_t)m_inv[mw] << 32) | m_inv[mw + 1]) << ms;
        if (invw >> (64 - k)) return false;
        const int cw = a >> 4, cs = 2 * (a & 15);
        uint64_t x = ((uint64_t)s_code[cw] << 32) | s_code[cw + 1];
        if (cs) x = (x << cs) | ((uint64_t)s_code[cw + 2] >> (32 - cs));
        key = rc_canonical(x >> (64 - 2 * k), k);
        sl = rc_count_slice(key, P);
        return true;
    };
    constexpr int ITER = RC_PROBE_TILE / RC_PROBE_THREADS, WAVES = RC_PROBE_THREADS / 64;
    if (MODE == 0) {
        for (int it = 0; it < ITER; ++it) {
            uint64_t key;
            uint32_t sl;
            if (window(it * RC_PROBE_THREADS + t, key, sl)) atomicAdd(&s_hist[sl & 63u], 1u);  // (P <= 64)
        }
        __syncthreads();
        if (t < 64 && s_hist[t]) atomicAdd(hist + t, (unsigned long long)s_hist[t]);
        return;
    }
    // MODE 1: ONE atomic on the output cursor per workgroup (one word sustains ~90 atomics per microsecond; a wave-level
    // reservation is 60 times as many): count the keys of slice p per (iteration, wave), reserve, then write
    __shared__ uint32_t s_n[ITER * WAVES + 1];
    __shared__ unsigned long long s_base;
    const int wv = t >> 6, lane = t & 63;
    for (int it = 0; it < ITER; ++it) {
        uint64_t key;
        uint32_t sl;
        const bool take = window(it * RC_PROBE_THREADS + t, key, sl) && sl == p;
        const unsigned long long m = __ballot(take);
        if (lane == 0) s_n[it * WAVES + wv] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (int i = 0; i < ITER * WAVES; ++i) {
            const uint32_t c = s_n[i];
            s_n[i] = run;
            run += c;
        }
        s_base = run ? atomicAdd(cursor, (unsigned long long)run) : 0ull;
    }
    __syncthreads();
    const unsigned long long base = s_base;
    for (int it = 0; it < ITER; ++it) {
        uint64_t key = 0;
        uint32_t sl;
        const bool take = window(it * RC_PROBE_THREADS + t, key, sl) && sl == p;
        const unsigned long long m = __ballot(take);
        if (take) out[base + s_n[it * WAVES + wv] + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull))] = key;
    }
}

__global__ void k_flag_keep(const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ cnt, size_t n, int min_count,
                            uint8_t *__restrict__ keep)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keep[i] = cnt[i] >= (uint32_t)min_count ? 1 : 0;
}

static void rc_count_release(rc_ctx *ctx)
{
    for (auto &a : ctx->cnt_chunks)
        if (a.p) (void)hipFree(a.p);
    ctx->cnt_chunks.clear();
    ctx->cnt_chunk_used = 0;
    ctx->cnt_arenas.clear();
    ctx->cnt_total = 0;
}

void rc_kept_release(rc_ctx *ctx)
{
    for (auto &a : ctx->kept_chunks)
        if (a.p) (void)hipFree(a.p);
    ctx->kept_chunks.clear();
    ctx->kept_arenas.clear();
}

int rc_count_begin(rc_ctx *ctx)
{
    rc_count_release(ctx);
    rc_kept_release(ctx);
    ctx->cnt_active = true;
    return RC_OK;
}

// keeps a copy of the arena in HBM (from_device: d_seq is device memory, else host memory)
int rc_count_add(rc_ctx *ctx, const uint8_t *seq, size_t nbytes, bool from_device)
{
    if (!ctx->cnt_active) {
        rc_set_error(ctx, "count_add: call rc_table_count_begin first");
        return RC_ERR_STATE;
    }
    if (nbytes == 0) return RC_OK;
    if (nbytes >= (1ull << 32)) {
        rc_set_error(ctx, "count: an arena must be below 2^32 bytes (add it in pieces)");
        return RC_ERR_ARG;
    }
    size_t cap = (size_t)128 << 30;  // what the counter may keep in HBM
    if (const char *e = getenv("RC_COUNT_RETAIN_MB")) cap = (size_t)atoll(e) << 20;
    if (ctx->cnt_total + nbytes > cap) {
        rc_set_error(ctx, "count: %zu MB of reads exceed what the k-mer counter keeps in HBM (%zu MB, RC_COUNT_RETAIN_MB): count them with "
                          "jellyfish and pass the dump (-c)", (ctx->cnt_total + nbytes) >> 20, cap >> 20);
        return RC_ERR_NOMEM;
    }
    // the arena's place: behind the last one in the current chunk (256-byte aligned, 64 bytes of slack), or a new chunk
    const size_t need = (nbytes + 64 + 255) & ~(size_t)255, chunk_bytes = (size_t)2 << 30;
    if (ctx->cnt_chunks.empty() || ctx->cnt_chunk_used + need > ctx->cnt_chunks.back().bytes) {
        rc_dbuf c;
        c.bytes = need > chunk_bytes ? need : chunk_bytes;
        RC_CHECK_HIP(ctx, hipMalloc(&c.p, c.bytes));
        ctx->cnt_chunks.push_back(c);
        ctx->cnt_chunk_used = 0;
    }
    rc_dbuf a;
    a.p = (char *)ctx->cnt_chunks.back().p + ctx->cnt_chunk_used;
    a.bytes = nbytes;
    hipError_t e = hipMemcpyAsync(a.p, seq, nbytes, from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // (the caller's buffer is its own again when this returns)
    if (e != hipSuccess) {
        rc_set_error(ctx, "count_add: copy failed: %s", hipGetErrorString(e));
        return RC_ERR_HIP;
    }
    ctx->cnt_chunk_used += need;
    ctx->cnt_arenas.push_back(a);
    ctx->cnt_total += nbytes;
    return RC_OK;
}

int rc_count_finish(rc_ctx *ctx, int min_count, int64_t *n_kmers)
{
    if (!ctx->cnt_active) {
        rc_set_error(ctx, "count_finish: call rc_table_count_begin first");
        return RC_ERR_STATE;
    }
    ctx->cnt_active = false;
    // RC_COUNT_TIMING=1 (dev): where finish() spends its time, on stderr
    static const bool timing = getenv("RC_COUNT_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    double t_alloc = 0, t_emit = 0, t_sort = 0, t_rle = 0, t_sel = 0;  // (t_sort / t_rle / t_sel: with RC_COUNT_TIMING's extra synchronisations)
    struct release_on_exit {
        rc_ctx *c;
        ~release_on_exit() { rc_count_release(c); }  // (an error leaves nothing behind; success with cnt_keep has moved the arenas out)
    } guard{ctx};
    const int k = ctx->k;
    // passes: a pass holds, per k-mer occurrence of its slice, the key (8 B), its sorted copy (8 B), the sort's scratch
    // (~8 B) and the run-length output (8 + 4 + 1 B)
    size_t mem = (size_t)24 << 30;
    if (const char *e = getenv("RC_COUNT_MEM_MB")) mem = (size_t)atoll(e) << 20;
    const double per_occ = 40.0;
    uint32_t P = (uint32_t)((double)ctx->cnt_total * per_occ * 1.15 / (double)mem) + 1;
    if (P > 64) P = 64;
    rc_dev_tmp b_hist, b_cursor;
    RC_CHECK_HIP(ctx, b_hist.alloc(64 * 8));
    RC_CHECK_HIP(ctx, b_cursor.alloc(8));
    RC_CHECK_HIP(ctx, hipMemsetAsync(b_hist.p, 0, 64 * 8, ctx->stream));
    for (const auto &a : ctx->cnt_arenas) {
        const unsigned G = (unsigned)((a.bytes + RC_PROBE_TILE - 1) / RC_PROBE_TILE);
        hipLaunchKernelGGL(k_count_scan<0>, dim3(G), dim3(RC_PROBE_THREADS), 0, ctx->stream, (const uint8_t *)a.p, a.bytes, k, P, 0u,
                           b_hist.as<unsigned long long>(), (uint64_t *)nullptr, (unsigned long long *)nullptr);
    }
    RC_CHECK_HIP(ctx, hipGetLastError());
    unsigned long long hist[64];
    RC_CHECK_HIP(ctx, hipMemcpyAsync(hist, b_hist.p, sizeof hist, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    size_t max_slice = 0;
    for (uint32_t p = 0; p < P; ++p) max_slice = std::max(max_slice, (size_t)hist[p]);
    if (max_slice >= (1ull << 32)) {
        rc_set_error(ctx, "count: a pass of %zu k-mer occurrences exceeds 2^32 (lower RC_COUNT_MEM_MB for more passes)", max_slice);
        return RC_ERR_ARG;
    }
    // The kept entries of all passes go straight into the two arrays the table is built from.  Their number is known only
    // at the end: the arrays are sized from the first pass that keeps anything (its share of the occurrences, + 15 %) and
    // regrown in the rare case a later pass does not fit.  One allocation holds the passes' scratch.  (Round 3 allocated
    // per pass and concatenated at the end: some forty hipMalloc / hipFree calls, each a round trip through the kernel
    // driver -- 0.1 s on a quiet host, 0.5 s and more on a busy one, against 0.2 s for the counting itself.)
    rc_dev_tmp b_allk, b_allc;
    size_t total_kept = 0, cap_kept = 0;
    if (max_slice > 0) {
        size_t ts_sort = 0, ts_rle = 0, ts_sel = 0;
        RC_CHECK_HIP(ctx, rocprim::radix_sort_keys(nullptr, ts_sort, (uint64_t *)nullptr, (uint64_t *)nullptr, max_slice, 0, 2 * k > 64 ? 64 : 2 * k, ctx->stream));
        RC_CHECK_HIP(ctx, rocprim::run_length_encode(nullptr, ts_rle, (uint64_t *)nullptr, (unsigned int)max_slice, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                                     (size_t *)nullptr, ctx->stream));
        RC_CHECK_HIP(ctx, rocprim::select(nullptr, ts_sel, (uint64_t *)nullptr, (uint8_t *)nullptr, (uint64_t *)nullptr, (size_t *)nullptr, max_slice, ctx->stream));
        const size_t tmp_bytes = std::max(ts_sort, std::max(ts_rle, ts_sel));
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t o_keys = 0, o_keys_s = o_keys + up(max_slice * 8), o_cnt = o_keys_s + up(max_slice * 8), o_keep = o_cnt + up(max_slice * 4),
                     o_runs = o_keep + up(max_slice), o_tmp = o_runs + 256, pool_bytes = o_tmp + up(tmp_bytes);
        rc_dev_tmp b_pool;
        const double ta0 = now();
        RC_CHECK_HIP(ctx, b_pool.alloc(pool_bytes));
        t_alloc += now() - ta0;
        char *pool = b_pool.as<char>();
        uint64_t *keys = (uint64_t *)(pool + o_keys), *keys_s = (uint64_t *)(pool + o_keys_s);
        uint32_t *cnt = (uint32_t *)(pool + o_cnt);
        uint8_t *keep = (uint8_t *)(pool + o_keep);
        size_t *d_runs = (size_t *)(pool + o_runs);
        void *tmp = pool + o_tmp;
        unsigned long long occ_total = 0;
        for (uint32_t p = 0; p < P; ++p) occ_total += hist[p];
        for (uint32_t p = 0; p < P; ++p) {
            const size_t m = (size_t)hist[p];
            if (m == 0) continue;
            RC_CHECK_HIP(ctx, hipMemsetAsync(b_cursor.p, 0, 8, ctx->stream));
            for (const auto &a : ctx->cnt_arenas) {
                const unsigned G = (unsigned)((a.bytes + RC_PROBE_TILE - 1) / RC_PROBE_TILE);
                hipLaunchKernelGGL(k_count_scan<1>, dim3(G), dim3(RC_PROBE_THREADS), 0, ctx->stream, (const uint8_t *)a.p, a.bytes, k, P, p,
                                   (unsigned long long *)nullptr, keys, b_cursor.as<unsigned long long>());
            }
            RC_CHECK_HIP(ctx, hipGetLastError());
            double tp = now();
            if (timing) {
                RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
                t_emit += now() - tp;
                tp = now();
            }
            size_t t1 = tmp_bytes;
            RC_CHECK_HIP(ctx, rocprim::radix_sort_keys(tmp, t1, keys, keys_s, m, 0, 2 * k > 64 ? 64 : 2 * k, ctx->stream));
            if (timing) {
                RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
                t_sort += now() - tp;
                tp = now();
            }
            t1 = tmp_bytes;
            RC_CHECK_HIP(ctx, rocprim::run_length_encode(tmp, t1, keys_s, (unsigned int)m, keys, cnt, d_runs, ctx->stream));
            size_t runs = 0;
            RC_CHECK_HIP(ctx, hipMemcpyAsync(&runs, d_runs, sizeof(size_t), hipMemcpyDeviceToHost, ctx->stream));
            RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            t_rle += now() - tp;
            if (runs == 0) continue;
            tp = now();
            hipLaunchKernelGGL(k_flag_keep, dim3((unsigned)((runs + 255) / 256)), dim3(256), 0, ctx->stream, keys, cnt, runs, min_count, keep);
            // the kept keys land in keys_s (free again) and are copied out; their counts go through a second select into the
            // same buffer and from there, clamped to int32, to their place
            t1 = tmp_bytes;
            RC_CHECK_HIP(ctx, rocprim::select(tmp, t1, keys, keep, keys_s, d_runs, runs, ctx->stream));
            size_t nsel = 0;
            RC_CHECK_HIP(ctx, hipMemcpyAsync(&nsel, d_runs, sizeof(size_t), hipMemcpyDeviceToHost, ctx->stream));
            RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            t_sel += now() - tp;
            if (nsel == 0) continue;
            if (total_kept + nsel > cap_kept) {
                const double ta1 = now();
                // what this pass kept of its occurrences, applied to the occurrences still to come
                unsigned long long seen = 0;
                for (uint32_t q = 0; q <= p; ++q) seen += hist[q];
                const double per_occ_kept = (double)(total_kept + nsel) / (double)(seen ? seen : 1);
                size_t want = (size_t)(per_occ_kept * (double)occ_total * (cap_kept ? 1.5 : 1.15)) + ((size_t)1 << 20);
                static const bool tight = getenv("RC_COUNT_TIGHT") != nullptr;  // tests: no slack, every pass regrows the arrays
                if (want < total_kept + nsel || tight) want = total_kept + nsel;
                rc_dev_tmp nk, nc;
                RC_CHECK_HIP(ctx, nk.alloc((want + 1) * 8));
                RC_CHECK_HIP(ctx, nc.alloc((want + 1) * 4));
                if (total_kept) {
                    RC_CHECK_HIP(ctx, hipMemcpyAsync(nk.p, b_allk.p, total_kept * 8, hipMemcpyDeviceToDevice, ctx->stream));
                    RC_CHECK_HIP(ctx, hipMemcpyAsync(nc.p, b_allc.p, total_kept * 4, hipMemcpyDeviceToDevice, ctx->stream));
                    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
                }
                std::swap(nk.p, b_allk.p);
                std::swap(nc.p, b_allc.p);
                cap_kept = want;
                t_alloc += now() - ta1;
            }
            RC_CHECK_HIP(ctx, hipMemcpyAsync(b_allk.as<uint64_t>() + total_kept, keys_s, nsel * 8, hipMemcpyDeviceToDevice, ctx->stream));
            uint32_t *selc = reinterpret_cast<uint32_t *>(keys_s);  // (keys_s was copied out: the stream orders the reuse)
            t1 = tmp_bytes;
            RC_CHECK_HIP(ctx, rocprim::select(tmp, t1, cnt, keep, selc, d_runs, runs, ctx->stream));
            hipLaunchKernelGGL(k_u32_to_i32_clamped, dim3((unsigned)((nsel + 255) / 256)), dim3(256), 0, ctx->stream, selc, b_allc.as<int32_t>() + total_kept, nsel);
            RC_CHECK_HIP(ctx, hipGetLastError());
            total_kept += nsel;
        }
        RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (!b_allk.p) {  // nothing kept: the build still wants its two arrays
        RC_CHECK_HIP(ctx, b_allk.alloc(8));
        RC_CHECK_HIP(ctx, b_allc.alloc(4));
    }
    if (ctx->cnt_keep) {  // the reads stay where they are for rc_submit_resident
        ctx->kept_arenas.swap(ctx->cnt_arenas);
        ctx->kept_chunks.swap(ctx->cnt_chunks);
        ctx->cnt_chunk_used = 0;
        ctx->cnt_total = 0;
    }
    const double t_passes = now();
    rc_count_release(ctx);  // the reads are no longer needed: their memory goes to the table build
    const double t_concat = now();
    int rc = rc_build_table_from_device_pairs(ctx, b_allk.as<uint64_t>(), b_allc.as<int32_t>(), total_kept);
    if (timing)
        fprintf(stderr, "[rc count timing] finish %.3f s: histogram + %u passes %.3f (emit %.3f, sort %.3f, run lengths %.3f, select %.3f, hipMalloc %.3f), reads released %.3f, table build %.3f\n",
                now() - t_begin, P, t_passes - t_begin, t_emit, t_sort, t_rle, t_sel, t_alloc, t_concat - t_passes, now() - t_concat);
    if (rc != RC_OK) rc_kept_release(ctx);
    if (rc == RC_OK && total_kept) {  // (rc_estimate_error_rate takes them from here; rc_table_release frees them)
        ctx->counted_codes = b_allk.p;
        ctx->counted_n = total_kept;
        b_allk.p = nullptr;
    }
    if (n_kmers) *n_kmers = (int64_t)total_kept;
    return rc;
}

// n bytes from device memory of one GPU to device memory of another (or the same), queued on `st`, a stream of the destination's
// device, which is the current one: device to device, peer to peer where the GPUs can, else through the host (synchronous)
static int rc_copy_across(rc_ctx *ctx, void *dst, int dst_dev, const void *src, int src_dev, size_t n, hipStream_t st)
{
    if (n == 0) return RC_OK;
    const bool force_staged = getenv("RC_REPLICATE_STAGED") != nullptr;  // tests: the path of GPUs without peer access
    if (dst_dev == src_dev && !force_staged) {
        RC_CHECK_HIP(ctx, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, st));
        return RC_OK;
    }
    int can = 0;
    if (!force_staged && dst_dev != src_dev) {
        if (hipDeviceCanAccessPeer(&can, dst_dev, src_dev) != hipSuccess) can = 0;
        if (can) {
            const hipError_t e = hipDeviceEnablePeerAccess(src_dev, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
            (void)hipGetLastError();
        }
    }
    if (can) {
        RC_CHECK_HIP(ctx, hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, n, st));
        return RC_OK;
    }
    const size_t CH = (size_t)64 << 20;
    char *h = nullptr;
    RC_CHECK_HIP(ctx, hipHostMalloc((void **)&h, std::min(CH, n), hipHostMallocPortable));
    hipError_t e = hipSuccess;
    for (size_t at = 0; at < n && e == hipSuccess; at += CH) {
        const size_t m = std::min(CH, n - at);
        e = hipSetDevice(src_dev);
        if (e == hipSuccess) e = hipMemcpy(h, (const char *)src + at, m, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipSetDevice(dst_dev);
        if (e == hipSuccess) e = hipMemcpyAsync((char *)dst + at, h, m, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    (void)hipSetDevice(dst_dev);
    (void)hipHostFree(h);
    if (e != hipSuccess) {
        rc_set_error(ctx, "count: copy between GPUs failed: %s", hipGetErrorString(e));
        return RC_ERR_HIP;
    }
    return RC_OK;
}

// rc_count_finish for reads that are spread over n contexts, one per GPU (`rcorrector -gpus N` in one pass: a batch's bases
// are uploaded to the GPU that will correct it, and nowhere else).  The key space is cut into the P slices one GPU would
// use; slice p belongs to GPU p % n: every GPU emits that slice's keys from its own arenas and sends them to the owner,
// which sorts, run-length encodes and selects them -- so each GPU scans a 1 / n share of the reads P times and sorts a
// 1 / n share of the keys, and every occurrence crosses xGMI once.  The kept entries are put end to end in slice order on
// cs[0], where the table is built: the same entries in the same order as rc_count_finish on one GPU holding all the reads
// (the ERROR_RATE sample and the dump depend on that order).  cs[0]'s min_count / keep settings apply to all.
int rc_count_finish_sharded(rc_ctx **cs, int n, int min_count, int64_t *n_kmers)
{
    rc_ctx *c0 = cs[0];
    for (int g = 0; g < n; ++g) {
        if (!cs[g] || !cs[g]->cnt_active) {
            rc_set_error(c0, "count_finish_sharded: every context needs an open counting session (rc_table_count_begin)");
            return RC_ERR_STATE;
        }
        if (cs[g]->k != c0->k) {
            rc_set_error(c0, "count_finish_sharded: contexts must have the same k");
            return RC_ERR_ARG;
        }
        for (int h = 0; h < g; ++h)
            if (cs[h] == cs[g]) {
                rc_set_error(c0, "count_finish_sharded: a context is listed twice");
                return RC_ERR_ARG;
            }
    }
    struct release_all {
        rc_ctx **cs;
        int n;
        ~release_all()
        {
            for (int g = 0; g < n; ++g) {
                (void)hipSetDevice(cs[g]->device);
                cs[g]->cnt_active = false;
                rc_count_release(cs[g]);  // (success with cnt_keep has moved the arenas out)
            }
            (void)hipSetDevice(cs[0]->device);
        }
    } guard{cs, n};
    const int k = c0->k;
    size_t total = 0;
    for (int g = 0; g < n; ++g) total += cs[g]->cnt_total;
    size_t mem = (size_t)24 << 30;
    if (const char *e = getenv("RC_COUNT_MEM_MB")) mem = (size_t)atoll(e) << 20;
    uint32_t P = (uint32_t)((double)total * 40.0 * 1.15 / (double)mem) + 1;  // (as rc_count_finish: the entries come out in the same order)
    if (P > 64) P = 64;
    auto fail_hip = [&](hipError_t e, const char *what) {
        rc_set_error(c0, "count_finish_sharded: %s failed: %s", what, hipGetErrorString(e));
        return RC_ERR_HIP;
    };
#define RC_SH_HIP(call, what)                        \
    do {                                             \
        const hipError_t e__ = (call);               \
        if (e__ != hipSuccess) return fail_hip(e__, what); \
    } while (0)
    // histograms: occurrences per slice on every GPU
    std::vector<std::vector<unsigned long long>> hist((size_t)n, std::vector<unsigned long long>(64, 0));
    {
        std::vector<rc_dev_tmp> b_hist((size_t)n);
        for (int g = 0; g < n; ++g) {
            RC_SH_HIP(hipSetDevice(cs[g]->device), "hipSetDevice");
            RC_SH_HIP(b_hist[(size_t)g].alloc(64 * 8), "hipMalloc");
            RC_SH_HIP(hipMemsetAsync(b_hist[(size_t)g].p, 0, 64 * 8, cs[g]->stream), "hipMemsetAsync");
            for (const auto &a : cs[g]->cnt_arenas) {
                const unsigned G = (unsigned)((a.bytes + RC_PROBE_TILE - 1) / RC_PROBE_TILE);
                hipLaunchKernelGGL(k_count_scan<0>, dim3(G), dim3(RC_PROBE_THREADS), 0, cs[g]->stream, (const uint8_t *)a.p, a.bytes, k, P, 0u,
                                   b_hist[(size_t)g].as<unsigned long long>(), (uint64_t *)nullptr, (unsigned long long *)nullptr);
            }
            RC_SH_HIP(hipGetLastError(), "histogram launch");
            RC_SH_HIP(hipMemcpyAsync(hist[(size_t)g].data(), b_hist[(size_t)g].p, 64 * 8, hipMemcpyDeviceToHost, cs[g]->stream), "hipMemcpyAsync");
        }
        for (int g = 0; g < n; ++g) {
            RC_SH_HIP(hipSetDevice(cs[g]->device), "hipSetDevice");
            RC_SH_HIP(hipStreamSynchronize(cs[g]->stream), "hipStreamSynchronize");
        }
    }
    std::vector<size_t> slice_total(P, 0);
    for (uint32_t p = 0; p < P; ++p)
        for (int g = 0; g < n; ++g) slice_total[p] += (size_t)hist[(size_t)g][p];
    // per owner: the scratch of its largest slice; per GPU: a staging buffer for the keys it emits for someone else
    struct Owner {
        rc_dev_tmp pool, allk, allc;
        size_t max_slice = 0, tmp_bytes = 0, kept = 0, cap = 0;
        size_t o_keys_s = 0, o_cnt = 0, o_keep = 0, o_runs = 0, o_tmp = 0;
    };
    std::vector<Owner> own((size_t)n);
    std::vector<rc_dev_tmp> stage((size_t)n), cursor((size_t)n);
    std::vector<size_t> stage_each((size_t)n, 0);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    for (int g = 0; g < n; ++g) {
        Owner &O = own[(size_t)g];
        size_t max_emit = 0;
        for (uint32_t p = 0; p < P; ++p) {
            if ((int)(p % (uint32_t)n) == g) O.max_slice = std::max(O.max_slice, slice_total[p]);
            else max_emit = std::max(max_emit, (size_t)hist[(size_t)g][p]);
        }
        if (O.max_slice >= (1ull << 32)) {
            rc_set_error(c0, "count: a pass of %zu k-mer occurrences exceeds 2^32 (lower RC_COUNT_MEM_MB for more passes)", O.max_slice);
            return RC_ERR_ARG;
        }
        RC_SH_HIP(hipSetDevice(cs[g]->device), "hipSetDevice");
        RC_SH_HIP(cursor[(size_t)g].alloc((size_t)n * 8), "hipMalloc");  // one cursor and one staging buffer per owner of a round
        stage_each[(size_t)g] = max_emit;
        if (max_emit) RC_SH_HIP(stage[(size_t)g].alloc((size_t)n * max_emit * 8), "hipMalloc");
        if (O.max_slice == 0) continue;
        size_t ts_sort = 0, ts_rle = 0, ts_sel = 0;
        RC_SH_HIP(rocprim::radix_sort_keys(nullptr, ts_sort, (uint64_t *)nullptr, (uint64_t *)nullptr, O.max_slice, 0, 2 * k > 64 ? 64 : 2 * k, cs[g]->stream), "sort size");
        RC_SH_HIP(rocprim::run_length_encode(nullptr, ts_rle, (uint64_t *)nullptr, (unsigned int)O.max_slice, (uint64_t *)nullptr, (uint32_t *)nullptr, (size_t *)nullptr,
                                             cs[g]->stream), "rle size");
        RC_SH_HIP(rocprim::select(nullptr, ts_sel, (uint64_t *)nullptr, (uint8_t *)nullptr, (uint64_t *)nullptr, (size_t *)nullptr, O.max_slice, cs[g]->stream), "select size");
        O.tmp_bytes = std::max(ts_sort, std::max(ts_rle, ts_sel));
        O.o_keys_s = up(O.max_slice * 8);
        O.o_cnt = O.o_keys_s + up(O.max_slice * 8);
        O.o_keep = O.o_cnt + up(O.max_slice * 4);
        O.o_runs = O.o_keep + up(O.max_slice);
        O.o_tmp = O.o_runs + 256;
        RC_SH_HIP(O.pool.alloc(O.o_tmp + up(O.tmp_bytes)), "hipMalloc");
    }
    // rounds: in round r GPU o owns slice r n + o
    struct Piece {
        int owner;
        size_t at, n;
    };
    std::vector<Piece> pieces(P, Piece{0, 0, 0});  // where slice p's kept entries lie in its owner's arrays
    for (uint32_t r0 = 0; r0 < P; r0 += (uint32_t)n) {
        // every GPU emits, for every owner of this round, the slice's keys from its own arenas: its own slice straight into its
        // sort buffer, the others' into a staging buffer each -- all GPUs at once, nothing waits for the host
        auto before_of = [&](int g, uint32_t p) {  // a GPU's keys follow those of the GPUs before it
            size_t b = 0;
            for (int h = 0; h < g; ++h) b += (size_t)hist[(size_t)h][p];
            return b;
        };
        for (int g = 0; g < n; ++g) {
            RC_SH_HIP(hipSetDevice(cs[g]->device), "hipSetDevice");
            RC_SH_HIP(hipMemsetAsync(cursor[(size_t)g].p, 0, (size_t)n * 8, cs[g]->stream), "hipMemsetAsync");
            for (int o = 0; o < n; ++o) {
                const uint32_t p = r0 + (uint32_t)o;
                if (p >= P) break;
                if (hist[(size_t)g][p] == 0) continue;
                uint64_t *dst_local = g == o ? own[(size_t)o].pool.as<uint64_t>() + before_of(g, p) : stage[(size_t)g].as<uint64_t>() + (size_t)o * stage_each[(size_t)g];
                for (const auto &a : cs[g]->cnt_arenas) {
                    const unsigned G = (unsigned)((a.bytes + RC_PROBE_TILE - 1) / RC_PROBE_TILE);
                    hipLaunchKernelGGL(k_count_scan<1>, dim3(G), dim3(RC_PROBE_THREADS), 0, cs[g]->stream, (const uint8_t *)a.p, a.bytes, k, P, p,
                                       (unsigned long long *)nullptr, dst_local, cursor[(size_t)g].as<unsigned long long>() + o);
                }
                RC_SH_HIP(hipGetLastError(), "emit launch");
            }
        }
        for (int g = 0; g < n; ++g) {
            RC_SH_HIP(hipSetDevice(cs[g]->device), "hipSetDevice");
            RC_SH_HIP(hipStreamSynchronize(cs[g]->stream), "hipStreamSynchronize");
        }
        // ... and every owner fetches what the others emitted for it (the owners' streams side by side)
        for (int o = 0; o < n; ++o) {
            const uint32_t p = r0 + (uint32_t)o;
            if (p >= P) break;
            RC_SH_HIP(hipSetDevice(cs[o]->device), "hipSetDevice");
            for (int g = 0; g < n; ++g) {
                const size_t m = (size_t)hist[(size_t)g][p];
                if (g == o || m == 0) continue;
                const int rc = rc_copy_across(c0, own[(size_t)o].pool.as<uint64_t>() + before_of(g, p), cs[o]->device,
                                              stage[(size_t)g].as<uint64_t>() + (size_t)o * stage_each[(size_t)g], cs[g]->device, m * 8, cs[o]->stream);
                if (rc) return rc;
            }
        }
        for (int g = 0; g < n; ++g) {
            RC_SH_HIP(hipSetDevice(cs[g]->device), "hipSetDevice");
            RC_SH_HIP(hipStreamSynchronize(cs[g]->stream), "hipStreamSynchronize");
        }
        // every owner reduces its slice (the owners' streams run side by side; the host waits for each in turn)
        std::vector<size_t> runs((size_t)n, 0), nsel((size_t)n, 0);
        for (int phase = 0; phase < 3; ++phase)
            for (int o = 0; o < n; ++o) {
                const uint32_t p = r0 + (uint32_t)o;
                if (p >= P || slice_total[p] == 0) continue;
                Owner &O = own[(size_t)o];
                const size_t m = slice_total[p];
                char *pool = O.pool.as<char>();
                uint64_t *keys = (uint64_t *)pool, *keys_s = (uint64_t *)(pool + O.o_keys_s);
                uint32_t *cnt = (uint32_t *)(pool + O.o_cnt);
                uint8_t *keep = (uint8_t *)(pool + O.o_keep);
                size_t *d_runs = (size_t *)(pool + O.o_runs);
                void *tmp = pool + O.o_tmp;
                hipStream_t st = cs[o]->stream;
                RC_SH_HIP(hipSetDevice(cs[o]->device), "hipSetDevice");
                size_t t1 = O.tmp_bytes;
                if (phase == 0) {
                    RC_SH_HIP(rocprim::radix_sort_keys(tmp, t1, keys, keys_s, m, 0, 2 * k > 64 ? 64 : 2 * k, st), "sort");
                    t1 = O.tmp_bytes;
                    RC_SH_HIP(rocprim::run_length_encode(tmp, t1, keys_s, (unsigned int)m, keys, cnt, d_runs, st), "run lengths");
                    RC_SH_HIP(hipMemcpyAsync(&runs[(size_t)o], d_runs, sizeof(size_t), hipMemcpyDeviceToHost, st), "hipMemcpyAsync");
                } else if (phase == 1) {
                    RC_SH_HIP(hipStreamSynchronize(st), "hipStreamSynchronize");
                    if (runs[(size_t)o] == 0) continue;
                    hipLaunchKernelGGL(k_flag_keep, dim3((unsigned)((runs[(size_t)o] + 255) / 256)), dim3(256), 0, st, keys, cnt, runs[(size_t)o], min_count, keep);
                    RC_SH_HIP(rocprim::select(tmp, t1, keys, keep, keys_s, d_runs, runs[(size_t)o], st), "select");
                    RC_SH_HIP(hipMemcpyAsync(&nsel[(size_t)o], d_runs, sizeof(size_t), hipMemcpyDeviceToHost, st), "hipMemcpyAsync");
                } else {
                    RC_SH_HIP(hipStreamSynchronize(st), "hipStreamSynchronize");
                    const size_t ns = runs[(size_t)o] ? nsel[(size_t)o] : 0;
                    pieces[p] = Piece{o, O.kept, ns};
                    if (ns == 0) continue;
                    if (O.kept + ns > O.cap) {  // (sized from what this owner has kept of what it has seen, + 50 %)
                        size_t seen = 0, todo = 0;
                        for (uint32_t q = (uint32_t)o; q < P; q += (uint32_t)n) (q <= p ? seen : todo) += slice_total[q];
                        size_t want = (size_t)((double)(O.kept + ns) * (1.0 + 1.5 * (double)todo / (double)(seen ? seen : 1))) + ((size_t)1 << 16);
                        static const bool tight = getenv("RC_COUNT_TIGHT") != nullptr;
                        if (want < O.kept + ns || tight) want = O.kept + ns;
                        rc_dev_tmp nk, nc;
                        RC_SH_HIP(nk.alloc((want + 1) * 8), "hipMalloc");
                        RC_SH_HIP(nc.alloc((want + 1) * 4), "hipMalloc");
                        if (O.kept) {
                            RC_SH_HIP(hipMemcpyAsync(nk.p, O.allk.p, O.kept * 8, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
                            RC_SH_HIP(hipMemcpyAsync(nc.p, O.allc.p, O.kept * 4, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
                            RC_SH_HIP(hipStreamSynchronize(st), "hipStreamSynchronize");
                        }
                        std::swap(nk.p, O.allk.p);
                        std::swap(nc.p, O.allc.p);
                        O.cap = want;
                    }
                    RC_SH_HIP(hipMemcpyAsync(O.allk.as<uint64_t>() + O.kept, keys_s, ns * 8, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
                    uint32_t *selc = reinterpret_cast<uint32_t *>(keys_s);  // (keys_s was copied out: the stream orders the reuse)
                    RC_SH_HIP(rocprim::select(tmp, t1, cnt, keep, selc, d_runs, runs[(size_t)o], st), "select");
                    hipLaunchKernelGGL(k_u32_to_i32_clamped, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, selc, O.allc.as<int32_t>() + O.kept, ns);
                    RC_SH_HIP(hipGetLastError(), "launch");
                    O.kept += ns;
                }
            }
        for (int o = 0; o < n; ++o) {
            RC_SH_HIP(hipSetDevice(cs[o]->device), "hipSetDevice");
            RC_SH_HIP(hipStreamSynchronize(cs[o]->stream), "hipStreamSynchronize");
        }
    }
    // the kept entries, slice after slice, on cs[0]
    size_t total_kept = 0;
    for (uint32_t p = 0; p < P; ++p) total_kept += pieces[p].n;
    RC_SH_HIP(hipSetDevice(c0->device), "hipSetDevice");
    rc_dev_tmp b_allk, b_allc;
    RC_SH_HIP(b_allk.alloc((total_kept + 1) * 8), "hipMalloc");
    RC_SH_HIP(b_allc.alloc((total_kept + 1) * 4), "hipMalloc");
    {
        size_t at = 0;
        for (uint32_t p = 0; p < P; ++p) {
            const Piece &pc = pieces[p];
            if (pc.n == 0) continue;
            const Owner &O = own[(size_t)pc.owner];
            int rc = rc_copy_across(c0, b_allk.as<uint64_t>() + at, c0->device, O.allk.as<uint64_t>() + pc.at, cs[pc.owner]->device, pc.n * 8, c0->stream);
            if (!rc) rc = rc_copy_across(c0, b_allc.as<int32_t>() + at, c0->device, O.allc.as<int32_t>() + pc.at, cs[pc.owner]->device, pc.n * 4, c0->stream);
            if (rc) return rc;
            at += pc.n;
        }
        RC_SH_HIP(hipStreamSynchronize(c0->stream), "hipStreamSynchronize");
    }
    for (int g = 0; g < n; ++g) {  // scratch back to its device; the reads stay where they are for rc_submit_resident, if asked
        RC_SH_HIP(hipSetDevice(cs[g]->device), "hipSetDevice");
        own[(size_t)g].pool.reset();
        own[(size_t)g].allk.reset();
        own[(size_t)g].allc.reset();
        stage[(size_t)g].reset();
        cursor[(size_t)g].reset();
        cs[g]->cnt_active = false;
        if (c0->cnt_keep) {
            cs[g]->kept_arenas.swap(cs[g]->cnt_arenas);
            cs[g]->kept_chunks.swap(cs[g]->cnt_chunks);
            cs[g]->cnt_chunk_used = 0;
            cs[g]->cnt_total = 0;
        }
        rc_count_release(cs[g]);
    }
    RC_SH_HIP(hipSetDevice(c0->device), "hipSetDevice");
    int rc = rc_build_table_from_device_pairs(c0, b_allk.as<uint64_t>(), b_allc.as<int32_t>(), total_kept);
    if (rc != RC_OK)
        for (int g = 0; g < n; ++g) {
            (void)hipSetDevice(cs[g]->device);
            rc_kept_release(cs[g]);
        }
    (void)hipSetDevice(c0->device);
    if (rc == RC_OK && total_kept) {
        c0->counted_codes = b_allk.p;
        c0->counted_n = total_kept;
        b_allk.p = nullptr;
    }
    if (n_kmers) *n_kmers = (int64_t)total_kept;
#undef RC_SH_HIP
    return rc;
}

// ends a counting session without counting: the arenas it was given become kept arenas (rc_submit_resident), no table is built
int rc_count_park(rc_ctx *ctx)
{
    if (!ctx->cnt_active) {
        rc_set_error(ctx, "count_park: call rc_table_count_begin first");
        return RC_ERR_STATE;
    }
    ctx->cnt_active = false;
    ctx->kept_arenas.swap(ctx->cnt_arenas);
    ctx->kept_chunks.swap(ctx->cnt_chunks);
    ctx->cnt_chunk_used = 0;
    ctx->cnt_total = 0;
    rc_count_release(ctx);
    return RC_OK;
}

int rc_count_reads(rc_ctx *ctx, const uint8_t *d_seq, size_t nbytes, int min_count, int64_t *n_kmers)
{
    if (nbytes == 0 || nbytes >= (1ull << 32)) {
        rc_set_error(ctx, "count: arena must be 1..2^32-1 bytes");
        return RC_ERR_ARG;
    }
    int rc = rc_count_begin(ctx);
    if (rc == RC_OK) rc = rc_count_add(ctx, d_seq, nbytes, true);
    if (rc == RC_OK) rc = rc_count_finish(ctx, min_count, n_kmers);
    return rc;
}
