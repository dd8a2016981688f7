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

#include <rocprim/rocprim.hpp>

#include "rc_internal.h"
#include "rc_device.h"

// ---- K0: build -----------------------------------------------------------------------------
__global__ void k_home_and_index(const uint64_t *__restrict__ canon, uint32_t *__restrict__ home,
                                 uint32_t *__restrict__ idx, size_t n, uint32_t nb_home)
{
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    size_t i = n - 1 - j;  // reversed input order (see header)
    home[j] = rc_home(canon[i], nb_home);
    idx[j] = (uint32_t)i;
}

__global__ void k_slot_seed(const uint32_t *__restrict__ home_sorted, long long *__restrict__ q, size_t n)
{
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    q[j] = (long long)RC_BUCKET_SLOTS * home_sorted[j] - (long long)j;
}

__global__ void k_scatter(const uint32_t *__restrict__ home_sorted, const uint32_t *__restrict__ idx_sorted,
                          const long long *__restrict__ qmax, const uint64_t *__restrict__ canon,
                          const int32_t *__restrict__ counts, uint32_t *__restrict__ buckets, size_t n)
{
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long long p = qmax[j] + (long long)j;
    const uint32_t b = (uint32_t)(p / RC_BUCKET_SLOTS), s = (uint32_t)(p % RC_BUCKET_SLOTS);
    const uint32_t i = idx_sorted[j];
    const uint64_t key = canon[i];
    uint32_t *w = buckets + (size_t)b * RC_BUCKET_DWORDS + s * 3;
    w[0] = (uint32_t)key;
    w[1] = (uint32_t)(key >> 32);
    w[2] = (uint32_t)counts[i];
    if (s == 0 && home_sorted[j] < b) buckets[(size_t)(b - 1) * RC_BUCKET_DWORDS + (RC_BUCKET_DWORDS - 1)] = 1u;
}

int rc_build_table_from_device_pairs(rc_ctx *ctx, const uint64_t *d_canon, const int32_t *d_counts, size_t n)
{
    if (n >= (1ull << 31)) {
        rc_set_error(ctx, "table build: %zu entries exceed the 2^31 limit", n);
        return RC_ERR_ARG;
    }
    if (ctx->d_buckets && !ctx->buckets_borrowed) (void)hipFree(ctx->d_buckets);
    ctx->d_buckets = nullptr;
    ctx->buckets_borrowed = false;
    // home buckets: n / (slots * load).  Random 64-byte gathers on MI355X are request-rate bound
    // (~55 G/s, tools/microbench_gather.hip) and fall off a cliff once the table outgrows the TLB
    // reach (~2 GiB), so a dense table wins: fewer bytes => more MALL/L2 hits per probe.
    double load = ctx->table_load;
    if (!(load > 0.05 && load <= 0.95)) load = 0.50;
    uint64_t want = (uint64_t)((double)n / (RC_BUCKET_SLOTS * load)) + 1;
    if (want < 64) want = 64;
    if (want >= (1ull << 32) - 8) {
        rc_set_error(ctx, "table build: bucket count overflow");
        return RC_ERR_ARG;
    }
    uint32_t nb_home = (uint32_t)want;
    ctx->nb_home = nb_home;
    int bits = 0;
    while ((1ull << bits) < (uint64_t)nb_home) ++bits;

    long long p_last = -1;
    rc_dev_tmp b_home, b_home_s, b_idx, b_idx_s, b_q, b_qm, b_tmp;
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
        uint32_t *home = b_home.as<uint32_t>(), *home_s = b_home_s.as<uint32_t>();
        uint32_t *idx = b_idx.as<uint32_t>(), *idx_s = b_idx_s.as<uint32_t>();
        long long *q = b_q.as<long long>(), *qm = b_qm.as<long long>();
        hipLaunchKernelGGL(k_home_and_index, dim3(G), dim3(B), 0, ctx->stream, d_canon, home, idx, n, ctx->nb_home);
        RC_CHECK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_sort, home, home_s, idx, idx_s, n, 0, bits > 0 ? bits : 1, ctx->stream));
        RC_CHECK_HIP(ctx, rocprim::inclusive_scan(nullptr, tmp_scan, q, qm, n, rocprim::maximum<long long>(), ctx->stream));
        RC_CHECK_HIP(ctx, b_tmp.alloc(tmp_sort > tmp_scan ? tmp_sort : tmp_scan));
        RC_CHECK_HIP(ctx, rocprim::radix_sort_pairs(b_tmp.p, tmp_sort, home, home_s, idx, idx_s, n, 0, bits > 0 ? bits : 1, ctx->stream));
        hipLaunchKernelGGL(k_slot_seed, dim3(G), dim3(B), 0, ctx->stream, home_s, q, n);
        RC_CHECK_HIP(ctx, rocprim::inclusive_scan(b_tmp.p, tmp_scan, q, qm, n, rocprim::maximum<long long>(), ctx->stream));
        long long q_last = 0;
        RC_CHECK_HIP(ctx, hipMemcpyAsync(&q_last, qm + (n - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
        RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        p_last = q_last + (long long)(n - 1);
    }
    uint64_t need = (uint64_t)(p_last / RC_BUCKET_SLOTS) + 2;  // +1 empty bucket after the last used
    uint64_t nb_alloc = need > (uint64_t)nb_home + 1 ? need : (uint64_t)nb_home + 1;
    if (nb_alloc >= (1ull << 32)) {
        rc_set_error(ctx, "table build: bucket count overflow");
        return RC_ERR_ARG;
    }
    ctx->nb_alloc = (uint32_t)nb_alloc;
    ctx->table_bytes = (size_t)nb_alloc * RC_BUCKET_BYTES;
    RC_CHECK_HIP(ctx, hipMalloc((void **)&ctx->d_buckets, ctx->table_bytes));
    RC_CHECK_HIP(ctx, hipMemsetAsync(ctx->d_buckets, 0, ctx->table_bytes, ctx->stream));
    if (n > 0) {
        hipLaunchKernelGGL(k_scatter, dim3(G), dim3(B), 0, ctx->stream, b_home_s.as<uint32_t>(), b_idx_s.as<uint32_t>(),
                           b_qm.as<long long>(), d_canon, d_counts, ctx->d_buckets, n);
        RC_CHECK_HIP(ctx, hipGetLastError());
    }
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_entries = n;
    return RC_OK;
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
        int cnt = rc_table_lookup(T, rc_canonical(base | (uint64_t)c, k));
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

// every stored (canonical code, count) pair, in unspecified order (jf_dump writer / test support)
__global__ void k_export(const uint32_t *__restrict__ buckets, size_t nslots, uint64_t *__restrict__ codes,
                         int32_t *__restrict__ counts, unsigned long long *__restrict__ n_out, size_t cap)
{
    size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    const uint32_t *w = buckets + (s / RC_BUCKET_SLOTS) * RC_BUCKET_DWORDS + (s % RC_BUCKET_SLOTS) * 3;
    if (w[2] == 0) return;
    unsigned long long at = atomicAdd(n_out, 1ull);
    if (at < cap) {
        codes[at] = ((uint64_t)w[1] << 32) | w[0];
        counts[at] = (int32_t)w[2];
    }
}

int rc_launch_export(rc_ctx *ctx, uint64_t *d_codes, int32_t *d_counts, unsigned long long *d_n, size_t cap)
{
    const size_t nslots = (size_t)ctx->nb_alloc * RC_BUCKET_SLOTS;
    hipLaunchKernelGGL(k_export, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_buckets, nslots, d_codes, d_counts, d_n, cap);
    RC_CHECK_HIP(ctx, hipGetLastError());
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

// ---- K1: probe kernel ----------------------------------------------------------------------
// counts[a] = GetCount(k-mer starting at arena byte a) for every a whose k-window lies inside one
// read (reads are NUL-terminated inside the arena, so "inside one read" == "no NUL in the
// window").  Windows holding a non-ACGT letter give 0 without touching the table (Store.h:61-62).
// One 256-thread workgroup owns a 4 KiB tile of the arena: it stages the tile (+32 B halo) into
// LDS as 2-bit codes plus two bit masks (non-ACGT, NUL), then every lane extracts its windows
// with funnel shifts, canonicalises with bit-reverse and probes one 64-byte bucket.
#define RC_PROBE_TILE 4096
#define RC_PROBE_THREADS 256

__device__ __forceinline__ void rc_pack16(const uint4 v, uint32_t &code, uint32_t &inv, uint32_t &nul)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    code = 0;
    inv = 0;
    nul = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t c = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        uint32_t b = 3, bad = 1;
        if (c == 'A') { b = 0; bad = 0; }
        if (c == 'C') { b = 1; bad = 0; }
        if (c == 'G') { b = 2; bad = 0; }
        if (c == 'T') { b = 3; bad = 0; }
        code |= b << (30 - 2 * j);
        inv |= bad << (15 - j);
        nul |= (c == 0 ? 1u : 0u) << (15 - j);
    }
}

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
            cnt = rc_table_lookup(T, rc_canonical(code, k));
        }
        __builtin_nontemporal_store(cnt, &counts[g]);  // streamed once: keep it out of the caches the table lives in
    }
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
    hipLaunchKernelGGL(k_probe, dim3(G), dim3(RC_PROBE_THREADS), 0, ctx->stream, rc_view(ctx), d_seq, nbytes, ctx->k, d_counts);
    rc_timer_end(ctx, RC_T_PROBE);
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}

// ---- k-mer counting from reads (replaces jellyfish bc/count/dump for in-HBM use) -----------------
// every valid k-mer window of the arena -> canonical code (else the all-ones sentinel, which is
// never a canonical code), radix sort, run-length encode, keep count >= min_count, build.
__global__ __launch_bounds__(RC_PROBE_THREADS) void k_emit_kmers(const uint8_t *__restrict__ seq, size_t nbytes, int k,
                                                                 uint64_t *__restrict__ out)
{
    __shared__ uint32_t s_code[RC_PROBE_TILE / 16 + 4];
    __shared__ uint16_t s_inv[RC_PROBE_TILE / 16 + 4];
    const size_t tile0 = (size_t)blockIdx.x * RC_PROBE_TILE;
    const int t = threadIdx.x;
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
    for (int a = t; a < RC_PROBE_TILE; a += RC_PROBE_THREADS) {
        const size_t g = tile0 + (size_t)a;
        if (g >= nbytes) break;
        uint64_t key = ~0ull;
        if (g + (size_t)k <= nbytes) {
            const int mw = a >> 5, ms = a & 31;
            const uint64_t invw = (((uint64_t)m_inv[mw] << 32) | m_inv[mw + 1]) << ms;
            if (!(invw >> (64 - k))) {
                const int cw = a >> 4, cs = 2 * (a & 15);
                uint64_t x = ((uint64_t)s_code[cw] << 32) | s_code[cw + 1];
                if (cs) x = (x << cs) | ((uint64_t)s_code[cw + 2] >> (32 - cs));
                key = rc_canonical(x >> (64 - 2 * k), k);
            }
        }
        out[g] = key;
    }
}

__global__ void k_flag_keep(const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ cnt, size_t n, int min_count,
                            uint8_t *__restrict__ keep)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keep[i] = (uniq[i] != ~0ull && cnt[i] >= (uint32_t)min_count) ? 1 : 0;
}

int rc_count_reads(rc_ctx *ctx, const uint8_t *d_seq, size_t nbytes, int min_count, int64_t *n_kmers)
{
    if (nbytes == 0 || nbytes >= (1ull << 32)) {
        rc_set_error(ctx, "count: arena must be 1..2^32-1 bytes");
        return RC_ERR_ARG;
    }
    const int k = ctx->k;
    rc_dev_tmp b_keys, b_keys_s, b_cnt, b_keep, b_selc, b_runs, b_tmp;
    size_t t_sort = 0, t_rle = 0, t_sel = 0;
    RC_CHECK_HIP(ctx, b_keys.alloc(nbytes * 8));
    RC_CHECK_HIP(ctx, b_keys_s.alloc(nbytes * 8));
    RC_CHECK_HIP(ctx, b_runs.alloc(sizeof(size_t) * 2));
    uint64_t *keys = b_keys.as<uint64_t>(), *keys_s = b_keys_s.as<uint64_t>();
    size_t *d_runs = b_runs.as<size_t>(), *d_nsel = d_runs + 1;
    const unsigned G = (unsigned)((nbytes + RC_PROBE_TILE - 1) / RC_PROBE_TILE);
    hipLaunchKernelGGL(k_emit_kmers, dim3(G), dim3(RC_PROBE_THREADS), 0, ctx->stream, d_seq, nbytes, k, keys);
    RC_CHECK_HIP(ctx, hipGetLastError());
    RC_CHECK_HIP(ctx, rocprim::radix_sort_keys(nullptr, t_sort, keys, keys_s, nbytes, 0, 64, ctx->stream));
    RC_CHECK_HIP(ctx, b_tmp.alloc(t_sort));
    RC_CHECK_HIP(ctx, rocprim::radix_sort_keys(b_tmp.p, t_sort, keys, keys_s, nbytes, 0, 64, ctx->stream));
    // run-length encode; unique keys reuse `keys`
    uint64_t *uniq = keys;
    RC_CHECK_HIP(ctx, b_cnt.alloc(nbytes * 4));
    uint32_t *cnt = b_cnt.as<uint32_t>();
    RC_CHECK_HIP(ctx, rocprim::run_length_encode(nullptr, t_rle, keys_s, (unsigned int)nbytes, uniq, cnt, d_runs, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RC_CHECK_HIP(ctx, b_tmp.alloc(t_rle));
    RC_CHECK_HIP(ctx, rocprim::run_length_encode(b_tmp.p, t_rle, keys_s, (unsigned int)nbytes, uniq, cnt, d_runs, ctx->stream));
    size_t runs = 0;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(&runs, d_runs, sizeof(size_t), hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // keep count >= min_count (and drop the sentinel run)
    RC_CHECK_HIP(ctx, b_keep.alloc(runs + 1));
    uint8_t *keep = b_keep.as<uint8_t>();
    hipLaunchKernelGGL(k_flag_keep, dim3((unsigned)((runs + 255) / 256)), dim3(256), 0, ctx->stream, uniq, cnt, runs, min_count, keep);
    uint64_t *sel_k = keys_s;  // sorted keys no longer needed
    RC_CHECK_HIP(ctx, b_selc.alloc((runs + 1) * 4));
    uint32_t *sel_c = b_selc.as<uint32_t>();
    RC_CHECK_HIP(ctx, rocprim::select(nullptr, t_sel, uniq, keep, sel_k, d_nsel, runs, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RC_CHECK_HIP(ctx, b_tmp.alloc(t_sel));
    RC_CHECK_HIP(ctx, rocprim::select(b_tmp.p, t_sel, uniq, keep, sel_k, d_nsel, runs, ctx->stream));
    RC_CHECK_HIP(ctx, rocprim::select(b_tmp.p, t_sel, cnt, keep, sel_c, d_nsel, runs, ctx->stream));
    size_t nsel = 0;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(&nsel, d_nsel, sizeof(size_t), hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    b_tmp.reset();
    b_keep.reset();
    b_cnt.reset();
    b_keys.reset();  // `uniq`: consumed by the select above
    int rc = rc_build_table_from_device_pairs(ctx, sel_k, reinterpret_cast<const int32_t *>(sel_c), nsel);
    if (n_kmers) *n_kmers = (int64_t)nsel;
    return rc;
}
