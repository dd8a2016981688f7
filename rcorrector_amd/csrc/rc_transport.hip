// rc_transport.hip -- the packed device boundary ("per batch only packed reads go down and
// (fix list | corrected seq, ret, l, m, h) come back"; it replaces the per-batch traffic of main.cpp:479-516 -- the
// reference hands its workers pointers into host memory, a GPU has to move the bytes).
//
// Up:   2 bits per base (16 per 32-bit word, arena byte p at bits 30 - 2 (p & 15) of word p >> 4), the letters that
//       are not A / C / G / T as a (position, letter) list, one quality bit per base (rc_pack_quality_bits), the read
//       offsets: 0.25 + 0.125 bytes per base + 4 per read instead of 2 per base -- 61 instead of 306 bytes for a
//       150-base read.
// Down: ret / l / m / h and the substitutions the correction made as (arena position, letter) pairs -- about 0.75 per
//       read at 0.5 % errors: 20 bytes instead of 167.
// Reads that are in HBM already -- the arenas the k-mer counter kept (rc_table_count_keep) -- do not travel at all
// (rc_submit_resident): the batch's ranges are copied device to device into the byte arena the kernels correct in place,
// and the fix list is the byte difference between that arena and the kept one (k_fix_list_bytes).
// On the device the packed arena is expanded into the byte arena the kernels read (k_unpack_bases: 1 byte written
// per base, ~1 ms for 25 M reads -- the kernels themselves are unchanged) and the fix list is the difference between
// the corrected arena and the packed one (k_fix_list).
#include "rc_internal.h"

// 16 arena bytes per thread: letters of the packed codes; NULs and the other letters are put in afterwards
__global__ __launch_bounds__(256) void k_unpack_bases(const uint32_t *__restrict__ packed, size_t n_words, uint8_t *__restrict__ seq)
{
    for (size_t w = (size_t)blockIdx.x * 256u + threadIdx.x; w < n_words; w += (size_t)gridDim.x * 256u) {
        const uint32_t v = packed[w];
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t x = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t c = (v >> (30 - 2 * (4 * q + j))) & 3u;
                // A C G T = 0x41 0x43 0x47 0x54
                const uint32_t ch = c == 0 ? 0x41u : (c == 1 ? 0x43u : (c == 2 ? 0x47u : 0x54u));
                x |= ch << (8 * j);
            }
            o[q] = x;
        }
        *reinterpret_cast<uint4 *>(seq + 16 * w) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// the NUL behind every read
__global__ __launch_bounds__(256) void k_put_nuls(const uint32_t *__restrict__ off, uint32_t n, uint8_t *__restrict__ seq)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) seq[off[i + 1] - 1] = 0;
}

// the letters outside ACGT
__global__ __launch_bounds__(256) void k_put_exceptions(const uint32_t *__restrict__ pos, const uint8_t *__restrict__ chr, uint32_t n, uint8_t *__restrict__ seq)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) seq[pos[i]] = chr[i];
}

__device__ __forceinline__ void rc_emit_fixes(bool has, uint32_t pos, uint32_t chr, uint32_t *__restrict__ n_fix, uint32_t cap,
                                              uint32_t *__restrict__ fix_pos, uint8_t *__restrict__ fix_chr)
{
    const uint64_t m = __ballot(has);
    if (!m) return;
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(n_fix, (uint32_t)__popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1, 64);
    if (has) {
        const uint32_t j = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (j < cap) {
            fix_pos[j] = pos;
            fix_chr[j] = (uint8_t)chr;
        }
    }
}

// Substitutions = the bytes of the corrected arena that are one of ACGT and differ from the letter of the packed code.
// A byte that is still a NUL or a letter outside ACGT is no fix; a letter outside ACGT that the correction replaced by
// C, G or T differs from its packed code (0 = 'A') and is found here, one replaced by 'A' is found by k_fix_exceptions.
// n_fix counts every fix, the arrays hold the first `cap`.
// A workgroup takes RC_FIX_WORDS words (16 KB of arena) at a time: every thread compares its words, the workgroup scans
// the counts and claims its stretch of the list with ONE atomic (an atomic per wave that holds a fix -- 0.5 M of them
// on one address for a 2 M-read batch -- took 3.9 ms, as long as the hash-probe kernel of the same batch).
#define RC_FIX_WPT 4                    // words per thread
#define RC_FIX_WORDS (256 * RC_FIX_WPT)
__global__ __launch_bounds__(256) void k_fix_list(const uint32_t *__restrict__ packed, size_t n_words, const uint8_t *__restrict__ seq,
                                                  uint32_t *__restrict__ n_fix, uint32_t cap, uint32_t *__restrict__ fix_pos, uint8_t *__restrict__ fix_chr)
{
    __shared__ uint32_t s_wave[4], s_base;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    for (size_t w0 = (size_t)blockIdx.x * RC_FIX_WORDS; w0 < n_words; w0 += (size_t)gridDim.x * RC_FIX_WORDS) {
        // thread t takes words w0 + q * 256 + t (coalesced 16-byte loads); its fixes: 2 bits of letter at 4 j of `let`
        uint32_t diff[RC_FIX_WPT];
        uint64_t let[RC_FIX_WPT];
        uint32_t mine = 0;
#pragma unroll
        for (int q = 0; q < RC_FIX_WPT; ++q) {
            const size_t w = w0 + (size_t)q * 256u + t;
            uint32_t v = 0;
            uint4 sv = make_uint4(0, 0, 0, 0);
            if (w < n_words) {
                v = packed[w];
                sv = *reinterpret_cast<const uint4 *>(seq + 16 * w);
            }
            const uint32_t sw[4] = {sv.x, sv.y, sv.z, sv.w};
            uint32_t d = 0;
            uint64_t lt = 0;
            // most words hold no fix: the letters of the packed codes (as k_unpack_bases writes them) against the arena's,
            // four bytes at a time -- a byte that differs and is not a NUL is looked at one by one below
            uint32_t cand = 0;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                uint32_t e = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t c = (v >> (30 - 2 * (4 * q4 + j))) & 3u;
                    e |= (c == 0 ? 0x41u : (c == 1 ? 0x43u : (c == 2 ? 0x47u : 0x54u))) << (8 * j);
                }
                const uint32_t x = sw[q4] ^ e;
                const uint32_t nz = ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x;                    // bit 7 of a byte: x's byte is not 0
                const uint32_t nn = ((sw[q4] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | sw[q4];          // ... the arena's byte is not NUL
                cand |= nz & nn & 0x80808080u;
            }
            if (cand) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t ch = (sw[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    const uint32_t c = (v >> (30 - 2 * j)) & 3u;
                    const int code = ch == 0x41u ? 0 : (ch == 0x43u ? 1 : (ch == 0x47u ? 2 : (ch == 0x54u ? 3 : -1)));
                    d |= (code >= 0 && (uint32_t)code != c ? 1u : 0u) << j;
                    lt |= (uint64_t)(code & 3) << (4 * j);
                }
            }
            diff[q] = d;
            let[q] = lt;
            mine += (uint32_t)__popc(d);
        }
        uint32_t inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(inc, o, 64);
            inc += lane >= (uint32_t)o ? y : 0u;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            before += (uint32_t)q < wave ? s_wave[q] : 0u;
            total += s_wave[q];
        }
        if (total) {  // (uniform)
            if (t == 0) s_base = atomicAdd(n_fix, total);
            __syncthreads();
            uint32_t j0 = s_base + before + inc - mine;
#pragma unroll
            for (int q = 0; q < RC_FIX_WPT; ++q) {
                uint32_t d = diff[q];
                const uint32_t p0 = (uint32_t)(16 * (w0 + (size_t)q * 256u + t));
                while (d) {
                    const int j = __ffs((int)d) - 1;
                    d &= d - 1;
                    if (j0 < cap) {
                        fix_pos[j0] = p0 + (uint32_t)j;
                        fix_chr[j0] = (uint8_t)("ACGT"[(let[q] >> (4 * j)) & 3u]);
                    }
                    ++j0;
                }
            }
        }
        __syncthreads();
    }
}

// 16 bytes from any address as two aligned 16-byte loads and a byte funnel (the compiler turns a 16-byte copy from an
// address it cannot prove aligned into sixteen byte loads).  Reads up to 31 bytes past p & ~15: the kept arenas carry
// 64 bytes of slack.  The shift is the same for every thread of a launch, so the switch is a uniform branch.
__device__ __forceinline__ uint4 rc_load16_any(const uint8_t *p)
{
    const uintptr_t a = (uintptr_t)p;
    const uint4 *q = reinterpret_cast<const uint4 *>(a & ~(uintptr_t)15);
    const uint32_t sh = (uint32_t)(a & 15u);
    const uint4 lo = q[0];
    if (sh == 0) return lo;
    const uint4 hi = q[1];
    const uint32_t r = sh & 3u;
    switch (sh >> 2) {
    case 0:
        return make_uint4(__builtin_amdgcn_alignbyte(lo.y, lo.x, r), __builtin_amdgcn_alignbyte(lo.z, lo.y, r), __builtin_amdgcn_alignbyte(lo.w, lo.z, r),
                          __builtin_amdgcn_alignbyte(hi.x, lo.w, r));
    case 1:
        return make_uint4(__builtin_amdgcn_alignbyte(lo.z, lo.y, r), __builtin_amdgcn_alignbyte(lo.w, lo.z, r), __builtin_amdgcn_alignbyte(hi.x, lo.w, r),
                          __builtin_amdgcn_alignbyte(hi.y, hi.x, r));
    case 2:
        return make_uint4(__builtin_amdgcn_alignbyte(lo.w, lo.z, r), __builtin_amdgcn_alignbyte(hi.x, lo.w, r), __builtin_amdgcn_alignbyte(hi.y, hi.x, r),
                          __builtin_amdgcn_alignbyte(hi.z, hi.y, r));
    default:
        return make_uint4(__builtin_amdgcn_alignbyte(hi.x, lo.w, r), __builtin_amdgcn_alignbyte(hi.y, hi.x, r), __builtin_amdgcn_alignbyte(hi.z, hi.y, r),
                          __builtin_amdgcn_alignbyte(hi.w, hi.z, r));
    }
}

// The same against the uncorrected bytes themselves: the corrected arena `seq` of na + nb bytes was copied from orig_a
// (its first na bytes) and orig_b (the rest, the second mates of a paired batch; nb may be 0).  A correction only ever
// writes one of ACGT over a different byte (ErrorCorrection.cpp:1468-1479), so every byte that differs is a fix.  16 bytes
// of `seq` per thread and step; the bytes they were copied from sit at any alignment (rc_load16_any).
__global__ __launch_bounds__(256) void k_fix_list_bytes(const uint8_t *__restrict__ orig_a, size_t na, const uint8_t *__restrict__ orig_b, size_t nb,
                                                        const uint8_t *__restrict__ seq, uint32_t *__restrict__ n_fix, uint32_t cap,
                                                        uint32_t *__restrict__ fix_pos, uint8_t *__restrict__ fix_chr)
{
    __shared__ uint32_t s_wave[4], s_base;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const size_t nbytes = na + nb, n_words = (nbytes + 15) / 16;
    for (size_t w0 = (size_t)blockIdx.x * RC_FIX_WORDS; w0 < n_words; w0 += (size_t)gridDim.x * RC_FIX_WORDS) {
        uint32_t diff[RC_FIX_WPT];
        uint4 now[RC_FIX_WPT];
        uint32_t mine = 0;
#pragma unroll
        for (int q = 0; q < RC_FIX_WPT; ++q) {
            const size_t w = w0 + (size_t)q * 256u + t, p = 16 * w;
            uint4 sv = make_uint4(0, 0, 0, 0), ov = make_uint4(0, 0, 0, 0);
            if (p + 16 <= nbytes) {
                sv = *reinterpret_cast<const uint4 *>(seq + p);
                if (p + 16 <= na) {
                    ov = rc_load16_any(orig_a + p);
                } else if (p >= na) {
                    ov = rc_load16_any(orig_b + (p - na));
                } else {
                    uint8_t o[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) o[j] = p + j < na ? orig_a[p + j] : orig_b[p + j - na];
                    __builtin_memcpy(&ov, o, 16);
                }
            } else if (p < nbytes) {  // the arena's last, partial word
                uint8_t o[16], c[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const size_t a = p + j;
                    c[j] = a < nbytes ? seq[a] : (uint8_t)0;
                    o[j] = a < nbytes ? (a < na ? orig_a[a] : orig_b[a - na]) : (uint8_t)0;
                }
                __builtin_memcpy(&sv, c, 16);
                __builtin_memcpy(&ov, o, 16);
            }
            const uint32_t sw[4] = {sv.x, sv.y, sv.z, sv.w}, ow[4] = {ov.x, ov.y, ov.z, ov.w};
            uint32_t d = 0;
            if ((sw[0] ^ ow[0]) | (sw[1] ^ ow[1]) | (sw[2] ^ ow[2]) | (sw[3] ^ ow[3])) {  // (most words hold no fix)
#pragma unroll
                for (int j = 0; j < 16; ++j) d |= ((((sw[j >> 2] ^ ow[j >> 2]) >> (8 * (j & 3))) & 0xffu) ? 1u : 0u) << j;
            }
            diff[q] = d;
            now[q] = sv;
            mine += (uint32_t)__popc(d);
        }
        uint32_t inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(inc, o, 64);
            inc += lane >= (uint32_t)o ? y : 0u;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            before += (uint32_t)q < wave ? s_wave[q] : 0u;
            total += s_wave[q];
        }
        if (total) {  // (uniform)
            if (t == 0) s_base = atomicAdd(n_fix, total);
            __syncthreads();
            uint32_t j0 = s_base + before + inc - mine;
#pragma unroll
            for (int q = 0; q < RC_FIX_WPT; ++q) {
                uint32_t d = diff[q];
                const uint32_t p0 = (uint32_t)(16 * (w0 + (size_t)q * 256u + t));
                const uint32_t sw[4] = {now[q].x, now[q].y, now[q].z, now[q].w};
                while (d) {
                    const int j = __ffs((int)d) - 1;
                    d &= d - 1;
                    if (j0 < cap) {
                        fix_pos[j0] = p0 + (uint32_t)j;
                        // (selected without indexing the array at run time: that would put it into scratch memory)
                        const uint32_t wsel = j < 8 ? (j < 4 ? sw[0] : sw[1]) : (j < 12 ? sw[2] : sw[3]);
                        fix_chr[j0] = (uint8_t)((wsel >> (8 * (j & 3))) & 0xffu);
                    }
                    ++j0;
                }
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_fix_exceptions(const uint32_t *__restrict__ pos, uint32_t n, const uint8_t *__restrict__ seq,
                                                        uint32_t *__restrict__ n_fix, uint32_t cap, uint32_t *__restrict__ fix_pos, uint8_t *__restrict__ fix_chr)
{
    const uint32_t n_end = (n + 255u) & ~255u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_end; i += gridDim.x * 256u) {
        const bool has = i < n && seq[pos[i]] == 0x41u;
        rc_emit_fixes(has, i < n ? pos[i] : 0u, 0x41u, n_fix, cap, fix_pos, fix_chr);
    }
}

int rc_launch_unpack(rc_ctx *ctx, const uint32_t *d_packed, size_t nbytes, const uint32_t *d_off, uint32_t n_reads, const uint32_t *d_exc_pos,
                     const uint8_t *d_exc_chr, uint32_t n_exc, uint8_t *d_seq)
{
    const size_t n_words = (nbytes + 15) / 16;
    if (n_words) {
        size_t g = (n_words + 255) / 256;
        if (g > (size_t)ctx->n_cu * 64) g = (size_t)ctx->n_cu * 64;
        hipLaunchKernelGGL(k_unpack_bases, dim3((unsigned)g), dim3(256), 0, ctx->stream, d_packed, n_words, d_seq);
    }
    if (n_reads) hipLaunchKernelGGL(k_put_nuls, dim3((n_reads + 255) / 256), dim3(256), 0, ctx->stream, d_off, n_reads, d_seq);
    if (n_exc) hipLaunchKernelGGL(k_put_exceptions, dim3((n_exc + 255) / 256), dim3(256), 0, ctx->stream, d_exc_pos, d_exc_chr, n_exc, d_seq);
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}

int rc_launch_fix_list_bytes(rc_ctx *ctx, const uint8_t *d_orig_a, size_t bytes_a, const uint8_t *d_orig_b, size_t bytes_b, const uint8_t *d_seq,
                              uint32_t *d_n_fix, uint32_t cap, uint32_t *d_fix_pos, uint8_t *d_fix_chr)
{
    RC_CHECK_HIP(ctx, hipMemsetAsync(d_n_fix, 0, 4, ctx->stream));
    const size_t n_words = (bytes_a + bytes_b + 15) / 16;
    if (n_words) {
        size_t g = (n_words + RC_FIX_WORDS - 1) / RC_FIX_WORDS;
        if (g > (size_t)ctx->n_cu * 32) g = (size_t)ctx->n_cu * 32;
        hipLaunchKernelGGL(k_fix_list_bytes, dim3((unsigned)g), dim3(256), 0, ctx->stream, d_orig_a, bytes_a, d_orig_b, bytes_b, d_seq, d_n_fix, cap, d_fix_pos,
                           d_fix_chr);
    }
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}

int rc_launch_fix_list(rc_ctx *ctx, const uint32_t *d_packed, size_t nbytes, const uint8_t *d_seq, const uint32_t *d_exc_pos, uint32_t n_exc,
                       uint32_t *d_n_fix, uint32_t cap, uint32_t *d_fix_pos, uint8_t *d_fix_chr)
{
    RC_CHECK_HIP(ctx, hipMemsetAsync(d_n_fix, 0, 4, ctx->stream));
    const size_t n_words = (nbytes + 15) / 16;
    if (n_words) {
        size_t g = (n_words + RC_FIX_WORDS - 1) / RC_FIX_WORDS;
        if (g > (size_t)ctx->n_cu * 32) g = (size_t)ctx->n_cu * 32;
        hipLaunchKernelGGL(k_fix_list, dim3((unsigned)g), dim3(256), 0, ctx->stream, d_packed, n_words, d_seq, d_n_fix, cap, d_fix_pos, d_fix_chr);
    }
    if (n_exc) {
        unsigned g = (n_exc + 255) / 256;
        if (g > (unsigned)ctx->n_cu * 16u) g = (unsigned)ctx->n_cu * 16u;
        hipLaunchKernelGGL(k_fix_exceptions, dim3(g), dim3(256), 0, ctx->stream, d_exc_pos, n_exc, d_seq, d_n_fix, cap, d_fix_pos, d_fix_chr);
    }
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}
