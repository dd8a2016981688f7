// rc_common.h -- shared scalar primitives of the MI355X correction path: 2-bit rolling k-mer
// code with the reference's one-slot invalid tracker, canonical form, bucket hash, table bucket
// layout, and GetBound in IEEE double.
//
// Reference behaviour restated (paths relative to /root/reference):
//   KmerCode.h:14-27,58-71, KmerCode.cpp:7-42   rolling code / canonical
//   Store.h:51-66                               Put / GetCount semantics
//   ErrorCorrection.cpp:139-142                 GetBound
//
// This header is compiled by hipcc for gfx950 (product) and by g++ for the lane-serial
// simulation harness under tests/hostsim (a test-only aid: there is no GPU in the build
// container, so the wave-uniform control flow of the search kernel is first checked against the
// oracle on the CPU).  RC_HD is the only portability hook.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RC_HD __host__ __device__ __forceinline__
#else
#define RC_HD inline
#endif

#define RC_MAX_READ_LENGTH 1024  // utils.h:7 (reads hold <=1023 bases)
#define RC_TRACE_WORDS 36         // int32 words per recorded iteration (rcorrector_amd.h: RC_TRACE_ITER_WORDS)
#define RC_MAX_TRIAL 1025        // ErrorCorrection.cpp:7
#define RC_INF 1000000000        // utils.h:10
#define RC_INT_MIN (-2147483647 - 1)

// ---- table buckets: 32 B (round 6; 64 B until round 5), two slot layouts ---------------------
// Why 32: a lane reads a bucket 16 bytes at a time, and each of those loads is a pass of 64 scattered addresses through the
// vector L1's address / tag stage -- what the probe kernels are bound by as much as by anything (profiles/
// r6_fused_where_the_time_goes.txt).  Two loads a probe instead of four: config 1 / 2 / 3 / 4 fused kernel 11.0 -> 10.0, 42.3 ->
// 38.2, 62.8 -> 56.6, 396 -> 375 ms, once the k-mers that must look at a second bucket (2.4 % at load 0.4, four slots a bucket;
// 0.6 % with eight) are finished apart from the main loop (rc_table_lookup_o: `more`).  Half a sector: two buckets share the
// 64 bytes HBM moves at least; the L2 fetches 128-byte lines either way.  -DRC_BUCKET_DWORDS=16 / 4 build 64- / 16-byte buckets.
// WIDE   (layout 0): (RC_BUCKET_DWORDS - 1) / 3 x {key_lo, key_hi, count} + 1 meta dword.  count == 0 marks an empty slot
//   (stored counts are >= 2 by construction, main.cpp:299); meta bit0 = "some key whose home is <=
//   this bucket lives in a later bucket" (probe goes on).  Any k, any count.
// PACKED (layout 1): RC_BUCKET_DWORDS / 2 x {rem, word}.  The canonical code goes through a bijection of the 2k-bit
//   key space; the home bucket is the top of (mixed * nb_home) and `rem` the next 32 bits of that
//   product, which together identify the code as long as nb_home >= 2^(2k-32) (rc_packed_addr).
//   word = count (27 bits) | displacement bucket - home (4 bits, 0..14; 15 = empty slot) | bit 31: the
//   continue flag, kept in the bucket's last slot.  A third less memory per k-mer: more of the table within the reach
//   of the TLB and the Infinity Cache.  Chosen per table when k, the counts and the placement allow.
//   Where nb_home < 2^(2k-32) (k >= 28 for tables of ordinary size) the remainder takes `ext` more
//   bits, the next ones of the same product, at the top of the count field: word = count (27-ext
//   bits) | remainder bits 32..32+ext (ext bits) | displacement | flag, for counts below 2^(27-ext).
#ifndef RC_BUCKET_DWORDS
#define RC_BUCKET_DWORDS 8
#endif
#define RC_BUCKET_BYTES (4 * RC_BUCKET_DWORDS)
#define RC_WIDE_SLOTS ((RC_BUCKET_DWORDS - 1) / 3)
#define RC_PACKED_SLOTS (RC_BUCKET_DWORDS / 2)
#define RC_PACKED_COUNT_MASK 0x07FFFFFFu
#define RC_PACKED_MAX_DISP 14            // displacement 15 marks an empty slot
#define RC_PACKED_EMPTY_WORD 0x78000000u

struct rc_table_view {
    const uint32_t *buckets;  // nbuckets_alloc * 16 dwords, 64-B aligned
    uint32_t nb_home;         // number of home buckets (any value)
    uint32_t nbuckets_alloc;  // home buckets + slack (no wrap-around)
    int layout;               // 0 wide, 1 packed
    int k;
    int ext;                  // PACKED: remainder bits beyond 32, kept above the count (0 .. RC_PACKED_MAX_EXT)
    // absence filter of a large PACKED table (nullptr: none): filter_words 32-bit words behind the bucket array.  The word
    // of a key = the top of (its mixed code * filter_words), i.e. next to its home bucket in key order; three bits
    // in it, taken from the key's remainder, are set.  A probe whose bits are not all set is a miss that never touches
    // the bucket array -- which, beyond the reach of the TLB, is what a miss costs (rc_table.hip: build_attempt).
    const uint32_t *filter;
    uint32_t filter_words;
    // 0: as above.  1 ("core", round 5): the word of a k-mer is chosen by its first k - 1 bases in ONE of its two orientations
    // (rc_filter_core_addr), and every entry is entered under both: the four k-mers a search node asks about -- one (k-1)-mer
    // extended by A, C, G, T (ErrorCorrection.cpp:286-301 / :523-538) -- then share a word, i.e. one request instead of
    // four, when the caller names the orientation in which the varying base comes last.  Any orientation gives the right
    // answer; 16 bits per entry instead of 10 (two insertions).
    int filter_kind;
    // 1: every lookup goes through the filter (tables beyond the TLB's reach: a miss costs a page walk).  0: only the
    // search's lookups do (rc_table_lookup_o<EXT, true>: k_correct) -- most of them are substitution candidates that do not
    // exist, four to a word -- while the probe kernels, nine tenths of whose k-mers are in the table, go straight to the
    // buckets (a filter word in front of every probe cost them 3 %).
    int filter_all;
};
// bits of a key in its filter word (three of 32, from 15 bits of the remainder)
RC_HD uint32_t rc_filter_mask(uint32_t rem)
{
    return (1u << (rem & 31u)) | (1u << ((rem >> 5) & 31u)) | (1u << ((rem >> 10) & 31u));
}
// kind 1: word and bits of the k-mer whose 2k-bit code, in the orientation the caller chose, is f
RC_HD void rc_filter_core_addr(uint64_t f, uint32_t words, uint32_t *word, uint32_t *mask)
{
    const uint64_t core = f >> 2;
    uint32_t h = (uint32_t)core * 0x9E3779B1u + (uint32_t)(core >> 32) * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 13;
#if defined(__HIP_DEVICE_COMPILE__)
    *word = __umulhi(h, words);
#else
    *word = (uint32_t)(((uint64_t)h * words) >> 32);
#endif
    const uint32_t r = (h ^ (((uint32_t)f & 3u) * 0x9E3779B1u)) * 0x297A2D39u;
    *mask = rc_filter_mask(r >> 17);
}
#define RC_PACKED_MAX_EXT 8   // counts keep at least 19 bits
// A count that does not fit the count field is stored as "all ones" and kept in full in a small
// array IN FRONT of the bucket array (same allocation, so a table is still one pointer): the
// RC_TABLE_PREFIX_BYTES before `buckets` hold up to RC_PACKED_OVF_MAX entries {code_lo, code_hi,
// count, 0}, ascending by code, from the start of the prefix, and their number in the dword 64 bytes
// before `buckets`.  (rRNA k-mers of a deep data set, with k >= 28: the alternative is the WIDE layout
// for the whole table, i.e. a table outside the reach of the TLB.)
#define RC_TABLE_PREFIX_BYTES 65536
#define RC_PACKED_OVF_MAX 4000

RC_HD int rc_layout_slots(int layout) { return layout ? RC_PACKED_SLOTS : RC_WIDE_SLOTS; }

RC_HD uint64_t rc_kmer_mask(int k) { return k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull); }

RC_HD uint64_t rc_brev64(uint64_t x)
{
#if defined(__clang__)
    return __builtin_bitreverse64(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
    return (x >> 32) | (x << 32);
#endif
}

// ---- 16 arena bytes -> 2-bit codes + letter masks, four bytes per operation (the fused probe kernel's staging) ----------
// rc_pack16m: the 2-bit codes of 16 bytes (first byte in the most significant bits, A C G T = 0 1 2 3, anything else -- the
// NUL behind a read and the padding between reads included -- as 3) and three 16-bit masks, bit j = byte j: is an A, is a T,
// is none of ACGT.  Bit arrays made of the masks (bit p % 32 of word p / 32 = arena byte p) give a k-mer window's "has a
// letter outside ACGT" as one funnel shift (k <= 32) and a read's letter masks (rc_quarter.h: rcq_lds_masks) as five.
// Per 32-bit word of four letters: ((c >> 1) ^ (c >> 2)) & 3 is the code of an ACGT letter; the letter that code stands for is
// 0x41 + 2 a + 6 b + 11 ab (a, b = the code's bits), and a byte that differs from it is not one of ACGT; the four 2-bit
// fields (and the four flags) are gathered into a byte by a multiply whose partial products do not overlap
// (rc_correct_core.h: rc_pack_read uses the same one).  tests/hostmath/pack16m.cpp: every byte value in every position.
RC_HD uint32_t rc_compress_even16(uint32_t x)  // bits 0, 2, 4, .. 30 of x -> bits 0 .. 15
{
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}
RC_HD uint32_t rc_brev32(uint32_t x)
{
#if defined(__clang__)
    return __builtin_bitreverse32(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
#endif
}
RC_HD void rc_pack16m(const uint32_t (&w)[4], uint32_t &code, uint32_t &am, uint32_t &tm, uint32_t &bad)
{
    uint32_t cw = 0, bw = 0;  // codes / "not ACGT" flags (low bit of the field), 2 bits per byte, first byte in the top bits
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t x = w[q];
        uint32_t c2 = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
        const uint32_t a = c2 & 0x01010101u, b = (c2 >> 1) & 0x01010101u, ab = a & b;
        const uint32_t e = 0x41414141u + (a << 1) + (b << 2) + (b << 1) + (ab << 3) + (ab << 1) + ab;  // the letter of the code
        const uint32_t d = x ^ e;
        const uint32_t nz = ((((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) >> 7) & 0x01010101u;  // 1 per byte that is not its code's letter
        c2 |= nz | (nz << 1);
        cw = (cw << 8) | ((c2 * 0x40100401u) >> 24);
        bw = (bw << 8) | ((nz * 0x40100401u) >> 24);
    }
    code = cw;
    // byte j's field sits at bits 31 - 2j, 30 - 2j: reversed, at 2j (the field's HIGH bit) and 2j + 1 (its low bit)
    const uint32_t r = rc_brev32(cw), rb = rc_brev32(bw);
    bad = rc_compress_even16(rb >> 1);
    tm = rc_compress_even16(r & (r >> 1)) & ~bad;
    am = rc_compress_even16(~(r | (r >> 1)));
}

// reverse complement of a k-mer code (first base in the most significant 2 bits, as the
// reference's KmerCode): reverse the 2-bit groups, complement, drop the unused low bits.
RC_HD uint64_t rc_revcomp(uint64_t code, int k)
{
    uint64_t x = rc_brev64(code);
    x = ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);
    return (~x) >> (64 - 2 * k);
}

// KmerCode::GetCanonicalKmerCode, KmerCode.h:58-71
RC_HD uint64_t rc_canonical(uint64_t code, int k)
{
    uint64_t rc = rc_revcomp(code, k);
    return rc < code ? rc : code;
}

// "dump order" of a table this library counted or writes out: ascending rc_dump_order_key(code).
// Jellyfish writes its dump in hash-table order, i.e. pseudo-random with respect to the k-mer
// text, and the ERROR_RATE pass (main.cpp:310-358) samples the first 100000 qualifying entries of
// that order; a bijective 64-bit mix (the splitmix64 finaliser) keeps that sample unbiased.
RC_HD uint64_t rc_dump_order_key(uint64_t z)
{
    z ^= z >> 30;
    z *= 0xbf58476d1ce4e5b9ull;
    z ^= z >> 27;
    z *= 0x94d049bb133111ebull;
    z ^= z >> 31;
    return z;
}

// 32-bit bucket hash of a canonical code (free choice: results do not depend on it)
RC_HD uint32_t rc_hash(uint64_t key)
{
    uint32_t h = (uint32_t)key * 0x9E3779B1u + (uint32_t)(key >> 32) * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 12;
    h *= 0x297A2D39u;
    h ^= h >> 15;
    return h;
}

// home bucket of a canonical code: multiply-shift range reduction of the 32-bit hash, so the
// bucket count (hence the load factor and the table's HBM/MALL footprint) can be any number
RC_HD uint32_t rc_home(uint64_t key, uint32_t nb_home)
{
    return (uint32_t)(((uint64_t)rc_hash(key) * (uint64_t)nb_home) >> 32);
}

// ---- PACKED layout addressing ---------------------------------------------------------------------
// The code goes through a bijection of [0, 2^2k) built from 32-bit multiplies only (the probe
// kernels are bound by VALU issue, and a 64-bit multiply is four quarter-rate instructions):
// for 2k <= 32 two rounds of "odd multiply modulo 2^2k, xor-shift by k"; for 2k > 32 an unbalanced
// Feistel network over the low 32 bits and the 2k-32 bits above them.  rc_unmix2k undoes it.
#define RC_MIX_A 0x9E3779B1u
#define RC_MIX_B 0x85EBCA77u
#define RC_MIX_C 0xC2B2AE3Du
#define RC_MIX_D 0x27D4EB2Fu
#define RC_MIX_E 0x165667B1u
RC_HD uint64_t rc_mix2k(uint64_t c, int k)
{
    const int kb = 2 * k;
    if (kb <= 32) {
        const uint32_t M = kb == 32 ? ~0u : ((1u << kb) - 1u);
        uint32_t x = (uint32_t)c;
        x = (x * RC_MIX_A) & M;
        x ^= x >> k;
        x = (x * RC_MIX_B) & M;
        x ^= x >> k;
        return x;
    }
    const int sh = 64 - kb;  // = 32 - (bits of hi), 0..31
    uint32_t lo = (uint32_t)c, hi = (uint32_t)(c >> 32);
    hi ^= (lo * RC_MIX_A) >> sh;
    lo = (lo ^ (hi * RC_MIX_B)) * RC_MIX_C;
    hi ^= (lo * RC_MIX_D) >> sh;
    lo ^= hi * RC_MIX_E;
    return ((uint64_t)hi << 32) | lo;
}
RC_HD uint64_t rc_unmix2k(uint64_t m, int k)
{
    const int kb = 2 * k;
    if (kb <= 32) {
        const uint32_t M = kb == 32 ? ~0u : ((1u << kb) - 1u);
        uint32_t x = (uint32_t)m;
        x ^= x >> k;
        x = (x * 0xB6C92F47u) & M;  // inverses modulo 2^32 of RC_MIX_B, RC_MIX_A
        x ^= x >> k;
        x = (x * 0x0E8B2F51u) & M;
        return x;
    }
    const int sh = 64 - kb;
    uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
    lo ^= hi * RC_MIX_E;
    hi ^= (lo * RC_MIX_D) >> sh;
    lo = (lo * 0xA89ED915u) ^ (hi * RC_MIX_B);  // inverse modulo 2^32 of RC_MIX_C
    hi ^= (lo * RC_MIX_A) >> sh;
    return ((uint64_t)hi << 32) | lo;
}
RC_HD uint32_t rc_mulhi32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
// home bucket and remainder of a canonical code: P = (mixed << (64 - 2k)) * nb_home as a 96-bit
// number, home = P >> 64, rem = bits 32..63 of P, xrem = the `ext` bits below those.  Codes that
// share a home differ by at least 2^(64-2k) * nb_home in P, so by at least one unit of (rem, xrem)
// when nb_home * 2^ext >= 2^(2k-32).
// (*top, optional: the most significant 32 bits of the mixed code -- what homes and filter words are cut from)
RC_HD void rc_packed_addr(uint64_t canon, int k, uint32_t nb_home, int ext, uint32_t *home, uint32_t *rem, uint32_t *xrem,
                          uint32_t *top = nullptr)
{
    const int kb = 2 * k;
    const uint64_t m = rc_mix2k(canon, k);
    *xrem = 0;
    if (kb <= 32) {
        const uint32_t a = (uint32_t)m << (32 - kb);
        if (top) *top = a;
        *home = rc_mulhi32(a, nb_home);
        *rem = a * nb_home;
        return;
    }
    const int sh = 64 - kb;
    const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
    const uint32_t a = sh ? ((hi << sh) | (lo >> (32 - sh))) : hi, b = lo << sh;  // m << sh as two words
    if (top) *top = a;
    const uint32_t tl = a * nb_home, th = rc_mulhi32(a, nb_home), u = rc_mulhi32(b, nb_home);
    const uint32_t sum = tl + u;
    *home = th + (sum < tl ? 1u : 0u);
    *rem = sum;
    if (ext) *xrem = (b * nb_home) >> (32 - ext);
}
// the inverse: the canonical code stored as (home, rem, xrem).  With m = mixed code, m * nb_home =
// home * 2^2k + f and (rem, xrem) = the top 32 + ext bits of the 2k-bit fraction f, so m = ceil(N /
// nb_home) for N = home * 2^2k + ((rem, xrem) aligned to the top of 2k bits) -- a 96-bit by 32-bit
// long division.
RC_HD uint64_t rc_packed_key(uint32_t home, uint32_t rem, uint32_t xrem, int ext, int k, uint32_t nb_home)
{
    const int kb = 2 * k;
    uint64_t hi, lo;  // N = hi * 2^64 + lo
    if (kb >= 32) {
        const uint64_t frac = (((uint64_t)rem << ext) | xrem) << (kb - 32 - ext);  // below 2^2k
        hi = kb == 64 ? home : ((uint64_t)home >> (64 - kb));
        lo = (kb == 64 ? 0 : ((uint64_t)home << kb)) | frac;
    } else {
        hi = 0;
        lo = ((uint64_t)home << kb) | (rem >> (32 - kb));
    }
    const uint64_t nb = nb_home;
    uint64_t r = hi % nb;  // hi / nb == 0: the quotient is below 2^2k <= 2^64
    uint64_t t = (r << 32) | (lo >> 32);
    const uint64_t q1 = t / nb;
    r = t % nb;
    t = (r << 32) | (lo & 0xFFFFFFFFull);
    const uint64_t q0 = t / nb;
    r = t % nb;
    uint64_t m = (q1 << 32) | q0;
    if (r) ++m;
    return rc_unmix2k(m, k);
}

// ---- rolling code with invalid tracker (KmerCode.cpp:7-42) ---------------------------------
// base codes: 0..3 = A,C,G,T; >= 4 = not ACGT (contributes bits 11 and sets the tracker)
struct rc_kmer {
    uint64_t code;
    int inv;
};

RC_HD rc_kmer rc_append(rc_kmer km, int k, int b)
{
    rc_kmer r;
    int inv = km.inv;
    if (inv != -1) ++inv;
    r.code = ((km.code << 2) & rc_kmer_mask(k)) | (uint64_t)(b >= 4 ? 3 : b);
    if (b >= 4) inv = 0;
    if (inv >= k) inv = -1;
    r.inv = inv;
    return r;
}

RC_HD rc_kmer rc_prepend(rc_kmer km, int k, int b)
{
    rc_kmer r;
    int inv = km.inv;
    if (inv != -1) inv -= 1;
    if (inv < 0) inv = -1;
    if (b >= 4) inv = k - 1;
    r.code = ((km.code >> 2) | ((uint64_t)(b >= 4 ? 3 : b) << (2 * (k - 1)))) & rc_kmer_mask(k);
    r.inv = inv;
    return r;
}

// ---- GetBound, ErrorCorrection.cpp:139-142 -------------------------------------------------
// c*E + 6*sqrt(c*E) + 1 in IEEE double with separately rounded mul/add/sqrt (the reference is
// x86-64 SSE2: no FMA contraction, correctly rounded sqrtsd).  On the device the explicit
// round-to-nearest intrinsics are used so hipcc cannot contract a*b+c into an FMA.
RC_HD double rc_bound_d(int c, double e)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double ce = __dmul_rn((double)c, e);
    double s = __dmul_rn(6.0, __dsqrt_rn(ce));
    return __dadd_rn(__dadd_rn(ce, s), 1.0);
#else
    volatile double ce = (double)c * e;
    volatile double s = 6.0 * __builtin_sqrt(ce);
    volatile double r = ce + s;
    return r + 1.0;
#endif
}

// the implicit double->int conversions at ErrorCorrection.cpp:164,798,815,821,1282 are
// cvttsd2si on the reference's platform: INT_MIN for NaN / out of range, truncation otherwise
RC_HD int rc_bound_i(int c, double e)
{
    if (c < 0) return RC_INT_MIN;  // sqrt(negative) = NaN
    double x = rc_bound_d(c, e);
    if (!(x < 2147483648.0)) return RC_INT_MIN;
    return (int)x;
}

// ---- the integer steps of GetBound ------------------------------------------------------------
// (int)GetBound(c) is a non-decreasing step function of the count c (every operation in it is
// monotone and correctly rounded), and ERROR_RATE is a constant of the run: B[v] = the smallest
// count whose bound reaches v, computed once on the host (x86 arithmetic: the reference's own) when
// the run parameters are set.  The search asks "is the bound of this count at least t" thousands of
// times per read and needs the bound's value only when the answer is no: one integer compare against
// B[t] replaces the double-precision multiply / square root / add chain in the common case.
//   B[v], v in [2, RC_BOUND_STEPS): as above, or RC_BOUND_NEVER if no count below 2^31 reaches v.
//   B[0] = number of valid entries (0: table unusable -- the bound overflows int for large counts
//   at this error rate, which breaks monotonicity -- every caller then evaluates GetBound itself).
#define RC_BOUND_STEPS 65536  // (thresholds up to there: counts of ~16 M at ERROR_RATE 0.004)
#define RC_BOUND_NEVER 0x80000000u
inline void rc_bound_steps_build(double e, uint32_t *B)  // host only
{
    for (int v = 0; v < RC_BOUND_STEPS; ++v) B[v] = RC_BOUND_NEVER;
    B[0] = 0;
    B[1] = 0;
    const double top = rc_bound_d(2147483647, e);
    if (!(e >= 0.0) || !(top < 2147483648.0)) return;  // NaN / negative rates, or the (int) conversion overflows: no table
    for (int v = 2; v < RC_BOUND_STEPS; ++v) {
        if (rc_bound_i(2147483647, e) < v) break;  // never reached (and no larger v is)
        // smallest c with bound_i(c) >= v.  c E + 6 sqrt(c E) + 1 = v solved for c gives a first guess a step or two from it
        // (sqrt(c E) = sqrt(8 + v) - 3); the walk to the exact step uses rc_bound_i itself, and a guess that is further off
        // than 64 steps (E tiny or huge) is settled by bisection as before.  (65 534 bisections of 31 steps each were 25 ms of
        // a two-second run, once per context.)
        long long lo = 0, hi = 2147483647;
        if (e > 0.0) {
            const double r = __builtin_sqrt(8.0 + (double)v) - 3.0, g = r > 0.0 ? r * r / e : 0.0;
            long long c = g >= 2147483647.0 ? 2147483647 : (long long)g;
            int steps = 0;
            while (steps < 64 && c > 0 && rc_bound_i((int)(c - 1), e) >= v) --c, ++steps;
            while (steps < 64 && c < 2147483647 && rc_bound_i((int)c, e) < v) ++c, ++steps;
            if (steps < 64 && rc_bound_i((int)c, e) >= v && (c == 0 || rc_bound_i((int)(c - 1), e) < v)) lo = hi = c;
        }
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (rc_bound_i((int)mid, e) >= v)
                hi = mid;
            else
                lo = mid + 1;
        }
        B[v] = (uint32_t)lo;
    }
    B[0] = RC_BOUND_STEPS;
}

// `b < GetBound(c)` as the double comparison at ErrorCorrection.cpp:1195 (NaN compares false)
RC_HD bool rc_less_than_bound(int b, int c, double e)
{
    if (c < 0) return false;
    return (double)b < rc_bound_d(c, e);
}
