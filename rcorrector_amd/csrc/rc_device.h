// rc_device.h -- device-side table probe shared by the kernels (rc_table.hip, rc_correct.hip).
#pragma once
#include "rc_common.h"

// Store::GetCount on a canonical code (Store.h:59-66): one 64-byte bucket = four 16-byte loads
// of the same sector; the slots are picked out of registers.  Of two equal keys the first in
// probe order wins (the build places the later Put first).  T.layout is wave-uniform.
// n_req (optional): incremented once per bucket read (profiling builds of k_correct)
// the full count of a k-mer whose slot holds the "all ones" count (rc_common.h): binary search in the
// table's prefix.  Rare by construction (at most RC_PACKED_OVF_MAX k-mers of a table).
__device__ inline int rc_packed_overflow_count(const rc_table_view &T, uint64_t canon)
{
    const uint32_t n = T.buckets[-16];
    const uint4 *E = reinterpret_cast<const uint4 *>(T.buckets) - RC_TABLE_PREFIX_BYTES / 16;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint4 e = E[mid];
        if ((((uint64_t)e.y << 32) | e.x) < canon)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo < n) {
        const uint4 e = E[lo];
        if ((((uint64_t)e.y << 32) | e.x) == canon) return (int)e.z;
    }
    return 0;  // (not reached for a slot that carries the mark)
}

// EXT = false: the caller knows that T.ext == 0 (the probe kernels, bound by VALU issue, are compiled
// both ways: the masks become constants and the extra multiply of rc_packed_addr disappears)
// orient: canon or its reverse complement -- the orientation in which the caller's neighbours differ in the LAST base (a
// search node's four extensions), for the "core" filter of rc_common.h; callers without such neighbours pass canon
// disp0 (PACKED, rc_table_lookup_quad's straggler path): the walk starts at the disp0-th bucket behind the home bucket -- the ones
// in front of it have been looked at, and so has the filter
// more (PACKED): only ONE bucket is looked at; where the walk would go on to the next one -- the k-mer not found, the bucket full and
// flagged -- *more is set and 0 returned: the caller finishes those k-mers later, gathered into full wavefronts (disp0 = 1), instead
// of sending the whole wavefront round the loop again for a lane or two (round 6, the fused probe kernel)
template <bool EXT = true, bool SEARCH = false>
__device__ __forceinline__ int rc_table_lookup_o(const rc_table_view &T, uint64_t canon, uint64_t orient, uint32_t *n_req = nullptr, uint32_t disp0 = 0,
                                                 bool *more = nullptr)
{
    // The slots are compared as 64-bit words, from the last slot to the first, so that the first slot in
    // probe order is assigned last and wins without a test (two instructions per slot: the probe
    // kernels are bound by VALU issue).
    if (T.layout) {  // PACKED: 8 x {rem, count | xrem | disp << 27}, continue flag in the last slot's bit 31
        uint32_t b, rem, xrem, top;
        const int ext = EXT ? T.ext : 0;
        rc_packed_addr(canon, T.k, T.nb_home, ext, &b, &rem, &xrem, &top);
#ifdef RC_EXP_ADDR_WINDOW  // dev (tools/exp/addr_window.md): every probe lands in the first RC_EXP_ADDR_WINDOW buckets -- WRONG counts; what
        b &= (RC_EXP_ADDR_WINDOW - 1);  // the probe kernel would cost if the table's lines were always in the L2 / Infinity Cache
#endif
        b += disp0;
        if (disp0 == 0 && T.filter && (SEARCH || T.filter_all)) {  // (wave-uniform) most misses end at one word of the filter
            if (T.filter_kind) {
                uint32_t fw, fm;
                rc_filter_core_addr(orient, T.filter_words, &fw, &fm);
                if ((T.filter[fw] & fm) != fm) return 0;
            } else {
                const uint32_t fm = rc_filter_mask(rem);
                if ((T.filter[rc_mulhi32(top, T.filter_words)] & fm) != fm) return 0;
            }
        }
        const uint32_t cmask = RC_PACKED_COUNT_MASK >> ext;  // (uniform)
        const uint32_t mhi = 0x7FFFFFFFu & ~cmask;
        const uint32_t xhi = xrem << (27 - ext);
        // A slot matches iff ((hi ^ whi) & mhi) | (lo ^ rem) is zero (an empty slot carries displacement 15, which no entry has): two
        // three-input bit operations (v_bitop3_b32: 0x28 = (a ^ b) & c, 0xF6 = a | (b ^ c)), a compare and a select per slot; the
        // count is masked out of the selected word once, behind the chain (round 6: the 64-bit compare of round 2 cost five
        // instructions a slot, a register copy among them).
        // -DRC_LOOKUP_HALF_BUCKET (round 6, measured and left out): the bucket read HALF BY HALF.  The build fills a bucket's slots from
        // the first on (k_scatter: positions in home order, rc_table.hip), so an empty fourth slot says that the second half is empty
        // too -- and that the bucket does not say "continue", which takes a full one: the lanes that found their k-mer in slots 0-3,
        // or whose bucket ends there (at load 0.4: 89 % of the k-mers that are in the table, 60 % of those that are not), are done
        // after two 16-byte loads.  Each of a probe's loads is a pass through the vector L1's address stage, which the probe kernels
        // feel (`RC_EXP_BUCKET_LOADS`: 42.0 / 38.1 / 36.9 / 33.0 ms with 4 / 3 / 2 / 1 of them) -- but the second half's loads still
        // issue for the wave whenever one lane wants them, now behind a dependent compare: config 1 10.76 -> 11.39 ms, config 2 41.3 ->
        // 42.3, config 3 62.0 -> 64.2, K2s and K3 + 2-3 % (profiles/r6_half_bucket_ab.txt; parity-green on 404 GPU tests).
        for (uint32_t disp = disp0;; ++disp, ++b) {
            const uint4 *p = reinterpret_cast<const uint4 *>(T.buckets + (size_t)b * RC_BUCKET_DWORDS);
            if (n_req) ++*n_req;
            const uint32_t whi = (disp << 27) | xhi;
#ifndef RC_LOOKUP_HALF_BUCKET
            uint32_t dlo[RC_PACKED_SLOTS], dhi[RC_PACKED_SLOTS];
#pragma unroll
            for (int q = 0; q < RC_BUCKET_DWORDS / 4; ++q) {
#ifdef RC_EXP_BUCKET_LOADS  // dev (WRONG counts): only the first RC_EXP_BUCKET_LOADS of a bucket's four 16-byte loads are issued -- what do the
                const uint4 v = q < RC_EXP_BUCKET_LOADS ? p[q] : make_uint4(0, 0x78000000u, 0, 0x78000000u);  // vector L1's address / tag stages cost the probe kernels?
#else
                const uint4 v = p[q];
#endif
                dlo[2 * q + 0] = v.x;
                dhi[2 * q + 0] = v.y;
                dlo[2 * q + 1] = v.z;
                dhi[2 * q + 1] = v.w;
            }
            uint32_t rh = 0;
#pragma unroll
            for (int s2 = RC_PACKED_SLOTS - 1; s2 >= 0; --s2) {
                const uint32_t t = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(dhi[s2], whi, mhi, 0x28), dlo[s2], rem, 0xF6);
                rh = t == 0 ? dhi[s2] : rh;
            }
            const uint32_t last = dhi[RC_PACKED_SLOTS - 1];
#else
            uint32_t rh = 0, last = 0;
            {
                const uint4 v0 = p[0], v1 = p[1];
                const uint32_t lo[4] = {v0.x, v0.z, v1.x, v1.z}, hi[4] = {v0.y, v0.w, v1.y, v1.w};
#pragma unroll
                for (int s2 = 3; s2 >= 0; --s2) {
                    const uint32_t t = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(hi[s2], whi, mhi, 0x28), lo[s2], rem, 0xF6);
                    rh = t == 0 ? hi[s2] : rh;
                }
                if (rh == 0 && (hi[3] & RC_PACKED_EMPTY_WORD) != RC_PACKED_EMPTY_WORD) {  // not found yet and the first half is full
                    const uint4 v2 = p[2], v3 = p[3];
                    const uint32_t lo2[4] = {v2.x, v2.z, v3.x, v3.z}, hi2[4] = {v2.y, v2.w, v3.y, v3.w};
#pragma unroll
                    for (int s2 = 3; s2 >= 0; --s2) {
                        const uint32_t t = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(hi2[s2], whi, mhi, 0x28), lo2[s2], rem, 0xF6);
                        rh = t == 0 ? hi2[s2] : rh;
                    }
                    last = hi2[3];
                }
            }
#endif
            const int r = (int)(rh & cmask);
            if (r != 0 || !(last >> 31) || disp == RC_PACKED_MAX_DISP) {
                if (r == (int)cmask) return rc_packed_overflow_count(T, canon);
                return r;
            }
            if (more) {
                *more = true;
                return 0;
            }
        }
    }
    uint32_t b = rc_home(canon, T.nb_home);
    const uint32_t klo = (uint32_t)canon, khi = (uint32_t)(canon >> 32);
    for (;;) {
        const uint4 *p = reinterpret_cast<const uint4 *>(T.buckets + (size_t)b * RC_BUCKET_DWORDS);
        uint32_t d[RC_BUCKET_DWORDS];
        if (n_req) ++*n_req;
#pragma unroll
        for (int q = 0; q < RC_BUCKET_DWORDS / 4; ++q) {
            const uint4 v = p[q];
            d[4 * q + 0] = v.x;
            d[4 * q + 1] = v.y;
            d[4 * q + 2] = v.z;
            d[4 * q + 3] = v.w;
        }
        // (an empty slot is {0, 0, 0}: it can only "match" the all-A code, with the count 0 of a miss)
        int r = 0;
#pragma unroll
        for (int s2 = RC_WIDE_SLOTS - 1; s2 >= 0; --s2)
            r = (d[3 * s2] == klo && d[3 * s2 + 1] == khi) ? (int)d[3 * s2 + 2] : r;  // (as one 64-bit compare the
                                                                                       // compiler re-slices the bucket read into a load per slot)
        if (r != 0 || !(d[RC_BUCKET_DWORDS - 1] & 1u)) return r;
        ++b;
    }
}

template <bool EXT = true>
__device__ __forceinline__ int rc_table_lookup(const rc_table_view &T, uint64_t canon, uint32_t *n_req = nullptr)
{
    return rc_table_lookup_o<EXT>(T, canon, canon, n_req);
}

// rc_canonical (rc_common.h: KmerCode::GetCanonicalKmerCode, KmerCode.h:58-71) in 32-bit halves for the probe loops, which are
// bound by vector-instruction issue: reversing 64 bits is two v_bfrev_b32 with the halves exchanged, and "swap the two bits
// of every base, complement" is one three-input bit operation per half on y << 1 and y >> 1 (0x27 = ~((a & ~c) | (b & c)),
// c = 0x55555555) -- 12 instructions instead of 19.
__device__ __forceinline__ uint64_t rc_canonical_dev(uint64_t code, int k)
{
    const uint32_t xl = (uint32_t)code, xh = (uint32_t)(code >> 32);
    const uint32_t yh = __builtin_bitreverse32(xl), yl = __builtin_bitreverse32(xh);
    const uint32_t zh = __builtin_amdgcn_bitop3_b32(yh << 1, yh >> 1, 0x55555555u, 0x27);
    const uint32_t zl = __builtin_amdgcn_bitop3_b32(yl << 1, yl >> 1, 0x55555555u, 0x27);
    const uint64_t rc = (((uint64_t)zh << 32) | zl) >> (64 - 2 * k);
    return rc < code ? rc : code;
}

// Store::GetCount (Store.h:59-66) for the 64 k-mers of a wavefront, one per lane, the four lanes of a QUAD reading a bucket
// together: in round r the quad looks at the home bucket of its lane r -- lane q loads the 16 bytes with slots 2q and 2q + 1,
// compares them, and the first match in slot order is passed round the quad (two DPP steps) -- so a load instruction touches 16
// lines, 64 contiguous bytes a quad, instead of 64 lines 16 bytes each, and a bucket costs one translation and one pass
// through the vector L1's address stage instead of four.  tools/microbench_bucket.hip (round 6, 2^30 random buckets, the same
// slot compare both ways): 217 -> 266 G buckets/s out of a table the L2 holds, 58 -> 56 G/s out of 128 MB - 1 GB (the
// fabric's request rate either way), 24 -> 51 G/s out of 4 GB, beyond the reach of the TLB.  The rare k-mer whose home
// bucket is full and says "continue" finishes the walk by itself (rc_table_lookup_o from the second bucket on).
// PACKED tables only (wave-uniform test; WIDE: the per-lane lookup).  `valid` = false: no lookup for this lane, result 0.
// MUST be called with every lane of the wavefront active (the lanes lend each other their loads).
template <int CTRL>
__device__ __forceinline__ uint32_t rc_dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <bool EXT = true>
__device__ __forceinline__ int rc_table_lookup_quad(const rc_table_view &T, uint64_t canon, bool valid)
{
    if (!T.layout) return valid ? rc_table_lookup<EXT>(T, canon) : 0;
    const int ql = threadIdx.x & 3;
    const int ext = EXT ? T.ext : 0;
    uint32_t b, rem, xrem, top;
    rc_packed_addr(canon, T.k, T.nb_home, ext, &b, &rem, &xrem, &top);
    if (T.filter && T.filter_all) {  // (wave-uniform) the absence filter first: a miss that ends there never touches the buckets
        bool pass = false;
        if (valid) {
            if (T.filter_kind) {
                uint32_t fw, fm;
                rc_filter_core_addr(canon, T.filter_words, &fw, &fm);
                pass = (T.filter[fw] & fm) == fm;
            } else {
                const uint32_t fm = rc_filter_mask(rem);
                pass = (T.filter[rc_mulhi32(top, T.filter_words)] & fm) == fm;
            }
        }
        valid = pass;
    }
    const uint32_t cmask = RC_PACKED_COUNT_MASK >> ext;  // (uniform)
    const uint32_t mhi = 0x7FFFFFFFu & ~cmask;
    const uint32_t xhi = xrem << (27 - ext);
    // what the quad needs to know about a lane's k-mer: bucket, remainder, the high word's wanted bits (displacement 0) -- and
    // whether there is anything to look up (bit 31 of the wanted bits, which no entry has: bit 31 is the continue flag, masked out)
    const uint32_t want = xhi | (valid ? 0u : 0x80000000u);
    uint32_t mine = 0, more = 0;
    auto round = [&](uint32_t bb, uint32_t rr, uint32_t ww, bool me) {
        uint4 v = make_uint4(0, RC_PACKED_EMPTY_WORD, 0, RC_PACKED_EMPTY_WORD);
        if (!(ww >> 31) && ql < RC_BUCKET_DWORDS / 4) v = reinterpret_cast<const uint4 *>(T.buckets + (size_t)bb * RC_BUCKET_DWORDS)[ql];
        const uint32_t t0 = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(v.y, ww, mhi, 0x28), v.x, rr, 0xF6);
        const uint32_t t1 = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(v.w, ww, mhi, 0x28), v.z, rr, 0xF6);
        uint32_t x = t1 == 0 ? v.w : 0u;
        x = t0 == 0 ? v.y : x;                       // the first of the lane's two slots wins
        const uint32_t y = rc_dpp_u32<0xB1>(x);      // quad_perm [1,0,3,2]
        const uint32_t p01 = (ql & 1) ? (y ? y : x) : (x ? x : y);
        const uint32_t z = rc_dpp_u32<0x4E>(p01);    // quad_perm [2,3,0,1]
        const uint32_t r = (ql & 2) ? (z ? z : p01) : (p01 ? p01 : z);  // the first match in slot order, in every lane of the quad
        const uint32_t cont = rc_dpp_u32<(RC_BUCKET_DWORDS / 4 - 1) * 0x55>(v.w) >> 31;  // the bucket's last slot (the quad's lane 3 with 64-byte buckets) carries the continue flag
        mine = me ? r : mine;
        more = me ? cont : more;
    };
    round(rc_dpp_u32<0x00>(b), rc_dpp_u32<0x00>(rem), rc_dpp_u32<0x00>(want), ql == 0);
    round(rc_dpp_u32<0x55>(b), rc_dpp_u32<0x55>(rem), rc_dpp_u32<0x55>(want), ql == 1);
    round(rc_dpp_u32<0xAA>(b), rc_dpp_u32<0xAA>(rem), rc_dpp_u32<0xAA>(want), ql == 2);
    round(rc_dpp_u32<0xFF>(b), rc_dpp_u32<0xFF>(rem), rc_dpp_u32<0xFF>(want), ql == 3);
    int r = (int)(mine & cmask);
    if (valid && r == 0 && more) r = rc_table_lookup_o<EXT>(T, canon, canon, nullptr, 1);  // (rare: the home bucket is full and the walk goes on)
    else if (r == (int)cmask) r = rc_packed_overflow_count(T, canon);
    return valid ? r : 0;
}

// ---- pieces of the probe kernels (K1) shared by rc_table.hip and rc_correct.hip ----------------------
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

#define RC_PLIST_MAX_READS 64  // reads per workgroup of the list-driven probe kernels

// the live entry stored in slot s of bucket b, if any: its canonical code and count.  A key that was
// Put more than once occupies several slots; the one a probe reaches first (the latest Put,
// Store.h:55) is the table's entry, the others are dead.
__device__ __forceinline__ bool rc_table_slot_entry(const rc_table_view &T, size_t b, int s, uint64_t *key, int32_t *count)
{
    const uint32_t *w = T.buckets + b * RC_BUCKET_DWORDS;
    uint64_t kk;
    int32_t cc;
    const uint32_t cmask = RC_PACKED_COUNT_MASK >> T.ext;
    if (T.layout) {
        const uint32_t word = w[2 * s + 1];
        cc = (int32_t)(word & cmask);
        if (cc == 0) return false;
        kk = rc_packed_key((uint32_t)(b - ((word >> 27) & 15u)), w[2 * s], (word & RC_PACKED_COUNT_MASK) >> (27 - T.ext), T.ext, T.k, T.nb_home);
        if (cc == (int32_t)cmask) cc = rc_packed_overflow_count(T, kk);
    } else {
        cc = (int32_t)w[3 * s + 2];
        if (cc == 0) return false;
        kk = ((uint64_t)w[3 * s + 1] << 32) | w[3 * s];
    }
    // walk the probe sequence from the key's home: live iff this slot is the first match
    uint32_t hb, rem = 0, xrem = 0;
    if (T.layout)
        rc_packed_addr(kk, T.k, T.nb_home, T.ext, &hb, &rem, &xrem);
    else
        hb = rc_home(kk, T.nb_home);
    const int S = rc_layout_slots(T.layout);
    for (size_t bb = hb;; ++bb) {
        const uint32_t *q = T.buckets + bb * RC_BUCKET_DWORDS;
        for (int i = 0; i < S; ++i) {
            bool match;
            if (T.layout) {
                const uint32_t word = q[2 * i + 1];
                match = (word & cmask) != 0 && q[2 * i] == rem && ((word >> 27) & 15u) == (uint32_t)(bb - hb) &&
                        (word & RC_PACKED_COUNT_MASK) >> (27 - T.ext) == xrem;
            } else {
                match = q[3 * i + 2] != 0 && q[3 * i] == (uint32_t)kk && q[3 * i + 1] == (uint32_t)(kk >> 32);
            }
            if (match) {
                if (bb != b || i != s) return false;  // shadowed by an earlier slot
                *key = kk;
                *count = cc;
                return true;
            }
        }
        if (bb > b) return false;  // (not reached: the slot itself matches at the latest)
    }
}
