// rc_correct_core.h -- per-read error correction as ONE WAVEFRONT PER READ.
//
// All 64 lanes of a wave walk the same control flow ("wave-uniform" scalars); lanes split the
// data-parallel parts (array fills, window scans, sort, the table probes of a search node).
// The per-read state lives in LDS (rc_read_state), the table in HBM.
//
// Reference behaviour restated (paths relative to /root/reference, v1.0.7):
//   GetStrongTrustedThreshold  ErrorCorrection.cpp:1482-1565   -> rc_front_end()
//   ErrorCorrection            ErrorCorrection.cpp:682-1480    -> rc_correct_read()
//   SearchPaths_Right/_Left    ErrorCorrection.cpp:201-678     -> rc_search() (explicit stack)
//   InferPosThreshold          ErrorCorrection.cpp:144-173     -> rc_pos_threshold()
//   GetKmerInformation         ErrorCorrection.cpp:1567-1602   -> rc_kmer_info()
//
// The template parameter W is the wave back end:
//   W::lane, W::STRIDE            lane id and lane count (64 on the device)
//   w.sync()                      orders LDS traffic between the lanes of the wave
//   w.ballot64(base, n, pred)     bit l = pred(base + l), one element per lane
//   w.for_lanes64(base, n, body)  body(base + l, l) on lane l
//   w.uni(x) / w.uni64(x)         a wave-uniform value -> scalar register (RC_U / RC_U64)
//   w.reduce_add(x)               sum over lanes
//   w.get(kmer) / w.lookup(code)  Store::GetCount of a k-mer state / of a valid forward code
//   w.sort(a, n)                  ascending in-LDS sort
//   w.stack_push/top/set_mask     search stack frames (HBM scratch)
//   w.phase(id) / w.stat(i, v)    optional accounting hooks
//   w.trace_passed/iter/strong    optional -verbose recording hooks
// The device back end is DevWaveT in rc_correct.hip; tests/hostsim has a lane-serial one
// (STRIDE = 1) that lets the CPU test-suite diff this exact control flow against the oracle.
#pragma once
#include "rc_common.h"

#define RC_BS_INLINE 16
// k as the wave back end knows it: a compile-time constant where the kernel was instantiated for one k
// (W::KT > 0: masks, shifts and window lengths become immediates instead of scalar registers), else the run's
#define RC_K(P) (W::KT > 0 ? W::KT : (P).k)

struct rc_run_params {
    int k;
    int max_fix_per_k;
    double error_rate;
    int bad_qual;  // badQualityThreshold as a signed char value
    // the first integer steps of GetBound at this error_rate (rc_common.h: rc_bound_steps_build): bs[v] = the
    // smallest count whose bound reaches v, for v in [2, RC_BS_INLINE); bs[0] = 0 if there is no table.
    // Part of the kernel arguments, i.e. read with scalar loads.
    uint32_t bs[RC_BS_INLINE];
    int flags;  // RC_PF_*
    // the whole table (RC_BOUND_STEPS entries, in device memory for the kernels): thresholds of RC_BS_INLINE and more --
    // reads of transcripts covered thousands of times -- take one load per read from here (nullptr: none)
    const uint32_t *bs_ext;
    // (int)GetBound(c) itself for the counts c < RC_BOUND_SMALL, one byte each, 255 = "evaluate it" (the value does not fit, or
    // would not be the host's): what the threshold rows and k_single turn a read's strong threshold into its weak one with
    // (ErrorCorrection.cpp:793-842) -- one load instead of a double-precision multiply / square root / add chain per read.
    // Device memory, behind bs_ext's entries (nullptr: none)
    const uint8_t *bound_small;
};
#define RC_BOUND_SMALL 4096
// the smallest count whose bound reaches t (t >= 2), or 0 if that is not known (no table, t beyond it, never reached)
RC_HD uint32_t rc_bs_lookup(const rc_run_params &P, int t)
{
    if (!P.bs[0] || t < 2) return 0;
    if (t < RC_BS_INLINE) return P.bs[t];
    if (!P.bs_ext || t >= RC_BOUND_STEPS) return 0;
    const uint32_t v = P.bs_ext[t];
    return v == RC_BOUND_NEVER ? 0 : v;
}
#define RC_PF_NO_ALT 1  // dev / tests: no alternative chains in the speculation rounds of the search

struct rc_island {
    short from, to;
};
struct rc_segment {
    short from, to, lanchor, ranchor;
    int top2[2];
};

// the four extension counts of a search node.  Named members, not an array: an array that is ever
// indexed at run time stays in private memory, and a load from private memory is a divergent value
// to the compiler -- one such load in the search loop turned its whole state into vector registers
struct rc_cnt4 {
    int c0, c1, c2, c3;
};

// search-stack frame: a node whose substitution alternatives are still pending
struct rc_frame {
    uint64_t code;
    int inv;
    int pos;
    int t;
    int threshold;
    int fix_cnt;
    int bottleneck;
    rc_cnt4 cnt;
    int mask;  // pending substitution candidates, bit c
};

// per-wave LDS state; arrays sized for `cap` bases (cap >= read length, multiple of 64) and
// `cap2` = cap rounded up to a power of two (sort buffer)
struct rc_read_state {
    unsigned char *base;    // [cap]   0..3 ACGT, 4 'N', 5 any other letter
    int *counts;            // [cap]   k-mer counts of the uncorrected read (K1 output)
    int *v;                 // [cap2]  sort buffer / fix positions
    signed char *path;      // [cap]   scratch path of the search (the reference's iBuffer)
    signed char *best;      // [cap]   accepted fixes (the reference's fix[])
    unsigned char *strongb; // [cap]   isStrongTrusted per base
    unsigned char *polya;   // [cap]   bit0 IsPolyA(.,k,2), bit1 IsPolyA(.,k,max(7,k/2))
    signed char *qual;      // [cap]   quality characters (only the vetoes read them)
    rc_island *isl;         // [cap/2+2]
    rc_segment *seg;        // [cap/2+2]
    // one bit per base (bit i%64 of word i/64), cap/64+1 words each, last word always 0
    uint64_t *m_a, *m_t;    // base is 'A' / 'T'
    uint64_t *m_n, *m_inv;  // base is 'N' / is not one of ACGT
    uint64_t *m_x;          // scratch: trusted k-mers, fixed positions
    // the read as 2-bit codes, 16 bases per word, first base most significant (a letter outside
    // ACGT contributes 3, as KmerCode::Append does); cap/16 + 3 words, the last two never written
    uint32_t *pk;
    // speculation cache of the search (rc_probe4_cached): what one gather round fetched.  Entries
    // [0, 4 n): the four extension counts of the next n <= RC_SPEC nodes of the keep-base path ("chain 0").
    // If the round expects the keep base of the chain's last node z to fail (K1 found its k-mer absent)
    // and z may be substituted, the other lanes fetch, for each of the three alternatives of z, the
    // keep-base counts of the nodes that follow on that alternative's path: three "alternative chains" of
    // spec_meta[0] entries each, from entry 4 (z + 1) on.
    int *spec_cnt;          // [RC_SPEC_ENTRIES]
    int *spec_meta;         // [4] entries per alternative chain (0 = none), nodes per alternative chain, z
    // counts of the corrected read's k-mers that a search already fetched, for GetKmerInformation:
    // [0] n (0 = empty), [1] first window, [2] position p and [3] base c of the one fix these windows
    // were probed with, [4 .. 4 + n) the counts of windows [1] .. [1] + n - 1
    int *memo;              // [4 + RC_MEMO_MAX]
    uint64_t *spec_code;    // [RC_SPEC]
    int *spec_inv;          // [RC_SPEC]
    int *spec_ret;          // [RC_SPEC] max(GetBound(max count),1) per cached node
    int *spec_keep;         // [RC_SPEC] count of the keep-base extension (-1: base not ACGT)
    int *spec_thr;          // [RC_SPEC] threshold of the node (InferPosThreshold)
    int *spec_mask;         // [RC_SPEC] substitution candidates of the node
    int len, kcnt;
};

#ifndef RC_SPEC
#define RC_SPEC 16           // nodes of chain 0 (tools/exp/spec_depth.py: what other depths cost in probes and rounds)
#endif
#define RC_SPEC_ENTRIES 128  // probes per gather round: two per lane
#define RC_MEMO_MAX 32

struct rc_spec_state {
    int n;    // cached positions (0 = empty)
    int pos;  // position of entry 0
    int dir;
};

RC_HD int rc_popc64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
RC_HD int rc_ctz64(uint64_t x)  // x != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((long long)x) - 1;
#else
    return __builtin_ctzll(x);
#endif
}
RC_HD int rc_clz64(uint64_t x)  // x != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}

// bits [i, i+n) of a mask array as the low n bits of the result (n <= 64; the array carries one
// zero word past the end)
RC_HD uint64_t rc_window(const uint64_t *m, int i, int n)
{
    const int w = i >> 6, sh = i & 63;
    uint64_t x = m[w] >> sh;
    if (sh) x |= m[w + 1] << (64 - sh);
    return n >= 64 ? x : (x & ((1ull << n) - 1ull));
}

RC_HD int rc_clz32(uint32_t x)  // x != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __clz((int)x);
#else
    return __builtin_clz(x);
#endif
}

// pk[] from base[] (rc_read_state::pk): one lane packs 16 bases = four dwords of base[].  A byte
// b in 0..5 becomes min(b, 3) by SWAR, and the multiply gathers the four 2-bit fields of a dword
// into its top byte in reading order (the partial products do not overlap, so nothing carries).
template <class W>
RC_HD void rc_pack_read(W &w, rc_read_state &S)
{
    const int nwords = (S.len + 15) >> 4;
    const uint32_t *b32 = reinterpret_cast<const uint32_t *>(S.base);
    for (int wi = w.lane; wi < nwords; wi += W::STRIDE) {
        uint32_t word = 0;
        for (int q = 0; q < 4; ++q) {
            uint32_t x = b32[4 * wi + q];
            const uint32_t t = (x >> 2) & 0x01010101u;
            x = (x | t | (t << 1)) & 0x03030303u;
            word = (word << 8) | ((x * 0x40100401u) >> 24);
        }
        S.pk[wi] = word;
    }
    w.sync();
}

// 2-bit code of the n bases starting at base p (1 <= n <= 32), first base most significant
RC_HD uint64_t rc_code_at(const uint32_t *pk, int p, int n)
{
    const int w0 = p >> 4, sh = 2 * (p & 15);
    uint64_t x = ((uint64_t)pk[w0] << 32) | pk[w0 + 1];
    if (sh) x = (x << sh) | ((uint64_t)pk[w0 + 2] >> (32 - sh));
    return x >> (64 - 2 * n);
}

// bit masks of the read's letters; every lane-parallel window test below runs on these
template <class W>
RC_HD void rc_build_masks(W &w, rc_read_state &S)
{
    const int nw = (S.len + 63) >> 6;
    for (int c = 0; c <= nw; ++c) {
        const int b0 = c << 6;
        const uint64_t ma = w.ballot64(b0, S.len, [&](int i) { return S.base[i] == 0; });
        const uint64_t mt = w.ballot64(b0, S.len, [&](int i) { return S.base[i] == 3; });
        const uint64_t mn = w.ballot64(b0, S.len, [&](int i) { return S.base[i] == 4; });
        const uint64_t mi = w.ballot64(b0, S.len, [&](int i) { return S.base[i] >= 4; });
        S.m_a[c] = ma;
        S.m_t[c] = mt;
        S.m_n[c] = mn;
        S.m_inv[c] = mi;
    }
    w.sync();
}

RC_HD int rc_min(int a, int b) { return a < b ? a : b; }

// A value every lane of the wave agrees on, read from LDS: RC_U moves it to a scalar register on
// the device (v_readfirstlane) so that everything computed from it -- k-mer shifts, compares,
// branches -- runs on the scalar unit instead of occupying the vector ALU 64 lanes wide.
#define RC_U(x) (w.uni((int)(x)))
#define RC_U64(x) (w.uni64((uint64_t)(x)))

// ---- front end ---------------------------------------------------------------------------------
// screens of ErrorCorrection.cpp:735-755 (=:1507-1527); returns 1 if the read is rejected
template <class W>
RC_HD int rc_screened(W &w, const rc_read_state &S, int k)
{
    (void)w;
    int n = 0, a = 0, t = 0;
    const int nw = (S.len + 63) >> 6;
    for (int c = 0; c < nw; ++c) {
        n += rc_popc64(S.m_n[c]);
        a += rc_popc64(S.m_a[c]);
        t += rc_popc64(S.m_t[c]);
    }
    return n > 5 || a > S.len - k || t > S.len - k;
}

// IsPolyA for every k-mer window, thresholds 2 and max(7,k/2) (ErrorCorrection.cpp:53-71,
// :776-779, :826): window popcounts of the A / T masks
template <class W>
RC_HD void rc_polya_flags(W &w, rc_read_state &S, int k)
{
    int thr7 = 7;
    if (k / 2 > thr7) thr7 = k / 2;
    for (int i = w.lane; i < S.kcnt; i += W::STRIDE) {
        const int a = rc_popc64(rc_window(S.m_a, i, k));
        const int t = rc_popc64(rc_window(S.m_t, i, k));
        int f = 0;
        if (a >= k - 2 || t >= k - 2) f |= 1;
        if (a >= k - thr7 || t >= k - thr7) f |= 2;
        S.polya[i] = (unsigned char)f;
    }
    w.sync();
}

// v[] = poly-A masked counts, sorted ascending (ErrorCorrection.cpp:774-784, :1247-1257)
template <class W>
RC_HD void rc_masked_sorted(W &w, rc_read_state &S)
{
    for (int i = w.lane; i < S.kcnt; i += W::STRIDE) S.v[i] = (S.polya[i] & 2) ? -1 : S.counts[i];
    w.sync();
    w.sort(S.v, S.kcnt);
}

// the "drop" scan of ErrorCorrection.cpp:787-816 / :1543-1563 on sorted v[].
// returns strong; *found = 1 if a drop was found, *prev = v[i-1] at the drop
template <class W>
RC_HD int rc_initial_strong(W &w, const rc_read_state &S, int *found, int *prev)
{
    const int kcnt = S.kcnt;
    // highest i in [1, kcnt) with v[i] > 2 v[i-1] && v[i] > 10
    for (int b0 = ((kcnt - 1) >> 6) << 6; b0 >= 0; b0 -= 64) {
        const uint64_t m = w.ballot64(b0, kcnt, [&](int i) { return i >= 1 && S.v[i] > 2 * S.v[i - 1] && S.v[i] > 10; });
        if (m) {
            const int i = b0 + 63 - rc_clz64(m);
            *found = 1;
            *prev = RC_U(S.v[i - 1]);
            return RC_U(S.v[i]);
        }
    }
    *found = 0;
    *prev = 0;
    int i = kcnt;  // lowest i with v[i] > 0, else kcnt
    for (int b0 = 0; b0 < kcnt; b0 += 64) {
        const uint64_t m = w.ballot64(b0, kcnt, [&](int j) { return S.v[j] > 0; });
        if (m) {
            i = b0 + rc_ctz64(m);
            break;
        }
    }
    return RC_U(S.v[(i + kcnt - 1) / 2]);
}

// GetStrongTrustedThreshold (ErrorCorrection.cpp:1482-1565).  Requires base[], counts[],
// len, kcnt loaded.  info bit0 = drop found, bit1 = v[i-1]==2 at the drop, bit2 = screened.
template <class W>
RC_HD int rc_front_end(W &w, rc_read_state &S, const rc_run_params &P, int *info)
{
    *info = 4;
    if (S.len < RC_K(P)) return -1;
    if (rc_screened(w, S, RC_K(P))) return -1;
    rc_polya_flags(w, S, RC_K(P));
    rc_masked_sorted(w, S);
    int found, prev;
    int strong = rc_initial_strong(w, S, &found, &prev);
    *info = (found ? 1 : 0) | ((found && prev == 2) ? 2 : 0);
    return strong;
}

// ---- search ------------------------------------------------------------------------------------
struct rc_search_ctx {
    int dir;  // +1 right, -1 left
    int start, to;
    int max_fix_cnt;
    int best_fix_cnt;
    int best_bottleneck;
    int trial_cnt;
    int top2a, top2b;  // top2FixBottleNeck[0], [1] of the segment being searched
};

// a wave-uniform value, pinned as a value: on the device v_readfirstlane (free for a value that
// already sits in a scalar register).  Without it the optimiser rewrites a chain of selects over
// adjacent members into ONE load with a computed address, which keeps the struct in private memory
RC_HD int rc_pin(int x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(x);
#else
    return x;
#endif
}

// cnt[idx] for a wave-uniform runtime idx
RC_HD int rc_sel4(const rc_cnt4 &c, int idx)
{
    const int c0 = rc_pin(c.c0), c1 = rc_pin(c.c1), c2 = rc_pin(c.c2), c3 = rc_pin(c.c3);
    int r = c0;
    r = idx == 1 ? c1 : r;
    r = idx == 2 ? c2 : r;
    r = idx == 3 ? c3 : r;
    return r;
}

// bit c set iff c != b and cnt[c] >= threshold (substitution candidates, :343-352 / :577-588)
RC_HD int rc_candidates(const rc_cnt4 &c, int b, int threshold)
{
    int m = 0;
    if (b != 0 && c.c0 >= threshold) m |= 1;
    if (b != 1 && c.c1 >= threshold) m |= 2;
    if (b != 2 && c.c2 >= threshold) m |= 4;
    if (b != 3 && c.c3 >= threshold) m |= 8;
    return m;
}

// InferPosThreshold (ErrorCorrection.cpp:144-173) given the four extension counts
RC_HD int rc_pos_threshold(const rc_cnt4 &cnt, int upper, double e)
{
    int mx = 0;
    mx = cnt.c0 > mx ? cnt.c0 : mx;
    mx = cnt.c1 > mx ? cnt.c1 : mx;
    mx = cnt.c2 > mx ? cnt.c2 : mx;
    mx = cnt.c3 > mx ? cnt.c3 : mx;
    int ret = rc_bound_i(mx, e);
    if (ret < 1) ret = 1;
    if (upper > ret || upper <= 0) return ret;
    return upper;
}

RC_HD int rc_max4(const rc_cnt4 &cnt)
{
    int mx = 0;
    mx = cnt.c0 > mx ? cnt.c0 : mx;
    mx = cnt.c1 > mx ? cnt.c1 : mx;
    mx = cnt.c2 > mx ? cnt.c2 : mx;
    mx = cnt.c3 > mx ? cnt.c3 : mx;
    return mx;
}

RC_HD rc_kmer rc_extend(rc_kmer km, int k, int dir, int b)
{
    return dir > 0 ? rc_append(km, k, b) : rc_prepend(km, k, b);
}

// kc after j rc_extend() steps with the read's own bases p0, p0+dir, ..., p0+dir*(j-1)
// (0 <= j <= 31), in closed form: the code is a shift plus a window of the packed read, and the
// one-slot invalid tracker (KmerCode.cpp:7-42) ends at the most recently appended invalid base /
// at the leftmost prepended one, or ages by j steps if the run holds none.
RC_HD rc_kmer rc_extend_run(const rc_read_state &S, rc_kmer kc, int k, int dir, int p0, int j)
{
    if (j <= 0) return kc;
    rc_kmer r;
    if (dir > 0) {
        r.code = ((kc.code << (2 * j)) | rc_code_at(S.pk, p0, j)) & rc_kmer_mask(k);
        const uint64_t im = rc_window(S.m_inv, p0, j);  // bit i = base p0+i is not ACGT
        int inv;
        if (im)
            inv = j - 1 - (63 - rc_clz64(im));
        else
            inv = kc.inv == -1 ? -1 : kc.inv + j;
        r.inv = inv >= k ? -1 : inv;
    } else {
        const int lo = p0 - j + 1;
        if (j >= k)
            r.code = rc_code_at(S.pk, lo, k);
        else
            r.code = (kc.code >> (2 * j)) | (rc_code_at(S.pk, lo, j) << (2 * (k - j)));
        const uint64_t im = rc_window(S.m_inv, lo, j);  // bit i = base lo+i, prepended at step j-1-i
        int inv;
        if (im)
            inv = k - 1 - rc_ctz64(im);
        else
            inv = kc.inv == -1 ? -1 : kc.inv - j;
        r.inv = inv < 0 ? -1 : inv;
    }
    return r;
}

// original k-mer of the read starting at base `a`, as the reference builds an anchor (Restart +
// k Appends, ErrorCorrection.cpp:1140-1142 / :1152-1154): a window of the packed read, the
// tracker at the last base of the window that is not ACGT.  Wave-uniform: scalar loads.
template <class W>
RC_HD rc_kmer rc_anchor(W &w, const rc_read_state &S, int k, int a)
{
    rc_kmer kc;
    const int w0 = a >> 4, sh = 2 * (a & 15);
    uint64_t x = ((uint64_t)(uint32_t)RC_U(S.pk[w0]) << 32) | (uint32_t)RC_U(S.pk[w0 + 1]);
    if (sh) x = (x << sh) | ((uint64_t)(uint32_t)RC_U(S.pk[w0 + 2]) >> (32 - sh));
    kc.code = x >> (64 - 2 * k);
    const int mw = a >> 6, ms = a & 63;
    uint64_t im = RC_U64(S.m_inv[mw]) >> ms;
    if (ms) im |= RC_U64(S.m_inv[mw + 1]) << (64 - ms);
    if (k < 64) im &= (1ull << k) - 1ull;
    kc.inv = im ? k - 1 - (63 - rc_clz64(im)) : -1;
    return kc;
}

// The four extension counts of search node (kc, pos).  InferPosThreshold and steps (1)/(3) of the
// reference all look at the same four k-mers, so they are fetched once per node -- and, because a
// node's successor along the keep-base path is known in advance (the read's own next base), the
// extensions of the next RC_SPEC positions are fetched in the SAME gather round: lane (j, c)
// extends kc by the read's bases pos .. pos+j-1 and probes extension c.  Later nodes hit the cache
// iff their k-mer state equals the speculated one; anything else (a substitution, a jump, a
// popped frame) misses and refills from there.  Pure memoisation: results cannot change.
//
// What a round speculates on is a matter of efficiency only, and K1 has already told us a lot: while
// the path is still the read itself, the keep-base count of node j is a count K1 fetched.  Where that
// count is 0 the keep base WILL fail (every threshold is >= 1), so nodes beyond such a node z are not
// worth fetching along the keep path; if z may be substituted (:343 / :577), what comes next are its
// alternatives, and the lanes go to them instead: for each of the three other bases the keep-base
// counts of the nodes behind z on that base's path ("alternative chains", rc_alt_run) -- one probe per
// node where no node of that stretch can have substitution candidates of its own (strong-trusted or
// poly-A positions: the usual case, the k-1 bases behind an isolated error).  A single-error segment
// then costs one gather round instead of three, and the counts it fetched are the new k-mer counts
// GetKmerInformation needs afterwards (rc_read_state::memo).
template <class W>
RC_HD int rc_probe4_cached(W &w, rc_read_state &S, rc_spec_state &Z, const rc_run_params &P, rc_kmer kc, int dir, int pos, int to,
                           rc_cnt4 &cnt)
{
    const int k = RC_K(P);
    if (Z.n > 0 && Z.dir == dir) {
        const int j = (pos - Z.pos) * dir;
        if (j >= 0 && j < Z.n && RC_U64(S.spec_code[j]) == kc.code && RC_U(S.spec_inv[j]) == kc.inv) {
            cnt.c0 = RC_U(S.spec_cnt[4 * j + 0]);
            cnt.c1 = RC_U(S.spec_cnt[4 * j + 1]);
            cnt.c2 = RC_U(S.spec_cnt[4 * j + 2]);
            cnt.c3 = RC_U(S.spec_cnt[4 * j + 3]);
            return j;
        }
    }
    w.stat(3, 1);
    int n = dir > 0 ? (to - pos) : (pos - to + 1);  // nodes left on this side of the segment end
    if (n > RC_SPEC) n = RC_SPEC;
    if (n < 1) n = 1;
    // alternative chains?  z = first node of the keep path whose k-mer K1 found absent
    int am = 0, ana = 0, az = 0, bz = 0;
    const int a0 = dir > 0 ? pos - k : pos + 1;  // first base of this node's k-mer if it is a k-mer of the read
    if (!(P.flags & RC_PF_NO_ALT) && kc.inv == -1 && a0 >= 0 && a0 + k <= S.len) {
        const rc_kmer o = rc_anchor(w, S, k, a0);
        if (o.code == kc.code && o.inv == -1) {
            const uint64_t zm = w.ballot64(0, n, [&](int j) { return (dir > 0 ? S.counts[pos + j - k + 1] : S.counts[pos - j]) == 0; });
            if (zm) {
                const int z = rc_ctz64(zm), pz = pos + dir * z;
                bz = RC_U(S.base[pz]);
                const int pa = RC_U(dir > 0 ? S.polya[pz - k + 1] : S.polya[pz]);
                const int na = dir > 0 ? to - pz - 1 : pz - to;                  // nodes behind z up to the end of the search
                int mm = dir > 0 ? S.len - 1 - pz : pz;                          // k-mers on that side that contain base pz ...
                if (mm > k - 1) mm = k - 1;
                int m = na > mm ? na : mm;                                       // ... are fetched too (for the memo)
                if (z + 1 + m > 31) m = 31 - (z + 1);                            // (rc_extend_run takes at most 31 steps)
                if (4 * (z + 1) + 3 * m > RC_SPEC_ENTRIES) m = (RC_SPEC_ENTRIES - 4 * (z + 1)) / 3;
                if (bz < 4 && !RC_U(S.strongb[pz]) && !(pa & 1) && na >= 0 && na <= m && m >= 1) {
                    // no node of the stretch may have substitution candidates of its own
                    const uint64_t open = w.ballot64(0, na, [&](int j) {
                        const int q = pz + dir * (1 + j);
                        return !S.strongb[q] && !((dir > 0 ? S.polya[q - k + 1] : S.polya[q]) & 1);
                    });
                    if (!open) {
                        am = m;
                        ana = na;
                        az = z;
                        n = z + 1;
                    }
                }
            }
        }
    }
    w.sync();
    const int n4 = 4 * n, E = n4 + 3 * am;
    w.stat(8, E);
    w.stat(9, am > 0 ? 1 : 0);
    // entry e: chain 0, node e / 4, extension e % 4 -- or alternative (e - n4) / am, its node (e - n4) % am,
    // whose k-mer is the read's own k-mer there with base z replaced
    auto entry = [&](int e, bool *live) -> rc_kmer {
        rc_kmer km;
        km.code = 0;
        km.inv = 0;
        *live = e < E;
        if (e < n4) {
            const int j = e >> 2, c = e & 3;
            const rc_kmer kj = rc_extend_run(S, kc, k, dir, pos, j);
            if (c == 0) {
                S.spec_code[j] = kj.code;
                S.spec_inv[j] = kj.inv;
            }
            km = rc_extend(kj, k, dir, c);
        } else if (e < E) {
            const int r = e - n4;
            const int a = (r >= am ? 1 : 0) + (r >= 2 * am ? 1 : 0), j = r - a * am;
            const int ca = a + (a >= bz ? 1 : 0);
            // the k-mer that ends (right) / starts (left) j + 1 positions past z: az + 2 + j steps along the
            // read from this node, with base z -- now j + 1 places inside -- replaced by ca
            km = rc_extend_run(S, kc, k, dir, pos, az + 2 + j);
            if (j + 1 < k) {
                const int sh = dir > 0 ? 2 * (j + 1) : 2 * (k - 1 - (j + 1));
                km.code ^= (uint64_t)(bz ^ ca) << sh;
            }
        }
        return km;
    };
    // (up to two probes per lane, one after the other: both in flight at once would need the registers of
    // two bucket reads, which the kernel does not have at 8 waves per SIMD)
    for (int e0 = 0; e0 < E; e0 += 64) {
        w.for_lanes64(e0, E, [&](int q, int) {
            bool live;
            const rc_kmer km = entry(q, &live);
            S.spec_cnt[q] = w.get(km, dir);
        });
    }
    if (w.lane == 0) {
        S.spec_meta[0] = am;
        S.spec_meta[1] = ana;
        S.spec_meta[2] = az;
    }
    w.sync();
    Z.n = n;
    Z.pos = pos;
    Z.dir = dir;
    cnt.c0 = RC_U(S.spec_cnt[0]);
    cnt.c1 = RC_U(S.spec_cnt[1]);
    cnt.c2 = RC_U(S.spec_cnt[2]);
    cnt.c3 = RC_U(S.spec_cnt[3]);
    return 0;
}

// The node just entered is child `a` (0..2: the alternatives of z in A, C, G, T order) of the cached
// node z: walk its alternative chain.  A node of the chain is known by its keep-base count x alone, and
// that is enough where (i) x >= t: the node's threshold, min(max(GetBound(max of the four counts), 1), t),
// cannot exceed t, so the keep base passes (:302-312 / :539-549); (ii) the position cannot offer
// substitutions (checked when the chain was fetched), so no frame is pending and nothing else happens at
// the node; (iii) left searches hand the node's threshold down as the next t (:546): x >= bs[t] means the
// bound of x, hence of the largest count, reaches t, so the threshold IS t and t stays.  Takes the
// leading nodes that qualify; the first one that does not is an ordinary node again (cache miss, refill).
// Returns the number of nodes taken; *full = the whole chain (also when it has no nodes at all).
template <class W>
RC_HD int rc_alt_run(W &w, rc_read_state &S, const rc_run_params &P, int a, int dir, rc_kmer &kc, int &pos, int t, int &bottleneck,
                     bool *full)
{
    const int am = RC_U(S.spec_meta[0]), na = RC_U(S.spec_meta[1]), az = RC_U(S.spec_meta[2]);
    *full = na == 0;
    if (na == 0 || t < 1) return 0;
    uint32_t Bt = 0;
    if (dir < 0) {
        Bt = RC_U(rc_bs_lookup(P, t));
        if (!Bt) return 0;
    }
    const int eb = 4 * (az + 1) + a * am;
    const uint64_t okm = w.ballot64(0, na, [&](int j) {
        const int x = S.spec_cnt[eb + j];
        return x >= t && (uint32_t)x >= Bt;
    });
    const uint64_t notok = ~okm;
    int R = notok ? rc_ctz64(notok) : 64;
    if (R > na) R = na;
    if (R == 0) return 0;
    w.for_lanes64(0, R, [&](int j, int) { S.path[pos + dir * j] = -1; });
    bottleneck = rc_min(bottleneck, w.min_range(S.spec_cnt, eb, eb + R));
    rc_kmer nk = rc_extend_run(S, kc, RC_K(P), dir, pos, R);
    kc.code = RC_U64(nk.code);
    kc.inv = RC_U(nk.inv);
    pos += dir * R;
    *full = R == na;
    w.stat(6, 1);
    w.stat(7, R == na ? 1 : 0);
    w.sync();
    return R;
}

// terminal bookkeeping, ErrorCorrection.cpp:243-284 / :483-523
// memo_a >= 0: the path ends with the substitution of the cached node z by its alternative memo_a and that
// alternative's whole chain -- if the path is accepted, the counts of the k-mers around z go to the memo
template <class W>
RC_HD void rc_search_terminal(W &w, rc_read_state &S, rc_search_ctx &C, const rc_spec_state &Z, int k, int pos, int t, int fix_cnt,
                              int bottleneck, int memo_a)
{
    if (bottleneck < t) ++fix_cnt;
    if (fix_cnt < C.max_fix_cnt) {
        C.top2a = bottleneck;
        C.top2b = -1;
    } else if (fix_cnt == C.max_fix_cnt) {
        if (bottleneck > C.top2a) {
            C.top2b = C.top2a;
            C.top2a = bottleneck;
        } else if (bottleneck > C.top2b) {
            C.top2b = bottleneck;
        }
    }
    if (fix_cnt < C.max_fix_cnt || (fix_cnt == C.max_fix_cnt && bottleneck > C.best_bottleneck)) {
        if (fix_cnt < C.max_fix_cnt) C.trial_cnt = -(C.max_fix_cnt - fix_cnt + 1) * RC_MAX_TRIAL;
        int lo, hi;  // copy path -> best over [lo, hi)
        if (C.dir > 0) {
            lo = C.start;
            hi = pos;
        } else {
            lo = pos + 1;
            hi = C.start + 1;
        }
        w.sync();
        for (int i = lo + w.lane; i < hi; i += W::STRIDE) S.best[i] = S.path[i];
        if (memo_a >= 0) {
            const int am = RC_U(S.spec_meta[0]), az = RC_U(S.spec_meta[2]);
            const int pz = Z.pos + C.dir * az, bz = RC_U(S.base[pz]);
            const int ca = memo_a + (memo_a >= bz ? 1 : 0);
            const int eb = 4 * (az + 1) + memo_a * am;
            // windows in ascending order: right search: z's own k-mer starts at pz - k + 1, entry j of the chain at
            // pz - k + 2 + j; left search: z's own at pz, entry j at pz - 1 - j
            const int lo_w = C.dir > 0 ? pz - k + 1 : pz - am;
            w.for_lanes64(0, am + 1, [&](int i, int) {
                int c;
                if (C.dir > 0)
                    c = i == 0 ? S.spec_cnt[4 * az + ca] : S.spec_cnt[eb + i - 1];
                else
                    c = i == am ? S.spec_cnt[4 * az + ca] : S.spec_cnt[eb + am - 1 - i];
                S.memo[4 + i] = c;
            });
            if (w.lane == 0) {
                S.memo[0] = am + 1;
                S.memo[1] = lo_w;
                S.memo[2] = pz;
                S.memo[3] = ca;
            }
        }
        w.sync();
        C.max_fix_cnt = fix_cnt;
        C.best_bottleneck = bottleneck;
        C.best_fix_cnt = 1;
    } else if (fix_cnt == C.max_fix_cnt && bottleneck == C.best_bottleneck) {
        C.best_fix_cnt += 1;
    }
}

// InferPosThreshold's clamp (ErrorCorrection.cpp:165-172) given ret = max(GetBound(max),1)
RC_HD int rc_clamp_threshold(int ret, int upper)
{
    if (upper > ret || upper <= 0) return ret;
    return upper;
}

// Keep-run fast path.  The cache holds the extension counts of nodes j0 .. Z.n-1 of the keep-base
// path that starts at the current node.  As long as a node's first child is "keep the base"
// (:302-312 / :539-549) the descent does nothing but: compute the node's threshold, remember a
// frame if substitution candidates exist, set fix[pos] = -1, lower the bottleneck, move on.  All of
// that is evaluated for every cached node at once (one lane per node); the number of leading
// nodes that really keep their base is read off a ballot.  Returns that number R and updates the
// current node to the node where the run stops (whose counts are still cached if R < Z.n - j0).
template <class W>
RC_HD int rc_keep_run(W &w, rc_read_state &S, const rc_run_params &P, const rc_spec_state &Z, int j0, int dir, rc_kmer &kc,
                      int &pos, int &t, int fix_cnt, int &bottleneck, int &sp)
{
    const int k = RC_K(P);
    const int n = Z.n;
    // per node: its threshold clamped to the t handed to the run (right searches hand t down unchanged,
    // so this IS the node's threshold; left searches hand the node's threshold down as the next t, :546,
    // i.e. a running minimum along the path) and the keep-base count
    w.for_lanes64(j0, n, [&](int jj, int) {
        const int c0 = S.spec_cnt[4 * jj], c1 = S.spec_cnt[4 * jj + 1], c2 = S.spec_cnt[4 * jj + 2], c3 = S.spec_cnt[4 * jj + 3];
        int mx = 0;
        mx = c0 > mx ? c0 : mx;
        mx = c1 > mx ? c1 : mx;
        mx = c2 > mx ? c2 : mx;
        mx = c3 > mx ? c3 : mx;
        int ret = rc_bound_i(mx, P.error_rate);
        if (ret < 1) ret = 1;
        S.spec_ret[jj] = rc_clamp_threshold(ret, t);
        const int b = S.base[Z.pos + dir * jj];
        int kc2 = -1;
        kc2 = b == 0 ? c0 : kc2;
        kc2 = b == 1 ? c1 : kc2;
        kc2 = b == 2 ? c2 : kc2;
        kc2 = b == 3 ? c3 : kc2;
        S.spec_keep[jj] = kc2;
    });
    w.sync();
    if (dir > 0) {
        w.for_lanes64(j0, n, [&](int jj, int) { S.spec_thr[jj] = S.spec_ret[jj]; });
    } else {
        // clamp(ret_j, clamp(ret_j-1, ... t)) = the minimum of the clamped values so far (every ret is >= 1, and a
        // non-positive t only lets the first node's own value through, which spec_ret already holds)
        w.prefix_min(S.spec_ret, S.spec_thr, j0, n);
    }
    w.sync();
    const uint64_t okm = w.ballot64(j0, n, [&](int jj) { return S.spec_keep[jj] >= S.spec_thr[jj]; });
    // leading ones of okm
    const uint64_t notok = ~okm;
    int R = notok ? rc_ctz64(notok) : 64;
    if (R > n - j0) R = n - j0;
    w.stat(0, 1);
    w.stat(1, R);
    w.stat(2, n - j0);
    if (R == 0) return 0;
    // substitution candidates of the kept nodes, fix[pos] = -1
    const uint64_t fm = w.ballot64(j0, j0 + R, [&](int jj) {
        const int p2 = Z.pos + dir * jj;
        const int b = S.base[p2];
        const int pa = dir > 0 ? S.polya[p2 - k + 1] : S.polya[p2];
        int m2 = 0;
        if (!S.strongb[p2] && !(pa & 1)) {
            const int thr = S.spec_thr[jj];
            for (int c = 0; c < 4; ++c)
                if (c != b && S.spec_cnt[4 * jj + c] >= thr) m2 |= 1 << c;
        }
        S.spec_mask[jj] = m2;
        S.path[p2] = -1;
        return m2 != 0;
    });
    w.sync();
    // frames for the branch points, in path order
    uint64_t f2 = fm;
    while (f2) {
        const int jj = j0 + rc_ctz64(f2);
        f2 &= f2 - 1;
        rc_frame f;
        f.code = RC_U64(S.spec_code[jj]);
        f.inv = RC_U(S.spec_inv[jj]);
        f.pos = Z.pos + dir * jj;
        f.t = (dir > 0 || jj == j0) ? t : RC_U(S.spec_thr[jj - 1]);
        f.threshold = RC_U(S.spec_thr[jj]);
        f.fix_cnt = fix_cnt;
        int bb = bottleneck;
        for (int i = j0; i < jj; ++i) bb = rc_min(bb, RC_U(S.spec_keep[i]));
        f.bottleneck = bb;
        f.cnt.c0 = RC_U(S.spec_cnt[4 * jj + 0]);
        f.cnt.c1 = RC_U(S.spec_cnt[4 * jj + 1]);
        f.cnt.c2 = RC_U(S.spec_cnt[4 * jj + 2]);
        f.cnt.c3 = RC_U(S.spec_cnt[4 * jj + 3]);
        f.mask = RC_U(S.spec_mask[jj]);
        w.stack_push(sp, f);
        ++sp;
    }
    // the node the run stops at
    const int idx = j0 + R;
    bottleneck = rc_min(bottleneck, w.min_range(S.spec_keep, j0, idx));
    if (dir < 0) t = RC_U(S.spec_thr[idx - 1]);
    if (idx < n) {
        kc.code = RC_U64(S.spec_code[idx]);
        kc.inv = RC_U(S.spec_inv[idx]);
    } else {
        rc_kmer last;
        last.code = RC_U64(S.spec_code[idx - 1]);
        last.inv = RC_U(S.spec_inv[idx - 1]);
        kc = rc_extend(last, k, dir, RC_U(S.base[Z.pos + dir * (idx - 1)]));
    }
    pos = Z.pos + dir * idx;
    return R;
}

// one SearchPaths_Right/_Left call tree (ErrorCorrection.cpp:201-442 / :444-678), depth-first
// in the reference's visiting order, with an explicit stack that only holds nodes whose
// substitution alternatives are still pending.
template <class W>
RC_HD void rc_search(W &w, rc_read_state &S, const rc_run_params &P, rc_search_ctx &C, rc_kmer kc0,
                     int t0)
{
    const int k = RC_K(P);
    const int dir = C.dir;
    int sp = 0;
    // current node
    rc_kmer kc = kc0;
    int pos = C.start, t = t0, fix_cnt = 0, bottleneck = 1000000000;
    bool have = true;
    rc_spec_state Z;
    Z.n = 0;
    Z.pos = 0;
    Z.dir = dir;
    int alt_a = -1;  // >= 0: the node being entered is this alternative of the cached node z (rc_probe4_cached)
    // is (code, inv, p) the cached node the alternative chains hang off?  then child c of it has a chain
    auto alt_of = [&](uint64_t code, int inv, int p, int c) -> int {
        if (Z.n <= 0 || RC_U(S.spec_meta[0]) <= 0) return -1;
        const int az = RC_U(S.spec_meta[2]);
        if (p != Z.pos + dir * az || RC_U64(S.spec_code[az]) != code || RC_U(S.spec_inv[az]) != inv) return -1;
        const int bz = RC_U(S.base[p]);
        return c < bz ? c : c - 1;
    };

    for (;;) {
        w.phase(8);
        if (!have) {
            if (sp == 0) break;
            rc_frame f;
            w.stack_top(sp - 1, f);
            int c = 0;
            while (!((f.mask >> c) & 1)) ++c;
            f.mask &= ~(1 << c);
            if (f.mask == 0)
                --sp;
            else
                w.stack_set_mask(sp - 1, f.mask);
            // substitution child c of node f (ErrorCorrection.cpp:345-371 / :579-607)
            rc_kmer fk;
            fk.code = f.code;
            fk.inv = f.inv;
            S.path[f.pos] = (signed char)c;
            w.sync();
            ++C.trial_cnt;
            alt_a = alt_of(f.code, f.inv, f.pos, c);
            kc = rc_extend(fk, k, dir, c);
            pos = f.pos + dir;
            t = dir > 0 ? f.t : f.threshold;
            fix_cnt = f.fix_cnt + (RC_U(S.base[f.pos]) < 4 ? 1 : 0);
            bottleneck = rc_min(f.bottleneck, rc_sel4(f.cnt, c));
            have = true;
        }

        // entry gate, ErrorCorrection.cpp:211-224 / :453-467
        if (C.trial_cnt > RC_MAX_TRIAL) {
            if (C.max_fix_cnt > 2) {
                --C.max_fix_cnt;
                C.best_fix_cnt = 0;
                C.best_bottleneck = -1;
                C.trial_cnt = 0;
            } else {
                have = false;
                continue;
            }
        }
        if (fix_cnt > C.max_fix_cnt) {
            have = false;
            alt_a = -1;
            continue;
        }
        int memo_a = -1;
        if (alt_a >= 0) {  // down the alternative's chain (no node of it changes trial_cnt or fix_cnt: the gate above holds for all)
            bool full;
            rc_alt_run(w, S, P, alt_a, dir, kc, pos, t, bottleneck, &full);
            if (full) memo_a = alt_a;
            alt_a = -1;
        }
        if (dir > 0 ? (pos >= C.to) : (pos < C.to)) {
            w.phase(14);
            rc_search_terminal(w, S, C, Z, k, pos, t, fix_cnt, bottleneck, memo_a);
            have = false;
            continue;
        }

        rc_cnt4 cnt;
        w.phase(9);
        const int j0 = rc_probe4_cached(w, S, Z, P, kc, dir, pos, C.to, cnt);
        w.phase(10);
        // descend along "keep the base" for as many cached nodes as take that branch
        if (rc_keep_run(w, S, P, Z, j0, dir, kc, pos, t, fix_cnt, bottleneck, sp) > 0) {
            have = true;
            continue;
        }
        w.phase(11);
        int threshold = RC_U(rc_pos_threshold(cnt, t, P.error_rate));  // :287 / :525
        const int b = RC_U(S.base[pos]);
        const bool bvalid = b < 4;

        int mask = 0;  // substitution candidates, :343-352 / :577-588
        {
            const int pa = RC_U(dir > 0 ? S.polya[pos - k + 1] : S.polya[pos]);
            if (!RC_U(S.strongb[pos]) && !(pa & 1)) mask = rc_candidates(cnt, b, threshold);
        }

        bool first = false;
        rc_kmer nkc = kc;
        int npos = pos, nt = t, nfix = fix_cnt, nbott = bottleneck;
        if (bvalid) {
            int c0 = rc_sel4(cnt, b);
            if (c0 >= threshold) {  // keep the base, :302-312 / :539-549
                S.path[pos] = -1;
                nkc = rc_extend(kc, k, dir, b);
                npos = pos + dir;
                nt = dir > 0 ? t : threshold;
                nbott = rc_min(bottleneck, c0);
                first = true;
            } else if (threshold == 1 && t <= 2) {  // accidental gap, :313-338 / :550-573
                // the reference extends one base at a time until the count recovers, giving up after
                // k probes (the k-th probe's answer is never used) or at the segment end.  All
                // candidate windows are probed in one gather round; the first hit is the one the
                // sequential loop would have stopped at.
                w.phase(12);
                rc_kmer tmp = rc_extend(kc, k, dir, b);
                int m = dir > 0 ? (C.to - 1 - pos) : (pos - C.to);  // positions left in range
                if (m > k - 1) m = k - 1;
                int steps = k, i = pos;
                (void)c0;
                if (m > 0) {
                    const rc_kmer tmp0 = tmp;
                    // window sN is the keep-base extension of the node sN steps down the keep path, and
                    // those were fetched with this node's round (spec_keep[], set by rc_keep_run above):
                    // only windows beyond the cached nodes need a round of their own
                    int mc = Z.n - 1 - j0;
                    if (mc > m) mc = m;
                    int sN = 0;
                    if (mc > 0) {
                        const uint64_t hit = w.ballot64(1, mc + 1, [&](int q) { return S.spec_keep[j0 + q] >= threshold; });
                        if (hit) sN = 1 + rc_ctz64(hit);
                    }
                    if (sN == 0 && m > mc) {
                        const uint64_t hit = w.ballot64(mc + 1, m + 1, [&](int q) {
                            return w.get(rc_extend_run(S, tmp0, k, dir, pos + dir, q), dir) >= threshold;
                        });
                        w.stat(4, 1);
                        w.stat(5, m - mc);
                        if (hit) sN = mc + 1 + rc_ctz64(hit);
                    }
                    if (sN > 0) {
                        tmp = rc_extend_run(S, tmp0, k, dir, pos + dir, sN);
                        tmp.code = RC_U64(tmp.code);
                        tmp.inv = RC_U(tmp.inv);
                        i = pos + dir * sN;
                        steps = sN;
                    }
                }
                if (steps < k && (dir > 0 ? (i < C.to) : (i >= C.to))) {
                    int lo = dir > 0 ? pos : i, hi = dir > 0 ? i : pos;
                    for (int j = lo + w.lane; j <= hi; j += W::STRIDE) S.path[j] = -1;
                    nkc = tmp;
                    npos = i + dir;
                    nt = dir > 0 ? t : threshold;
                    nfix = fix_cnt + 1;
                    first = true;
                }
            }
        }

        if (!first && mask) {  // first child is a substitution
            int c = 0;
            while (!((mask >> c) & 1)) ++c;
            mask &= ~(1 << c);
            S.path[pos] = (signed char)c;
            ++C.trial_cnt;
            alt_a = alt_of(kc.code, kc.inv, pos, c);
            nkc = rc_extend(kc, k, dir, c);
            npos = pos + dir;
            nt = dir > 0 ? t : threshold;
            nfix = fix_cnt + (bvalid ? 1 : 0);
            nbott = rc_min(bottleneck, rc_sel4(cnt, c));
            first = true;
        }

        if (first) {
            if (mask) {
                rc_frame f;
                f.code = kc.code;
                f.inv = kc.inv;
                f.pos = pos;
                f.t = t;
                f.threshold = threshold;
                f.fix_cnt = fix_cnt;
                f.bottleneck = bottleneck;
                f.cnt = cnt;
                f.mask = mask;
                w.stack_push(sp, f);
                ++sp;
            }
        } else {
            // jump over an unfixable stretch, :393-441 / :629-677
            w.phase(13);
            rc_kmer tmp = kc;
            int i, c1 = 0;
            int thr = threshold;
            if (dir > 0) {
                for (i = pos; i < C.to; ++i) {
                    rc_cnt4 cn;
                    if (i == pos)
                        cn = cnt;
                    else
                        rc_probe4_cached(w, S, Z, P, tmp, dir, i, C.to, cn);
                    thr = RC_U(rc_pos_threshold(cn, t, P.error_rate));
                    int bb = RC_U(S.base[i]);
                    tmp = rc_append(tmp, k, bb);
                    c1 = (bb < 4) ? rc_sel4(cn, bb) : 0;
                    S.path[i] = -1;
                    if (c1 >= thr) break;
                }
                int pen = (i < S.len) ? (i - pos - k + 1) : ((i - pos) / 2);
                if (pen <= 0) pen = 1;
                nfix = fix_cnt + pen;
                if (i >= C.to) i -= 1;
                npos = i + 1;
                nt = t;
            } else {
                for (i = pos; i >= C.to; --i) {
                    rc_cnt4 cn;
                    if (i == pos)
                        cn = cnt;
                    else
                        rc_probe4_cached(w, S, Z, P, tmp, dir, i, C.to, cn);
                    thr = RC_U(rc_pos_threshold(cn, t, P.error_rate));
                    int bb = RC_U(S.base[i]);
                    tmp = rc_prepend(tmp, k, bb);
                    c1 = (bb < 4) ? rc_sel4(cn, bb) : 0;
                    S.path[i] = -1;
                    if (c1 >= thr) break;
                }
                int pen = (i >= 0) ? (pos - i - k + 1) : (pos - 1);
                if (pen <= 0) pen = 1;
                nfix = fix_cnt + pen;
                if (i <= C.to) ++i;  // :672
                npos = i - 1;
                nt = thr;
            }
            nkc = tmp;
            nbott = bottleneck;
        }
        w.sync();
        kc = nkc;
        pos = npos;
        t = nt;
        fix_cnt = nfix;
        bottleneck = nbott;
        have = true;
    }
}

// ErrorCorrection (ErrorCorrection.cpp:682-1480).  On entry base[], counts[], polya[] are
// loaded; strong0/info0 are this read's rc_front_end() results (from the threshold kernel) and
// pair_t is min(strong of both mates) or -1.  Returns the reference's return value; best[]
// holds the fixes to apply when the return value is > 0.
template <class W>
RC_HD int rc_correct_read(W &w, rc_read_state &S, const rc_run_params &P, int pair_t, int strong0,
                          int info0)
{
    const int k = RC_K(P);
    const int len = S.len, kcnt = S.kcnt;
    if (w.lane == 0) S.memo[0] = 0;
    if (len < k) return -1;   // :713
    if (info0 & 4) return -1; // screens, :735-755
    w.trace_passed();         // -verbose prints "Before correction" from here on, :759-770

    // initial thresholds, :793-842
    int strong = strong0, trust;
    bool flag = false;
    trust = RC_U(rc_bound_i(strong, P.error_rate));
    if (info0 & 1) {
        if (strong >= 20 && (info0 & 2) && trust < 3) {
            flag = true;
            trust = 3;
        }
    }
    if (pair_t >= 1 && strong > pair_t) {
        if (!flag || pair_t < 20) trust = RC_U(rc_bound_i(pair_t, P.error_rate));
        strong = pair_t;
    }
    if (trust < 2) trust = 2;

    int iter = 0, total_fix = 0, bad_segment_cnt = 0;
    int tstart = 0, tend = 0;
    for (;;) {  // :854-1291
        w.phase(2);
        w.trace_iter(strong, trust);  // :856-857
        const int allowed_fix = len;
        total_fix = 0;
        bool unfixable = false, force_next = false;
        int isl_cnt = 0, longest = -1, j = 0, i;

        for (i = w.lane; i < len; i += W::STRIDE) {
            S.strongb[i] = 0;
            S.path[i] = -1;
        }
        w.sync();

        // trusted k-mer islands, :870-931.  trusted[i] = counts[i] >= strong && !IsPolyA(i); the
        // maximal runs come out of the bit mask (run starts / ends), so the cost is O(#runs).
        {
            const int nwk = (kcnt + 63) >> 6;
            for (int c = 0; c <= nwk; ++c) {
                const uint64_t tm = w.ballot64(c << 6, kcnt, [&](int q) { return S.counts[q] >= strong && !(S.polya[q] & 1); });
                S.m_x[c] = tm;
            }
            w.sync();
            int *rs = S.v;                      // run starts
            int *re = S.v + (kcnt + 1) / 2 + 1; // run ends
            int ns = 0, ne = 0;
            for (int c = 0; c < nwk; ++c) {
                const uint64_t T = RC_U64(S.m_x[c]);
                const uint64_t prev = c ? (RC_U64(S.m_x[c - 1]) >> 63) : 0ull;
                const uint64_t next = RC_U64(S.m_x[c + 1]) & 1ull;
                uint64_t st = T & ~((T << 1) | prev);
                uint64_t en = T & ~((T >> 1) | (next << 63));
                while (st) {
                    rs[ns++] = (c << 6) + rc_ctz64(st);
                    st &= st - 1;
                }
                while (en) {
                    re[ne++] = (c << 6) + rc_ctz64(en);
                    en &= en - 1;
                }
            }
            w.sync();
            if (!(RC_U64(S.m_x[0]) & 1ull)) {  // first k-mer untrusted: the scan records a run of length 0 first
                longest = 0;
                tstart = 0;
                tend = -1;
            }
            for (int r = 0; r < ns; ++r) {
                const int f = RC_U(rs[r]), t2 = RC_U(re[r]);
                j = t2 - f + 1;
                if (j > longest) {
                    longest = j;
                    tstart = f;
                    tend = t2;
                }
                if (j >= 2) {
                    S.isl[isl_cnt].from = (short)f;
                    S.isl[isl_cnt].to = (short)t2;
                    ++isl_cnt;
                }
            }
            w.sync();
        }

        // boundary adjustment, :934-965
        for (i = 1; i < isl_cnt; ++i) {
            int pf = RC_U(S.isl[i - 1].from), pt = RC_U(S.isl[i - 1].to), cf = RC_U(S.isl[i].from), ct = RC_U(S.isl[i].to);
            if (cf <= pt + k) {
                int len1 = pt - pf, len2 = ct - cf, overlap = pt + k - cf;
                // is there a k-mer strictly between with counts <= 2 && counts < trust?  (gap < k <= 32)
                const uint64_t weak = w.ballot64(pt + 1, cf, [&](int q) { return S.counts[q] <= 2 && S.counts[q] < trust; });
                if (!weak) continue;
                if (overlap > 3) continue;
                if (len1 < len2)
                    S.isl[i - 1].to = (short)(pt - (overlap + 1));
                else
                    S.isl[i].from = (short)(cf + (overlap + 1));
                w.sync();
            }
        }

        // to base space, :968-1007: island [from,to] of k-mers covers bases [from, to+k-1]; the
        // maximal runs of covered bases are the union of these (sorted) intervals
        {
            int nb = 0, cura = 0, curb = -2;
            for (i = 0; i < isl_cnt; ++i) {
                const int f = RC_U(S.isl[i].from), tt = RC_U(S.isl[i].to);
                if (f > tt) continue;
                const int a2 = f, b2 = tt + k - 1;
                for (j = a2 + w.lane; j <= b2; j += W::STRIDE) S.strongb[j] = 1;
                if (nb > 0 && a2 <= curb + 1) {
                    if (b2 > curb) curb = b2;
                } else {
                    if (nb > 0) {
                        S.isl[nb - 1].from = (short)cura;
                        S.isl[nb - 1].to = (short)curb;
                    }
                    cura = a2;
                    curb = b2;
                    ++nb;
                }
            }
            if (nb > 0) {
                S.isl[nb - 1].from = (short)cura;
                S.isl[nb - 1].to = (short)curb;
            }
            isl_cnt = nb;
        }
        if (isl_cnt == 0) {
            S.isl[0].from = (short)tstart;
            S.isl[0].to = (short)(tend + k - 1);
            isl_cnt = 1;
        }
        w.sync();

        // segments, :1009-1046
        int seg_cnt = 0;
        {
            int f0 = RC_U(S.isl[0].from), t0 = RC_U(S.isl[0].to);
            if (f0 > 0) {
                S.seg[seg_cnt].from = 0;
                S.seg[seg_cnt].to = (short)(f0 - 1);
                S.seg[seg_cnt].lanchor = 0;
                S.seg[seg_cnt].ranchor = (short)(t0 - f0 + 1);
                ++seg_cnt;
            }
            for (i = 0; i < isl_cnt - 1; ++i) {
                int af = RC_U(S.isl[i].from), at = RC_U(S.isl[i].to), bf = RC_U(S.isl[i + 1].from), bt = RC_U(S.isl[i + 1].to);
                S.seg[seg_cnt].from = (short)(at + 1);
                S.seg[seg_cnt].to = (short)(bf - 1);
                S.seg[seg_cnt].lanchor = (short)(at - af + 1);
                S.seg[seg_cnt].ranchor = (short)(bt - bf + 1);
                ++seg_cnt;
            }
            int lf = RC_U(S.isl[i].from), lt = RC_U(S.isl[i].to);
            if (lt < len - 1) {
                S.seg[seg_cnt].from = (short)(lt + 1);
                S.seg[seg_cnt].to = (short)len;
                S.seg[seg_cnt].lanchor = (short)(lt - lf + 1);
                S.seg[seg_cnt].ranchor = 0;
                ++seg_cnt;
            }
            for (i = 0; i < seg_cnt; ++i) S.seg[i].top2[0] = S.seg[i].top2[1] = -1;
        }
        w.sync();
        w.trace_strong(S.strongb, len);  // :1088-1094

        if (longest == -1) return -1;  // :1107
        if (longest == kcnt) return 0; // :1110
#if defined(RC_EXP_STOP) && RC_EXP_STOP == 3  // dev builds (tools/exp_stops.sh): cost of the phases up to here
        return 0;
#endif

        for (i = w.lane; i < len; i += W::STRIDE) S.best[i] = -1;
        w.sync();
        bad_segment_cnt = 0;
        w.phase(3);
        if (seg_cnt > 0) {  // :1118-1230
            rc_search_ctx C;
            int best_bottleneck = RC_INF;
            C.best_fix_cnt = -1;
            C.trial_cnt = 0;
            C.max_fix_cnt = allowed_fix;
            for (int si = 0; si < seg_cnt; ++si) {
                const int sf = RC_U(S.seg[si].from), st = RC_U(S.seg[si].to);
                C.top2a = C.top2b = -1;
                C.trial_cnt = 0;
                C.max_fix_cnt = (st - sf + 1) * P.max_fix_per_k / k * 2 + 1;
                if (C.max_fix_cnt < P.max_fix_per_k) C.max_fix_cnt = P.max_fix_per_k;
                C.best_bottleneck = -1;
                int a;  // first base of the anchor k-mer
                if (RC_U(S.seg[si].lanchor) >= RC_U(S.seg[si].ranchor)) {
                    int extend = (st == len) ? 0 : (k - 1);
                    a = sf - k;
                    if (a < 0) return -1;  // the reference reads seq[-1] here (undefined)
                    C.dir = 1;
                    C.start = a + k;
                    C.to = st + extend;
                } else {
                    int extend = (sf == 0) ? 0 : (k - 1);
                    a = st + 1;
                    if (a + k > len) return -1;  // undefined in the reference
                    C.dir = -1;
                    C.start = a - 1;
                    C.to = sf - extend;
                }
                rc_search(w, S, P, C, rc_anchor(w, S, k, a), trust);  // one body for both directions
                w.phase(3);
                S.seg[si].top2[0] = C.top2a;
                S.seg[si].top2[1] = C.top2b;
                w.sync();
                if (C.best_bottleneck == -1) {
                    ++bad_segment_cnt;
                    continue;
                }
                if (C.best_bottleneck < best_bottleneck) best_bottleneck = C.best_bottleneck;
                if (best_bottleneck == -1) break;
                if (C.trial_cnt > RC_MAX_TRIAL) return -1;
                total_fix += C.max_fix_cnt;
            }
            int best_fix_cnt = C.best_fix_cnt;
            if (best_bottleneck != -1) {  // :1178-1192
                best_fix_cnt = 1;
                for (i = 0; i < seg_cnt; ++i)
                    if (RC_U(S.seg[i].top2[1]) >= best_bottleneck) best_fix_cnt *= 2;
            }
            if (best_bottleneck != -1 && iter == 0 &&
                RC_U(rc_less_than_bound(best_bottleneck, strong, P.error_rate) ? 1 : 0))  // :1195
                force_next = true;
            if (best_fix_cnt >= 2)
                return -1;
            else if (best_fix_cnt <= 0)
                unfixable = true;
        }
        if (total_fix == 0 && force_next) return 0;  // :1231
        if (total_fix > allowed_fix) unfixable = true;
        if (!unfixable && !force_next) break;
        if (trust < 10 && !force_next) return -1;  // :1241

        // lower the thresholds, :1247-1289
        w.phase(4);
        rc_masked_sorted(w, S);
        // the scan of :1258-1277 from the top of the sorted counts: a "drop" at i is v[i] <= strong with
        // v[i] > 2 v[i-1] && v[i] > 10, or v[i-1] == 0 && v[i] >= 5; the scan stops at the highest drop
        // below strong (at index 0 if there is none) and any drop at all lowers the thresholds to v[stop]
        bool has_drop = false;
        i = 0;
        for (int b0 = ((kcnt - 1) >> 6) << 6; b0 >= 0; b0 -= 64) {
            const int st = strong;
            const uint64_t dm = w.ballot64(b0, kcnt, [&](int q) {
                if (q < 1) return false;
                const int vi = S.v[q], vp = S.v[q - 1];
                return vi <= st && ((vi > 2 * vp && vi > 10) || (vp == 0 && vi >= 5));
            });
            const uint64_t bm = dm & w.ballot64(b0, kcnt, [&](int q) { return S.v[q] < st; });
            if (bm) {  // drops above the stop (in this chunk) count too
                has_drop = true;
                i = b0 + 63 - rc_clz64(bm);
                break;
            }
            if (dm) has_drop = true;
        }
        if (has_drop) {
            ++iter;
            int vi = RC_U(S.v[i]);
            trust = RC_U(rc_bound_i(vi, P.error_rate));
            strong = vi;
        } else
            break;
    }

    // ---- post filters (positions list in v[]) ----
    w.phase(5);
#if defined(RC_EXP_STOP) && RC_EXP_STOP == 4
    return total_fix;
#endif
    int cnt = 0;  // positions with a fix on a non-N base, ascending, :1296-1303
    for (int b0 = 0; b0 < len; b0 += 64) {
        const uint64_t fm = w.ballot64(b0, len, [&](int q) { return S.base[q] != 4 && S.best[q] != -1; });
        if (fm) {
            w.for_lanes64(b0, len, [&](int q, int ln) {
                if ((fm >> ln) & 1ull) S.v[cnt + rc_popc64(fm & ((1ull << ln) - 1ull))] = q;
            });
            cnt += rc_popc64(fm);
        }
    }
    w.sync();
    const int badq = P.bad_qual;
    const int q0 = RC_U(S.qual[0]);
    for (int i = 1; i < cnt; ++i) {  // pairwise veto, :1314-1398
        // two neighbouring fixes pp < pi within one k-mer span: compare the weakest k-mer that covers
        // only one of them with the weakest that covers both (k-mers reaching the fix before pp or the
        // one after pi are left out).  All candidate k-mers lie in [pp-k+1, pi] -- at most 2k-1 <= 63 of
        // them, one per lane -- so the three scans of the reference are two wave minima.
        const int pi = RC_U(S.v[i]), pp = RC_U(S.v[i - 1]);
        if (q0 != 0 && (RC_U(S.qual[pi]) <= badq && RC_U(S.qual[pp]) <= badq)) continue;
        if (pi - pp + 1 > k) continue;
        int lo = pp - k + 1;                     // first k-mer that covers pp ...
        if (lo < 0) lo = 0;
        if (i >= 2) {                            // ... and starts after the previous fix
            const int pprev = RC_U(S.v[i - 2]);
            if (lo < pprev + 1) lo = pprev + 1;
        }
        int hi = kcnt;                           // k-mers must end before the next fix
        if (i < cnt - 1) {
            const int pnext = RC_U(S.v[i + 1]);
            if (hi > pnext - k + 1) hi = pnext - k + 1;
        }
        uint32_t ms = 0xFFFFFFFFu, md = 0xFFFFFFFFu;
        w.for_lanes64(pp - k + 1, hi, [&](int j, int) {
            if (j < lo || j > pi) return;
            const uint32_t c = (uint32_t)S.counts[j];
            if (j + k - 1 >= pi && j <= pp)      // covers both fixes
                md = c < md ? c : md;
            else                                 // covers pp only (ends before pi) or pi only (starts after pp)
                ms = c < ms ? c : ms;
        });
        ms = w.wave_min_u32(ms);
        md = w.wave_min_u32(md);
        if (ms == 0xFFFFFFFFu || md == 0xFFFFFFFFu) continue;
        const int min_single = (int)ms, min_double = (int)md;
        if (min_single > 1 && min_double > 1 && min_single > min_double / 2 && min_single < 2 * min_double) {
            // drop both, and every fix chained to them by gaps of at most k
            S.best[pi] = -1;
            S.best[pp] = -1;
            for (int jj = i - 2; jj >= 0 && RC_U(S.v[jj + 1]) - RC_U(S.v[jj]) + 1 <= k; --jj) S.best[RC_U(S.v[jj])] = -1;
            while (i + 1 < cnt && RC_U(S.v[i + 1]) - RC_U(S.v[i]) + 1 <= k) {
                S.best[RC_U(S.v[i + 1])] = -1;
                ++i;
            }
            w.sync();
        }
    }

    if (total_fix > 3 && len > 10) {  // end-of-read veto, :1407-1430
        int tmp = rc_popc64(w.ballot64(0, 10, [&](int q) { return S.best[q] != -1 && S.base[q] != 4 && (int)S.qual[q] > badq; }));
        if (tmp >= 2)
            w.for_lanes64(0, 10, [&](int q, int) {
                if (S.base[q] != 4) S.best[q] = -1;
            });
        w.sync();
        tmp = rc_popc64(w.ballot64(len - 10, len, [&](int q) { return S.best[q] != -1 && S.base[q] != 4 && (int)S.qual[q] > badq; }));
        if (tmp >= 3)
            w.for_lanes64(len - 10, len, [&](int q, int) {
                if (S.base[q] != 4) S.best[q] = -1;
            });
        w.sync();
    }

    if (total_fix >= P.max_fix_per_k) {  // density veto, :1432-1466, weights x2
        // a fix on a non-N base weighs 1, or 2 where the quality is good; a k-window heavier than
        // 2 MAX_FIX_PER_K rejects the read.  Two bit masks (the letter masks m_a / m_t are no longer
        // needed at this point and serve as scratch) turn the window sums into popcounts.
        const int nw = (len + 63) >> 6;
        for (int c = 0; c <= nw; ++c) {
            const uint64_t f = w.ballot64(c << 6, len, [&](int q) { return S.base[q] != 4 && S.best[q] != -1; });
            const uint64_t g = f & w.ballot64(c << 6, len, [&](int q) { return (int)S.qual[q] > badq; });
            S.m_a[c] = f;
            S.m_t[c] = g;
        }
        w.sync();
        int bad = 0;
        for (int i = w.lane; i < kcnt; i += W::STRIDE)
            if (rc_popc64(rc_window(S.m_a, i, k)) + rc_popc64(rc_window(S.m_t, i, k)) > 2 * P.max_fix_per_k) bad = 1;
        if (w.reduce_add(bad)) return -1;
    }

    int ret = 0;  // :1468-1479
    for (int i = w.lane; i < len; i += W::STRIDE)
        if (S.best[i] != -1) ++ret;
    ret = w.reduce_add(ret);
    if (ret == 0 && bad_segment_cnt > 0) return -1;
    return ret;
}

// rank-`kth` (0-based, ascending) of the n unsigned values v[], none above `hi` except
// 0xFFFFFFFF place holders: a radix descent over the bits of `hi`, one ballot per bit and 64
// values -- for the few bits k-mer counts use this is an order of magnitude cheaper than sorting
template <class W>
RC_HD int rc_select_kth(W &w, const int *v, int n, int kth, uint32_t hi)
{
    uint32_t prefix = 0;
    for (int b = 31 - rc_clz32(hi | 1u); b >= 0; --b) {
        const uint32_t cand = prefix | (1u << b);
        int below = 0;
        for (int b0 = 0; b0 < n; b0 += 64) below += rc_popc64(w.ballot64(b0, n, [&](int q) { return (uint32_t)v[q] < cand; }));
        if (below <= kth) prefix = cand;
    }
    return (int)prefix;
}

// GetKmerInformation (ErrorCorrection.cpp:1567-1602) on the read after `ret` fixes were
// applied to base[] and pk[] (so both already hold the corrected bases where best[i] != -1).
// counts[] still holds the pre-correction counts; only windows touching a fix are re-probed.
// l and h are wave reductions, m = sorted[n/2] is a rank selection: no sort.
template <class W>
RC_HD void rc_kmer_info(W &w, rc_read_state &S, const rc_run_params &P, int ret, int *l, int *m,
                        int *h)
{
    const int k = RC_K(P);
    *l = *m = *h = 0;
    if (S.kcnt <= 0) return;
    // fixed positions (their base is now one of ACGT, so they leave the invalid mask)
    const int nw = (S.len + 63) >> 6;
    for (int c = 0; c <= nw; ++c) {
        uint64_t fm = 0;
        if (ret > 0) fm = w.ballot64(c << 6, S.len, [&](int q) { return S.best[q] != -1; });
        S.m_x[c] = fm;
    }
    w.sync();
    // counts a search already fetched for the windows around its one fix p -> c: good for a window whose only
    // fix in the final result is that one (the count of a k-mer is a function of its letters)
    int memo_n = 0, memo_lo = 0, memo_p = 0;
    if (ret > 0) {
        memo_n = RC_U(S.memo[0]);
        if (memo_n > 0) {
            memo_lo = RC_U(S.memo[1]);
            memo_p = RC_U(S.memo[2]);
            if (RC_U(S.best[memo_p]) != RC_U(S.memo[3])) memo_n = 0;
        }
    }
    int nvalid = 0;
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    for (int b0 = 0; b0 < S.kcnt; b0 += 64) {
        const uint64_t vm = w.ballot64(b0, S.kcnt, [&](int q) {
            return (rc_window(S.m_inv, q, k) & ~rc_window(S.m_x, q, k)) == 0;
        });
        nvalid += rc_popc64(vm);
        w.for_lanes64(b0, S.kcnt, [&](int q, int ln) {
            uint32_t c = 0xFFFFFFFFu;
            if ((vm >> ln) & 1ull) {
                int cc;
                const uint64_t fw = ret > 0 ? rc_window(S.m_x, q, k) : 0ull;
                if (fw != 0) {  // window holds a fix: the new k-mer's count, from the memo or the table
                    const int d = memo_p - q;
                    if (q >= memo_lo && q < memo_lo + memo_n && d >= 0 && d < k && fw == (1ull << d))
                        cc = S.memo[4 + q - memo_lo];
                    else
                        cc = w.lookup(rc_code_at(S.pk, q, k));
                } else
                    cc = S.counts[q];
                if (cc == 0) cc = 1;
                c = (uint32_t)cc;
                lo = c < lo ? c : lo;
                hi = c > hi ? c : hi;
            }
            S.v[q] = (int)c;
        });
    }
    w.sync();
    if (nvalid == 0) return;
    lo = w.wave_min_u32(lo);
    hi = w.wave_max_u32(hi);
    *l = (int)lo;
    *h = (int)hi;
    *m = rc_select_kth(w, S.v, S.kcnt, nvalid / 2, hi);
}
