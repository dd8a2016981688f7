// rc_single.h -- K2s, "isolated substitutions": reads whose whole correction is one substitution per untrusted stretch
// are finished here, four reads per wavefront (one per 16-lane row, as rc_quarter.h), and never reach k_correct.
//
// What a sequencing error in the middle of a well-covered read looks like to ErrorCorrection (ErrorCorrection.cpp:682-1480):
// the k k-mers that contain base p fall below the strong threshold, every other k-mer stays above it.  The function then
// builds two islands around p (:870-1007), one single-base segment [p, p] (:1009-1046), searches it from the larger
// anchor (:1125-1175): the first node offers the base's alternatives (:343-371 / :577-607), the k-1 (right) or k (left)
// nodes behind it lie inside the other island -- no substitutions there (:343 / :577), only "keep the base" -- the path
// ends with one fix, nothing else is ever explored, no veto applies to a lone fix (:1296-1466), the base is replaced
// (:1468-1479) and GetKmerInformation re-counts the k k-mers around it (:1567-1602).  This kernel replays exactly that --
// and ONLY that: every step is guarded by the condition under which the general code takes the same turn, and a read that
// fails any of them is left untouched for k_correct.  The conditions, with s / t the iteration's strong and weak
// thresholds (:793-842, pair override included):
//   (1) every letter is one of ACGT; the screens passed (:735-755);   [(1) and (2) are checked by the threshold kernel,
//       rc_quarter.h, which has the letters, the counts and the mask in registers: it flags the read as a candidate and
//       leaves the stretches it found in rc_kernel_args::runs]
//   (2) the trusted mask T[i] = counts[i] >= s && !IsPolyA(i, 2) (:870-931) has 1-runs that are all at least 2 long (each
//       is an island: no fall-back island, :1002-1007) and 0-runs that are exactly k long inside the read -- so
//       consecutive islands are k + 1 k-mers apart and no boundary moves (:934-965 needs a distance <= k), the base space
//       islands leave exactly the base p = last k-mer of the 0-run between them -- or at most k long at either end of
//       the read: an error less than k bases from the end, whose segment runs from p to the read's end and is searched
//       from its only anchor with `extend` = 0 (:1016-1022, :1038-1044, :1138, :1152): p's node, then one node per
//       remaining base.  There are at most min(3, MAX_FIX_PER_K - 1) 0-runs (below every veto's trigger, :1407, :1432;
//       fixes of different segments are more than k apart, so the pairwise veto :1314 sees none);
//   (3) at p's node: the own base fails its threshold (count < threshold, so "keep" is not taken, :302 / :539) and it
//       is not the accidental-gap case (:313 / :550: threshold == 1 && t <= 2); p's k-mer is not poly-A (:343 / :577
//       would forbid substitutions); exactly ONE alternative reaches the threshold (a second one would open a second
//       path), and its count is >= t;
//   (4) at every node behind p: the keep-base count x is >= t -- the node's threshold never exceeds the t handed down
//       (:165-172), so the base is kept whatever the other three extensions count -- and, in left searches, which hand the
//       node's threshold down as the next t (:546), x >= bs[t]: the bound of x, hence of the largest extension, reaches t,
//       the threshold IS t, t stays (rc_bs_lookup: the inverse of GetBound, a table built on the host);
//   (5) the path's bottleneck (the smallest count on it) is >= t, so the terminal adds no fix (:243 / :483) and the
//       path is accepted with fix count 1 < maxFixCnt (MAX_FIX_PER_K >= 2); with one candidate there is no frame to pop;
//   (6) the smallest bottleneck over the segments is not below GetBound(s) (:1195 would force another iteration).
// Then ret = number of segments, the bases are replaced, and l / m / h come from the counts with the k windows around
// each fix replaced by the counts the path just fetched (the k-th window, which a right search never visits, is
// fetched for this purpose).  tests: every GPU parity set and fuzz seed runs with this kernel in the path;
// RC_NO_SINGLE=1 takes it out (knob matrix); a Python restatement of the same conditions was checked against the
// oracle on every data set first.
#pragma once

#ifndef RC_K2S_WAVES
#define RC_K2S_WAVES 8  // waves per SIMD the register allocation is held to (25 M x 150 bp pairs, round 3: 12.4 / 11.4 / 11.1 / 11.4 ms at 5 / 6 / 7 / 8;
                        // round 6, 32-byte buckets -- a lookup holds half the registers: 11.4 / 10.5 / 10.0 at 6 / 7 / 8, config 3 19.5 / 17.6 / 16.9)
#endif
#ifndef RC_K2S_GRID
#define RC_K2S_GRID 128  // workgroups per CU of the grid that walks the list (25 M x 150 bp pairs: 11.5 / 11.0 / 10.4 / 10.2 / 10.2 ms at 12 / 24 / 48 / 96 / 384: the sections by number of stretches are uneven work)
#endif
namespace rcs {

// instances for reads of up to EC * 16 k-mers, EC = 8 / 9 / 10 count registers per lane (rc_quarter.h: the same three
// as the threshold code); every read of up to 160 bases fits the last
constexpr int MAX_KCNT = 160, MAX_LEN = 160, PK_WORDS = MAX_LEN / 16 + 2, MAX_SEG = 3;

}  // namespace rcs

// one 16-lane row, one read (r = 0xFFFFFFFF: none); s_cnt / s_pk: the row's LDS (rows of a wave do not share any)
template <bool EXT, int EC>
__device__ __forceinline__ void rcs_row(const rc_kernel_args &A, uint32_t r, int32_t *s_cnt_row, uint32_t *s_pk_row)
{
    using namespace rcs;
    constexpr int MAX_KCNT = EC * 16;
    const int lane = threadIdx.x & 63, row = lane >> 4, l = lane & 15, row_lane0 = row << 4;
    const int k = A.P.k, mfk = A.P.max_fix_per_k;
    const double er = A.P.error_rate;
    auto wsync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto row_any = [&](bool p) { return rcq::row_bits(__ballot(p), row) != 0; };

    bool ok = r < A.n && mfk >= 2;
    uint32_t o = 0;
    int len = 0, strong0 = 0, info0 = 4, pair_t = -1;
    if (ok) {
        o = A.off[r];
        len = (int)(A.off[r + 1] - o) - 1;
        strong0 = A.strong[r];
        info0 = A.info[r];
        if (A.mode == 1) {
            const uint32_t half = A.n >> 1;
            const int ms = A.strong[r < half ? r + half : r - half];
            pair_t = strong0 < ms ? strong0 : ms;
        } else if (A.mode == 2) {
            const int ms = A.strong[r ^ 1u];
            pair_t = strong0 < ms ? strong0 : ms;
        }
    }
    const int kcnt = len - k + 1;
    ok = ok && !(info0 & 4) && len >= k && len <= MAX_LEN && kcnt <= MAX_KCNT && kcnt >= 5;

    // thresholds of the first iteration, ErrorCorrection.cpp:793-842
    int s = strong0, t = 2;
    if (ok) {
        bool flag = false;
        int trust = rc_bound_i(s, er);
        if ((info0 & 1) && s >= 20 && (info0 & 2) && trust < 3) {
            flag = true;
            trust = 3;
        }
        if (pair_t >= 1 && s > pair_t) {
            if (!flag || pair_t < 20) trust = rc_bound_i(pair_t, er);
            s = pair_t;
        }
        t = trust < 2 ? 2 : trust;
    }

    // condition (2) was checked by the threshold kernel, which leaves the stretches it found in A.runs (rc_quarter.h: the
    // same mask -- counts >= s, not poly-A at 2 -- from the same s; condition (1) too: a candidate has ACGT letters only)
    int ns = 0;
    uint32_t run01 = 0, run2 = 0;
    if (ok) {
        const uint2 rw = A.runs[r];
        run01 = rw.x;
        run2 = rw.y & 0xffffu;
        ns = (int)(rw.y >> 16);
        const int max_seg = mfk - 1 < MAX_SEG ? mfk - 1 : MAX_SEG;
        ok = ns >= 1 && ns <= max_seg;
    }
    if (!__ballot(ok)) return;  // (no row of this wave has a read: nothing below touches another wave)
    // the packed read (ds_or of every base's two bits) and K1's counts -> the row's LDS
    if (l < PK_WORDS) s_pk_row[l] = 0;
    wsync();
#pragma unroll
    for (int e = 0; e < 10; ++e) {
        const int p = e * 16 + l;
        if (ok && p < len) atomicOr(&s_pk_row[e], (uint32_t)(rc_base_code(A.seq[o + p]) & 3) << (30 - 2 * l));
    }
#pragma unroll
    for (int e = 0; e < EC; ++e) {
        const int g = e * 16 + l;
        int v = 0;
        if (ok && g < kcnt) v = A.counts[o + g];
        s_cnt_row[g] = v;
    }
    wsync();
#if defined(RC_K2S_STOP) && RC_K2S_STOP == 1  // dev builds: cost of the stages up to here
    return;
#endif
    uint32_t Bt = 0;  // condition (4), left searches
    if (ok) Bt = rc_bs_lookup(A.P, t);
    const bool bs_ok = Bt != 0;

    // 2-bit code of the n <= 32 bases starting at base q (cf. rc_code_at)
    auto code_at = [&](int q) -> uint64_t {
        const uint32_t *pk = s_pk_row;
        const int w0 = q >> 4, sh = 2 * (q & 15);
        uint64_t x = ((uint64_t)pk[w0] << 32) | pk[w0 + 1];
        if (sh) x = (x << sh) | ((uint64_t)pk[w0 + 2] >> (32 - sh));
        return x >> (64 - 2 * k);
    };
    auto base_at = [&](int q) -> int { return (int)((s_pk_row[q >> 4] >> (30 - 2 * (q & 15))) & 3u); };
    // IsPolyA(window g, 2), ErrorCorrection.cpp:53-71: >= k - 2 A's (code 0) or T's (code 3) among the window's k letters
    auto polya2 = [&](int g) -> bool {
        const uint64_t x = code_at(g);
        const uint64_t m = (k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull)) & 0x5555555555555555ull;
        const uint64_t lo = x & m, hi = (x >> 1) & m;
        return __popcll(lo & hi) >= k - 2 || __popcll(~(lo | hi) & m) >= k - 2;
    };

    int fixp0 = 0, fixp1 = 0, fixp2 = 0, fixc0 = 0, fixc1 = 0, fixc2 = 0;  // (named: a run-time index would put them in scratch)
    int best_bott = 2147483647, prev_z1 = -1;
    for (int si = 0; si < MAX_SEG; ++si) {
        if (!__ballot(ok && si < ns)) break;  // (no row of the wave has a segment left)
        const bool act = ok && si < ns;   // (row-uniform)
        int z0 = 0, z1 = 0, next_z0 = kcnt;
        if (act) {
            const uint32_t rs = si == 0 ? (run01 & 0xffffu) : (si == 1 ? (run01 >> 16) : run2);
            const uint32_t rn = si == 0 ? (run01 >> 16) : run2;   // the run after this one
            z0 = (int)(rs & 0xffu);
            z1 = z0 + (int)(rs >> 8) - 1;
            if (si + 1 < ns) next_z0 = (int)(rn & 0xffu);
        }
        // a 0-run inside the read is k long; one at either end of the read is at most k long (the error is less than k
        // bases from that end) -- the segment then runs to the read's end (:1009-1046, `extend` = 0 at :1138 / :1152)
        const int zlen = z1 - z0 + 1;
        const bool at_start = z0 == 0, at_end = z1 == kcnt - 1;
        bool good = act && (zlen == k || ((at_start || at_end) && zlen < k));
        const int p = at_start ? z1 : z0 + k - 1;                           // the base every k-mer of the 0-run holds
        const bool right = (z0 - prev_z1 - 1) >= (next_z0 - z1 - 1);        // lanchor >= ranchor (:1136): the islands' lengths
        prev_z1 = z1;
        const int win = right ? z0 : z1;                                    // k-mer of p's node: it ends (right) / starts (left) at p
        const int b = good ? base_at(p) : 0;
        // round A: the four extensions of the anchor at p (the own base's count is K1's)
        int cA = 0;
        if (good && l < 4) {
            if (l == b)
                cA = s_cnt_row[win];
            else {
                const uint64_t km = code_at(win) ^ ((uint64_t)(b ^ l) << (2 * (k - 1 - (p - win))));
                cA = rc_table_lookup<EXT>(A.T, rc_canonical(km, k));
            }
        }
        const int c0 = __shfl(cA, row_lane0 + 0, 64), c1 = __shfl(cA, row_lane0 + 1, 64), c2 = __shfl(cA, row_lane0 + 2, 64), c3 = __shfl(cA, row_lane0 + 3, 64);
        int cstar = 0, cnt_star = 0;
        if (good) {
            int mx = c0 > c1 ? c0 : c1;
            mx = c2 > mx ? c2 : mx;
            mx = c3 > mx ? c3 : mx;
            mx = mx > 0 ? mx : 0;
            int ret = rc_bound_i(mx, er);
            if (ret < 1) ret = 1;
            const int thr = (t > ret || t <= 0) ? ret : t;                  // InferPosThreshold, :165-172
            const int own = b == 0 ? c0 : (b == 1 ? c1 : (b == 2 ? c2 : c3));
            const int m0 = (b != 0 && c0 >= thr) ? 1 : 0, m1 = (b != 1 && c1 >= thr) ? 1 : 0, m2 = (b != 2 && c2 >= thr) ? 1 : 0, m3 = (b != 3 && c3 >= thr) ? 1 : 0;
            cstar = m0 ? 0 : (m1 ? 1 : (m2 ? 2 : 3));
            cnt_star = cstar == 0 ? c0 : (cstar == 1 ? c1 : (cstar == 2 ? c2 : c3));
            good = own < thr && !(thr == 1 && t <= 2) && !polya2(win) && (m0 + m1 + m2 + m3) == 1 && cnt_star >= t;   // condition (3)
            if (!right) good = good && bs_ok;
        }
        // round B: the other k - 1 windows of the 0-run with p replaced: the nodes behind p and, in a right search, the
        // window that starts at p (for GetKmerInformation only)
        int bott = cnt_star;
        bool fail = false;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = h * 16 + l;
            const int wj = z0 + j;
            if (good && j < zlen && wj != win) {
                const uint64_t km = code_at(wj) ^ ((uint64_t)(b ^ cstar) << (2 * (k - 1 - (p - wj))));
                const int x = rc_table_lookup<EXT>(A.T, rc_canonical(km, k));
                const bool node = (right && !at_end) ? wj < z1 : true;      // (right, inside the read: wj == z1 is the extra window)
                if (node) {
                    if (x < t || (!right && (uint32_t)x < Bt)) fail = true;  // condition (4)
                    bott = x < bott ? x : bott;
                }
                s_cnt_row[wj] = x;
            }
        }
        if (good && l == 0) s_cnt_row[win] = cnt_star;
        good = good && !row_any(fail);
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            const int y = __shfl_xor(bott, m, 16);
            bott = y < bott ? y : bott;
        }
        good = good && bott >= t;                                           // condition (5)
        if (act) {
            ok = good;
            fixp0 = si == 0 ? p : fixp0;
            fixp1 = si == 1 ? p : fixp1;
            fixp2 = si == 2 ? p : fixp2;
            fixc0 = si == 0 ? cstar : fixc0;
            fixc1 = si == 1 ? cstar : fixc1;
            fixc2 = si == 2 ? cstar : fixc2;
            best_bott = bott < best_bott ? bott : best_bott;
        }
        wsync();
    }
    if (ok && rc_less_than_bound(best_bott, s, er)) ok = false;             // condition (6)
    if (!__ballot(ok)) return;
#if defined(RC_K2S_STOP) && RC_K2S_STOP == 2
    return;
#endif

    // GetKmerInformation of the corrected read (:1567-1602): min / element kcnt/2 / max of the counts in ascending order,
    // 0 shown as 1 (:1583).  Minimum and maximum are row reductions; the element of rank kcnt/2 comes from a descent over
    // the bits of the largest count (one count-the-smaller-ones row reduction per bit: counts of a few hundred make that
    // a third of the sorting network's instructions); counts of 2^14 and more take the network.
    int x[EC];
    int vmin = 2147483647, vmax = 1;
#pragma unroll
    for (int e = 0; e < EC; ++e) {
        const int g = e * 16 + l;
        int v = s_cnt_row[g];
        v = v == 0 ? 1 : v;
        const bool in = ok && g < kcnt;
        x[e] = in ? v : 2147483647;
        vmin = x[e] < vmin ? x[e] : vmin;
        vmax = in && v > vmax ? v : vmax;
    }
    auto row_min = [&](int v) {
        int y;
        y = rcq::row_xor<1>(v); v = y < v ? y : v;
        y = rcq::row_xor<2>(v); v = y < v ? y : v;
        y = rcq::row_xor<4>(v); v = y < v ? y : v;
        y = rcq::row_xor<8>(v); v = y < v ? y : v;
        return v;
    };
    auto row_max = [&](int v) {
        int y;
        y = rcq::row_xor<1>(v); v = y > v ? y : v;
        y = rcq::row_xor<2>(v); v = y > v ? y : v;
        y = rcq::row_xor<4>(v); v = y > v ? y : v;
        y = rcq::row_xor<8>(v); v = y > v ? y : v;
        return v;
    };
    auto row_sum = [&](int v) {
        v += rcq::row_xor<1>(v);
        v += rcq::row_xor<2>(v);
        v += rcq::row_xor<4>(v);
        v += rcq::row_xor<8>(v);
        return v;
    };
    const int v0 = row_min(vmin), vh = row_max(vmax);
    const int im = kcnt > 0 ? kcnt >> 1 : 0;
    int vm;
    const int wmax = max(max(__builtin_amdgcn_readlane(vh, 0), __builtin_amdgcn_readlane(vh, 16)),
                         max(__builtin_amdgcn_readlane(vh, 32), __builtin_amdgcn_readlane(vh, 48)));  // (uniform)
    if (wmax < (1 << 14)) {
        int prefix = 0;
        for (int b = 31 - __builtin_clz((unsigned)wmax | 1u); b >= 0; --b) {
            const int cand = prefix | (1 << b);
            int below = 0;
#pragma unroll
            for (int e = 0; e < EC; ++e) below += x[e] < cand ? 1 : 0;
            below = row_sum(below);
            prefix = below <= im ? cand : prefix;   // the largest value with at most im counts below it: the element of rank im
        }
        vm = prefix;
    } else {
        int c4[4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) c4[bb] = __builtin_amdgcn_sbfe(l, bb, 1) ^ (int)0x80000000;
        rcq::merges<EC, 2>(x, c4);
        int sm = x[0];
#pragma unroll
        for (int e = 1; e < EC; ++e) sm = (im >> 4) == e ? x[e] : sm;
        vm = __builtin_amdgcn_ds_bpermute((row_lane0 + (im & 15)) << 2, sm);
    }
    if (ok) {
        if (l < ns) {
            const int fp = l == 0 ? fixp0 : (l == 1 ? fixp1 : fixp2), fc = l == 0 ? fixc0 : (l == 1 ? fixc1 : fixc2);
            A.seq[o + (uint32_t)fp] = (uint8_t)("ACGT"[fc]);
        }
        if (l == 0) {
            A.ret[r] = ns;
            A.l[r] = v0;
            A.m[r] = vm;
            A.h[r] = vh;
            A.cls[r] = 0;
        }
    }
}

// reads of up to 160 bases / EC * 16 k-mers; 256 threads = 16 reads per pass.  The reads come off the work list
// (RC_WORK_CLASSES sections, rc_internal.h), whose length stays on the device: a fixed grid of workgroups walks the
// concatenated sections (a workgroup per 16 reads would be 1.5 M workgroups for 25 M reads, a third of them empty)
template <bool EXT, int EC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RC_K2S_WAVES, RC_K2S_WAVES))) void k_single(rc_kernel_args A)
{
    using namespace rcs;
    __shared__ int32_t s_cnt[16][EC * 16];
    __shared__ uint32_t s_pk[16][PK_WORDS];
    const int rr = (int)(threadIdx.x >> 6) * 4 + (int)((threadIdx.x & 63) >> 4);
    const uint32_t n0 = A.n_work[0], n1 = A.n_work[1], n2 = A.n_work[2], n3 = A.n_work[3];
    const uint32_t total = n0 + n1 + n2 + n3;
    for (uint32_t base = blockIdx.x * 16u; base < total; base += gridDim.x * 16u) {
        uint32_t g = base + (uint32_t)rr, r = 0xFFFFFFFFu;
        if (g < total) {
            int c = 0;
            if (g >= n0) {
                g -= n0;
                c = 1;
                if (g >= n1) {
                    g -= n1;
                    c = 2;
                    if (g >= n2) {
                        g -= n2;
                        c = 3;
                    }
                }
            }
            r = A.worklist[(size_t)c * A.work_stride + g];
        }
        rcs_row<EXT, EC>(A, r, s_cnt[rr], s_pk[rr]);
    }
}
