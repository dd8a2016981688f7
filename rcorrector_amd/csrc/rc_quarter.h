// rc_quarter.h -- "quarter-wave" kernels: FOUR reads per 64-lane wavefront, one per 16-lane DPP
// row, for the per-read array work whose control flow does not depend on the data
// (GetStrongTrustedThreshold, ErrorCorrection.cpp:1482-1565).
//
// Why: with one read per wave (rc_correct_core.h) a 150-base read gives every lane two or three
// elements, so most instructions of these phases are executed for a handful of useful lanes, and
// every lane exchange is an LDS round trip (ds_bpermute) on a serial dependency chain -- the
// wave-per-read threshold kernel spends ~17 000 cycles per read on ~900 instructions.  Here element
// g of a read lives in register g/16 of lane g%16 of the read's row: the sort network exchanges
// through DPP row operations (quad_perm, row_shr/shl, row_ror, row_mirror: plain VALU, no LDS),
// strides >= 16 are register-to-register, the letter masks are 32-bit words in registers and a
// window's A/T count is one v_alignbit + v_and + v_bcnt.  No LDS allocation at all.
//
// Limits of this layout: k-mer windows per read <= 256 (16 registers x 16 lanes) and read length
// <= 320; batches with longer reads take the wave-per-read kernel (same results, rc_front_end()).
// Included by rc_correct.hip only.
#pragma once

namespace rcq {

// instantiations: <8, 10> = up to 128 windows / 160 bases per read, <9, 10> = 144 / 160 (151-base reads at
// k = 23 have 129 windows), <10, 10> = every read of up to 160 bases, <16, 20> = 256 / 320
// (EC count registers and EB base registers per lane, 16 lanes per read)
constexpr int MAX_KCNT = 16 * 16;
constexpr int MAX_LEN = 20 * 16;
static_assert(MAX_KCNT == RC_Q_MAX_KCNT && MAX_LEN == RC_Q_MAX_LEN, "rc_internal.h");

template <int CTRL>
__device__ __forceinline__ int dpp(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
// value of lane (l ^ X) of the same 16-lane row
template <int X>
__device__ __forceinline__ int row_xor(int v)
{
    if constexpr (X == 1) {
        return dpp<0xB1>(v);  // quad_perm [1,0,3,2]
    } else if constexpr (X == 2) {
        return dpp<0x4E>(v);  // quad_perm [2,3,0,1]
    } else if constexpr (X == 3) {
        return dpp<0x1B>(v);  // quad_perm [3,2,1,0]
    } else if constexpr (X == 4) {
        const int t = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);  // row_shl:4 into banks 0,2
        return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);        // row_shr:4 into banks 1,3
    } else if constexpr (X == 7) {
        return dpp<0x141>(v);  // row_half_mirror
    } else if constexpr (X == 8) {
        return dpp<0x128>(v);  // row_ror:8
    } else {
        static_assert(X == 15, "row_xor: unsupported pattern");
        return dpp<0x140>(v);  // row_mirror
    }
}
__device__ __forceinline__ int med3(int a, int b, int c)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

// ascending bitonic network over the EC * 16 elements (e, l) of each row, all-ascending ("flip")
// formulation: the first stage of a merge pairs g with g ^ (size-1), the others g with g ^ stride,
// the lower index keeps the minimum.  c[b] = bit b of l ? INT_MAX : INT_MIN turns one v_med3_i32
// into "min on the lower side, max on the upper side".
// EC need not be a power of two: the network is the one of the next power of two with the registers
// beyond EC left out.  They would hold the padding value INT_MAX (every caller pads its own tail
// with it), and since the lower index always keeps the minimum, a compare-exchange between a real
// register and a padding one changes neither: 9 registers (144 windows: 151-base reads at k = 23)
// cost 9/8 of the in-row steps of 8 and 13 register-to-register exchanges more, not the 2.6 x of 16.
template <int EC, int X, int BIT>
__device__ __forceinline__ void cx_row(int (&x)[EC], const int (&c)[4])
{
#pragma unroll
    for (int e = 0; e < EC; ++e) x[e] = med3(x[e], row_xor<X>(x[e]), c[BIT]);
}
template <int EC, int STRIDE>
__device__ __forceinline__ void strides(int (&x)[EC], const int (&c)[4])
{
    if constexpr (STRIDE >= 16) {
#pragma unroll
        for (int e = 0; e < EC; ++e) {
            const int pe = e ^ (STRIDE >> 4);
            if (pe > e && pe < EC) {
                const int lo = x[e] < x[pe] ? x[e] : x[pe];
                const int hi = x[e] < x[pe] ? x[pe] : x[e];
                x[e] = lo;
                x[pe] = hi;
            }
        }
    } else {
        cx_row<EC, STRIDE, ilog2(STRIDE)>(x, c);
    }
    if constexpr (STRIDE > 1) strides<EC, STRIDE / 2>(x, c);
}
template <int EC, int SIZE>
__device__ __forceinline__ void merges(int (&x)[EC], const int (&c)[4])
{
    if constexpr (SIZE <= 16) {
        cx_row<EC, SIZE - 1, ilog2(SIZE) - 1>(x, c);
    } else {
#pragma unroll
        for (int e = 0; e < EC; ++e) {
            const int pe = e ^ ((SIZE >> 4) - 1);
            if (pe > e && pe < EC) {
                const int ye = row_xor<15>(x[pe]), yp = row_xor<15>(x[e]);
                x[e] = x[e] < ye ? x[e] : ye;
                x[pe] = x[pe] < yp ? yp : x[pe];
            }
        }
    }
    if constexpr (SIZE >= 4) strides<EC, SIZE / 4>(x, c);
    if constexpr (SIZE < EC * 16) merges<EC, SIZE * 2>(x, c);
}

// The same for 8 registers in the TRANSPOSED layout (round 4): element g of the sorted row in register g % 8 of lane g / 8.
// A bitonic network does not care where its input comes from -- the counts are loaded (and poly-A-masked) in the layout
// above and simply read as this one -- and with the low index bits in the registers 18 of the 28 stages of 128 elements
// are register-to-register compare-exchanges (v_min + v_max per pair: 4 cycles per element) instead of 6; only the 10
// stages with a stride of 8 and more go through a DPP move + v_med3 (8 cycles per element): 1 250 instead of 1 820 cycles
// per wave, and the scan for the drop below finds an element's predecessor in the register next to it.
__device__ __forceinline__ void cx(int &a, int &b)
{
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo;
    b = hi;
}
__device__ __forceinline__ void t8_tail(int (&x)[8])  // strides 4, 2, 1
{
    cx(x[0], x[4]); cx(x[1], x[5]); cx(x[2], x[6]); cx(x[3], x[7]);
    cx(x[0], x[2]); cx(x[1], x[3]); cx(x[4], x[6]); cx(x[5], x[7]);
    cx(x[0], x[1]); cx(x[2], x[3]); cx(x[4], x[5]); cx(x[6], x[7]);
}
// first stage of a merge of 8 (X + 1) elements: g pairs with g ^ (8 (X + 1) - 1) = register 7 - e of lane l ^ X; BIT = the
// highest bit of X: the lane that has it clear holds the lower index and keeps the minimum
template <int X, int BIT>
__device__ __forceinline__ void t8_flip(int (&x)[8], const int (&c)[4])
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ya = row_xor<X>(x[7 - e]), yb = row_xor<X>(x[e]);
        x[e] = med3(x[e], ya, c[BIT]);
        x[7 - e] = med3(x[7 - e], yb, c[BIT]);
    }
}
template <int X, int BIT>
__device__ __forceinline__ void t8_lane(int (&x)[8], const int (&c)[4])  // stride 8 X
{
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = med3(x[e], row_xor<X>(x[e]), c[BIT]);
}
__device__ __forceinline__ void sort_t8(int (&x)[8], const int (&c)[4])
{
    cx(x[0], x[1]); cx(x[2], x[3]); cx(x[4], x[5]); cx(x[6], x[7]);  // 2
    cx(x[0], x[3]); cx(x[1], x[2]); cx(x[4], x[7]); cx(x[5], x[6]);  // 4: flip, stride 1
    cx(x[0], x[1]); cx(x[2], x[3]); cx(x[4], x[5]); cx(x[6], x[7]);
    cx(x[0], x[7]); cx(x[1], x[6]); cx(x[2], x[5]); cx(x[3], x[4]);  // 8: flip, strides 2, 1
    cx(x[0], x[2]); cx(x[1], x[3]); cx(x[4], x[6]); cx(x[5], x[7]);
    cx(x[0], x[1]); cx(x[2], x[3]); cx(x[4], x[5]); cx(x[6], x[7]);
    t8_flip<1, 0>(x, c);  // 16
    t8_tail(x);
    t8_flip<3, 1>(x, c);  // 32
    t8_lane<1, 0>(x, c);
    t8_tail(x);
    t8_flip<7, 2>(x, c);  // 64
    t8_lane<2, 1>(x, c);
    t8_lane<1, 0>(x, c);
    t8_tail(x);
    t8_flip<15, 3>(x, c);  // 128
    t8_lane<4, 2>(x, c);
    t8_lane<2, 1>(x, c);
    t8_lane<1, 0>(x, c);
    t8_tail(x);
}

// the 16 bits of a wave-wide ballot that belong to this lane's row
__device__ __forceinline__ uint32_t row_bits(uint64_t ballot, int row)
{
    return (uint32_t)(ballot >> (row << 4)) & 0xffffu;
}

}  // namespace rcq

// dev (tools/fused_stops.sh): -DRC_FUSED_STOP=<n> cuts the fused probe + threshold kernel off after stage n -- everything computed up to
// there is folded into one value that a store nobody takes depends on, so the compiler keeps the stages in front of the cut
// and drops the ones behind it.  Wrong results; what is measured is the kernel's time and instruction count up to the cut.
#ifdef RC_FUSED_STOP
#define RCQ_STOP(n, ...)                                                  \
    do {                                                                  \
        if constexpr (RC_FUSED_STOP == (n)) {                             \
            int sink_ = 0;                                                \
            const int vals_[] = {__VA_ARGS__};                            \
            for (int v_ : vals_) sink_ = sink_ * 31 + v_;                 \
            if (sink_ == 0x5bd1e995 && A.strong) A.strong[0] = sink_;     \
            return 1;                                                     \
        }                                                                 \
    } while (0)
#else
#define RCQ_STOP(n, ...) do { } while (0)
#endif

// GetStrongTrustedThreshold + classification of the read held by this lane's 16-lane row (four reads
// per wave; rows 2j and 2j+1 of a wave hold the two mates of a pair).  base_at(p) = letter of base p,
// count_at(g) = K1's count of k-mer g; the caller decides where they come from (HBM for
// k_threshold_q, the workgroup's LDS for the fused probe kernel).  Writes strong / info / cls and, for
// reads it can finish, ret / l / m / h of read r; returns the class (1 = k_correct has work to do).
// KT: the k the caller is compiled for (0: A.P.k)
// (int)GetBound(c, ERROR_RATE) of a row's threshold: a byte of the host's table (rc_run_params::bound_small) where it holds
// the value, the double-precision chain of rc_common.h where it does not (counts of RC_BOUND_SMALL and more, screened reads'
// -1, no table).  Call in wave-uniform control flow.
__device__ __forceinline__ int rcq_bound(const rc_kernel_args &A, int c)
{
    int v = 255;
    if (A.P.bound_small && (uint32_t)c < (uint32_t)RC_BOUND_SMALL) v = A.P.bound_small[c];
    if (__ballot(v == 255)) {
        const int slow = rc_bound_i(c, A.P.error_rate);
        v = v == 255 ? slow : v;
    }
    return v;
}

// MS = where the letter masks come from: rcq_no_masks -- built here from base_at(), a compare chain and three ballots per
// base (the batch in HBM: k_threshold_q) -- or rcq_lds_masks: the fused probe kernel has packed its arena's letters into
// bit arrays already (rc_pack16m: bit p % 32 of word p / 32 = arena byte p is an A / is a T / is neither of ACGT), and a row's
// words are five funnel shifts of them (round 6: the compare chains were 230 of the 1 516 vector instructions of a row pass).
struct rcq_no_masks {
    static constexpr bool present = false;
};
struct rcq_lds_masks {
    static constexpr bool present = true;
    const uint32_t *am, *tm, *bad;  // LDS
    uint32_t lp;                    // the row's first base in the arena
};
template <int E_CNT, int E_BASE, int KT = 0, class FB, class FC, class MS = rcq_no_masks>
__device__ __forceinline__ int rcq_threshold_row(const rc_kernel_args &A, uint32_t r, bool live, int len, FB base_at, FC count_at, MS msrc = MS())
{
    using namespace rcq;
    const int lane = threadIdx.x & 63, row = lane >> 4, l = lane & 15;
    const int k = KT ? KT : A.P.k;
    const int kcnt = len >= k ? len - k + 1 : 0;

    // K1's counts, element g in register g/16 of lane g%16
    int x[E_CNT];
#pragma unroll
    for (int e = 0; e < E_CNT; ++e) {
        const int g = e * 16 + l;
        x[e] = g < kcnt ? count_at(g) : 0;
    }

    // letter masks of the row's read as 32-bit words (bit p%32 of word p/32 = base p is the letter); `other` != 0: the read has
    // a letter outside ACGT
    uint32_t ma[E_BASE / 2 + 1], mt[E_BASE / 2 + 1];
    uint32_t other = 0;
    int n_cnt = 0;
    if constexpr (MS::present) {
        const uint32_t w0 = msrc.lp >> 5, sh = msrc.lp & 31u;
        uint32_t wa[E_BASE / 2 + 1], wt[E_BASE / 2 + 1], wb[E_BASE / 2 + 1];
#pragma unroll
        for (int j = 0; j <= E_BASE / 2; ++j) {
            wa[j] = msrc.am[w0 + j];
            wt[j] = msrc.tm[w0 + j];
            wb[j] = msrc.bad[w0 + j];
        }
#pragma unroll
        for (int j = 0; j < E_BASE / 2; ++j) {
            const int left = len - 32 * j;  // bases of the read from bit 0 of this word on
            const uint32_t lm = left >= 32 ? 0xffffffffu : (left > 0 ? (1u << left) - 1u : 0u);
            ma[j] = __builtin_amdgcn_alignbit(wa[j + 1], wa[j], sh) & lm;
            mt[j] = __builtin_amdgcn_alignbit(wt[j + 1], wt[j], sh) & lm;
            other |= __builtin_amdgcn_alignbit(wb[j + 1], wb[j], sh) & lm;
        }
        ma[E_BASE / 2] = mt[E_BASE / 2] = 0;
        if (__ballot(other != 0)) {  // (wave-uniform, rare) how many of them are N's   :1507-1510
#pragma unroll
            for (int e = 0; e < E_BASE; ++e) {
                const int p = e * 16 + l;
                n_cnt += __popc(row_bits(__ballot(p < len && base_at(p) == (uint32_t)'N'), row));
            }
        }
    } else {
        int code[E_BASE];
#pragma unroll
        for (int e = 0; e < E_BASE; ++e) {
            const int p = e * 16 + l;
            code[e] = p < len ? rc_base_code(base_at(p)) : 7;
        }
        uint32_t fa[E_BASE], ft[E_BASE];
#pragma unroll
        for (int e = 0; e < E_BASE; ++e) {
            fa[e] = row_bits(__ballot(code[e] == 0), row);
            ft[e] = row_bits(__ballot(code[e] == 3), row);
            n_cnt += __popc(row_bits(__ballot(code[e] == 4), row));
            other |= row_bits(__ballot(code[e] >= 4 && e * 16 + l < len), row);
        }
#pragma unroll
        for (int j = 0; j < E_BASE / 2; ++j) {
            ma[j] = fa[2 * j] | (fa[2 * j + 1] << 16);
            mt[j] = ft[2 * j] | (ft[2 * j + 1] << 16);
        }
        ma[E_BASE / 2] = mt[E_BASE / 2] = 0;
    }
    RCQ_STOP(10, (int)other, n_cnt, x[0] ^ x[1] ^ x[2] ^ x[3] ^ x[E_CNT - 4] ^ x[E_CNT - 3] ^ x[E_CNT - 2] ^ x[E_CNT - 1]);
    int a_cnt = 0, t_cnt = 0;
#pragma unroll
    for (int j = 0; j < E_BASE / 2; ++j) {
        a_cnt += __popc(ma[j]);
        t_cnt += __popc(mt[j]);
    }
    const bool screened = len < k || n_cnt > 5 || a_cnt > len - k || t_cnt > len - k;  // :1491,1507-1527
    RCQ_STOP(11, (int)screened, (int)(ma[0] ^ ma[1] ^ ma[2] ^ ma[3] ^ ma[4]), (int)(mt[0] ^ mt[1] ^ mt[2] ^ mt[3] ^ mt[4]),
             x[0] ^ x[1] ^ x[2] ^ x[3] ^ x[E_CNT - 4] ^ x[E_CNT - 3] ^ x[E_CNT - 2] ^ x[E_CNT - 1]);

    // poly-A masked counts (:1530-1541): a window with >= k - max(7, k/2) A's or T's counts as -1;
    // window g = bits [g, g+k) of the mask = one funnel shift of two adjacent words (k <= 32)
    int thr7 = 7;
    if (k / 2 > thr7) thr7 = k / 2;
    const uint32_t kmask = k >= 32 ? 0xffffffffu : ((1u << k) - 1u);
    uint32_t not_polya2 = 0;  // bit e: window e * 16 + l has fewer than k - 2 A's and fewer than k - 2 T's (IsPolyA(.., 2) of the island code, :870-931)
#pragma unroll
    for (int e = 0; e < E_CNT; ++e) {
        const int g = e * 16 + l;
        const int w = e >> 1;                     // (e*16 + l) / 32
        const uint32_t sh = (uint32_t)((e & 1) * 16 + l);
        const int a = __popc(__builtin_amdgcn_alignbit(ma[w + 1], ma[w], sh) & kmask);
        const int t = __popc(__builtin_amdgcn_alignbit(mt[w + 1], mt[w], sh) & kmask);
        const int at = a > t ? a : t;
        const bool polya = at >= k - thr7;
        not_polya2 |= (at < k - 2 ? 1u : 0u) << e;
        x[e] = g < kcnt ? (polya ? -1 : x[e]) : 2147483647;
    }
    RCQ_STOP(12, (int)screened, x[0], x[1], x[2], x[3], x[E_CNT - 4], x[E_CNT - 3], x[E_CNT - 2], x[E_CNT - 1]);

    int c[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) c[b] = __builtin_amdgcn_sbfe(l, b, 1) ^ (int)0x80000000;  // bit ? INT_MAX : INT_MIN
    // T8: 8 count registers take the transposed network -- from here on sorted element g sits in register g % 8 of lane g / 8
    // (RC_SORT_LAYOUT_A at compile time keeps the layout of the load, for A/B runs)
#ifdef RC_SORT_LAYOUT_A
    constexpr bool T8 = false;
#else
    constexpr bool T8 = E_CNT == 8;
#endif
    if constexpr (T8)
        sort_t8(x, c);
    else
        merges<E_CNT, 2>(x, c);
    RCQ_STOP(13, (int)screened, x[0], x[1], x[2], x[3], x[E_CNT - 4], x[E_CNT - 3], x[E_CNT - 2], x[E_CNT - 1]);
    // sorted element idx, read by every lane of the row: its register (a chain of selects: x[] must stay in registers) and lane
#define RCQ_SORTED_AT(dst, idx_)                                                                       \
    do {                                                                                               \
        const int i_ = (idx_);                                                                         \
        int s_ = x[0];                                                                                 \
        _Pragma("unroll") for (int e = 1; e < E_CNT; ++e) s_ = (T8 ? (i_ & 7) : (i_ >> 4)) == e ? x[e] : s_; \
        dst = __builtin_amdgcn_ds_bpermute(((row << 4) + (T8 ? (i_ >> 3) : (i_ & 15))) << 2, s_);      \
    } while (0)

    // the "drop" scan (:1543-1563): highest g in [1, kcnt) with v[g] > 2 v[g-1] && v[g] > 10
    const int row_lane0 = row << 4;
    int strong = 0, prev = 0;
    bool found = false;
    int i0 = kcnt;  // lowest g with v[g] > 0
    if constexpr (T8) {
        // an element's predecessor is the register next to it (register 0: register 7 of the lane before); every lane keeps
        // its own highest drop, the row takes the highest lane that has one
        const int below = dpp<0x111>(x[7]);  // row_shr:1 (lane 0 of the row: 0, and its g = 0 is never a drop)
        int hx = 0, hp = 0, npos = 0;
        bool hf = false;
#pragma unroll
        for (int e = 7; e >= 0; --e) {
            const int g = l * 8 + e;
            const int p = e > 0 ? x[e > 0 ? e - 1 : 0] : below;
            const bool drop = g >= 1 && g < kcnt && x[e] > 2 * p && x[e] > 10;
            const bool take = drop && !hf;
            hx = take ? x[e] : hx;
            hp = take ? p : hp;
            hf = hf || drop;
            npos += x[e] <= 0 ? 1 : 0;  // (the padding behind kcnt is INT_MAX)
        }
        const uint32_t fm = row_bits(__ballot(hf), row);
        found = fm != 0;
        const int src = (row_lane0 + (31 - __clz((int)(fm | 1u)))) << 2;
        const int s = __builtin_amdgcn_ds_bpermute(src, hx), pp = __builtin_amdgcn_ds_bpermute(src, hp);
        strong = found ? s : 0;
        prev = found ? pp : 0;
        // ascending order: what is not positive comes first, so the lowest positive element is as far in as there are others
        npos += row_xor<1>(npos);
        npos += row_xor<2>(npos);
        npos += row_xor<4>(npos);
        npos += row_xor<8>(npos);
        i0 = npos < kcnt ? npos : kcnt;
    } else {
        bool have_pos = false;
        uint32_t fdrop[E_CNT], fpos[E_CNT];
        int pv[E_CNT];
#pragma unroll
        for (int e = 0; e < E_CNT; ++e) {
            const int g = e * 16 + l;
            const int same = dpp<0x111>(x[e]);                  // row_shr:1 -- the element before, same register
            const int wrap = e > 0 ? dpp<0x121>(x[e > 0 ? e - 1 : 0]) : 0;  // row_ror:1 -- lane 0 sees lane 15 of the register before
            const int p = l == 0 ? wrap : same;
            pv[e] = p;
            const bool drop = g >= 1 && g < kcnt && x[e] > 2 * p && x[e] > 10;
            fdrop[e] = row_bits(__ballot(drop), row);
            fpos[e] = row_bits(__ballot(g < kcnt && x[e] > 0), row);
        }
#pragma unroll
        for (int e = E_CNT - 1; e >= 0; --e) {
            const bool hit = !found && fdrop[e] != 0;
            const int li = 31 - __clz((int)(fdrop[e] | 1u));
            const int src = (row_lane0 + li) << 2;
            const int s = __builtin_amdgcn_ds_bpermute(src, x[e]);
            const int pp = __builtin_amdgcn_ds_bpermute(src, pv[e]);
            strong = hit ? s : strong;
            prev = hit ? pp : prev;
            found = found || hit;
        }
#pragma unroll
        for (int e = 0; e < E_CNT; ++e) {
            const bool hit = !have_pos && fpos[e] != 0;
            i0 = hit ? e * 16 + (__ffs((int)fpos[e]) - 1) : i0;
            have_pos = have_pos || hit;
        }
    }
    {
        // no drop: the median of the positive part, v[(i0 + kcnt - 1) / 2]   (:1556-1563)
        const int idx = kcnt > 0 ? (i0 + kcnt - 1) / 2 : 0;
        int med;
        RCQ_SORTED_AT(med, idx);
        strong = found ? strong : med;
    }
    const int strong_self = screened ? -1 : strong;
    RCQ_STOP(14, strong_self, prev, (int)found, x[0], x[1], x[2], x[3], x[E_CNT - 4], x[E_CNT - 3], x[E_CNT - 2], x[E_CNT - 1]);
    // Class of the read.  Replay the first threshold iteration of ErrorCorrection (:793-842):
    // `s` = the strong threshold it starts with (its own, lowered to the pair's), `t0` = the weak
    // one ("trust").  If no k-mer of the read lies below t0 (v[0] >= t0 >= 2, which also means
    // every window is in the table: no letter outside ACGT, no poly-A mask) and two adjacent k-mers
    // are trusted (count >= s, not poly-A at 2), the function's result is known without running it:
    //  * two adjacent trusted k-mers are a real island (:870-931: a run of >= 2; no fall-back island, :1002-1007),
    //    no boundary is moved (:934-965 needs a count < trust) and every segment has its anchor k-mer
    //    inside the read (:1140-1154).  [Rounds 2-3 asked for "more than ceil(kcnt/2) k-mers reach s", which
    //    implies the adjacency but fails for every read with an odd number of k-mers whose s is the median
    //    of its counts without a tie -- 151-base reads at k = 23: 4.6 % of the reads of a batch went to
    //    k_correct for nothing; the mask is the one the k_single candidates are found from anyway];
    //  * in every segment search the keep-base child is taken at every node, because its count is
    //    a count of the unchanged read, >= t0 >= the node's threshold (InferPosThreshold never
    //    returns more than the threshold handed down, :165-172); the zero-fix path ends first
    //    (:243-284), sets maxFixCnt = 0, and every substitution alternative is cut at its entry
    //    (:211-224) before it can touch the result; trialCnt stays negative;
    //  * so total_fix = 0, no segment is bad, and the function returns 0 at :1110, :1231 or :1479
    //    without changing a base.  GetKmerInformation (:1567-1602) of the unchanged read is min /
    //    element kcnt/2 / max of the counts sorted above.
    // Such reads are finished here; k_correct only sees the others, cls = 1 .. RC_WORK_CLASSES: the
    // higher, the earlier in its launch.  What makes a read expensive is a search that has to cross
    // most of the read from a single island (:1157-1250: every node offers substitutions, MAX_TRIAL
    // trials per fix-count level), i.e. few k-mers reaching s.  Classes by the share of k-mers below s:
    // >= 7/8, >= 3/4, >= 1/2, the rest.  Measured on 4 M reads of 150 bp, k = 31, 5 % errors: the first
    // class holds 39 % of the reads, 59 % of the gather rounds and 864 of the 1000 most expensive
    // reads (the worst: 34 565 rounds); no read of the third class exceeds 5 300 rounds, none of the
    // last (11 % of the reads, 1.3 % of the rounds) 2 100.
    int cls = 1;
    if (A.cls) {
        int s = strong_self;
        int t0 = rcq_bound(A, s);
        bool flag = false;
        if (found && s >= 20 && prev == 2 && t0 < 3) {
            flag = true;
            t0 = 3;
        }
        if (A.mode != 0) {  // (uniform)
            const int mate = __builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, strong_self);
            const int pair_t = strong_self < mate ? strong_self : mate;
            const int tp = rcq_bound(A, pair_t);
            if (pair_t >= 1 && s > pair_t) {
                if (!flag || pair_t < 20) t0 = tp;
                s = pair_t;
            }
        }
        if (t0 < 2) t0 = 2;
        const int v0 = __builtin_amdgcn_ds_bpermute(row_lane0 << 2, x[0]);
        int n_below = 0;  // k-mers with count < s
#pragma unroll
        for (int e = 0; e < E_CNT; ++e) n_below += __popc(row_bits(__ballot(x[e] < s), row));
        RCQ_STOP(15, strong_self, prev, (int)found, s, t0, v0, n_below, x[0], x[1], x[2], x[3], x[E_CNT - 4], x[E_CNT - 3], x[E_CNT - 2], x[E_CNT - 1]);
        // the trusted mask ErrorCorrection builds its islands from (:870-931): counts[g] >= s && !IsPolyA(g, 2), bit g % 16 of
        // word g / 16; `adj` = it has two adjacent 1-bits, i.e. a real island (a run of >= 2 trusted k-mers) exists
        constexpr int NTB = 4 * ((E_CNT + 3) / 4);
        uint32_t tb[NTB];
        int y[E_CNT];  // the real (unmasked, unsorted) counts
        int rmin = 2147483647, rmax = 0;
#pragma unroll
        for (int e = 0; e < NTB; ++e) tb[e] = 0;
        uint32_t adj_bits = 0;
#pragma unroll
        for (int e = 0; e < E_CNT; ++e) {
            const int g = e * 16 + l;
            y[e] = g < kcnt ? count_at(g) : 2147483647;
            rmin = y[e] < rmin ? y[e] : rmin;
            rmax = g < kcnt && y[e] > rmax ? y[e] : rmax;
            tb[e] = row_bits(__ballot(g < kcnt && y[e] >= s && ((not_polya2 >> e) & 1u)), row);
            adj_bits |= tb[e] & (tb[e] >> 1);
            if (e > 0) adj_bits |= (tb[e - 1] >> 15) & tb[e];
        }
        const bool adj = adj_bits != 0;
        const bool clean = !screened && v0 >= t0 && adj;
        const int im = kcnt >> 1, ih = kcnt > 0 ? kcnt - 1 : 0;
        int vm, vh;
        RCQ_SORTED_AT(vm, im);
        RCQ_SORTED_AT(vh, ih);
        if (!screened && n_below < kcnt) cls = n_below >= kcnt - (kcnt >> 3) ? 4 : (n_below >= kcnt - (kcnt >> 2) ? 3 : (n_below >= kcnt - (kcnt >> 1) ? 2 : 1));
        RCQ_STOP(16, strong_self, prev, (int)found, s, t0, v0, n_below, vm, vh, (int)clean, cls, rmin, rmax, (int)(tb[0] ^ tb[1] ^ tb[2] ^ tb[3] ^ tb[NTB - 4] ^ tb[NTB - 3] ^ tb[NTB - 2] ^ tb[NTB - 1]),
                 y[0] ^ y[1] ^ y[2] ^ y[3] ^ y[E_CNT - 4] ^ y[E_CNT - 3] ^ y[E_CNT - 2] ^ y[E_CNT - 1], (int)(ma[0] ^ mt[0]));
        // The same on the REAL counts.  The sorted array hides the windows the threshold scan masks as poly-A (:1530-1541:
        // >= k - max(7, k/2) A's or T's -- one read in eight has such a window), so v0 < 0 for a read that is clean in every
        // other respect.  ErrorCorrection itself never looks at that mask (its own, :870-931, asks for >= k - 2): if every
        // real count reaches t0 and its own mask has two adjacent trusted k-mers, the argument above holds word for word.  l / m / h of the real counts: two row reductions and a
        // descent over the bits of the largest count for the element of rank kcnt / 2 (no second sort).
        bool clean2 = false;
        {
            int q;
            q = row_xor<1>(rmin); rmin = q < rmin ? q : rmin;
            q = row_xor<2>(rmin); rmin = q < rmin ? q : rmin;
            q = row_xor<4>(rmin); rmin = q < rmin ? q : rmin;
            q = row_xor<8>(rmin); rmin = q < rmin ? q : rmin;
            clean2 = !clean && !screened && kcnt > 0 && rmin >= t0 && adj;
            if (__ballot(clean2)) {  // (wave-uniform)
                q = row_xor<1>(rmax); rmax = q > rmax ? q : rmax;
                q = row_xor<2>(rmax); rmax = q > rmax ? q : rmax;
                q = row_xor<4>(rmax); rmax = q > rmax ? q : rmax;
                q = row_xor<8>(rmax); rmax = q > rmax ? q : rmax;
                const int wmax = max(max(__builtin_amdgcn_readlane(rmax, 0), __builtin_amdgcn_readlane(rmax, 16)),
                                     max(__builtin_amdgcn_readlane(rmax, 32), __builtin_amdgcn_readlane(rmax, 48)));
                int prefix = 0;
                for (int b = 31 - __builtin_clz((unsigned)wmax | 1u); b >= 0; --b) {
                    const int cd = prefix | (1 << b);
                    int below = 0;
#pragma unroll
                    for (int e = 0; e < E_CNT; ++e) below += y[e] < cd ? 1 : 0;
                    below += row_xor<1>(below);
                    below += row_xor<2>(below);
                    below += row_xor<4>(below);
                    below += row_xor<8>(below);
                    prefix = below <= im ? cd : prefix;  // the largest value with at most im counts below it = the element of rank im
                }
                if (clean2) {
                    cls = 0;
                    if (live && l == 0) {
                        A.ret[r] = 0;
                        A.l[r] = rmin;
                        A.m[r] = prefix;
                        A.h[r] = rmax;
                    }
                }
            }
        }
        RCQ_STOP(17, strong_self, prev, (int)found, s, t0, v0, vm, vh, (int)clean, (int)clean2, cls, (int)(tb[0] ^ tb[1] ^ tb[2] ^ tb[3] ^ tb[NTB - 4] ^ tb[NTB - 3] ^ tb[NTB - 2] ^ tb[NTB - 1]),
                 (int)other);
        // Candidate for k_single (rc_single.h, condition (2)): every letter ACGT, and the trusted mask -- counts >= s and
        // not poly-A at threshold 2 (:870-931) -- has no 1-run of length one and only 0-runs of exactly k, or of at most k
        // at either end of the read.  k_single checks it again (it needs the runs' positions anyway); this flag only keeps the
        // reads that cannot pass out of its work list, so that its rows are filled with reads that mostly do.
        if (A.cand) {
            bool cand = false;
            int cand_runs = 1;  // 0-runs of a candidate = segments k_single will walk: its work list is grouped by that number
            if constexpr (E_CNT <= 12) {
                constexpr int NW = (E_CNT + 3) / 4;  // 64-bit words of the mask
                static_assert(4 * NW == NTB, "the trusted mask above");
                if (!clean && !clean2 && !screened && other == 0 && kcnt >= 5) {
                    // T = the trusted mask, Z = its zero bits inside [0, kcnt); shifts run over the NW-word number
                    uint64_t T[NW], Z[NW], S[NW], E[NW];
                    uint64_t any = 0, iso = 0;
#pragma unroll
                    for (int q = 0; q < NW; ++q) {
                        T[q] = (uint64_t)tb[4 * q] | ((uint64_t)tb[4 * q + 1] << 16) | ((uint64_t)tb[4 * q + 2] << 32) | ((uint64_t)tb[4 * q + 3] << 48);
                        const int left = kcnt - 64 * q;
                        const uint64_t mq = left >= 64 ? ~0ull : (left > 0 ? ((1ull << left) - 1ull) : 0ull);
                        Z[q] = ~T[q] & mq;
                        any |= T[q];
                    }
                    int nruns = 0;
#pragma unroll
                    for (int q = 0; q < NW; ++q) {
                        const uint64_t t_up = (T[q] << 1) | (q > 0 ? T[q - 1] >> 63 : 0ull), t_dn = (T[q] >> 1) | (q + 1 < NW ? T[q + 1] << 63 : 0ull);
                        const uint64_t z_up = (Z[q] << 1) | (q > 0 ? Z[q - 1] >> 63 : 0ull), z_dn = (Z[q] >> 1) | (q + 1 < NW ? Z[q + 1] << 63 : 0ull);
                        iso |= T[q] & ~t_up & ~t_dn;
                        S[q] = Z[q] & ~z_up;  // starts and ends of the 0-runs: paired in order, every end must be its start + k - 1
                        E[q] = Z[q] & ~z_dn;
                        nruns += __popcll(S[q]);
                    }
                    bool shape = any != 0 && iso == 0 && nruns >= 1 && nruns <= 3;
                    // The runs' first and last k-mers: the (q + 1)-th set bit of S and of E, q < 3.  Every lane of the row holds
                    // the same words, so the six extractions are dealt out to six lanes -- lane q takes run q's start, lane 4 + q
                    // its end (clear the lowest set bit q times, then find the lowest) -- instead of every lane walking all six
                    // (round 6: the walk was 190 of the row pass's 1 516 vector instructions).
                    const int q = l & 3;
                    uint64_t M[NW];
#pragma unroll
                    for (int v = 0; v < NW; ++v) M[v] = (l & 4) ? E[v] : S[v];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        bool done = i >= q;
#pragma unroll
                        for (int v = 0; v < NW; ++v) {
                            const bool here = !done && M[v] != 0;
                            M[v] = here ? (M[v] & (M[v] - 1)) : M[v];
                            done = done || here;
                        }
                    }
                    int z = 0;
                    {
                        bool got = false;
#pragma unroll
                        for (int v = 0; v < NW; ++v) {
                            const bool here = !got && M[v] != 0;
                            z = here ? 64 * v + __ffsll((long long)M[v]) - 1 : z;
                            got = got || here;
                        }
                    }
                    const int z0 = z, z1 = dpp<0x104>(z);  // row_shl:4 -- lane q reads lane q + 4
                    const int rl = z1 - z0 + 1;
                    const bool mine = q < nruns && l < 3;
                    const bool good = rl == k || ((z0 == 0 || z1 == kcnt - 1) && rl < k);
                    shape = shape && row_bits(__ballot(mine && !good), row) == 0;
                    const int rrq = mine ? (int)((uint32_t)z0 | ((uint32_t)rl << 8)) : 0;  // per run: first k-mer | length << 8 (what k_single walks)
                    const uint32_t rr[3] = {(uint32_t)rrq, (uint32_t)dpp<0x101>(rrq), (uint32_t)dpp<0x102>(rrq)};  // (lane 0: its own, lane 1's, lane 2's)
                    cand = shape;
                    cand_runs = nruns;
                    if (cand && live && l == 0) A.runs[r] = make_uint2(rr[0] | (rr[1] << 16), rr[2] | ((uint32_t)nruns << 16));
                }
            }
            if (live && l == 0) A.cand[r] = cand ? (uint8_t)cand_runs : 0;
        }
        RCQ_STOP(18, strong_self, prev, (int)found, v0, vm, vh, (int)clean, cls);
        if (clean) {
            cls = 0;
            if (live && l == 0) {
                A.ret[r] = 0;
                A.l[r] = v0;
                A.m[r] = vm;
                A.h[r] = vh;
            }
        }
    }
    if (live && l == 0) {
        A.strong[r] = strong_self;
        A.info[r] = screened ? 4 : ((found ? 1 : 0) | ((found && prev == 2) ? 2 : 0));
        if (A.cls) A.cls[r] = (uint8_t)cls;
    }
    return cls;
}

// the threshold kernel over a batch in HBM: 256-thread workgroups = 16 reads
template <int E_CNT, int E_BASE>
__global__ __launch_bounds__(256) void k_threshold_q(rc_kernel_args A)
{
    const int row = (threadIdx.x & 63) >> 4;
    if (A.worklist) {
        // the pass of a length tier over its list (rc_launch_tier_lists: the tier's reads in locality order, the mates of a
        // pair adjacent -- rows 2j and 2j+1 again); the list's length lives on the device, a fixed grid walks it
        const uint32_t n_list = *A.n_work;
        for (uint32_t wv = (uint32_t)blockIdx.x * 4u + (threadIdx.x >> 6); wv * 4u < n_list; wv += gridDim.x * 4u) {
            const uint32_t i = wv * 4u + (uint32_t)row;
            const bool live = i < n_list;
            const uint32_t r = live ? A.worklist[i] : 0u;
            uint32_t o = 0;
            int len = 0;
            if (live) {
                o = A.off[r];
                len = (int)(A.off[r + 1] - o) - 1;
            }
            rcq_threshold_row<E_CNT, E_BASE>(
                A, r, live, len, [&](int p) { return (uint32_t)A.seq[o + p]; }, [&](int g) { return A.counts[o + g]; });
        }
        return;
    }
    // rows 2j and 2j+1 of a wave hold the two mates of a pair (paired: reads u and n/2 + u;
    // interleaved: reads 2u and 2u+1), so the pair threshold is one lane exchange away
    const uint32_t wv = (uint32_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    uint32_t r;
    bool live;
    if (A.mode == 1) {
        const uint32_t half = A.n >> 1, u = wv * 2u + (uint32_t)(row >> 1);
        live = u < half;
        r = (row & 1) ? half + u : u;
    } else {
        r = wv * 4u + (uint32_t)row;
        live = r < A.n;
    }
    uint32_t o = 0;
    int len = 0;
    if (live) {
        o = A.off[r];
        len = (int)(A.off[r + 1] - o) - 1;
    }
    if (A.tier_hi != RC_TIER_ALL || A.tier_lo >= 0) {  // (uniform) another tier's read: cls = 0, nothing else
        int ml = len;
        if (A.mode != 0) {
            const int mm = __shfl(len, (threadIdx.x & 63) ^ 16, 64);
            ml = mm > ml ? mm : ml;
        }
        if (live && !rc_in_tier(A, ml)) {
            if ((threadIdx.x & 15) == 0) {
                if (A.cls) A.cls[r] = 0;
                if (A.cand) A.cand[r] = 0;
            }
            live = false;
            len = 0;
        }
        if (!__ballot(live)) return;
    }
    rcq_threshold_row<E_CNT, E_BASE>(
        A, r, live, len, [&](int p) { return (uint32_t)A.seq[o + p]; }, [&](int g) { return A.counts[o + g]; });
}
