// rc_correct_kernel.h -- the device back end of rc_correct_core.h (DevWaveT), the LDS layout of a read's state and the
// correction kernel k_correct (K3: ErrorCorrection + GetKmerInformation, ErrorCorrection.cpp:682-1480, :1567-1602, i.e. the
// body of ErrorCorrection_Thread, :73-136).  A header because the kernel is instantiated in several translation units:
// rc_correct.hip holds the instances for any k and table layout, rc_correct_k*.hip those compiled for one k and one
// layout (k, its masks and shifts, and the slot format are then immediates instead of scalar registers the kernel
// does not have: 78 SGPRs at 8 waves per SIMD against several hundred live scalar values).
#pragma once
#include <cstddef>
#include <cstdio>
#include <cstdlib>

#include "rc_internal.h"
#include "rc_device.h"

// PROF = per-phase s_memtime accounting (dev aid, RC_PHASE_PROF=1); compiled out otherwise
// TRACE = record what the reference prints under -verbose for every threshold iteration
//         (ErrorCorrection.cpp:856-857, :1088-1094) into a per-read record; compiled out otherwise
// KT_ = the k this instance is compiled for (0: any k, read from the run parameters)
template <bool PROF, bool TRACE = false, bool SORT_REGS = true, int KT_ = 0>
struct DevWaveT {
    static const int STRIDE = 64;
    static constexpr int KT = KT_;
    int lane;
    unsigned long long t_last = 0;
    int cur_phase = 0;
    unsigned long long *acc = nullptr;  // PROF builds: 16 accumulators in LDS (0-7 phases of a read, 8-15 inside the search); a
                                        // private array indexed by the phase would live in scratch memory
#ifdef RC_EXP_ROUNDS  // dev builds: gather rounds per read, reported in place of l (tools/rounds_hist.py)
    int rounds = 0;
    __device__ __forceinline__ void stat(int i, int v)
    {
        if (i == 3) rounds += v;
    }
#else
    int rounds = 0, rounds_max = 0;  // PROF builds: gather rounds of the current read / of the wave's worst read
    long long rounds_sum = 0;
    // PROF builds: what the search's speculation bought (rc_correct_core.h: w.stat): [0] keep-runs, [1] nodes they kept, [2] cached
    // nodes they were offered, [4] gap-window rounds, [5] their probes, [6] alternative chains walked, [7] of them to the end,
    // [8] probes of the gather rounds, [9] rounds with alternative chains
    long long stv[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    __device__ __forceinline__ void stat(int i, int v)
    {
        if (PROF && i == 3) rounds += v;
        if (PROF && i != 3 && i < 10) stv[i] += v;
    }
#endif
    // trace record of the current read: [0] flags (bit 0: passed the screens, i.e. "Before
    // correction" is printed), [1] iterations seen, then RC_TRACE_WORDS per recorded iteration:
    // strong, trust, has_bitmap, 0, 32 words of the per-base "strong trusted" bitmap
    int32_t *tr = nullptr;
    int tr_cap = 0;
    __device__ __forceinline__ void trace_passed()
    {
        if (TRACE && lane == 0) tr[0] |= 1;
    }
    __device__ __forceinline__ void trace_iter(int strong, int trust)
    {
        if (TRACE) {
            const int it = uni(tr[1]);
            if (lane == 0) {
                if (it < tr_cap) {
                    int32_t *e = tr + 2 + (size_t)it * RC_TRACE_WORDS;
                    e[0] = strong;
                    e[1] = trust;
                    e[2] = 0;
                    e[3] = 0;
                }
                tr[1] = it + 1;
            }
            sync();
        }
    }
    __device__ __forceinline__ void trace_strong(const unsigned char *strongb, int len)
    {
        if (TRACE) {
            const int it = uni(tr[1]) - 1;
            if (it >= 0 && it < tr_cap) {
                int32_t *e = tr + 2 + (size_t)it * RC_TRACE_WORDS;
                for (int c = 0; c < RC_MAX_READ_LENGTH / 64; ++c) {
                    const uint64_t m = ballot64(c << 6, len, [&](int q) { return strongb[q] != 0; });
                    if (lane == 0) {
                        e[4 + 2 * c] = (int32_t)(uint32_t)m;
                        e[5 + 2 * c] = (int32_t)(uint32_t)(m >> 32);
                    }
                }
                if (lane == 0) e[2] = 1;
            }
            sync();
        }
    }
    __device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
    __device__ __forceinline__ uint64_t uni64(uint64_t x)
    {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32));
        return ((uint64_t)hi << 32) | lo;
    }
    __device__ __forceinline__ void phase(int id)
    {
        // At the boundaries between the phases of a read the lane number becomes a new value to the
        // compiler, so that nothing derived from it (lane * 24, ~lane, 1 << lane, an LDS address ...) is
        // computed once before the per-read loop and kept alive -- i.e. spilled to scratch -- across the
        // search: each phase recomputes its few from the one register that holds the lane.  (Not inside
        // the search, ids 8 and up: there the hoisting is wanted.)
        if (id < 8) asm volatile("" : "+v"(lane));
        if (PROF) {
            unsigned long long t = __builtin_readcyclecounter();
            if (lane == 0) acc[cur_phase] += t - t_last;
            t_last = t;
            cur_phase = id;
        }
    }
    rc_table_view T;
    int k;
    rc_frame *stack;  // this wave's frames in HBM scratch

    // Lanes of ONE wave exchange data through LDS.  DS instructions of a wave execute in issue
    // order, so no hardware wait is needed -- only the compiler must keep LDS accesses on their
    // side of this point (a wavefront-scope fence; __syncthreads() would also drain every pending
    // global load/store with s_waitcnt vmcnt(0), dozens of times per read).
    __device__ __forceinline__ void sync()
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // bit l of the result = pred(base + l) for base + l < n (one element per lane)
    template <class F>
    __device__ __forceinline__ uint64_t ballot64(int base, int n, F pred)
    {
        const int i = base + lane;
        bool p = false;
        if (i < n) p = pred(i);
        return __ballot(p);
    }
    // body(base + l, l) on lane l for base + l < n
    template <class F>
    __device__ __forceinline__ void for_lanes64(int base, int n, F body)
    {
        const int i = base + lane;
        if (i < n) body(i, lane);
    }

    __device__ __forceinline__ int reduce_add(int x)
    {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        return __builtin_amdgcn_readfirstlane(x);
    }

    // min / max over the 64 lanes: DPP inside each 16-lane row (no LDS), v_readlane across rows
    template <bool MAX>
    __device__ __forceinline__ uint32_t wave_minmax_u32(uint32_t x)
    {
        auto op = [](uint32_t a, uint32_t b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
        x = op(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        x = op(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
        x = op(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true));  // row_half_mirror
        x = op(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, true));  // row_mirror
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)x, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)x, 16);
        const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)x, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
        return op(op(r0, r1), op(r2, r3));
    }
    __device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) { return wave_minmax_u32<false>(x); }
    __device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) { return wave_minmax_u32<true>(x); }

    // out[j] = min(in[j0 .. j]) for j in [j0, n), n <= 64 (one element per lane): DPP row_shr scan inside
    // each 16-lane row, the totals of the rows before through v_readlane
    __device__ __forceinline__ void prefix_min(const int *in, int *out, int j0, int n)
    {
        const bool live = lane >= j0 && lane < n;
        int x = live ? in[lane] : 2147483647;
        auto step = [&](int y) { x = y < x ? y : x; };
        step(__builtin_amdgcn_update_dpp(2147483647, x, 0x111, 0xF, 0xF, false));  // row_shr:1
        step(__builtin_amdgcn_update_dpp(2147483647, x, 0x112, 0xF, 0xF, false));  // row_shr:2
        step(__builtin_amdgcn_update_dpp(2147483647, x, 0x114, 0xF, 0xF, false));  // row_shr:4
        step(__builtin_amdgcn_update_dpp(2147483647, x, 0x118, 0xF, 0xF, false));  // row_shr:8
        if (n > 16) {
            const int r0 = __builtin_amdgcn_readlane(x, 15), r1 = __builtin_amdgcn_readlane(x, 31), r2 = __builtin_amdgcn_readlane(x, 47);
            const int m01 = r0 < r1 ? r0 : r1, m012 = m01 < r2 ? m01 : r2;
            const int row = lane >> 4;
            const int before = row == 0 ? 2147483647 : (row == 1 ? r0 : (row == 2 ? m01 : m012));
            x = before < x ? before : x;
        }
        if (live) out[lane] = x;
    }
    // min(a[lo .. hi)), hi - lo <= 64 (INT_MAX for an empty range)
    __device__ __forceinline__ int min_range(const int *a, int lo, int hi)
    {
        const int i = lo + lane;
        const uint32_t x = i < hi ? ((uint32_t)a[i] ^ 0x80000000u) : 0xFFFFFFFFu;
        return (int)(wave_min_u32(x) ^ 0x80000000u);
    }

    uint32_t n_req = 0;  // PROF builds: bucket reads issued by this lane
    uint32_t n_probe = 0, n_absent = 0;  // PROF builds: k-mers looked up by this lane / of them not in the table
    // dir: the search direction the k-mer was extended in (+1: its last base is the one that varies between the lanes of a
    // gather round, -1: its first -- i.e. the last of its reverse complement), 0: no such neighbours
    __device__ __forceinline__ int get(rc_kmer km, int dir = 0)
    {
        if (km.inv != -1) return 0;
        const uint64_t rv = rc_revcomp(km.code, KT ? KT : k);
        const uint64_t canon = rv < km.code ? rv : km.code;
        const int c = rc_table_lookup_o<true, true>(T, canon, dir > 0 ? km.code : (dir < 0 ? rv : canon), PROF ? &n_req : nullptr);
        if (PROF) {
            ++n_probe;
            n_absent += c == 0 ? 1u : 0u;
        }
        return c;
    }

    __device__ __forceinline__ int lookup(uint64_t code) { return rc_table_lookup(T, rc_canonical(code, KT ? KT : k), PROF ? &n_req : nullptr); }

    // in-register bitonic network over E*64 elements (element g = e*64 + lane lives in x[e]):
    // strides below 64 exchange through the lane crossbar, strides >= 64 between a lane's own
    // registers.  No LDS traffic inside the network, ~5 VALU per compare-exchange.
    template <int E>
    __device__ __forceinline__ void bitonic_regs(int *a, int n)
    {
        int x[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int g = e * 64 + lane;
            x[e] = g < n ? a[g] : 2147483647;
        }
#pragma unroll
        for (int size = 2; size <= E * 64; size <<= 1) {
#pragma unroll
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                if (stride >= 64) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const int pe = e ^ (stride >> 6);
                        if (pe > e) {
                            const bool up = ((e * 64) & size) == 0;
                            const int lo = x[e] < x[pe] ? x[e] : x[pe];
                            const int hi = x[e] < x[pe] ? x[pe] : x[e];
                            x[e] = up ? lo : hi;
                            x[pe] = up ? hi : lo;
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const int g = e * 64 + lane;
                        const int y = __shfl_xor(x[e], stride, 64);
                        const bool up = (g & size) == 0;
                        const bool lower = (lane & stride) == 0;
                        const int lo = x[e] < y ? x[e] : y;
                        const int hi = x[e] < y ? y : x[e];
                        x[e] = (up == lower) ? lo : hi;
                    }
                }
            }
        }
        sync();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int g = e * 64 + lane;
            if (g < n) a[g] = x[e];
        }
        sync();
    }

    // ascending sort of a[0..n) in LDS (register network up to 256 elements, LDS bitonic above;
    // a[] has room for the next power of two)
    // SORT_REGS = false: LDS network only -- k_correct sorts on cold paths only (threshold retries, the
    // fused front end of long single-end reads) and should not carry three unrolled networks
    __device__ __forceinline__ void sort(int *a, int n)
    {
        if (SORT_REGS) {
            if (n <= 64) return bitonic_regs<1>(a, n);
            if (n <= 128) return bitonic_regs<2>(a, n);
            if (n <= 256) return bitonic_regs<4>(a, n);
        }
        int n2 = 1;
        while (n2 < n) n2 <<= 1;
        for (int i = n + lane; i < n2; i += 64) a[i] = 2147483647;
        sync();
        for (int size = 2; size <= n2; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = lane; t < (n2 >> 1); t += 64) {
                    const int pos = 2 * t - (t & (stride - 1));
                    const int par = pos + stride;
                    const bool up = (pos & size) == 0;
                    const int x = a[pos], y = a[par];
                    if ((x > y) == up) {
                        a[pos] = y;
                        a[par] = x;
                    }
                }
                sync();
            }
        }
    }

    // Search stack.  The first lstack_n frames live in LDS -- in the sort buffer v[], which nothing
    // touches while a search runs -- so that a push or a pop is a handful of LDS operations; deeper
    // frames (rare) go to the wave's HBM scratch, where every push has to wait for its store.
    uint32_t *lstack = nullptr;
    int lstack_n = 0;
    static constexpr int FRAME_DWORDS = (int)(sizeof(rc_frame) / 4);
    __device__ __forceinline__ void stack_push(int sp, const rc_frame &f)
    {
        if (sp < lstack_n) {
            if (lane == 0) {
                uint32_t *d = lstack + sp * FRAME_DWORDS;
                d[0] = (uint32_t)f.code;
                d[1] = (uint32_t)(f.code >> 32);
                d[2] = (uint32_t)f.inv;
                d[3] = (uint32_t)f.pos;
                d[4] = (uint32_t)f.t;
                d[5] = (uint32_t)f.threshold;
                d[6] = (uint32_t)f.fix_cnt;
                d[7] = (uint32_t)f.bottleneck;
                d[8] = (uint32_t)f.cnt.c0;
                d[9] = (uint32_t)f.cnt.c1;
                d[10] = (uint32_t)f.cnt.c2;
                d[11] = (uint32_t)f.cnt.c3;
                d[12] = (uint32_t)f.mask;
            }
            sync();
            return;
        }
        if (lane == 0) stack[sp] = f;
        __threadfence_block();
    }
    __device__ __forceinline__ void stack_top(int idx, rc_frame &f)
    {
        static_assert(sizeof(rc_frame) == 56 && offsetof(rc_frame, mask) == 48, "rc_frame layout");
        uint32_t d[13];
        if (idx < lstack_n) {
            const uint32_t *p = lstack + idx * FRAME_DWORDS;
#pragma unroll
            for (int q = 0; q < 13; ++q) d[q] = p[q];
        } else {
            // all loads go out before the first value is used (one round trip, not thirteen).  The frame
            // was written by lane 0 of this very wave: workgroup-scope loads (served by the XCD's L2,
            // past the CU's L1) see it -- system-scope (volatile) loads went all the way to memory
            const uint32_t *p = reinterpret_cast<const uint32_t *>(stack + idx);
#pragma unroll
            for (int q = 0; q < 13; ++q) d[q] = __hip_atomic_load(p + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        f.code = ((uint64_t)(uint32_t)uni((int)d[1]) << 32) | (uint32_t)uni((int)d[0]);
        f.inv = uni((int)d[2]);
        f.pos = uni((int)d[3]);
        f.t = uni((int)d[4]);
        f.threshold = uni((int)d[5]);
        f.fix_cnt = uni((int)d[6]);
        f.bottleneck = uni((int)d[7]);
        f.cnt.c0 = uni((int)d[8]);
        f.cnt.c1 = uni((int)d[9]);
        f.cnt.c2 = uni((int)d[10]);
        f.cnt.c3 = uni((int)d[11]);
        f.mask = uni((int)d[12]);
    }
    __device__ __forceinline__ void stack_set_mask(int idx, int mask)
    {
        if (idx < lstack_n) {
            if (lane == 0) lstack[idx * FRAME_DWORDS + 12] = (uint32_t)mask;
            sync();
            return;
        }
        if (lane == 0) stack[idx].mask = mask;
        __threadfence_block();
    }
};
typedef DevWaveT<false> DevWave;

#ifndef RC_DEQUEUE
#define RC_DEQUEUE 8  // reads per work-counter atomic
#endif
#define RC_META_WORDS 6  // per read of a dequeued chunk: index, offset, end offset, strong, info, mate's strong

struct rc_lds_layout {
    int cap, cap2;
    size_t o_counts, o_v, o_isl, o_seg, o_base, o_path, o_best, o_strongb, o_polya, o_qual, o_masks, o_spec, o_pk, o_meta, total;
    int mask_words;
};

// segments of a read: at most (len + 1) / (k + 1) + 1 (fill_args checks it).  The 192-base class keeps room for
// 20 -- enough down to k = 10 -- and leaves smaller k to the next class: its LDS is what allows 8 waves per SIMD
static __host__ __device__ constexpr inline int rc_seg_capacity(int cap) { return cap <= 192 ? 20 : cap / 6 + 4; }

static __host__ __device__ constexpr inline rc_lds_layout rc_layout(int cap)
{
    rc_lds_layout L{};
    L.cap = cap;
    int c2 = 64;
    while (c2 < cap) c2 <<= 1;
    L.cap2 = c2;
    // islands are runs of >= 2 trusted k-mers separated by >= 1 other: at most (kcnt+1)/3 (+1 for the
    // fall-back island); segments lie between islands that cover >= k bases each and are >= 1 base
    // apart: at most (len+1)/(k+1) + 1, which fill_args() checks against this capacity
    const int nseg = rc_seg_capacity(cap);
    const int nisl = cap / 3 + 2;
    size_t o = 0;
    L.mask_words = cap / 64 + 2;
    L.o_masks = o;
    o += (size_t)L.mask_words * 8 * 5;
    L.o_spec = o;
    o += (size_t)RC_SPEC_ENTRIES * 4 + (size_t)RC_SPEC * (8 + 4 + 4 * 4) + 16 + (size_t)(4 + RC_MEMO_MAX) * 4;
    L.o_pk = o;
    o += (size_t)(cap / 16 + 4) * 4;
    L.o_meta = o;
    o += (size_t)RC_DEQUEUE * RC_META_WORDS * 4;
    L.o_counts = o;
    o += (size_t)cap * 4;
    L.o_v = o;
    o += (size_t)c2 * 4;
    L.o_seg = o;
    o += (size_t)nseg * sizeof(rc_segment);
    // the islands exist between two searches only, the entries of the speculation cache during one: same bytes
    // where they fit
    if ((size_t)nisl * sizeof(rc_island) <= (size_t)RC_SPEC_ENTRIES * 4 + (size_t)RC_SPEC * 8) {
        L.o_isl = L.o_spec;
    } else {
        L.o_isl = o;
        o += (size_t)nisl * sizeof(rc_island);
    }
    o = (o + 15) & ~(size_t)15;  // rc_pack_read reads base[] as dwords
    L.o_base = o;
    o += cap;
    L.o_path = o;
    o += cap;
    L.o_best = o;
    o += cap;
    L.o_strongb = o;
    o += cap;
    L.o_polya = o;
    o += cap;
    L.o_qual = o;
    o += cap;
    L.total = (o + 15) & ~(size_t)15;
    return L;
}

__device__ __forceinline__ void rc_carve(uint8_t *lds, const rc_lds_layout &L, rc_read_state &S)
{
    S.counts = reinterpret_cast<int *>(lds + L.o_counts);
    S.v = reinterpret_cast<int *>(lds + L.o_v);
    S.seg = reinterpret_cast<rc_segment *>(lds + L.o_seg);
    S.isl = reinterpret_cast<rc_island *>(lds + L.o_isl);
    S.base = lds + L.o_base;
    S.path = reinterpret_cast<signed char *>(lds + L.o_path);
    S.best = reinterpret_cast<signed char *>(lds + L.o_best);
    S.strongb = lds + L.o_strongb;
    S.polya = lds + L.o_polya;
    S.qual = reinterpret_cast<signed char *>(lds + L.o_qual);
    uint64_t *mm = reinterpret_cast<uint64_t *>(lds + L.o_masks);
    S.m_a = mm;
    S.m_t = mm + L.mask_words;
    S.m_n = mm + 2 * L.mask_words;
    S.m_inv = mm + 3 * L.mask_words;
    S.m_x = mm + 4 * L.mask_words;
    S.pk = reinterpret_cast<uint32_t *>(lds + L.o_pk);
    S.spec_code = reinterpret_cast<uint64_t *>(lds + L.o_spec);
    S.spec_cnt = reinterpret_cast<int *>(lds + L.o_spec + RC_SPEC * 8);
    S.spec_inv = S.spec_cnt + RC_SPEC_ENTRIES;
    S.spec_ret = S.spec_inv + RC_SPEC;
    S.spec_keep = S.spec_ret + RC_SPEC;
    S.spec_thr = S.spec_keep + RC_SPEC;
    S.spec_mask = S.spec_thr + RC_SPEC;
    S.spec_meta = S.spec_mask + RC_SPEC;
    S.memo = S.spec_meta + 4;
}

__device__ __forceinline__ int rc_base_code(uint32_t c)
{
    int b = 5;
    b = c == 'A' ? 0 : b;
    b = c == 'C' ? 1 : b;
    b = c == 'G' ? 2 : b;
    b = c == 'T' ? 3 : b;
    b = c == 'N' ? 4 : b;
    return b;
}

struct rc_kernel_args {
    rc_table_view T;
    rc_run_params P;
    int mode;
    uint32_t n;
    uint8_t *seq;
    const uint8_t *qual;   // one byte per arena byte, or (qual_bits) one BIT per arena byte: quality > badQualityThreshold
    int qual_bits;
    uint32_t qual_split, qual_base2;  // bit mode: arena bytes >= qual_split have their bits at byte qual_base2 on (second arena of a host batch)
    const uint32_t *off;
    const int32_t *counts;  // K1 output, indexed like seq
    int32_t *strong, *info;
    uint8_t *cls;              // K2 -> compaction: 1 = the read still needs k_correct (nullptr: no classification)
    uint8_t *cand;             // K2 -> k_single: > 0 = the read's trusted k-mers have the shape of isolated substitutions: the number of its untrusted stretches (nullptr: not computed)
    uint2 *runs;               // K2 -> k_single, candidates only: the stretches, (first k-mer | length << 8) 16 bits each: x = run 0 | run 1 << 16, y = run 2 | count << 16
    const uint32_t *worklist;  // k_correct: the reads to process (nullptr: all of [0, n))
    const uint32_t *n_work;    // k_correct: number of entries of worklist (device memory)
    int32_t *ret, *l, *m, *h;
    rc_frame *stack;
    int stack_frames;  // per wave
    uint32_t *work;
    int cap;        // LDS capacity of the runtime-layout kernels (k_threshold)
    int cap_class;  // capacity class of k_correct (192 / 320 / 1024)
    unsigned long long *phase_cycles;  // [8], PROF builds only
    int32_t *rounds_out;               // PROF builds only: gather rounds of every read k_correct processed (nullptr: not wanted)
    uint32_t work_stride;              // entries between the sections of worklist (rc_internal.h)
    int fused_front_end;               // 1: k_correct computes the read's own threshold (single-end, no threshold kernel ran)
    int32_t *trace;                    // TRACE builds only: n x (2 + trace_cap * RC_TRACE_WORDS) words
    int trace_cap;
    // Length tiers (rc_api_batch.hip: rc_correct_device_impl).  A batch whose reads do not all fit the fastest kernels is
    // processed tier by tier; a unit (a read, or a pair: mates need each other's threshold) belongs to the tier
    // (tier_lo, tier_hi] that holds the length of its longer read, and a threshold kernel launched for one tier
    // leaves the reads of the others alone (cls = 0: not on this pass's work list).  RC_TIER_ALL: no tiers.
    int tier_lo, tier_hi;
    // single-end batches only: the pairStrongTrustThreshold every read is corrected with (ErrorCorrection.h:27; -1 = none,
    // what ErrorCorrection_Thread passes for a read without a mate, ErrorCorrection.cpp:121) -- rc_correct_read
    int pair_override;
};
// the unit of read r: its own length and, in paired / interleaved batches, its mate's (off: n + 1 offsets, NULs counted)
__device__ __forceinline__ int rc_unit_max_len(const rc_kernel_args &A, uint32_t r, int len)
{
    if (A.mode == 0) return len;
    const uint32_t half = A.n >> 1, mr = A.mode == 1 ? (r < half ? r + half : r - half) : (r ^ 1u);
    const int ml = (int)(A.off[mr + 1] - A.off[mr]) - 1;
    return ml > len ? ml : len;
}
__device__ __forceinline__ bool rc_in_tier(const rc_kernel_args &A, int unit_max_len) { return unit_max_len > A.tier_lo && unit_max_len <= A.tier_hi; }

template <class W>
__device__ __forceinline__ void rc_load_read(W &w, const rc_kernel_args &A, rc_read_state &S, uint32_t o, int len, int lane, bool with_qual)
{
    S.len = len;
    S.kcnt = len >= RC_K(A.P) ? len - RC_K(A.P) + 1 : 0;
    for (int i = lane; i < len; i += 64) {
        S.base[i] = (unsigned char)rc_base_code(A.seq[o + i]);
        S.counts[i] = i < S.kcnt ? A.counts[o + i] : 0;
        if (with_qual) {
            if (A.qual_bits) {
                // the vetoes only compare a quality with badQualityThreshold (ErrorCorrection.cpp:1313-1466) and
                // test qual[0] != 0 (FASTQ marker): a bit per base stands in for the byte
                uint32_t p = o + (uint32_t)i;
                const uint8_t *qb = A.qual;
                if (p >= A.qual_split) {
                    p -= A.qual_split;
                    qb += A.qual_base2;
                }
                S.qual[i] = ((qb[p >> 3] >> (p & 7u)) & 1u) ? (signed char)127 : (signed char)-128;
            } else
                S.qual[i] = (signed char)A.qual[o + i];
        }
    }
    w.sync();
    rc_build_masks(w, S);
    rc_pack_read(w, S);
}

#ifndef RC_HEADS
#define RC_HEADS 8    // work-queue heads (one per XCD)
#endif
#ifndef RC_K3_WAVES
#define RC_K3_WAVES 8  // waves per SIMD the register allocation of k_correct is held to (measured: 5: 104, 6: 92, 7: 86, 8: 81 ms)
#endif

// CAP = LDS capacity class (bases per read, a multiple of 64): the layout is a compile-time
// constant, so every array of rc_read_state is an immediate LDS address instead of a scalar
// register (the kernel's scalar state does not fit the 102 SGPRs a wave has as it is)
// resident waves per SIMD the register allocation is held to: RC_K3_WAVES where the LDS of the
// capacity class allows that many, else what the LDS allows
static __host__ __device__ constexpr int rc_k3_waves(int cap)
{
    const int by_lds = cap <= 192 ? 8 : (cap <= 320 ? 5 : 1);
    return by_lds < RC_K3_WAVES ? by_lds : RC_K3_WAVES;
}

// KT = the k the instance is compiled for (0: any, from the run parameters); TL = the table's slot layout as the
// instance knows it: 0 any (from the table view), 1 PACKED without remainder extension, 2 PACKED with one
template <int CAP, bool PROF, bool TRACE, int KT = 0, int TL = 0>
__global__ __launch_bounds__(64, rc_k3_waves(CAP)) void k_correct(rc_kernel_args A)
{
    constexpr rc_lds_layout L = rc_layout(CAP);
    __shared__ __attribute__((aligned(16))) uint8_t lds[L.total + (PROF ? 128 : 0)];
    rc_read_state S;
    rc_carve(lds, L, S);
    typedef DevWaveT<PROF, TRACE, false, KT> W;
    W w;
    w.lane = threadIdx.x;
    w.tr_cap = A.trace_cap;
    if (PROF) {
        w.acc = reinterpret_cast<unsigned long long *>(lds + L.total);
        if (w.lane < 16) w.acc[w.lane] = 0;
        w.sync();
        w.t_last = __builtin_readcyclecounter();
    }
    w.T = A.T;  // (a copy in registers: what the instance knows at compile time becomes a constant in it)
    if (TL != 0) w.T.layout = 1;
    if (TL == 1) w.T.ext = 0;
    if (KT != 0) w.T.k = KT;
    w.k = A.P.k;
    w.stack = A.stack + (size_t)blockIdx.x * A.stack_frames;
    w.lstack = reinterpret_cast<uint32_t *>(S.v);
    w.lstack_n = (int)((size_t)L.cap2 * 4 / sizeof(rc_frame));
    if (PROF && w.lane == 0) atomicMin(A.phase_cycles + 8, (unsigned long long)wall_clock64());
    // Work distribution.  The reference hands out read indices from one mutex-protected counter
    // (ErrorCorrection.cpp:87-90); one device-scope atomic word sustains only ~88 dequeues/us on
    // MI355X, so the queue [0, n_work) is cut into RC_HEADS slices with a head word each (128 B
    // apart).  A wave starts on the slice of its XCD (workgroup b runs on XCD b % 8 -- an affinity
    // for speed, nothing depends on it), takes RC_DEQUEUE entries per atomic, and moves on to the
    // next slice when one is exhausted, so no slice is left behind whatever the placement.
    // The list has RC_WORK_CLASSES sections, taken one after the other (the reads the threshold kernel
    // expects to search longest come first: a launch ends when its last read does, and a read that
    // runs for tens of thousands of gather rounds had better not be the last one started).  Without
    // a list (no classification ran) there is one section, the reads themselves.
    const int n_sections = A.n_work ? RC_WORK_CLASSES : 1;
    int section = 0;
    uint32_t n_work = A.n_work ? (uint32_t)__builtin_amdgcn_readfirstlane((int)A.n_work[0]) : A.n;
    const uint32_t *list = A.worklist;
    uint32_t *heads = A.work;
    uint32_t chunk_lo = 0, chunk_hi = 0, chunk_base = 0;
    uint32_t *meta = reinterpret_cast<uint32_t *>(lds + L.o_meta);
    int head = (int)(blockIdx.x % RC_HEADS), heads_done = 0;
    for (;;) {
        if (chunk_lo >= chunk_hi) {
            bool got = false;
            for (;;) {
                while (heads_done < RC_HEADS) {
                    const uint32_t lo = (uint32_t)(((uint64_t)n_work * (uint32_t)head) / RC_HEADS);
                    const uint32_t hi = (uint32_t)(((uint64_t)n_work * (uint32_t)(head + 1)) / RC_HEADS);
                    uint32_t r0 = hi - lo;
                    if (hi > lo) {  // (an empty slice costs no atomic)
                        if (w.lane == 0) r0 = atomicAdd(heads + head * 32, (uint32_t)RC_DEQUEUE);
                        r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r0);
                    }
                    if (r0 < hi - lo) {
                        chunk_lo = lo + r0;
                        chunk_hi = hi - chunk_lo > (uint32_t)RC_DEQUEUE ? chunk_lo + RC_DEQUEUE : hi;
                        got = true;
                        break;
                    }
                    head = head + 1 == RC_HEADS ? 0 : head + 1;
                    ++heads_done;
                }
                if (got || ++section >= n_sections) break;
                n_work = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.n_work[section]);
                list = A.worklist + (size_t)section * A.work_stride;
                heads = A.work + section * (RC_HEADS * 32);
                head = (int)(blockIdx.x % RC_HEADS);
                heads_done = 0;
            }
            if (!got) {
                if (PROF && w.lane == 0) atomicMin(A.phase_cycles + 9, (unsigned long long)wall_clock64());
                break;
            }
            // everything k_correct needs to know about the chunk's reads before it can load them, fetched
            // by one lane per read in two dependent round trips per chunk instead of six per read
            chunk_base = chunk_lo;
            if ((uint32_t)w.lane < chunk_hi - chunk_lo) {
                uint32_t ri = chunk_lo + (uint32_t)w.lane;
                if (list) ri = list[ri];
                uint32_t *mw = meta + w.lane * RC_META_WORDS;
                mw[0] = ri;
                mw[1] = A.off[ri];
                mw[2] = A.off[ri + 1];
                if (!A.fused_front_end) {
                    mw[3] = (uint32_t)A.strong[ri];
                    mw[4] = (uint32_t)A.info[ri];
                    if (A.mode == 1) {
                        const uint32_t half = A.n >> 1;
                        mw[5] = (uint32_t)A.strong[ri < half ? ri + half : ri - half];
                    } else if (A.mode == 2)
                        mw[5] = (uint32_t)A.strong[ri ^ 1u];
                }
            }
            w.sync();
        }
        const int ci = (int)(chunk_lo - chunk_base);
        ++chunk_lo;
        const uint32_t *me = meta + ci * RC_META_WORDS;
        const uint32_t r = (uint32_t)w.uni((int)me[0]);
        const uint32_t o = (uint32_t)w.uni((int)me[1]);
        w.phase(0);
        if (TRACE) {
            w.tr = A.trace + (size_t)r * (2 + (size_t)A.trace_cap * RC_TRACE_WORDS);
            if (w.lane == 0) w.tr[0] = w.tr[1] = 0;
            w.sync();
        }
#if defined(RC_EXP_STOP) && RC_EXP_STOP == 0
        if (w.lane == 0) A.ret[r] = (int)o;
        continue;
#endif
        rc_load_read(w, A, S, o, w.uni((int)(me[2] - o) - 1), w.lane, true);
#if defined(RC_EXP_STOP) && RC_EXP_STOP == 1
        if (w.lane == 0) A.ret[r] = (int)S.m_inv[0] + (int)S.pk[1] + S.counts[5] + S.qual[7];
        w.sync();
        continue;
#endif
        int strong0, info0;
        if (A.fused_front_end) {  // single-end: no mate to wait for, the threshold pass runs right here
            w.phase(1);
            strong0 = rc_front_end(w, S, A.P, &info0);
        } else {
            strong0 = w.uni((int)me[3]);
            info0 = w.uni((int)me[4]);
        }
        int pair_t = A.pair_override;
        if (A.mode != 0) pair_t = rc_min(strong0, w.uni((int)me[5]));
        w.phase(1);
        if (!A.fused_front_end && S.kcnt > 0 && !(info0 & 4)) rc_polya_flags(w, S, RC_K(A.P));
#if defined(RC_EXP_STOP) && RC_EXP_STOP == 2
        if (w.lane == 0) A.ret[r] = (int)S.polya[3] + pair_t;
        w.sync();
        continue;
#endif
        const int ret = rc_correct_read(w, S, A.P, pair_t, strong0, info0);
#if defined(RC_EXP_STOP) && (RC_EXP_STOP == 3 || RC_EXP_STOP == 4 || RC_EXP_STOP == 5)
        if (w.lane == 0) A.ret[r] = ret + S.best[3] + S.seg[0].from;
        w.sync();
        continue;
#endif
        w.phase(6);
        w.sync();
        if (ret > 0) {
            for (int i = w.lane; i < S.len; i += 64) {
                const int f = S.best[i];
                if (f != -1) {
                    A.seq[o + i] = (uint8_t)("ACGT"[f]);
                    S.base[i] = (unsigned char)f;
                }
            }
            w.sync();
            rc_pack_read(w, S);
        }
        int l, m, h;
        rc_kmer_info(w, S, A.P, ret, &l, &m, &h);
        if (w.lane == 0) {
            A.ret[r] = ret;
#ifdef RC_EXP_ROUNDS
            A.l[r] = w.rounds;
            w.rounds = 0;
#else
            A.l[r] = l;
#endif
            A.m[r] = m;
            A.h[r] = h;
        }
        w.sync();
        w.phase(7);
#ifndef RC_EXP_ROUNDS
        if (PROF) {
            if (A.rounds_out && w.lane == 0) A.rounds_out[r] = w.rounds;
            w.rounds_sum += w.rounds;
            w.rounds_max = w.rounds > w.rounds_max ? w.rounds : w.rounds_max;
            w.rounds = 0;
        }
#endif
    }
    if (PROF && w.lane == 0) {
        w.phase(7);
        for (int i = 0; i < 8; ++i) atomicAdd(A.phase_cycles + i, w.acc[i]);
        for (int i = 8; i < 16; ++i) atomicAdd(A.phase_cycles + 8 + i, w.acc[i]);  // slots 16..23
        atomicMax(A.phase_cycles + 10, (unsigned long long)wall_clock64());
#ifndef RC_EXP_ROUNDS
        atomicAdd(A.phase_cycles + 11, (unsigned long long)w.rounds_sum);
        atomicMax(A.phase_cycles + 12, (unsigned long long)w.rounds_max);
        for (int i = 0; i < 10; ++i)
            if (i != 3) atomicAdd(A.phase_cycles + 24 + i, (unsigned long long)w.stv[i]);
#endif
    }
    if (PROF) {
        const int req = w.reduce_add((int)w.n_req), prb = w.reduce_add((int)w.n_probe), abs_ = w.reduce_add((int)w.n_absent);
        if (w.lane == 0) {
            atomicAdd(A.phase_cycles + 13, (unsigned long long)(uint32_t)req);
            atomicAdd(A.phase_cycles + 14, (unsigned long long)(uint32_t)prb);
            atomicAdd(A.phase_cycles + 15, (unsigned long long)(uint32_t)abs_);
        }
    }
}

// the instances compiled for one k and one table layout (rc_correct_k*.hip), reads up to 191 bases: nullptr if there
// is none for this k / layout
typedef void (*rc_k3_launcher)(dim3 grid, hipStream_t stream, const rc_kernel_args &A);
rc_k3_launcher rc_k3_special(int k, int layout, int ext);
#define RC_K3_SPECIAL(KT, TL)                                                                                     \
    void rc_k3_launch_##KT##_##TL(dim3 grid, hipStream_t stream, const rc_kernel_args &A)                          \
    {                                                                                                             \
        hipLaunchKernelGGL((k_correct<192, false, false, KT, TL>), grid, dim3(64), 0, stream, A);                  \
    }
