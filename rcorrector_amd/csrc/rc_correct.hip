// rc_correct.hip -- device back end of rc_correct_core.h and the two per-read kernels:
//   k_threshold (K2): GetStrongTrustedThreshold for every read   (ErrorCorrection.cpp:1482-1565)
//   k_correct   (K3): ErrorCorrection + GetKmerInformation        (ErrorCorrection.cpp:682-1480,
//                     :1567-1602), i.e. the body of ErrorCorrection_Thread (:73-136)
// One 64-lane wavefront (= one workgroup) per read; persistent waves pull read indices from an
// atomic counter, which is the reference's mutex-protected batchUsed counter (:87-90) and
// absorbs the heavy tail of the search.  Per-read state is carved out of dynamic LDS.
#include <cstddef>
#include <cstdio>
#include <cstdlib>

#include "rc_internal.h"
#include "rc_device.h"

// PROF = per-phase s_memtime accounting (dev aid, RC_PHASE_PROF=1); compiled out otherwise
// TRACE = record what the reference prints under -verbose for every threshold iteration
//         (ErrorCorrection.cpp:856-857, :1088-1094) into a per-read record; compiled out otherwise
template <bool PROF, bool TRACE = false, bool SORT_REGS = true>
struct DevWaveT {
    static const int STRIDE = 64;
    int lane;
    unsigned long long t_last = 0;
    int cur_phase = 0;
    unsigned long long acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // 0-7 phases of a read, 8-15 inside the search
#ifdef RC_EXP_ROUNDS  // dev builds: gather rounds per read, reported in place of l (tools/rounds_hist.py)
    int rounds = 0;
    __device__ __forceinline__ void stat(int i, int v)
    {
        if (i == 3) rounds += v;
    }
#else
    int rounds = 0, rounds_max = 0;  // PROF builds: gather rounds of the current read / of the wave's worst read
    long long rounds_sum = 0;
    __device__ __forceinline__ void stat(int i, int v)
    {
        if (PROF && i == 3) rounds += v;
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
            acc[cur_phase] += t - t_last;
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

    uint32_t n_req = 0;  // PROF builds: bucket reads issued by this lane
    __device__ __forceinline__ int get(rc_kmer km)
    {
        return km.inv == -1 ? rc_table_lookup(T, rc_canonical(km.code, k), PROF ? &n_req : nullptr) : 0;
    }

    __device__ __forceinline__ int lookup(uint64_t code) { return rc_table_lookup(T, rc_canonical(code, k), PROF ? &n_req : nullptr); }

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

static __host__ __device__ constexpr inline int rc_seg_capacity(int cap) { return cap / 6 + 4; }

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
    o += (size_t)RC_SPEC * (8 + 4 * 4 + 4 + 4 * 4);
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
    L.o_isl = o;
    o += (size_t)nisl * sizeof(rc_island);
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
    S.spec_inv = S.spec_cnt + RC_SPEC * 4;
    S.spec_ret = S.spec_inv + RC_SPEC;
    S.spec_keep = S.spec_ret + RC_SPEC;
    S.spec_thr = S.spec_keep + RC_SPEC;
    S.spec_mask = S.spec_thr + RC_SPEC;
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
    const uint32_t *worklist;  // k_correct: the reads to process (nullptr: all of [0, n))
    const uint32_t *n_work;    // k_correct: number of entries of worklist (device memory)
    int32_t *ret, *l, *m, *h;
    rc_frame *stack;
    int stack_frames;  // per wave
    uint32_t *work;
    int cap;        // LDS capacity of the runtime-layout kernels (k_threshold)
    int cap_class;  // capacity class of k_correct (192 / 320 / 1024)
    unsigned long long *phase_cycles;  // [8], PROF builds only
    uint32_t work_stride;              // entries between the sections of worklist (rc_internal.h)
    int fused_front_end;               // 1: k_correct computes the read's own threshold (single-end, no threshold kernel ran)
    int32_t *trace;                    // TRACE builds only: n x (2 + trace_cap * RC_TRACE_WORDS) words
    int trace_cap;
};

template <class W>
__device__ __forceinline__ void rc_load_read(W &w, const rc_kernel_args &A, rc_read_state &S, uint32_t o, int len, int lane, bool with_qual)
{
    S.len = len;
    S.kcnt = len >= A.P.k ? len - A.P.k + 1 : 0;
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

// Software-pipelined over the reads of a wave: the kernel has no table probes, so a read costs two
// dependent HBM round trips (its offsets, then its bases and counts) against a few hundred
// instructions of work -- the offsets are fetched two reads ahead and the bases / counts one read
// ahead, into registers, while the current read is processed out of LDS.
#define RC_K2_PF 4  // 64-base chunks prefetched per read (longer reads load their tail directly)
__global__ __launch_bounds__(64) void k_threshold(rc_kernel_args A)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const rc_lds_layout L = rc_layout(A.cap);
    rc_read_state S;
    rc_carve(lds, L, S);
    DevWave w;
    w.lane = threadIdx.x;
    w.T = A.T;
    w.k = A.P.k;
    w.stack = nullptr;
    const int lane = w.lane, k = A.P.k;
    const uint32_t G = gridDim.x;
    uint32_t r = blockIdx.x;
    if (r >= A.n) return;
    uint32_t o_cur = A.off[r], e_cur = A.off[r + 1];
    uint32_t o_nxt = 0, e_nxt = 0;
    if (r + G < A.n) {
        o_nxt = A.off[r + G];
        e_nxt = A.off[r + G + 1];
    }
    uint8_t pb[RC_K2_PF];
    int32_t pc[RC_K2_PF];
    auto fetch = [&](uint32_t o, uint32_t e) {
        const int len = (int)(e - o) - 1, kcnt = len >= k ? len - k + 1 : 0;
#pragma unroll
        for (int c = 0; c < RC_K2_PF; ++c) {
            const int i = c * 64 + lane;
            pb[c] = i < len ? A.seq[o + i] : (uint8_t)0;
            pc[c] = i < kcnt ? A.counts[o + i] : 0;
        }
    };
    fetch(o_cur, e_cur);
    for (;;) {
        const uint32_t o = o_cur;
        const int len = (int)(e_cur - o) - 1;
        S.len = len;
        S.kcnt = len >= k ? len - k + 1 : 0;
#pragma unroll
        for (int c = 0; c < RC_K2_PF; ++c) {
            const int i = c * 64 + lane;
            if (i < len) {
                S.base[i] = (unsigned char)rc_base_code(pb[c]);
                S.counts[i] = pc[c];
            }
        }
        for (int i = RC_K2_PF * 64 + lane; i < len; i += 64) {
            S.base[i] = (unsigned char)rc_base_code(A.seq[o + i]);
            S.counts[i] = i < S.kcnt ? A.counts[o + i] : 0;
        }
        // next read's data and the offsets of the one after it go out now
        const uint32_t rn = r + G;
        const bool more = rn < A.n;
        if (more) {
            o_cur = o_nxt;
            e_cur = e_nxt;
            fetch(o_cur, e_cur);
            if (rn + G < A.n) {
                o_nxt = A.off[rn + G];
                e_nxt = A.off[rn + G + 1];
            }
        }
        w.sync();
        rc_build_masks(w, S);
        int info;
        const int strong = rc_front_end(w, S, A.P, &info);
        if (lane == 0) {
            A.strong[r] = strong;
            A.info[r] = info;
        }
        w.sync();
        if (!more) break;
        r = rn;
    }
}

#include "rc_quarter.h"

// ---- K1 + K2 fused, for batches probed in locality order ------------------------------------------
// The list-driven probe kernel (rc_table.hip: k_probe_list) has its workgroup's reads and their
// counts in LDS anyway; running the quarter-wave threshold / classification on them right there
// saves K2's pass over the counts (4 bytes per base, written and read back through HBM) and the
// write itself for every read the classification finishes.  Reads up to 160 bases / 128 k-mers.
// RC_FUSED_TILE = bytes of the workgroup's local arena, WAVES = resident waves per SIMD the register
// allocation is held to: 2816 B = 16 reads of up to 160 bases, one full pass of the 16-row threshold
// code, 17 KB of LDS; 4096 B = 32 reads of up to 119 bases, two passes, 24 KB.  Six waves (80 VGPRs)
// for both: the small arena would allow eight by LDS, but at 64 VGPRs the compiler no longer keeps the
// eight 16-byte loads of two probes in flight -- measured on 25 M x 150 bp against a 1.26 GB table:
// 44.1 / 42.3 / 46.2 / 47.7 ms at 5 / 6 / 7 / 8 waves (a 545 MB table, mostly cache hits, prefers
// eight: 65.5 -> 62.1 ms).
#ifndef RC_FUSED_SMALL_WAVES
#define RC_FUSED_SMALL_WAVES 6
#endif
template <int RC_FUSED_TILE, int WAVES, bool EXT>
__global__ __launch_bounds__(RC_PROBE_THREADS) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_probe_threshold_list(rc_kernel_args A, size_t nbytes, const uint32_t *__restrict__ list,
                                                                           uint32_t reads_per_block, int32_t *__restrict__ counts)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_raw[(RC_FUSED_TILE + 64) / 4];
    __shared__ uint32_t s_code[RC_FUSED_TILE / 16 + 4];
    __shared__ uint16_t s_inv[RC_FUSED_TILE / 16 + 4];
    __shared__ uint16_t s_nul[RC_FUSED_TILE / 16 + 4];
    __shared__ uint32_t s_lpos[RC_PLIST_MAX_READS + 1], s_gpos[RC_PLIST_MAX_READS], s_len1[RC_PLIST_MAX_READS], s_rid[RC_PLIST_MAX_READS];
    __shared__ __attribute__((aligned(16))) int32_t s_cnt[RC_FUSED_TILE + 64];
    __shared__ uint8_t s_cls[RC_PLIST_MAX_READS];
    const int t = threadIdx.x, k = A.P.k;
    const uint8_t *seq = A.seq;
    const uint32_t i0 = blockIdx.x * reads_per_block;
    const uint32_t nr = A.n - i0 < reads_per_block ? A.n - i0 : reads_per_block;
    for (int c = t; c < (RC_FUSED_TILE + 64) / 4; c += RC_PROBE_THREADS) s_raw[c] = 0;
    if ((uint32_t)t < nr) {
        const uint32_t r = list[i0 + t], g0 = A.off[r];
        s_rid[t] = r;
        s_gpos[t] = g0;
        s_len1[t] = A.off[r + 1] - g0;  // bases + the NUL
    }
    __syncthreads();
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
    for (uint32_t j = (uint32_t)t >> 6; j < nr; j += RC_PROBE_THREADS / 64) {  // copy, aligned dwords, outside bytes masked to NUL
        const uint32_t g0 = s_gpos[j], lp = s_lpos[j], g1 = g0 + s_len1[j] - 1;
        const uint32_t w0 = g0 >> 2, w1 = (g1 + 3) >> 2;
        for (uint32_t w = w0 + ((uint32_t)t & 63u); w < w1; w += 64u) {
            uint32_t v;
            if ((size_t)4 * w + 4 <= nbytes) {
                v = *reinterpret_cast<const uint32_t *>(seq + (size_t)4 * w);
            } else {
                v = 0;
                for (size_t q = 0; (size_t)4 * w + q < nbytes; ++q) v |= (uint32_t)seq[(size_t)4 * w + q] << (8 * q);
            }
            const uint32_t lo = 4 * w < g0 ? g0 - 4 * w : 0, hi = 4 * w + 4 > g1 ? 4 * w + 4 - g1 : 0;
            uint32_t m = 0xFFFFFFFFu;
            if (lo) m &= 0xFFFFFFFFu << (8 * lo);
            if (hi) m &= 0xFFFFFFFFu >> (8 * hi);
            s_raw[(lp >> 2) + (w - w0)] = v & m;
        }
    }
    __syncthreads();
    const uint32_t total = s_lpos[nr];
    for (int chunk = t; chunk < RC_FUSED_TILE / 16 + 2; chunk += RC_PROBE_THREADS) {
        const uint4 v = *reinterpret_cast<const uint4 *>(s_raw + 4 * chunk);
        uint32_t code, inv, nul;
        rc_pack16(v, code, inv, nul);
        s_code[chunk] = code;
        s_inv[chunk ^ 1] = (uint16_t)inv;
        s_nul[chunk ^ 1] = (uint16_t)nul;
    }
    if (t < 2) s_code[RC_FUSED_TILE / 16 + 2 + t] = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t *m_inv = reinterpret_cast<const uint32_t *>(s_inv);
    const uint32_t *m_nul = reinterpret_cast<const uint32_t *>(s_nul);
#pragma unroll 2
    for (uint32_t a = 4 + (uint32_t)t; a + (uint32_t)k <= total; a += RC_PROBE_THREADS) {  // probe: counts stay in LDS
        const int mw = a >> 5, ms = a & 31;
        const uint64_t nulw = (((uint64_t)m_nul[mw] << 32) | m_nul[mw + 1]) << ms;
        if (nulw >> (64 - k)) continue;
        const uint64_t invw = (((uint64_t)m_inv[mw] << 32) | m_inv[mw + 1]) << ms;
        int cnt = 0;
        if (!(invw >> (64 - k))) {
            const int cw = a >> 4, cs = 2 * (a & 15);
            uint64_t x = ((uint64_t)s_code[cw] << 32) | s_code[cw + 1];
            if (cs) x = (x << cs) | ((uint64_t)s_code[cw + 2] >> (32 - cs));
            cnt = rc_table_lookup<EXT>(A.T, rc_canonical(x >> (64 - 2 * k), k));
        }
        s_cnt[a] = cnt;
    }
    __syncthreads();
    // thresholds + classes: 16 reads per pass (one per 16-lane row; the list keeps mates adjacent)
    const uint8_t *raw8 = reinterpret_cast<const uint8_t *>(s_raw);
    for (uint32_t j0 = 0; j0 < nr; j0 += RC_PROBE_THREADS / 16) {
        const uint32_t j = j0 + ((uint32_t)t >> 4);
        const bool live = j < nr;
        const uint32_t lp = live ? s_lpos[j] : 0;
        const int len = live ? (int)s_len1[j] - 1 : 0;
        const int cls = rcq_threshold_row<8, 10>(
            A, live ? s_rid[j] : 0, live, len, [&](int p) { return (uint32_t)raw8[lp + p]; }, [&](int g) { return s_cnt[lp + g]; });
        if (live && (t & 15) == 0) s_cls[j] = (uint8_t)cls;
    }
    __syncthreads();
    // the counts k_correct will read: those of the reads that still need it, four per lane (a read
    // starts at the same offset modulo 4 here and in the arena; the up to three words in front of its
    // first count and behind its last one belong to NULs and to the last k-1 positions of a read,
    // which hold no count -- k >= 4, rc_launch_probe_threshold_list)
    for (uint32_t j = (uint32_t)t >> 6; j < nr; j += RC_PROBE_THREADS / 64) {
        if (A.cls && !s_cls[j]) continue;
        const int kcnt = (int)s_len1[j] - 1 - k + 1;
        const uint32_t lp = s_lpos[j], g0 = s_gpos[j], head = lp & 3u;
        const int4 *src = reinterpret_cast<const int4 *>(s_cnt + (lp - head));
        int4 *dst = reinterpret_cast<int4 *>(counts + (size_t)(g0 - head));
        for (int q = t & 63; 4 * q < kcnt + (int)head; q += 64) dst[q] = src[q];
    }
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
static constexpr int rc_k3_waves(int cap)
{
    const int by_lds = cap <= 192 ? 8 : (cap <= 320 ? 5 : 1);
    return by_lds < RC_K3_WAVES ? by_lds : RC_K3_WAVES;
}

template <int CAP, bool PROF, bool TRACE>
__global__ __launch_bounds__(64, rc_k3_waves(CAP)) void k_correct(rc_kernel_args A)
{
    constexpr rc_lds_layout L = rc_layout(CAP);
    __shared__ __attribute__((aligned(16))) uint8_t lds[L.total];
    rc_read_state S;
    rc_carve(lds, L, S);
    DevWaveT<PROF, TRACE, false> w;
    w.lane = threadIdx.x;
    w.tr_cap = A.trace_cap;
    if (PROF) w.t_last = __builtin_readcyclecounter();
    w.T = A.T;
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
        rc_load_read(w, A, S, o, w.uni((int)(me[2] - o) - 1), w.lane, true);
        int strong0, info0;
        if (A.fused_front_end) {  // single-end: no mate to wait for, the threshold pass runs right here
            w.phase(1);
            strong0 = rc_front_end(w, S, A.P, &info0);
        } else {
            strong0 = w.uni((int)me[3]);
            info0 = w.uni((int)me[4]);
        }
        int pair_t = -1;
        if (A.mode != 0) pair_t = rc_min(strong0, w.uni((int)me[5]));
        w.phase(1);
        if (!A.fused_front_end && S.kcnt > 0 && !(info0 & 4)) rc_polya_flags(w, S, A.P.k);
        const int ret = rc_correct_read(w, S, A.P, pair_t, strong0, info0);
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
#endif
    }
    if (PROF) {
        const int req = w.reduce_add((int)w.n_req);
        if (w.lane == 0) atomicAdd(A.phase_cycles + 13, (unsigned long long)(uint32_t)req);
    }
}

static int rc_cap_for(int max_len)
{
    int cap = ((max_len + 1 + 63) / 64) * 64;
    if (cap < 64) cap = 64;
    return cap;
}

static int fill_args(rc_ctx *ctx, const rc_device_batch_args &a, rc_kernel_args &A)
{
    if (!ctx->d_buckets) {
        rc_set_error(ctx, "correct: no k-mer table loaded");
        return RC_ERR_STATE;
    }
    if (!ctx->params_set) {
        rc_set_error(ctx, "correct: run parameters not set (rc_set_run_params)");
        return RC_ERR_STATE;
    }
    if (a.max_len >= RC_MAX_READ_LENGTH) {
        rc_set_error(ctx, "correct: read of %d bases exceeds the %d-base limit (utils.h:7)", a.max_len, RC_MAX_READ_LENGTH - 1);
        return RC_ERR_ARG;
    }
    if (a.mode == 1 && (a.n & 1)) {
        rc_set_error(ctx, "correct: paired mode needs an even number of reads");
        return RC_ERR_ARG;
    }
    if (a.mode == 2 && (a.n & 1)) {
        rc_set_error(ctx, "correct: interleaved mode needs an even number of reads");
        return RC_ERR_ARG;
    }
    A.T = rc_view(ctx);
    A.P = ctx->P;
    A.mode = a.mode;
    A.n = a.n;
    A.seq = a.seq;
    A.qual = a.qual;
    A.qual_bits = a.qual_bits;
    A.qual_split = a.qual_split;
    A.qual_base2 = a.qual_base2;
    A.off = a.off;
    A.counts = (const int32_t *)ctx->counts.p;
    A.strong = (int32_t *)ctx->strong.p;
    A.info = (int32_t *)ctx->info.p;
    A.cls = nullptr;
    A.worklist = nullptr;
    A.work_stride = 0;
    A.n_work = nullptr;
    A.ret = a.ret;
    A.l = a.l;
    A.m = a.m;
    A.h = a.h;
    A.stack = nullptr;
    A.stack_frames = 0;
    A.work = (uint32_t *)ctx->work.p;
    A.cap = rc_cap_for(a.max_len);
    {   // capacity class of k_correct: room for the read and for its segments (tiny k only)
        const int need_seg = (a.max_len + 1) / (ctx->k + 1) + 1;
        int cls = A.cap <= 192 ? 192 : (A.cap <= 320 ? 320 : 1024);
        while (cls < 1024 && rc_seg_capacity(cls) < need_seg) cls = cls == 192 ? 320 : 1024;
        if (rc_seg_capacity(cls) < need_seg) {
            rc_set_error(ctx, "correct: reads of %d bases with k = %d are not supported (too many segments)", a.max_len, ctx->k);
            return RC_ERR_ARG;
        }
        A.cap_class = cls;
    }
    A.phase_cycles = nullptr;
    A.trace = nullptr;
    A.trace_cap = 0;
    A.fused_front_end = a.mode == 0 && !ctx->thr_ready;
    return RC_OK;
}

// classify: let the quarter-wave kernel finish the reads that need no correction (ret, l, m, h
// written there) and flag the others in ctx->cls; ctx->cls_ready tells the caller whether it did
int rc_launch_threshold(rc_ctx *ctx, const rc_device_batch_args &a, bool classify)
{
    ctx->cls_ready = false;
    if (a.n == 0) return RC_OK;
    rc_kernel_args A;
    int rc = fill_args(ctx, a, A);
    if (rc) return rc;
    const rc_lds_layout L = rc_layout(A.cap);
    // four reads per wave when every read of the batch fits the quarter-wave layout (rc_quarter.h)
    const bool quarter = a.max_len <= rcq::MAX_LEN && a.max_len - A.P.k + 1 <= rcq::MAX_KCNT && !ctx->env_k2_wave_per_read;
    if (quarter && classify && a.ret && ctx->trace_cap == 0 && !ctx->env_no_classify) {
        if ((rc = rc_dbuf_reserve(ctx, &ctx->cls, (size_t)a.n + 256))) return rc;
        A.cls = (uint8_t *)ctx->cls.p;
        ctx->cls_ready = true;
    }
    rc_timer_begin(ctx);
    if (quarter && a.max_len <= 160 && a.max_len - A.P.k + 1 <= 128) {
        hipLaunchKernelGGL((k_threshold_q<8, 10>), dim3((a.n + 15) / 16), dim3(256), 0, ctx->stream, A);
    } else if (quarter) {
        hipLaunchKernelGGL((k_threshold_q<16, 20>), dim3((a.n + 15) / 16), dim3(256), 0, ctx->stream, A);
    } else {
        unsigned grid = (unsigned)ctx->n_cu * 32u;
        if (grid > a.n) grid = a.n;
        hipLaunchKernelGGL(k_threshold, dim3(grid), dim3(64), L.total, ctx->stream, A);
    }
    rc_timer_end(ctx, RC_T_THRESH);
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}

// K1 + K2 in one kernel over the locality list (ctx->loc_list); *done = false if the batch does not fit it
int rc_launch_probe_threshold_list(rc_ctx *ctx, const rc_device_batch_args &a, size_t nbytes, bool *done)
{
    *done = false;
    ctx->cls_ready = false;
    if (a.n == 0 || a.max_len > 160 || a.max_len - ctx->k + 1 > 128 || ctx->k < 4 || ctx->env_k2_wave_per_read || ctx->env_no_fuse) return RC_OK;
    rc_kernel_args A;
    int rc = fill_args(ctx, a, A);
    if (rc) return rc;
    if (a.ret && ctx->trace_cap == 0 && !ctx->env_no_classify) {
        if ((rc = rc_dbuf_reserve(ctx, &ctx->cls, (size_t)a.n + 256))) return rc;
        A.cls = (uint8_t *)ctx->cls.p;
        ctx->cls_ready = true;
    }
    // reads per workgroup: a read takes its bases, the NUL and up to 6 bytes of alignment; whole passes
    // of the 16-row threshold code (mates stay together); the small arena unless the large one holds
    // twice the reads
    auto fit = [&](int tile) {
        uint32_t r = (uint32_t)((tile - 8) / (a.max_len + 8));
        if (r > RC_PLIST_MAX_READS) r = RC_PLIST_MAX_READS;
        return r & ~15u;
    };
    const bool large = fit(2816) < 32 && fit(4096) >= 32;
    const uint32_t rpb = large ? fit(4096) : fit(2816);
    if (rpb == 0) return RC_OK;  // (not reached: 16 reads of 160 bases fit)
    rc_timer_begin(ctx);
    const dim3 grid((a.n + rpb - 1) / rpb), block(RC_PROBE_THREADS);
    const uint32_t *list = (const uint32_t *)ctx->loc_list.p;
    int32_t *counts = (int32_t *)ctx->counts.p;
    if (large && ctx->ext)
        hipLaunchKernelGGL((k_probe_threshold_list<4096, 6, true>), grid, block, 0, ctx->stream, A, nbytes, list, rpb, counts);
    else if (large)
        hipLaunchKernelGGL((k_probe_threshold_list<4096, 6, false>), grid, block, 0, ctx->stream, A, nbytes, list, rpb, counts);
    else if (ctx->ext)
        hipLaunchKernelGGL((k_probe_threshold_list<2816, RC_FUSED_SMALL_WAVES, true>), grid, block, 0, ctx->stream, A, nbytes, list, rpb, counts);
    else
        hipLaunchKernelGGL((k_probe_threshold_list<2816, RC_FUSED_SMALL_WAVES, false>), grid, block, 0, ctx->stream, A, nbytes, list, rpb, counts);
    rc_timer_end(ctx, RC_T_PROBE);
    RC_CHECK_HIP(ctx, hipGetLastError());
    *done = true;
    return RC_OK;
}

int rc_launch_correct(rc_ctx *ctx, const rc_device_batch_args &a)
{
    if (a.n == 0) return RC_OK;
    rc_kernel_args A;
    int rc = fill_args(ctx, a, A);
    if (rc) return rc;
    unsigned grid = (unsigned)ctx->n_cu * 4u * (unsigned)rc_k3_waves(ctx->trace_cap > 0 ? 1024 : A.cap_class);
    if (ctx->env_k3_grid_waves > 0 && ctx->env_k3_grid_waves < RC_K3_WAVES) grid = (unsigned)ctx->n_cu * 4u * (unsigned)ctx->env_k3_grid_waves;
    if (grid > a.n) grid = a.n;
    A.stack_frames = A.cap + 64;
    rc = rc_dbuf_reserve(ctx, &ctx->stack, (size_t)grid * A.stack_frames * sizeof(rc_frame));
    if (rc) return rc;
    A.stack = (rc_frame *)ctx->stack.p;
    // queue heads and phase counters to zero; the work-list length (written by the compaction) stays
    RC_CHECK_HIP(ctx, hipMemsetAsync(ctx->work.p, 0, RC_WORK_NWORK_OFF, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemsetAsync((char *)ctx->work.p + RC_WORK_PHASE_OFF, 0, 192, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemsetAsync((char *)ctx->work.p + RC_WORK_PHASE_OFF + 64, 0xff, 16, ctx->stream));  // the two minima
    A.phase_cycles = (unsigned long long *)((char *)ctx->work.p + RC_WORK_PHASE_OFF);
    if (ctx->cls_ready) {
        A.worklist = (const uint32_t *)ctx->worklist.p;
        A.work_stride = (uint32_t)ctx->work_stride;
        A.n_work = (const uint32_t *)((char *)ctx->work.p + RC_WORK_NWORK_OFF);
    }
    if (ctx->trace_cap > 0) {
        rc = rc_dbuf_reserve(ctx, &ctx->trace, (size_t)a.n * (2 + (size_t)ctx->trace_cap * RC_TRACE_WORDS) * 4);
        if (rc) return rc;
        A.trace = (int32_t *)ctx->trace.p;
        A.trace_cap = ctx->trace_cap;
    }
    rc_timer_begin(ctx);
    if (ctx->trace_cap > 0)
        hipLaunchKernelGGL((k_correct<1024, false, true>), dim3(grid), dim3(64), 0, ctx->stream, A);
    else if (ctx->phase_prof && A.cap_class == 192)
        hipLaunchKernelGGL((k_correct<192, true, false>), dim3(grid), dim3(64), 0, ctx->stream, A);
    else if (A.cap_class == 192)
        hipLaunchKernelGGL((k_correct<192, false, false>), dim3(grid), dim3(64), 0, ctx->stream, A);
    else if (A.cap_class == 320)
        hipLaunchKernelGGL((k_correct<320, false, false>), dim3(grid), dim3(64), 0, ctx->stream, A);
    else
        hipLaunchKernelGGL((k_correct<1024, false, false>), dim3(grid), dim3(64), 0, ctx->stream, A);
    rc_timer_end(ctx, RC_T_CORRECT);
    if (ctx->phase_prof && A.cap_class == 192) {
        unsigned long long pc[24];
        uint32_t nwork = a.n, nsec[RC_WORK_CLASSES] = {0};
        RC_CHECK_HIP(ctx, hipMemcpyAsync(pc, A.phase_cycles, sizeof pc, hipMemcpyDeviceToHost, ctx->stream));
        if (A.n_work) RC_CHECK_HIP(ctx, hipMemcpyAsync(nsec, A.n_work, sizeof nsec, hipMemcpyDeviceToHost, ctx->stream));
        RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (A.n_work) {
            nwork = 0;
            for (int i = 0; i < RC_WORK_CLASSES; ++i) nwork += nsec[i];
        }
        static const char *names[8] = {"dequeue+load", "polya", "islands/segments", "search", "lower-thresholds", "post-filters", "apply+kmerinfo", "store"};
        unsigned long long tot = 0;
        for (int i = 0; i < 8; ++i) tot += pc[i];
        ctx->k3_listed += nwork;
        ctx->k3_rounds += pc[11];
        ctx->k3_requests += pc[13];
        if (ctx->phase_prof_print) {
        fprintf(stderr, "[rc phase prof] k_correct, %u reads, cycles/read:", a.n);
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %s=%.0f(%.0f%%)", names[i], (double)pc[i] / a.n, 100.0 * pc[i] / (tot ? tot : 1));
        // wall_clock64 ticks at 100 MHz: when the queue ran dry and when the last wave ended
        fprintf(stderr, "\n[rc phase prof] work list %u of %u reads; queue empty at %.2f ms, last wave done at %.2f ms; gather rounds: %.2f per listed read, worst read %llu\n",
                nwork, a.n, (double)(pc[9] - pc[8]) / 1e5, (double)(pc[10] - pc[8]) / 1e5, (double)pc[11] / (nwork ? nwork : 1), pc[12]);
        fprintf(stderr, "[rc phase prof] bucket reads: %.1f per listed read; work-list sections (first to last): %u / %u / %u / %u reads\n",
                (double)pc[13] / (nwork ? nwork : 1), nsec[0], nsec[1], nsec[2], nsec[3]);
        static const char *sn[8] = {"node entry+pop", "refill (gather round)", "keep-run", "single node", "gap windows", "jump", "terminal", "-"};
        fprintf(stderr, "[rc phase prof] inside the search, cycles/read:");
        for (int i = 0; i < 7; ++i) fprintf(stderr, " %s=%.0f", sn[i], (double)pc[16 + i] / a.n);
        fprintf(stderr, "\n");
        }
    }
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}

// UpdateSummary (main.cpp:73-79) over a batch's return values: reads += n, bases += sum of ret > 0
__global__ __launch_bounds__(256) void k_summary(const int32_t *__restrict__ ret, uint32_t n, unsigned long long *__restrict__ out)
{
    unsigned long long acc = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const int r = ret[i];
        if (r > 0) acc += (unsigned long long)r;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out + 1, acc);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(out, (unsigned long long)n);
}

int rc_launch_summary(rc_ctx *ctx, const int32_t *d_ret, uint32_t n)
{
    if (n == 0) return RC_OK;
    unsigned grid = (n + 255u) / 256u;
    if (grid > 2048u) grid = 2048u;
    hipLaunchKernelGGL(k_summary, dim3(grid), dim3(256), 0, ctx->stream, d_ret, n,
                       (unsigned long long *)((char *)ctx->work.p + RC_WORK_SUMMARY_OFF));
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}
