// rc_correct.hip -- device back end of rc_correct_core.h and the two per-read kernels:
//   k_threshold (K2): GetStrongTrustedThreshold for every read   (ErrorCorrection.cpp:1482-1565)
//   k_correct   (K3): ErrorCorrection + GetKmerInformation        (ErrorCorrection.cpp:682-1480,
//                     :1567-1602), i.e. the body of ErrorCorrection_Thread (:73-136)
// One 64-lane wavefront (= one workgroup) per read; persistent waves pull read indices from an
// atomic counter, which is the reference's mutex-protected batchUsed counter (:87-90) and
// absorbs the heavy tail of the search.  Per-read state is carved out of dynamic LDS.
#include "rc_correct_kernel.h"


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
    const bool tiers = A.tier_hi != RC_TIER_ALL || A.tier_lo >= 0;
    for (;;) {
        const uint32_t o = o_cur;
        const int len = (int)(e_cur - o) - 1;
        // another tier's read: not touched (its own pass computes its threshold), and not on this pass's work list
        const bool mine = !tiers || rc_in_tier(A, rc_unit_max_len(A, r, len));
        if (A.cls && lane == 0) A.cls[r] = mine ? 1 : 0;
        S.len = len;
        S.kcnt = len >= k ? len - k + 1 : 0;
#pragma unroll
        for (int c = 0; c < RC_K2_PF; ++c) {
            const int i = c * 64 + lane;
            if (mine && i < len) {
                S.base[i] = (unsigned char)rc_base_code(pb[c]);
                S.counts[i] = pc[c];
            }
        }
        for (int i = RC_K2_PF * 64 + lane; mine && i < len; i += 64) {
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
        if (mine) {  // (wave-uniform)
            rc_build_masks(w, S);
            int info;
            const int strong = rc_front_end(w, S, A.P, &info);
            if (lane == 0) {
                A.strong[r] = strong;
                A.info[r] = info;
            }
        }
        w.sync();
        if (!more) break;
        r = rn;
    }
}

// GetKmerInformation (ErrorCorrection.h:28, ErrorCorrection.cpp:1567-1602) of every read of a batch AS IT IS (no
// correction): one wave per read over K1's counts -- the code k_correct ends every read with (rc_kmer_info with no fix).
__global__ __launch_bounds__(64) void k_kmer_info(rc_kernel_args A)
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
    for (uint32_t r = blockIdx.x; r < A.n; r += gridDim.x) {
        const uint32_t o = A.off[r];
        rc_load_read(w, A, S, o, (int)(A.off[r + 1] - o) - 1, w.lane, false);
        int l, m, h;
        rc_kmer_info(w, S, A.P, 0, &l, &m, &h);
        if (w.lane == 0) {
            A.l[r] = l;
            A.m[r] = m;
            A.h[r] = h;
        }
        w.sync();
    }
}

int rc_launch_kmer_info(rc_ctx *ctx, const rc_device_batch_args &a);

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
// Round 6, 32-byte buckets (a probe in flight holds 8 VGPRs instead of 16): 38.1 / 36.9-38.0 / 35.9 / 41.6 ms at 5 / 6 / 7 / 8 on that
// table, 61.5 / 56.5 / 53.7 / 55.6 on the 545 MB one -- seven for both (72 VGPRs, 7 x 17.6 KB of LDS a CU).
#ifndef RC_FUSED_SMALL_WAVES
#define RC_FUSED_SMALL_WAVES 7
#endif
#define RC_FUSED_EXT_WAVES 6  // (the instances for tables with remainder-extension bits -- config 4's -- keep six: 372 vs 374-398 ms at seven)
// EC = count registers per lane of the threshold rows (rc_quarter.h): 8 for reads of up to 128 k-mers, 9 for 144 (151-base
// reads at k = 23), 10 for every read of up to 160 bases.  A read the tier of this launch does not hold (rc_kernel_args::
// tier_lo / tier_hi: a longer read, or the mate of one) is left out -- no bases copied, no counts, cls = 0.
// DEDUP = slots of the workgroup's k-mer set (0: none).  The list puts reads that share their minimal m-mer next to each other,
// so the 16 reads of a tile overlap: 73 % of a tile's probes ask for a k-mer another position of the tile asks for as well
// (tools/exp/tile_duplicates.py), and the probe loop is bound by requests -- to the L2, whose channels a tile's 2 048 random
// buckets keep busy even when every line is there (19 of its 28 ms with a table that fits the L2), and behind it to the fabric
// (0.45 misses per probe: a tile's working set is 128 KB, six tiles a CU, 4 MB of L2 an XCD).  So every position first enters
// its canonical k-mer into an open-addressed set in LDS (a 64-bit compare-and-swap: the first to arrive owns the slot), the
// owners' k-mers -- a compact list -- are looked up, one bucket read per distinct k-mer of the tile, and every position takes
// its count from its slot.  A k-mer that finds no slot within four steps is looked up on the spot.
// QUAD: the probes go through rc_table_lookup_quad (rc_device.h: four lanes read a bucket together)
// NT: threads of the workgroup -- 256 (16 reads a tile), or 64 (RC_FUSED_WAVE_TILES=1, dev: a wavefront is a workgroup of its own with
// four reads, so no wave ever waits at a barrier for another)
template <int RC_FUSED_TILE, int WAVES, bool EXT, int EC = 8, int DEDUP = 0, bool QUAD = false, int NT = RC_PROBE_THREADS>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_probe_threshold_list(rc_kernel_args A, size_t nbytes, const uint32_t *__restrict__ list,
                                                                           uint32_t reads_per_block, int32_t *__restrict__ counts, uint32_t tiles_per_xcd,
                                                                           const uint2 *__restrict__ span)
{
    constexpr int NS = DEDUP ? DEDUP : 1;
    __shared__ unsigned long long s_key[NS];
    __shared__ int32_t s_val[NS];
    __shared__ uint16_t s_own[NS];
    __shared__ uint32_t s_nown;
    constexpr uint32_t RC_LATE_CAP = 64;                 // k-mers per wavefront whose walk goes on behind the home bucket (the probe loop)
    __shared__ uint16_t s_late[NT / 64][RC_LATE_CAP];
    __shared__ uint32_t s_nlate[NT / 64];
    __shared__ __attribute__((aligned(16))) uint32_t s_raw[(RC_FUSED_TILE + 64) / 4];
    __shared__ uint32_t s_code[RC_FUSED_TILE / 16 + 4];
    // letter masks of the arena, bit p % 32 of word p / 32 = arena byte p (rc_pack16m): not one of ACGT / an A / a T
    __shared__ __attribute__((aligned(4))) uint16_t s_bad[RC_FUSED_TILE / 16 + 20], s_am[RC_FUSED_TILE / 16 + 20], s_tm[RC_FUSED_TILE / 16 + 20];
    __shared__ uint32_t s_lpos[RC_PLIST_MAX_READS + 1], s_gpos[RC_PLIST_MAX_READS], s_len1[RC_PLIST_MAX_READS], s_rid[RC_PLIST_MAX_READS];
    __shared__ uint32_t s_vpos[RC_PLIST_MAX_READS + 1];
    __shared__ uint32_t s_kuni;
    __shared__ __attribute__((aligned(16))) int32_t s_cnt[RC_FUSED_TILE + 64];
    __shared__ uint8_t s_cls[RC_PLIST_MAX_READS];
    const int t = threadIdx.x, k = A.P.k;  // (an instance compiled for k = 23 was measured and dropped: 46.0 vs 41.9 ms)
    const uint8_t *seq = A.seq;
    // tiles_per_xcd != 0 (RC_FUSED_XCD=1): workgroup b runs on XCD b % 8, so tile (b % 8) * tiles_per_xcd + b / 8 hands each XCD's L2
    // one contiguous stretch of the locality list (neighbouring tiles share most of their k-mers)
    const uint32_t tile = tiles_per_xcd ? (blockIdx.x & 7u) * tiles_per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    if ((uint64_t)tile * reads_per_block >= A.n) return;  // (uniform: the last XCD's stretch may be short)
    const uint32_t i0 = tile * reads_per_block;
    const uint32_t nr = A.n - i0 < reads_per_block ? A.n - i0 : reads_per_block;
    for (int c = t; c < (RC_FUSED_TILE + 64) / 4; c += NT) s_raw[c] = 0;
    if constexpr (DEDUP != 0) {
        for (int c = t; c < NS; c += NT) s_key[c] = ~0ull;  // (no canonical code is all ones: TT..T is the larger strand of AA..A)
        if (t == 0) s_nown = 0;
    }
    if ((uint32_t)t < nr) {
        const uint32_t r = list[i0 + t];
        const uint2 sp = span[i0 + t];  // (k_probe_order left it next to the list: no gather of the offsets behind the list's load)
        s_rid[t] = r;
        s_gpos[t] = sp.x;
        s_len1[t] = sp.y;  // bases + the NUL
    }
    __syncthreads();
    if (A.tier_hi != RC_TIER_ALL) {  // (uniform) the list keeps mates adjacent and nr is even in paired / interleaved batches
        bool other = false;
        if ((uint32_t)t < nr) {
            int ml = (int)s_len1[t] - 1;
            if (A.mode != 0) {
                const int mm = (int)s_len1[t ^ 1] - 1;
                ml = mm > ml ? mm : ml;
            }
            other = !rc_in_tier(A, ml);
        }
        __syncthreads();
        if (other) s_len1[t] = 0;  // 0 = not a read of this launch
        __syncthreads();
    }
    // local start of each read: same alignment modulo 4 as in memory, a NUL in front.  Read j starts at base_j + (gpos_j & 3) with
    // base_0 = 4 and base_{j+1} = base_j + (((gpos_j & 3) + len1_j + 3) & ~3): a prefix sum, done by the first wave (round 6: one
    // thread used to walk the reads while 255 waited at the barrier)
    // ... and s_vpos[j] = the k-mers of the reads in front of read j (the same scan: bytes in the low half of the word, k-mers in the
    // high half -- 64 reads of 160 bases stay below 2^16 either way): the probe loop walks k-mers, not arena positions
    if (t < 64) {
        const bool in = (uint32_t)t < nr;
        const uint32_t sj = in ? (s_gpos[t] & 3u) + s_len1[t] : 0u;
        const uint32_t kc = in && s_len1[t] > (uint32_t)k ? s_len1[t] - (uint32_t)k : 0u;  // (len1 counts the NUL: kcnt = len - k + 1)
        const uint32_t own = ((sj + 3u) & ~3u) | (kc << 16);
        uint32_t inc = own;  // inclusive scan over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o, 64);
            inc += t >= o ? up : 0u;
        }
        const uint32_t base = 4u + ((inc - own) & 0xFFFFu);
        if (in) {
            s_lpos[t] = base + (s_gpos[t] & 3u);
            s_vpos[t] = (inc - own) >> 16;
        }
        if ((uint32_t)t + 1 == nr) {
            s_lpos[nr] = base + sj;  // the end of the last read
            s_vpos[nr] = inc >> 16;
        }
        // every read of the tile with the same number of k-mers (untrimmed reads of one length: the usual batch)?  Then k-mer v is
        // k-mer v % K of read v / K, and the probe loop walks the tile's k-mers instead of its arena positions
        const uint32_t kc0 = __shfl(kc, 0, 64);
        const bool same = __ballot(in && kc != kc0) == 0;
        if (t == 0) s_kuni = same ? kc0 : 0u;
    }
    __syncthreads();
#define RC_FUSED_CUT(n, v)                                                       \
    do {                                                                         \
        if constexpr (RC_FUSED_STOP_ == (n)) {                                   \
            if ((uint32_t)(v) == 0x5bd1e995u && A.strong) A.strong[0] = (int)(v); \
            return;                                                              \
        }                                                                        \
    } while (0)
#ifdef RC_FUSED_STOP  // dev: see rc_quarter.h
    constexpr int RC_FUSED_STOP_ = RC_FUSED_STOP;
#else
    constexpr int RC_FUSED_STOP_ = -1;
#endif
    RC_FUSED_CUT(0, s_lpos[t & 15] ^ s_gpos[t & 15] ^ s_rid[t & 15]);
    for (uint32_t j = (uint32_t)t >> 6; j < nr; j += NT / 64) {  // copy, aligned dwords, outside bytes masked to NUL
        if (!s_len1[j]) continue;
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
    RC_FUSED_CUT(1, s_raw[t]);
    const uint32_t total = s_lpos[nr];
    for (int chunk = t; chunk < RC_FUSED_TILE / 16 + 2; chunk += NT) {
        const uint4 v = *reinterpret_cast<const uint4 *>(s_raw + 4 * chunk);
        uint32_t code, am, tm, bad;
        const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
        rc_pack16m(vw, code, am, tm, bad);
        s_code[chunk] = code;
        s_bad[chunk] = (uint16_t)bad;
        s_am[chunk] = (uint16_t)am;
        s_tm[chunk] = (uint16_t)tm;
    }
    if (t < 2) s_code[RC_FUSED_TILE / 16 + 2 + t] = 0xFFFFFFFFu;
    if (t < 18) {  // (the rows read whole words up to 160 bases behind a read's start)
        s_bad[RC_FUSED_TILE / 16 + 2 + t] = 0xFFFFu;
        s_am[RC_FUSED_TILE / 16 + 2 + t] = 0;
        s_tm[RC_FUSED_TILE / 16 + 2 + t] = 0;
    }
    __syncthreads();
    RC_FUSED_CUT(2, s_code[t & 127] ^ s_bad[t & 127] ^ s_am[t & 127] ^ s_tm[t & 127]);
    const uint32_t *m_bad = reinterpret_cast<const uint32_t *>(s_bad);
    const uint32_t kmask = k >= 32 ? 0xffffffffu : ((1u << k) - 1u);
#ifndef RC_PROBE_UNROLL
#define RC_PROBE_UNROLL 2
#endif
    if constexpr (DEDUP != 0) {
        for (uint32_t a = 4 + (uint32_t)t; a + (uint32_t)k <= total; a += NT) {  // every position enters its k-mer into the set
            const int mw = a >> 5;
            int cnt = 0;
            if (!(__builtin_amdgcn_alignbit(m_bad[mw + 1], m_bad[mw], (uint32_t)a & 31u) & kmask)) {
                const int cw = a >> 4, cs = 2 * (a & 15);
                const uint64_t x = ((((uint64_t)s_code[cw] << 32) | s_code[cw + 1]) << cs) | (((uint64_t)s_code[cw + 2] << cs) >> 32);
                const uint64_t canon = rc_canonical_dev(x >> (64 - 2 * k), k);
                uint32_t h = (((uint32_t)canon * 0x9E3779B1u) ^ ((uint32_t)(canon >> 32) * 0x85EBCA77u)) >> (32 - rcq::ilog2(NS));
                int slot = -1;
                bool own = false;
#pragma unroll 1
                for (int tr = 0; tr < 4; ++tr) {
                    const unsigned long long old = atomicCAS(&s_key[h], ~0ull, (unsigned long long)canon);
                    if (old == ~0ull || old == canon) {
                        own = old == ~0ull;
                        slot = (int)h;
                        break;
                    }
                    h = (h + 1u) & (uint32_t)(NS - 1);
                }
                if (own) s_own[atomicAdd(&s_nown, 1u)] = (uint16_t)slot;
                cnt = slot >= 0 ? -1 - slot : rc_table_lookup<EXT>(A.T, canon);  // (no slot within four steps: a crowded set)
            }
            s_cnt[a] = cnt;
        }
        __syncthreads();
        const uint32_t nown = s_nown;
#pragma unroll RC_PROBE_UNROLL
        for (uint32_t i = (uint32_t)t; i < nown; i += NT) {  // one bucket read per distinct k-mer of the tile
            const uint32_t slot = s_own[i];
            s_val[slot] = rc_table_lookup<EXT>(A.T, (uint64_t)s_key[slot]);
        }
        __syncthreads();
        for (uint32_t a = 4 + (uint32_t)t; a + (uint32_t)k <= total; a += NT) {
            const int v = s_cnt[a];
            if (v < 0) s_cnt[a] = s_val[-1 - v];
        }
    } else if (QUAD) {
        // probe, the four lanes of a quad reading each bucket together (rc_table_lookup_quad): the loop's trip count is the
        // wavefront's, not the lane's -- a lane past the end of the arena still lends its loads
#pragma unroll RC_PROBE_UNROLL
        for (uint32_t a0 = 4 + ((uint32_t)t & ~63u); a0 + (uint32_t)k <= total; a0 += NT) {
            const uint32_t a = a0 + ((uint32_t)t & 63u);
            const bool inside = a + (uint32_t)k <= total;
            const uint32_t ac = inside ? a : 4u;  // (keeps the LDS reads of a lane past the end inside the arrays)
            const int mw = ac >> 5;
            // a window with a letter outside ACGT -- the NUL behind a read included: a position that is no k-mer of any read -- counts 0
            const bool valid = inside && !(__builtin_amdgcn_alignbit(m_bad[mw + 1], m_bad[mw], ac & 31u) & kmask);
            const int cw = ac >> 4, cs = 2 * (ac & 15);
            const uint64_t x = ((((uint64_t)s_code[cw] << 32) | s_code[cw + 1]) << cs) | (((uint64_t)s_code[cw + 2] << cs) >> 32);
            const int cnt = rc_table_lookup_quad<EXT>(A.T, rc_canonical_dev(x >> (64 - 2 * k), k), valid);
            if (inside) s_cnt[a] = cnt;
        }
    } else {
    // probe: counts stay in LDS.  A k-mer that is not in its home bucket while the bucket says "continue" (a full bucket: 0.6 % of the
    // probes at load 0.4) is NOT followed here -- one such lane would send its whole wavefront through a second bucket read, a third
    // of all loop iterations -- but noted in the wavefront's list and finished below, the stragglers of a wavefront side by side.
    const int wv = t >> 6;
    if ((t & 63) == 0) s_nlate[wv] = 0;
    __builtin_amdgcn_wave_barrier();
    // A sixth of a tile's arena positions start no k-mer -- the last k - 1 of every read, its NUL, the padding -- and walking them
    // costs two of the loop's ten iterations with their loads.  Where every read of the tile has the same number K of k-mers
    // (s_kuni: untrimmed reads of one length) the loop walks the K-MERS instead: k-mer v = k-mer v % K of read v / K (one
    // multiply-high by 2^32 / K, exact below 2^14 k-mers a tile).  A ragged tile walks its positions as before.  [Mapping v to its
    // read by binary search over the reads' k-mer prefix sums -- any tile -- was measured first: six dependent LDS reads an
    // iteration, 35.5 -> 36.9 ms; -DRC_PROBE_BY_POSITION keeps every tile on the position walk.]
#ifdef RC_PROBE_BY_POSITION
    const uint32_t kuni = 0;
#else
    const uint32_t kuni = s_kuni;
#endif
    const uint32_t krcp = kuni ? (uint32_t)(0x100000000ull / kuni) + 1u : 0u;   // ceil(2^32 / K) (K > 1; K = 1: the position walk)
    const uint32_t n_it = kuni > 1 ? s_vpos[nr] : (total >= (uint32_t)k + 4u ? total - (uint32_t)k - 4u + 1u : 0u);  // k-mers, or positions 4 .. total - k
#pragma unroll RC_PROBE_UNROLL
    for (uint32_t v = (uint32_t)t; v < n_it; v += NT) {
        uint32_t a = 4u + v;
        if (kuni > 1) {  // (uniform)
            const uint32_t jr = __umulhi(v, krcp);
            a = s_lpos[jr] + (v - jr * kuni);
        }
        // a window with a letter outside ACGT counts 0 (walking positions: the NUL behind a read included -- a position that is no k-mer)
        const int mw = a >> 5;
        int cnt = 0;
        if (!(__builtin_amdgcn_alignbit(m_bad[mw + 1], m_bad[mw], (uint32_t)a & 31u) & kmask)) {
            const int cw = a >> 4, cs = 2 * (a & 15);  // the 64 bits from base a on: two words shifted up, the third fills in (no branch on cs)
            const uint64_t x = ((((uint64_t)s_code[cw] << 32) | s_code[cw + 1]) << cs) | (((uint64_t)s_code[cw + 2] << cs) >> 32);
            const uint64_t canon = rc_canonical_dev(x >> (64 - 2 * k), k);
            bool more = false;
            cnt = rc_table_lookup_o<EXT>(A.T, canon, canon, nullptr, 0, &more);
            if (more) {
                const uint32_t slot = atomicAdd(&s_nlate[wv], 1u);
                if (slot < RC_LATE_CAP)
                    s_late[wv][slot] = (uint16_t)a;
                else
                    cnt = rc_table_lookup_o<EXT>(A.T, canon, canon, nullptr, 1);  // (a crowded list: on the spot)
            }
        }
        s_cnt[a] = cnt;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        const uint32_t nl = s_nlate[wv] < RC_LATE_CAP ? s_nlate[wv] : RC_LATE_CAP;
        for (uint32_t i = (uint32_t)t & 63u; i < nl; i += 64u) {
            const uint32_t a = s_late[wv][i];
            const int cw = a >> 4, cs = 2 * (a & 15);
            const uint64_t x = ((((uint64_t)s_code[cw] << 32) | s_code[cw + 1]) << cs) | (((uint64_t)s_code[cw + 2] << cs) >> 32);
            const uint64_t canon = rc_canonical_dev(x >> (64 - 2 * k), k);
            s_cnt[a] = rc_table_lookup_o<EXT>(A.T, canon, canon, nullptr, 1);
        }
    }
    }
    __syncthreads();
    RC_FUSED_CUT(3, s_cnt[t] ^ s_cnt[t + 256] ^ s_cnt[t + 512] ^ s_cnt[t + 768] ^ s_cnt[t + 1024] ^ s_cnt[t + 1280] ^ s_cnt[t + 1536] ^ s_cnt[t + 1792] ^ s_cnt[t + 2048] ^ s_cnt[t + 2304] ^ s_cnt[t + 2560]);
    // thresholds + classes: 16 reads per pass (one per 16-lane row; the list keeps mates adjacent)
    const uint8_t *raw8 = reinterpret_cast<const uint8_t *>(s_raw);
    for (uint32_t j0 = 0; j0 < nr; j0 += NT / 16) {
        const uint32_t j = j0 + ((uint32_t)t >> 4);
        const bool live = j < nr && s_len1[j] != 0;
        const uint32_t lp = live ? s_lpos[j] : 0;
        const int len = live ? (int)s_len1[j] - 1 : 0;
        const rcq_lds_masks msrc = {reinterpret_cast<const uint32_t *>(s_am), reinterpret_cast<const uint32_t *>(s_tm), m_bad, lp};
        const int cls = rcq_threshold_row<EC, 10>(
            A, live ? s_rid[j] : 0, live, len, [&](int p) { return (uint32_t)raw8[lp + p]; }, [&](int g) { return s_cnt[lp + g]; }, msrc);
        if (live && (t & 15) == 0) s_cls[j] = (uint8_t)cls;
        if (j < nr && !live && (t & 15) == 0) {  // another tier's read
            s_cls[j] = 0;
            if (A.cls) A.cls[s_rid[j]] = 0;
            if (A.cand) A.cand[s_rid[j]] = 0;
        }
    }
    __syncthreads();
    RC_FUSED_CUT(19, s_cls[t & 15]);
    // the counts k_correct will read: those of the reads that still need it, four per lane (a read
    // starts at the same offset modulo 4 here and in the arena; the up to three words in front of its
    // first count and behind its last one belong to NULs and to the last k-1 positions of a read,
    // which hold no count -- k >= 4, rc_launch_probe_threshold_list)
    for (uint32_t j = (uint32_t)t >> 6; j < nr; j += NT / 64) {
        if (!s_len1[j] || (A.cls && !s_cls[j])) continue;
        const int kcnt = (int)s_len1[j] - 1 - k + 1;
        const uint32_t lp = s_lpos[j], g0 = s_gpos[j], head = lp & 3u;
        const int4 *src = reinterpret_cast<const int4 *>(s_cnt + (lp - head));
        int4 *dst = reinterpret_cast<int4 *>(counts + (size_t)(g0 - head));
        for (int q = t & 63; 4 * q < kcnt + (int)head; q += 64) dst[q] = src[q];
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
    A.cand = nullptr;
    A.runs = nullptr;
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
    A.rounds_out = ctx->phase_prof ? ctx->rounds_out : nullptr;
    A.trace = nullptr;
    A.trace_cap = 0;
    A.fused_front_end = a.mode == 0 && !ctx->thr_ready;
    A.tier_lo = a.tier_lo;
    A.tier_hi = a.tier_hi;
    A.pair_override = a.pair_override;
    return RC_OK;
}

// count registers per lane the threshold rows / k_single need for reads of up to max_len bases (rc_quarter.h): 8 / 9 / 10,
// 0 = the reads do not fit the 160-base instances
static int rc_short_ec(const rc_ctx *ctx, int max_len, int k)
{
    if (max_len > 160) return 0;
    const int kcnt = max_len - k + 1;
    const int ec = kcnt <= 128 ? 8 : (kcnt <= 144 ? 9 : 10);
    return ec < ctx->env_force_ec ? ctx->env_force_ec : ec;  // RC_FORCE_EC=9|10 (dev / tests): the wider instances on shorter reads
}

// classify: let the quarter-wave kernel finish the reads that need no correction (ret, l, m, h
// written there) and flag the others in ctx->cls; ctx->cls_ready tells the caller whether it did.
// a.tier_lo / a.tier_hi: the pass of one length tier (a.max_len = the longest read of that tier): every kernel
// then writes cls -- 0 for the reads of the other tiers.
int rc_launch_threshold(rc_ctx *ctx, const rc_device_batch_args &a, bool classify)
{
    ctx->cls_ready = false;
    ctx->cand_ready = false;
    if (a.n == 0) return RC_OK;
    rc_kernel_args A;
    int rc = fill_args(ctx, a, A);
    if (rc) return rc;
    const rc_lds_layout L = rc_layout(A.cap);
    const bool tiered = a.tier_hi != RC_TIER_ALL || a.tier_lo >= 0;
    // four reads per wave when every read of the batch fits the quarter-wave layout (rc_quarter.h)
    const bool quarter = a.max_len <= rcq::MAX_LEN && a.max_len - A.P.k + 1 <= rcq::MAX_KCNT && !ctx->env_k2_wave_per_read;
    const int ec = rc_short_ec(ctx, a.max_len, A.P.k);
    if ((quarter || tiered) && classify && a.ret && ctx->trace_cap == 0 && !ctx->env_no_classify) {
        if ((rc = rc_dbuf_reserve(ctx, &ctx->cls, (size_t)a.n + 256))) return rc;
        A.cls = (uint8_t *)ctx->cls.p;
        ctx->cls_ready = true;
        if (quarter && !ctx->env_no_single && ec) {  // candidates of k_single (rc_single.h)
            if ((rc = rc_dbuf_reserve(ctx, &ctx->cand, (size_t)a.n + 256))) return rc;
            if ((rc = rc_dbuf_reserve(ctx, &ctx->runs, (size_t)a.n * 8 + 256))) return rc;
            A.cand = (uint8_t *)ctx->cand.p;
            A.runs = (uint2 *)ctx->runs.p;
            ctx->cand_ready = true;
        }
    }
    dim3 qgrid((a.n + 15) / 16), qblock(256);
    if (quarter && a.tier_list && A.cls) {
        // list-driven pass of a tier: only its reads are touched, so the classes of the other tiers' reads (the passes
        // before this one left them there) are cleared first -- the compaction below reads the whole array
        RC_CHECK_HIP(ctx, hipMemsetAsync(A.cls, 0, a.n, ctx->stream));
        A.worklist = a.tier_list;
        A.n_work = a.tier_n;
        if (qgrid.x > (unsigned)ctx->n_cu * 64u) qgrid.x = (unsigned)ctx->n_cu * 64u;
    }
    rc_timer_begin(ctx);
    if (quarter && ec == 8) {
        hipLaunchKernelGGL((k_threshold_q<8, 10>), qgrid, qblock, 0, ctx->stream, A);
    } else if (quarter && ec == 9) {
        hipLaunchKernelGGL((k_threshold_q<9, 10>), qgrid, qblock, 0, ctx->stream, A);
    } else if (quarter && ec == 10) {
        hipLaunchKernelGGL((k_threshold_q<10, 10>), qgrid, qblock, 0, ctx->stream, A);
    } else if (quarter) {
        hipLaunchKernelGGL((k_threshold_q<16, 20>), qgrid, qblock, 0, ctx->stream, A);
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
// (a.max_len: the longest read of the tier the launch is for, at most 160 bases)
int rc_launch_probe_threshold_list(rc_ctx *ctx, const rc_device_batch_args &a, size_t nbytes, bool *done)
{
    *done = false;
    ctx->cls_ready = false;
    ctx->cand_ready = false;
    const int ec = rc_short_ec(ctx, a.max_len, ctx->k);
    if (a.n == 0 || !ec || ctx->k < 4 || ctx->env_k2_wave_per_read || ctx->env_no_fuse) return RC_OK;
    rc_kernel_args A;
    int rc = fill_args(ctx, a, A);
    if (rc) return rc;
    if (a.ret && ctx->trace_cap == 0 && !ctx->env_no_classify) {
        if ((rc = rc_dbuf_reserve(ctx, &ctx->cls, (size_t)a.n + 256))) return rc;
        A.cls = (uint8_t *)ctx->cls.p;
        ctx->cls_ready = true;
        if (!ctx->env_no_single) {  // candidates of k_single (rc_single.h)
            if ((rc = rc_dbuf_reserve(ctx, &ctx->cand, (size_t)a.n + 256))) return rc;
            if ((rc = rc_dbuf_reserve(ctx, &ctx->runs, (size_t)a.n * 8 + 256))) return rc;
            A.cand = (uint8_t *)ctx->cand.p;
            A.runs = (uint2 *)ctx->runs.p;
            ctx->cand_ready = true;
        }
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
    const uint32_t n_tiles = (a.n + rpb - 1) / rpb;
    const uint32_t txcd = ctx->env_fused_xcd ? (n_tiles + 7) / 8 : 0;
    const dim3 grid(txcd ? txcd * 8 : n_tiles), block(RC_PROBE_THREADS);
    const uint32_t *list = (const uint32_t *)ctx->loc_list.p;
    int32_t *counts = (int32_t *)ctx->counts.p;
    // Three ways of cutting the probes' requests were built and measured in round 6 (profiles/r6_fused_where_the_time_goes.txt);
    // none pays on the bench presets, all are parity-green and stay behind their switches (tests: knob matrix):
    //  * RC_FUSED_DEDUP=1, the tile's k-mer set: L2 requests -47 %, fabric requests -18 % -- and config 2 41.4 -> 46.9 ms, config 3
    //    62.5 -> 71.0 (two barriers, an LDS compare-and-swap per position, five workgroups a CU instead of six); config 4 411.7 ->
    //    366.8 ms in one call, 413.0 -> 411.7 in another;
    //  * RC_FUSED_XCD=1, tiles in XCD-contiguous order: fabric requests -37 % (1.42 -> 0.91 G), the kernel's time unchanged;
    //  * RC_PROBE_QUAD=1, a quad of lanes per bucket (rc_table_lookup_quad): +23 % on L2 hits and 2.1 x beyond the TLB's reach in
    //    tools/microbench_bucket.hip, but 57 more vector instructions per probe here: config 2 41.2 -> 47.0 ms, config 3 62.5 -> 77.4.
    const bool dedup = ctx->env_dedup > 0;
    const bool quad = ctx->env_quad > 0;
    if (ctx->env_wave_tiles && ec == 8 && !ctx->ext) {  // dev: one wavefront, four reads, no inter-wave barrier (k <= 23-ish tables without extension bits)
        const uint32_t rpw = 4;
        const uint32_t nt = (a.n + rpw - 1) / rpw, tx = ctx->env_fused_xcd ? (nt + 7) / 8 : 0;
        rc_timer_begin(ctx);
        hipLaunchKernelGGL((k_probe_threshold_list<704, 6, false, 8, 0, false, 64>), dim3(tx ? tx * 8 : nt), dim3(64), 0, ctx->stream, A, nbytes,
                           (const uint32_t *)ctx->loc_list.p, rpw, (int32_t *)ctx->counts.p, tx, (const uint2 *)ctx->loc_span.p);
        rc_timer_end(ctx, RC_T_PROBE);
        RC_CHECK_HIP(ctx, hipGetLastError());
        *done = true;
        return RC_OK;
    }
#define RC_FUSED_LAUNCH(TILE, WAVES, EXT, EC)                                                                                                        \
    do {                                                                                                                                             \
        if (dedup)                                                                                                                                   \
            hipLaunchKernelGGL((k_probe_threshold_list<TILE, 5, EXT, EC, 1024>), grid, block, 0, ctx->stream, A, nbytes, list, rpb, counts, txcd, (const uint2 *)ctx->loc_span.p);         \
        else if (quad)                                                                                                                               \
            hipLaunchKernelGGL((k_probe_threshold_list<TILE, WAVES, EXT, EC, 0, true>), grid, block, 0, ctx->stream, A, nbytes, list, rpb, counts, txcd, (const uint2 *)ctx->loc_span.p);  \
        else                                                                                                                                         \
            hipLaunchKernelGGL((k_probe_threshold_list<TILE, WAVES, EXT, EC, 0>), grid, block, 0, ctx->stream, A, nbytes, list, rpb, counts, txcd, (const uint2 *)ctx->loc_span.p);        \
    } while (0)
    if (large) {  // (reads of up to 119 bases: at most 116 k-mers; 32 reads a tile: their k-mers would want a set of 2 048 slots -- not built)
        if (ctx->ext)
            hipLaunchKernelGGL((k_probe_threshold_list<4096, 6, true, 8, 0>), grid, block, 0, ctx->stream, A, nbytes, list, rpb, counts, txcd, (const uint2 *)ctx->loc_span.p);
        else
            hipLaunchKernelGGL((k_probe_threshold_list<4096, 6, false, 8, 0>), grid, block, 0, ctx->stream, A, nbytes, list, rpb, counts, txcd, (const uint2 *)ctx->loc_span.p);
    } else if (ec == 8) {
        if (ctx->ext)
            RC_FUSED_LAUNCH(2816, RC_FUSED_EXT_WAVES, true, 8);
        else
            RC_FUSED_LAUNCH(2816, RC_FUSED_SMALL_WAVES, false, 8);
    } else if (ec == 9) {
        if (ctx->ext)
            RC_FUSED_LAUNCH(2816, RC_FUSED_EXT_WAVES, true, 9);
        else
            RC_FUSED_LAUNCH(2816, RC_FUSED_SMALL_WAVES, false, 9);
    } else {
        if (ctx->ext)
            RC_FUSED_LAUNCH(2816, RC_FUSED_EXT_WAVES, true, 10);
        else
            RC_FUSED_LAUNCH(2816, RC_FUSED_SMALL_WAVES, false, 10);
    }
#undef RC_FUSED_LAUNCH
    rc_timer_end(ctx, RC_T_PROBE);
    RC_CHECK_HIP(ctx, hipGetLastError());
    *done = true;
    return RC_OK;
}

#include "rc_single.h"

// K2s: finish the reads whose correction is one isolated substitution per untrusted stretch (rc_single.h); they leave
// the work list before it is compacted.  Needs the classification (cls) and K1's counts of the listed reads.
int rc_launch_single(rc_ctx *ctx, const rc_device_batch_args &a, bool *ran)
{
    *ran = false;
    if (a.n == 0 || !ctx->cls_ready || !ctx->cand_ready || ctx->env_no_single) return RC_OK;
    const int ec = rc_short_ec(ctx, a.max_len, ctx->k);
    if (!ec || ctx->k < 4 || ctx->P.max_fix_per_k < 2) return RC_OK;
    rc_kernel_args A;
    int rc = fill_args(ctx, a, A);
    if (rc) return rc;
    // its work list: the reads the threshold kernel flagged as candidates, grouped by their number of untrusted stretches
    // (the four reads of a wavefront walk their stretches in lock step: the wave takes as long as its longest read)
    const size_t stride = ((size_t)a.n + 63) & ~(size_t)63;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->single_list, stride * 3 * 4 + 256))) return rc;
    uint32_t *d_n = (uint32_t *)((char *)ctx->work.p + RC_WORK_NSINGLE_OFF);
    RC_CHECK_HIP(ctx, hipMemsetAsync(d_n, 0, 16, ctx->stream));
    if ((rc = rc_launch_compact_flag(ctx, (const uint8_t *)ctx->cand.p, a.n, (uint32_t *)ctx->single_list.p, stride, d_n))) return rc;
    A.cls = (uint8_t *)ctx->cls.p;
    A.runs = (uint2 *)ctx->runs.p;
    A.worklist = (const uint32_t *)ctx->single_list.p;
    A.work_stride = stride;
    A.n_work = d_n;
    rc_timer_begin(ctx);
    unsigned g = (unsigned)ctx->n_cu * (unsigned)RC_K2S_GRID;  // (a few workgroups per CU slot; they walk the list, whose length stays on the device)
    if (g > (a.n + 15) / 16) g = (a.n + 15) / 16;
    const dim3 grid(g), block(256);
    if (ec == 8) {
        if (ctx->ext)
            hipLaunchKernelGGL((k_single<true, 8>), grid, block, 0, ctx->stream, A);
        else
            hipLaunchKernelGGL((k_single<false, 8>), grid, block, 0, ctx->stream, A);
    } else if (ec == 9) {
        if (ctx->ext)
            hipLaunchKernelGGL((k_single<true, 9>), grid, block, 0, ctx->stream, A);
        else
            hipLaunchKernelGGL((k_single<false, 9>), grid, block, 0, ctx->stream, A);
    } else {
        if (ctx->ext)
            hipLaunchKernelGGL((k_single<true, 10>), grid, block, 0, ctx->stream, A);
        else
            hipLaunchKernelGGL((k_single<false, 10>), grid, block, 0, ctx->stream, A);
    }
    rc_timer_end(ctx, RC_T_SINGLE);
    RC_CHECK_HIP(ctx, hipGetLastError());
    *ran = true;
    return RC_OK;
}

int rc_launch_kmer_info(rc_ctx *ctx, const rc_device_batch_args &a)
{
    if (a.n == 0) return RC_OK;
    rc_kernel_args A;
    int rc = fill_args(ctx, a, A);
    if (rc) return rc;
    const rc_lds_layout L = rc_layout(A.cap);
    unsigned grid = (unsigned)ctx->n_cu * 16u;
    if (grid > a.n) grid = a.n;
    hipLaunchKernelGGL(k_kmer_info, dim3(grid), dim3(64), L.total, ctx->stream, A);
    RC_CHECK_HIP(ctx, hipGetLastError());
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
    RC_CHECK_HIP(ctx, hipMemsetAsync((char *)ctx->work.p + RC_WORK_PHASE_OFF, 0, 34 * 8, ctx->stream));
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
    rc_k3_launcher special = nullptr;  // an instance compiled for this k and this table layout?
    if (ctx->trace_cap == 0 && !ctx->phase_prof && A.cap_class == 192 && !ctx->env_k3_generic) special = rc_k3_special(ctx->k, ctx->layout, ctx->ext);
    if (special)
        special(dim3(grid), ctx->stream, A);
    else if (ctx->trace_cap > 0)
        hipLaunchKernelGGL((k_correct<1024, false, true>), dim3(grid), dim3(64), 0, ctx->stream, A);
    else if (ctx->phase_prof && A.cap_class == 192)
        hipLaunchKernelGGL((k_correct<192, true, false>), dim3(grid), dim3(64), 0, ctx->stream, A);
    else if (ctx->phase_prof && A.cap_class == 320)
        hipLaunchKernelGGL((k_correct<320, true, false>), dim3(grid), dim3(64), 0, ctx->stream, A);
    else if (ctx->phase_prof)
        hipLaunchKernelGGL((k_correct<1024, true, false>), dim3(grid), dim3(64), 0, ctx->stream, A);
    else if (A.cap_class == 192)
        hipLaunchKernelGGL((k_correct<192, false, false>), dim3(grid), dim3(64), 0, ctx->stream, A);
    else if (A.cap_class == 320)
        hipLaunchKernelGGL((k_correct<320, false, false>), dim3(grid), dim3(64), 0, ctx->stream, A);
    else
        hipLaunchKernelGGL((k_correct<1024, false, false>), dim3(grid), dim3(64), 0, ctx->stream, A);
    rc_timer_end(ctx, RC_T_CORRECT);
    if (ctx->phase_prof) {
        unsigned long long pc[34];
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
        fprintf(stderr, "[rc phase prof] k-mers looked up by the searches: %.1f per listed read, %.1f %% of them not in the table\n",
                (double)pc[14] / (nwork ? nwork : 1), 100.0 * (double)pc[15] / (double)(pc[14] ? pc[14] : 1));
        fprintf(stderr, "[rc phase prof] bucket reads: %.1f per listed read; work-list sections (first to last): %u / %u / %u / %u reads\n",
                (double)pc[13] / (nwork ? nwork : 1), nsec[0], nsec[1], nsec[2], nsec[3]);
        static const char *sn[8] = {"node entry+pop", "refill (gather round)", "keep-run", "single node", "gap windows", "jump", "terminal", "-"};
        fprintf(stderr, "[rc phase prof] inside the search, cycles/read:");
        for (int i = 0; i < 7; ++i) fprintf(stderr, " %s=%.0f", sn[i], (double)pc[16 + i] / a.n);
        fprintf(stderr, "\n");
        const double R = (double)(pc[11] ? pc[11] : 1), KR = (double)(pc[24] ? pc[24] : 1);
        fprintf(stderr, "[rc phase prof] a gather round: %.1f probes, %.1f %% with alternative chains (%.2f walked per round, %.0f %% of them to the end); a keep-run (%.2f per round) keeps %.2f of the %.2f cached nodes it is offered; gap-window rounds %.3f per round (%.1f probes)\n",
                (double)pc[32] / R, 100.0 * (double)pc[33] / R, (double)pc[30] / R, 100.0 * (double)pc[31] / (double)(pc[30] ? pc[30] : 1), KR / R, (double)pc[25] / KR,
                (double)pc[26] / KR, (double)pc[28] / R, (double)pc[29] / (double)(pc[28] ? pc[28] : 1));
        }
    }
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}

void rc_k3_launch_23_1(dim3, hipStream_t, const rc_kernel_args &);
void rc_k3_launch_25_1(dim3, hipStream_t, const rc_kernel_args &);
void rc_k3_launch_31_2(dim3, hipStream_t, const rc_kernel_args &);
rc_k3_launcher rc_k3_special(int k, int layout, int ext)
{
    if (layout != 1) return nullptr;
    if (k == 23 && ext == 0) return rc_k3_launch_23_1;  // run_rcorrector.pl's default k
    if (k == 25 && ext == 0) return rc_k3_launch_25_1;  // Trinity's k
    if (k == 31 && ext > 0) return rc_k3_launch_31_2;
    return nullptr;
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
    __shared__ unsigned long long s_acc[4];
    if ((threadIdx.x & 63) == 0) s_acc[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {  // one atomic per workgroup (same-address atomics cost ~12 ns each: an atomic per wave was 100 us)
        const unsigned long long tot = s_acc[0] + s_acc[1] + s_acc[2] + s_acc[3];
        if (tot) atomicAdd(out + 1, tot);
        if (blockIdx.x == 0) atomicAdd(out, (unsigned long long)n);
    }
}

int rc_launch_summary(rc_ctx *ctx, const int32_t *d_ret, uint32_t n)
{
    if (n == 0) return RC_OK;
    unsigned grid = (n + 255u) / 256u;
    if (grid > 1024u) grid = 1024u;
    hipLaunchKernelGGL(k_summary, dim3(grid), dim3(256), 0, ctx->stream, d_ret, n,
                       (unsigned long long *)((char *)ctx->work.p + RC_WORK_SUMMARY_OFF));
    RC_CHECK_HIP(ctx, hipGetLastError());
    return RC_OK;
}
