// rc_internal.h -- context object and cross-TU declarations of librcorrector_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "rc_correct_core.h"

#define RC_CHECK_HIP(ctx, expr)                                                                  \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            rc_set_error((ctx), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                         __LINE__);                                                              \
            return RC_ERR_HIP;                                                                   \
        }                                                                                        \
    } while (0)

enum { RC_OK = 0, RC_ERR_ARG = -1, RC_ERR_HIP = -2, RC_ERR_IO = -3, RC_ERR_STATE = -4, RC_ERR_NOMEM = -5, RC_ERR_NOSPACE = -6 };

// scoped device allocation: freed on every exit path of the function that owns it
struct rc_dev_tmp {
    void *p = nullptr;
    rc_dev_tmp() = default;
    rc_dev_tmp(const rc_dev_tmp &) = delete;
    rc_dev_tmp &operator=(const rc_dev_tmp &) = delete;
    ~rc_dev_tmp() { reset(); }
    void reset()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    hipError_t alloc(size_t bytes)
    {
        reset();
        return hipMalloc(&p, bytes ? bytes : 1);
    }
    template <class T>
    T *as() const { return static_cast<T *>(p); }
};

// grow-only device buffer
struct rc_dbuf {
    void *p = nullptr;
    size_t bytes = 0;
};

#define RC_MAX_SLOTS 4

struct rc_kernel_timer {
    double ms = 0;       // accumulated
    uint64_t launches = 0;
};

enum { RC_T_PROBE = 0, RC_T_THRESH = 1, RC_T_CORRECT = 2, RC_T_SINGLE = 3, RC_T_COUNT };

struct rc_ctx {
    int device = 0;
    int k = 23;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool profile = false;
    bool phase_prof = false;  // the instrumented build of k_correct runs (RC_PHASE_PROF=1 or rc_profile_enable(ctx, 2))
    bool phase_prof_print = false;  // ... and prints its per-phase cycle accounting (RC_PHASE_PROF=1, dev aid)
    uint64_t k3_listed = 0, k3_rounds = 0, k3_requests = 0;  // accumulated by the instrumented build
    int32_t *rounds_out = nullptr;  // rc_profile_read_rounds: where the instrumented build leaves every read's gather rounds
    rc_kernel_timer timers[RC_T_COUNT];

    rc_run_params P;
    bool params_set = false;
    bool qual_bits = false;  // quality arenas are bit arrays (rc_set_quality_bits)
    // Slot lanes: slot s > 0 of the asynchronous entry points runs in a context of its own (lane[s], created at its first use:
    // own streams, events and scratch; the table, the run parameters and the kept arenas are this context's, lent), so that
    // the kernels of batches in different slots overlap on the GPU -- a batch's last waves on its slowest reads under the
    // next batch's probe kernel.  RC_SLOT_LANES=0: every slot in this context, one compute stream (round 4's behaviour).
    rc_ctx *lane[4] = {nullptr, nullptr, nullptr, nullptr};
    rc_ctx *slot_home[4] = {nullptr, nullptr, nullptr, nullptr};  // where the batch a slot holds was submitted (lanes can be switched)
    bool is_lane = false;
    bool env_slot_lanes = true;

    // k-mer table in HBM
    uint32_t *d_buckets = nullptr;
    bool buckets_borrowed = false;  // rc_table_share: another context owns d_buckets
    uint32_t nb_home = 0;
    double table_load = 0.50;  // target slot load factor of the next build (WIDE layout)
    bool table_load_set = false;  // RC_TABLE_LOAD given: no size-dependent choice
    double table_load_packed = 0.50;  // ... (PACKED layout)
    int layout = 0;       // slot layout of the table (rc_common.h): 0 WIDE, 1 PACKED
    int ext = 0;          // PACKED: remainder bits kept above the count (rc_table_view::ext)
    int layout_pref = 1;  // 0: always WIDE (RC_TABLE_LAYOUT=wide)
    uint32_t nb_alloc = 0;
    size_t n_entries = 0;   // accepted entries (duplicates included)
    size_t table_bytes = 0;
    // the canonical codes of the entries a table was counted from (rc_count_finish), kept until the ERROR_RATE pass has used them
    // or the table goes: that pass wants every entry's code, and reading them back out of the buckets (k_export: a decode and
    // a probe walk per slot) costs more than the 8 bytes per entry it frees
    void *counted_codes = nullptr;
    size_t counted_n = 0;
    uint32_t filter_words = 0;  // absence filter behind the bucket array (rc_common.h: rc_table_view::filter), 0 = none
    int filter_kind = 0;        // rc_table_view::filter_kind of the filter that is there
    int filter_all = 0;         // rc_table_view::filter_all

    // -verbose support: iterations recorded per read by k_correct (0 = off) and the record buffer
    int trace_cap = 0;
    rc_dbuf trace;

    // k-mer counter (rc_table_count_begin/add/finish): the arenas handed over so far, kept in HBM until finish
    // (the arenas are views into a few large allocations, cnt_chunks: a hipMalloc / hipFree per arena -- dozens per run, each
    // a round trip through the kernel driver -- cost more than counting them on a busy host)
    std::vector<rc_dbuf> cnt_arenas;
    std::vector<rc_dbuf> cnt_chunks;
    size_t cnt_chunk_used = 0;  // bytes of the last chunk handed out
    size_t cnt_total = 0;  // bytes
    bool cnt_active = false;
    // rc_table_count_keep(on): finish() leaves the arenas here instead of releasing them -- the reads of a data set
    // that was counted on this GPU are corrected where they lie (rc_submit_resident)
    bool cnt_keep = false;
    std::vector<rc_dbuf> kept_arenas, kept_chunks;

    // batch scratch
    rc_dbuf counts;   // int32 per arena byte
    rc_dbuf strong;   // int32 per read
    rc_dbuf info;     // int32 per read
    bool thr_ready = false;  // strong / info hold this batch's thresholds (the threshold kernel ran)
    rc_dbuf cls;      // uint8 per read: 1 = still needs k_correct (written by the threshold kernel)
    rc_dbuf cand;     // uint8 per read: 1 = candidate of k_single (written by the threshold kernel with cls)
    rc_dbuf single_list;  // the candidates' read indices, ascending (compaction of cand)
    rc_dbuf runs;     // uint2 per read, candidates only: their untrusted stretches (rc_kernel_args::runs)
    rc_dbuf bs_dev;   // RC_BOUND_STEPS uint32: the inverse of GetBound at the run's ERROR_RATE (rc_run_params::bs_ext)
    bool cand_ready = false;
    rc_dbuf worklist; // RC_WORK_CLASSES sections of work_stride uint32 each: the reads with cls == 4, 3, 2, 1, ascending within a section
    rc_dbuf sel_tmp;  // rocPRIM scratch of the compaction
    rc_dbuf loc_a, loc_list, loc_span;  // locality order of a batch (rc_launch_locality_order): the reads, and where each lies in the arena
    rc_dbuf tier_flag, tier_list;  // mixed-length batches: the reads of the middle / long tier in locality order (rc_launch_tier_lists)
    size_t tier_stride = 0;        // uint32 entries between the two sections of tier_list
    bool env_fused_xcd = false;  // RC_FUSED_XCD=1 (dev): the fused probe kernel's tiles in XCD-contiguous order
    bool env_wave_tiles = false;  // RC_FUSED_WAVE_TILES=1 (dev): the fused probe kernel with one wavefront (four reads) per workgroup
    bool env_k3_local = false;  // RC_K3_LOCAL=1 (dev): k_correct's work list in the batch's locality order (non-tiered batches)
    int env_quad = -1;   // RC_PROBE_QUAD=0 / 1 (dev / tests): bucket reads by the lane / by the quad (rc_table_lookup_quad); -1: the default
    int env_dedup = -1;  // RC_FUSED_DEDUP=0 / 1 (dev / tests): the fused probe kernel without / with its per-tile k-mer set; -1: the table decides (rc_launch_probe_threshold_list)
    bool env_no_fuse = false;  // RC_NO_FUSE=1 (dev): separate probe and threshold kernels in locality order too
    int env_force_ec = 0;      // RC_FORCE_EC=9|10 (dev / tests): at least this many count registers per lane in the 160-base instances
    bool env_no_tier = false;  // RC_NO_TIER=1 (dev / tests): no length tiers, the longest read of a batch decides every kernel
    bool env_k3_generic = false;  // RC_K3_GENERIC=1 (dev / tests): the any-k instance of k_correct even where a compiled-for-k one exists
    bool env_no_single = false;  // RC_NO_SINGLE=1 (dev / tests): no isolated-substitution kernel, every listed read goes to k_correct
    bool env_no_alt = false;  // RC_NO_ALT=1 (dev / tests): rc_run_params::flags |= RC_PF_NO_ALT
    int locality_mode = 0;  // 0: large batches over large tables, 1: always (RC_LOCALITY=force), -1: never (RC_LOCALITY=off)
    bool cls_ready = false;  // cls / worklist describe this batch
    size_t work_stride = 0;  // uint32 entries between the sections of worklist
    // getenv() results, read once at rc_create
    bool env_k2_wave_per_read = false, env_no_classify = false, env_timing = false;
    int env_k3_grid_waves = 0;  // dev: persistent k_correct waves per SIMD actually launched (0 = as compiled)
    rc_dbuf stack;    // search stack frames
    rc_dbuf work;     // work counters
    rc_dbuf h_seq, h_qual, h_off, h_res;  // device staging for the host-buffer entry point (traced path)

    // asynchronous host-buffer path (rc_submit / rc_wait): copy streams and per-slot staging
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    struct rc_slot *slots = nullptr;  // RC_MAX_SLOTS of them, created on first use

    char err[512];
};

void rc_set_error(rc_ctx *ctx, const char *fmt, ...);
int rc_dbuf_reserve(rc_ctx *ctx, rc_dbuf *b, size_t bytes);
rc_table_view rc_view(const rc_ctx *ctx);
// frees the table this context owns (the allocation starts RC_TABLE_PREFIX_BYTES before d_buckets)
void rc_table_release(rc_ctx *ctx);
void rc_timer_begin(rc_ctx *ctx);
void rc_timer_end(rc_ctx *ctx, int which);

// rc_table.hip
int rc_build_table_from_device_pairs(rc_ctx *ctx, const uint64_t *d_canon, const int32_t *d_counts,
                                     size_t n);
int rc_launch_canonicalize(rc_ctx *ctx, uint64_t *d_codes, size_t n);
int rc_launch_lookup(rc_ctx *ctx, const uint64_t *d_codes, size_t n, int32_t *d_out);
int rc_launch_probe(rc_ctx *ctx, const uint8_t *d_seq, size_t nbytes, int32_t *d_counts);
int rc_count_begin(rc_ctx *ctx);
void rc_kept_release(rc_ctx *ctx);
int rc_count_add(rc_ctx *ctx, const uint8_t *seq, size_t nbytes, bool from_device);
int rc_count_finish(rc_ctx *ctx, int min_count, int64_t *n_kmers);
int rc_count_park(rc_ctx *ctx);
int rc_count_finish_sharded(rc_ctx **cs, int n, int min_count, int64_t *n_kmers);
int rc_count_reads(rc_ctx *ctx, const uint8_t *d_seq, size_t nbytes, int min_count, int64_t *n_kmers);
int rc_launch_selftest_bound(rc_ctx *ctx, const int32_t *d_c, size_t n, double e, int32_t *d_oi, double *d_od);
int rc_launch_export(rc_ctx *ctx, uint64_t *d_codes, int32_t *d_counts, unsigned long long *d_n, size_t cap);
int rc_table_entries_in_dump_order(rc_ctx *ctx, std::vector<uint64_t> *codes, std::vector<int32_t> *counts);
int rc_error_rate_candidates(rc_ctx *ctx, const uint64_t *d_codes, size_t n, bool by_hash, size_t want, std::vector<uint64_t> *vals);
int rc_table_codes_device(rc_ctx *ctx, uint64_t **d_codes, size_t *n);
int rc_launch_last_base_variants(rc_ctx *ctx, const uint64_t *d_codes, size_t n, int32_t *d_max2);
int rc_launch_digest(rc_ctx *ctx, unsigned long long *d_out);
int rc_launch_locality_order(rc_ctx *ctx, const struct rc_device_batch_args &a, size_t nbytes);
int rc_launch_probe_list(rc_ctx *ctx, const struct rc_device_batch_args &a, size_t nbytes, int32_t *d_counts, int skip_hi = -1);
int rc_launch_tier_lists(rc_ctx *ctx, const struct rc_device_batch_args &a, int s_hi, int m_hi);
// K1 over section `section` (0: middle tier, 1: long tier) of ctx->tier_list; a.max_len = the longest read of that tier
int rc_launch_probe_tier(rc_ctx *ctx, const struct rc_device_batch_args &a, size_t nbytes, int32_t *d_counts, int section);
int rc_launch_compact(rc_ctx *ctx, const uint8_t *d_cls, uint32_t n, uint32_t *d_list, size_t stride, uint32_t *d_count);
int rc_launch_compact_local(rc_ctx *ctx, const uint8_t *d_cls, uint32_t n, uint32_t *d_list, size_t stride, uint32_t *d_count);
int rc_launch_compact_flag(rc_ctx *ctx, const uint8_t *d_flag, uint32_t n, uint32_t *d_list, size_t stride, uint32_t *d_count);

// rc_correct.hip
#define RC_TIER_ALL 0x7fffffff  // rc_kernel_args::tier_hi: no length tiers
#define RC_Q_MAX_KCNT 256        // the quarter-wave threshold kernel's widest instance (rc_quarter.h): k-mers / bases per read
#define RC_Q_MAX_LEN 320
struct rc_device_batch_args {
    int mode;            // 0 single, 1 paired (reads [0,n/2) are mates of [n/2,n)), 2 interleaved
    uint32_t n;          // reads
    uint8_t *seq;        // arena, reads NUL-terminated
    const uint8_t *qual; // same offsets (or one bit per arena byte, see rc_set_quality_bits)
    int qual_bits = 0;
    uint32_t qual_split = 0xFFFFFFFFu, qual_base2 = 0;
    const uint32_t *off; // n+1
    int32_t *ret, *l, *m, *h;
    int max_len;         // longest read in the batch (bases)
    int tier_lo = -1, tier_hi = RC_TIER_ALL;  // length tier of this pass (rc_kernel_args::tier_lo / tier_hi)
    int pair_override = -1;                   // rc_kernel_args::pair_override
    // the tier's reads as a list in locality order and its length on the device (rc_launch_tier_lists), or nullptr: the
    // threshold kernel of the pass then walks the whole batch and leaves the other tiers' reads out
    const uint32_t *tier_list = nullptr, *tier_n = nullptr;
};
int rc_launch_threshold(rc_ctx *ctx, const rc_device_batch_args &a, bool classify);
int rc_launch_correct(rc_ctx *ctx, const rc_device_batch_args &a);
int rc_launch_single(rc_ctx *ctx, const rc_device_batch_args &a, bool *ran);
int rc_launch_probe_threshold_list(rc_ctx *ctx, const rc_device_batch_args &a, size_t nbytes, bool *done);
int rc_launch_summary(rc_ctx *ctx, const int32_t *d_ret, uint32_t n);
int rc_launch_kmer_info(rc_ctx *ctx, const rc_device_batch_args &a);

// rc_transport.hip: the packed boundary (include/rcorrector_amd.h: rc_packed_batch)
int rc_launch_unpack(rc_ctx *ctx, const uint32_t *d_packed, size_t nbytes, const uint32_t *d_off, uint32_t n_reads, const uint32_t *d_exc_pos,
                     const uint8_t *d_exc_chr, uint32_t n_exc, uint8_t *d_seq);
int rc_launch_fix_list_bytes(rc_ctx *ctx, const uint8_t *d_orig_a, size_t bytes_a, const uint8_t *d_orig_b, size_t bytes_b, const uint8_t *d_seq,
                              uint32_t *d_n_fix, uint32_t cap, uint32_t *d_fix_pos, uint8_t *d_fix_chr);
int rc_launch_fix_list(rc_ctx *ctx, const uint32_t *d_packed, size_t nbytes, const uint8_t *d_seq, const uint32_t *d_exc_pos, uint32_t n_exc,
                       uint32_t *d_n_fix, uint32_t cap, uint32_t *d_fix_pos, uint8_t *d_fix_chr);

// k_correct's work list comes in RC_WORK_CLASSES sections, taken in order: the reads expected to be
// the most expensive first, so that the last waves of a launch are not left alone with them
// (cls value of a read = 1 + its section counted from the back; cls 0 = finished by the threshold kernel)
#define RC_WORK_CLASSES 4
#ifndef RC_FILTER_KIND_DEFAULT
#define RC_FILTER_KIND_DEFAULT 1  // "core": config 4's k_correct 6.31 -> 5.61 s (RC_TABLE_FILTER_KIND=plain for the other)
#endif
// layout of rc_ctx::work (bytes): RC_WORK_CLASSES x RC_HEADS queue heads of k_correct, 128 B apart,
// then the lengths of the work-list sections, then the phase counters of PROF builds
#define RC_WORK_BYTES 5120
#define RC_WORK_NWORK_OFF 4096
#define RC_WORK_NSINGLE_OFF 4160  // 4 x uint32: lengths of the three sections of k_single's list, 0 (it reads them like the four section lengths)
#define RC_WORK_NTIER_OFF 4192    // 2 x uint32: reads of the middle / long length tier (rc_launch_tier_lists)
#define RC_WORK_PHASE_OFF 4224
#define RC_WORK_SUMMARY_OFF 4608  // 2 x uint64: reads, corrected bases (never reset)
