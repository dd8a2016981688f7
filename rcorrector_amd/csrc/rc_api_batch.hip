// rc_api_batch.hip -- C ABI, the correction entry points (include/rcorrector_amd.h): HBM-resident batches, host batches
// (synchronous, traced, per read) and the asynchronous host-buffer path (rc_submit / rc_wait slots).
#include "rc_api_internal.h"

extern "C" {

// ---- correction ------------------------------------------------------------------------------
int rc_probe_device(rc_ctx *ctx, const uint8_t *d_seq, uint64_t nbytes, int32_t *d_counts)
{
    if (!ctx || !d_seq || !d_counts) return RC_ERR_ARG;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    return rc_launch_probe(ctx, d_seq, (size_t)nbytes, d_counts);
}


int rc_correct_device(rc_ctx *ctx, const rc_device_batch *b) { return rc_correct_device_impl(ctx, b, 0xFFFFFFFFu, 0); }

// qual_split / qual_base2 (quality-bit mode only): arena bytes from qual_split on have their bits at
// byte qual_base2 of d_qual -- the second arena of a paired host batch, whose bit array is separate
// qual_bits: -1 = as rc_set_quality_bits says, 0 / 1 = this batch's quality arena holds bytes / bits (the packed boundary)
int rc_correct_device_impl(rc_ctx *ctx, const rc_device_batch *b, uint32_t qual_split, uint32_t qual_base2, int qual_bits)
{
    if (!ctx || !b) return RC_ERR_ARG;
    if (b->n_reads == 0) return RC_OK;
    if (b->mode < 0 || b->mode > 2 || !b->d_seq || !b->d_qual || !b->d_off || !b->d_ret || !b->d_l || !b->d_m || !b->d_h) {
        rc_set_error(ctx, "correct_device: bad batch descriptor");
        return RC_ERR_ARG;
    }
    if (b->nbytes >= (1ull << 32)) {
        rc_set_error(ctx, "correct_device: arena of %llu bytes exceeds the 4 GiB batch limit", (unsigned long long)b->nbytes);
        return RC_ERR_ARG;
    }
    if (!ctx->d_buckets) {  // (before anything is launched: every probe kernel dereferences the table)
        rc_set_error(ctx, "correct: no k-mer table loaded");
        return RC_ERR_STATE;
    }
    // mates travel together (main.cpp:441, :459-468): an odd read count in a paired or interleaved batch has a
    // read without a mate -- refused before the locality order or the pair exchange of the threshold kernel see it
    if (b->mode != 0 && (b->n_reads & 1u)) {
        rc_set_error(ctx, "correct: %s mode needs an even number of reads (got %u)", b->mode == 1 ? "paired" : "interleaved", b->n_reads);
        return RC_ERR_ARG;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    bool fused = false;  // probe and threshold kernels ran as one
    if ((rc = rc_dbuf_reserve(ctx, &ctx->counts, (size_t)b->nbytes * 4 + 256))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->strong, (size_t)b->n_reads * 4 + 256))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->info, (size_t)b->n_reads * 4 + 256))) return rc;
    rc_device_batch_args a;
    a.mode = b->mode;
    a.n = b->n_reads;
    a.seq = b->d_seq;
    a.qual = b->d_qual;
    a.qual_bits = qual_bits >= 0 ? qual_bits : (ctx->qual_bits ? 1 : 0);
    a.qual_split = qual_split;
    a.qual_base2 = qual_base2;
    a.off = b->d_off;
    a.ret = b->d_ret;
    a.l = b->d_l;
    a.m = b->d_m;
    a.h = b->d_h;
    a.max_len = b->max_read_len;
    // the reads the threshold kernel could not finish, as a work list; isolated substitutions are finished four reads to
    // a wave first (rc_single.h: it clears their cls), what is left is k_correct's list
    bool loc_order_valid = false;  // ctx->loc_list holds this batch's locality order (every read once) and no length tiers are in play
    auto single_and_compact = [&](const rc_device_batch_args &at) -> int {
        if (!ctx->cls_ready) return RC_OK;
        ctx->work_stride = ((size_t)at.n + 63) & ~(size_t)63;
        int e;
        if ((e = rc_dbuf_reserve(ctx, &ctx->worklist, ctx->work_stride * RC_WORK_CLASSES * 4 + 256))) return e;
        bool ran = false;
        if ((e = rc_launch_single(ctx, at, &ran))) return e;
        if (ctx->env_k3_local && loc_order_valid)
            return rc_launch_compact_local(ctx, (const uint8_t *)ctx->cls.p, at.n, (uint32_t *)ctx->worklist.p, ctx->work_stride,
                                           (uint32_t *)((char *)ctx->work.p + RC_WORK_NWORK_OFF));
        return rc_launch_compact(ctx, (const uint8_t *)ctx->cls.p, at.n, (uint32_t *)ctx->worklist.p, ctx->work_stride,
                                 (uint32_t *)((char *)ctx->work.p + RC_WORK_NWORK_OFF));
    };
    // large batches over a table that does not fit the caches are probed in min-hash order (rc_table.hip),
    // so that overlapping reads meet in the L2 / Infinity Cache
    const bool locality = ctx->locality_mode >= 0 && (ctx->locality_mode > 0 || (a.n >= (1u << 18) && ctx->table_bytes > ((size_t)128 << 20))) &&
                          a.max_len + 8 <= 4000;
    // Length tiers.  The reference treats every read of up to 1 023 bases alike (utils.h:7, ErrorCorrection.cpp:682-1480);
    // here the fast kernels -- the fused probe + threshold kernel, k_single, the compiled-for-k k_correct -- hold reads
    // of up to 160 bases, the quarter-wave threshold kernel 320, and the longest read of a batch used to decide for all
    // of them.  A batch with longer reads is now processed in up to three passes over the same arena, one per tier a
    // unit's longer read falls into: S (<= 160 bases), M (the quarter-wave layout: <= 320 bases / 256 k-mers), L (the
    // rest); each pass = threshold kernel -> (k_single) -> compaction -> k_correct with the tier's capacity class, and the
    // threshold kernel of a pass marks the other tiers' reads cls = 0.  Same results (a read's result depends on its unit
    // alone), the short reads of a mixed batch keep their kernels.  Needs the classification (work lists).
    const int S_HI = 160;
    const int m_hi = RC_Q_MAX_KCNT - 1 + ctx->k < RC_Q_MAX_LEN ? RC_Q_MAX_KCNT - 1 + ctx->k : RC_Q_MAX_LEN;
    const bool tiered = a.max_len > S_HI && !ctx->env_no_tier && !ctx->env_no_classify && !ctx->env_k2_wave_per_read && ctx->trace_cap == 0;
    if (tiered) {
        rc_device_batch_args at = a;
        at.tier_lo = -1;
        at.tier_hi = S_HI;
        at.max_len = S_HI;
        bool lists = false;  // the middle / long tier's reads as lists in locality order (rc_launch_tier_lists)
        if (locality) {
            if ((rc = rc_launch_locality_order(ctx, a, (size_t)b->nbytes))) return rc;
            if ((rc = rc_launch_probe_threshold_list(ctx, at, (size_t)b->nbytes, &fused))) return rc;
            if (fused) {
                // the other tiers' reads are a few per cent of a typical mixed batch: probed (and, the middle tier,
                // thresholded) through compact lists -- walking the whole batch for them cost 3.7 + 3.1 ms of a 25 M-read step
                rc_device_batch_args al = a;
                if ((rc = rc_launch_tier_lists(ctx, a, S_HI, m_hi))) return rc;
                al.max_len = a.max_len < m_hi ? a.max_len : m_hi;
                if ((rc = rc_launch_probe_tier(ctx, al, (size_t)b->nbytes, (int32_t *)ctx->counts.p, 0))) return rc;
                al.max_len = a.max_len;
                if (a.max_len > m_hi && (rc = rc_launch_probe_tier(ctx, al, (size_t)b->nbytes, (int32_t *)ctx->counts.p, 1))) return rc;
                lists = true;
            } else if ((rc = rc_launch_probe_list(ctx, a, (size_t)b->nbytes, (int32_t *)ctx->counts.p, -1)))
                return rc;
        } else if ((rc = rc_launch_probe(ctx, b->d_seq, (size_t)b->nbytes, (int32_t *)ctx->counts.p)))
            return rc;
        ctx->thr_ready = true;  // every pass runs a threshold kernel: k_correct never computes a threshold itself
        for (int tier = 0; tier < 3; ++tier) {
            if (tier == 1) {
                at.tier_lo = S_HI;
                at.tier_hi = m_hi;
                at.max_len = a.max_len < m_hi ? a.max_len : m_hi;
                if (lists) {
                    at.tier_list = (const uint32_t *)ctx->tier_list.p;
                    at.tier_n = (const uint32_t *)((char *)ctx->work.p + RC_WORK_NTIER_OFF);
                }
            } else if (tier == 2) {
                if (a.max_len <= m_hi) break;
                at.tier_lo = m_hi;
                at.tier_hi = RC_TIER_ALL;
                at.max_len = a.max_len;
                at.tier_list = at.tier_n = nullptr;  // (the wave-per-read threshold kernel walks the batch)
            }
            if (!(tier == 0 && fused) && (rc = rc_launch_threshold(ctx, at, true))) return rc;
            if (!ctx->cls_ready) {
                rc_set_error(ctx, "correct: internal: a length tier ran without classification");
                return RC_ERR_STATE;
            }
            if ((rc = single_and_compact(at))) return rc;
            if ((rc = rc_launch_correct(ctx, at))) return rc;
        }
        return rc_launch_summary(ctx, a.ret, a.n);
    }
    if (locality) {
        if ((rc = rc_launch_locality_order(ctx, a, (size_t)b->nbytes))) return rc;
        loc_order_valid = true;
        // probe + threshold + classification in one kernel where the reads fit it
        if ((rc = rc_launch_probe_threshold_list(ctx, a, (size_t)b->nbytes, &fused))) return rc;
        if (!fused && (rc = rc_launch_probe_list(ctx, a, (size_t)b->nbytes, (int32_t *)ctx->counts.p))) return rc;
    } else if ((rc = rc_launch_probe(ctx, b->d_seq, (size_t)b->nbytes, (int32_t *)ctx->counts.p)))
        return rc;
#ifdef RC_EXP_PROBE_ONLY  // dev (with RC_EXP_ADDR_WINDOW, whose counts are wrong by design): nothing behind the probe kernel runs
    return RC_OK;
#endif
    // thresholds: mates need each other's before either can be corrected, so paired / interleaved
    // batches always run the threshold kernel first; single-end batches do too when every read fits
    // the four-reads-per-wave kernel (cheaper there than inside k_correct), else k_correct computes them
    ctx->thr_ready = fused;
    if (!fused) ctx->cls_ready = false;
    const bool quarter_ok = a.max_len <= 320 && a.max_len - ctx->k + 1 <= 256 && !ctx->env_k2_wave_per_read;
    if (!fused && (a.mode != 0 || quarter_ok)) {
        if ((rc = rc_launch_threshold(ctx, a, true))) return rc;
        ctx->thr_ready = true;
    }
    if ((rc = single_and_compact(a))) return rc;
    if ((rc = rc_launch_correct(ctx, a))) return rc;
    // UpdateSummary (main.cpp:73-79), on the device: the counters live in HBM until rc_summary() asks
    return rc_launch_summary(ctx, a.ret, a.n);
}

// GetStrongTrustedThreshold (ErrorCorrection.h:26, ErrorCorrection.cpp:1482-1565) for every read of
// an arena in HBM: probe kernel + threshold kernel, the per-read values copied to d_strong
int rc_strong_threshold_device(rc_ctx *ctx, const uint8_t *d_seq, const uint32_t *d_off, uint32_t n_reads, uint64_t nbytes,
                               int32_t max_read_len, int32_t *d_strong)
{
    if (!ctx || !d_seq || !d_off || !d_strong) return RC_ERR_ARG;
    if (n_reads == 0) return RC_OK;
    if (nbytes >= (1ull << 32)) {
        rc_set_error(ctx, "strong_threshold_device: arena of %llu bytes exceeds the 4 GiB batch limit", (unsigned long long)nbytes);
        return RC_ERR_ARG;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->counts, (size_t)nbytes * 4 + 256))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->strong, (size_t)n_reads * 4 + 256))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->info, (size_t)n_reads * 4 + 256))) return rc;
    rc_device_batch_args a = rc_device_batch_args();  // (value-initialised: zeros, and the members with defaults -- no tiers)
    a.mode = 0;
    a.n = n_reads;
    a.seq = const_cast<uint8_t *>(d_seq);
    a.off = d_off;
    a.max_len = max_read_len;
    if ((rc = rc_launch_probe(ctx, d_seq, (size_t)nbytes, (int32_t *)ctx->counts.p))) return rc;
    if ((rc = rc_launch_threshold(ctx, a, false))) return rc;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(d_strong, ctx->strong.p, (size_t)n_reads * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return RC_OK;
}

// ---- per-read entry points: the granularity of ErrorCorrection.h:26-28, each a batch of one through the kernels above
// (a launch and two copies per call -- for bindings that work read by read and for spot checks, not for throughput)
static int one_read_upload(rc_ctx *ctx, const char *seq, const char *qual, rc_device_batch_args &a, size_t *len1)
{
    if (!seq) return RC_ERR_ARG;
    const size_t n1 = strlen(seq) + 1;
    if (n1 > RC_MAX_READ_LENGTH) {
        rc_set_error(ctx, "read of %zu bases exceeds the %d-base limit (utils.h:7)", n1 - 1, RC_MAX_READ_LENGTH - 1);
        return RC_ERR_ARG;
    }
    if (ctx->qual_bits) {
        rc_set_error(ctx, "the per-read entry points take quality bytes (rc_set_quality_bits is on)");
        return RC_ERR_STATE;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->h_seq, n1 + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->h_qual, n1 + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->h_off, 2 * 4))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->h_res, 4 * 4))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->counts, n1 * 4 + 256))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->strong, 4 + 256))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->info, 4 + 256))) return rc;
    const uint32_t off[2] = {0u, (uint32_t)n1};
    RC_CHECK_HIP(ctx, hipMemcpyAsync(ctx->h_seq.p, seq, n1, hipMemcpyHostToDevice, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemsetAsync(ctx->h_qual.p, 0, n1, ctx->stream));  // (no qualities: the FASTA marker qual[0] == 0)
    if (qual) {
        const size_t q1 = strnlen(qual, n1 - 1);
        RC_CHECK_HIP(ctx, hipMemcpyAsync(ctx->h_qual.p, qual, q1, hipMemcpyHostToDevice, ctx->stream));
    }
    RC_CHECK_HIP(ctx, hipMemcpyAsync(ctx->h_off.p, off, sizeof off, hipMemcpyHostToDevice, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (off and possibly seq are on the caller's stack)
    int32_t *d_res = (int32_t *)ctx->h_res.p;
    a = rc_device_batch_args();
    a.mode = 0;
    a.n = 1;
    a.seq = (uint8_t *)ctx->h_seq.p;
    a.qual = (const uint8_t *)ctx->h_qual.p;
    a.off = (const uint32_t *)ctx->h_off.p;
    a.ret = d_res;
    a.l = d_res + 1;
    a.m = d_res + 2;
    a.h = d_res + 3;
    a.max_len = (int)n1 - 1;
    *len1 = n1;
    return RC_OK;
}

int rc_strong_threshold_read(rc_ctx *ctx, const char *seq, int32_t *strong)
{
    if (!ctx || !seq || !strong) return RC_ERR_ARG;
    rc_device_batch_args a;
    size_t n1;
    int rc = one_read_upload(ctx, seq, nullptr, a, &n1);
    if (rc) return rc;
    if ((rc = rc_launch_probe(ctx, a.seq, n1, (int32_t *)ctx->counts.p))) return rc;
    if ((rc = rc_launch_threshold(ctx, a, false))) return rc;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(strong, ctx->strong.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

int rc_correct_read(rc_ctx *ctx, char *seq, const char *qual, int32_t pair_strong_threshold, int32_t *ret)
{
    if (!ctx || !seq || !ret) return RC_ERR_ARG;
    rc_device_batch_args a;
    size_t n1;
    int rc = one_read_upload(ctx, seq, qual, a, &n1);
    if (rc) return rc;
    a.pair_override = pair_strong_threshold;
    if ((rc = rc_launch_probe(ctx, a.seq, n1, (int32_t *)ctx->counts.p))) return rc;
    // no threshold kernel, no classification: k_correct computes the read's own threshold (its single-end front end) and
    // takes the pair's from the argument, exactly the reference's call
    ctx->thr_ready = false;
    ctx->cls_ready = false;
    ctx->cand_ready = false;
    if ((rc = rc_launch_correct(ctx, a))) return rc;
    if ((rc = rc_launch_summary(ctx, a.ret, 1))) return rc;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(seq, a.seq, n1 - 1, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(ret, a.ret, 4, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

int rc_kmer_info_read(rc_ctx *ctx, const char *seq, int32_t *l, int32_t *m, int32_t *h)
{
    if (!ctx || !seq || !l || !m || !h) return RC_ERR_ARG;
    rc_device_batch_args a;
    size_t n1;
    int rc = one_read_upload(ctx, seq, nullptr, a, &n1);
    if (rc) return rc;
    if ((rc = rc_launch_probe(ctx, a.seq, n1, (int32_t *)ctx->counts.p))) return rc;
    if ((rc = rc_launch_kmer_info(ctx, a))) return rc;
    int32_t out[3];
    RC_CHECK_HIP(ctx, hipMemcpyAsync(out, a.l, 12, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *l = out[0];
    *m = out[1];
    *h = out[2];
    return RC_OK;
}

int rc_sync(rc_ctx *ctx)
{
    if (!ctx) return RC_ERR_ARG;
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (rc_ctx *ln : ctx->lane)
        if (ln) RC_CHECK_HIP(ctx, hipStreamSynchronize(ln->stream));
    return RC_OK;
}

static int correct_batch_impl(rc_ctx *c, rc_batch *b, rc_trace *t);

int rc_submit(rc_ctx *c, const rc_batch *b, int slot);
int rc_wait(rc_ctx *c, int slot);

int rc_correct_batch(rc_ctx *c, rc_batch *b)
{
    int rc = rc_submit(c, b, 0);
    if (rc) return rc;
    return rc_wait(c, 0);
}

// rc_correct_batch + what the reference prints under -verbose (VERBOSE, ErrorCorrection.cpp:15):
// the counts before (:759-770) and after (:1590-1597) come from two extra runs of the probe
// kernel, the per-iteration thresholds and bitmaps (:856-857, :1088-1094) from the TRACE build of
// k_correct
int rc_correct_batch_traced(rc_ctx *c, rc_batch *b, rc_trace *t)
{
    if (!c || !b || !t) return RC_ERR_ARG;
    if (t->max_iter < 1 || !t->counts_before || !t->counts_after || !t->flags || !t->n_iter || !t->iter) {
        rc_set_error(c, "correct_batch_traced: bad trace descriptor");
        return RC_ERR_ARG;
    }
    if (c->qual_bits) {
        rc_set_error(c, "correct_batch_traced: not available in quality-bit mode");
        return RC_ERR_STATE;
    }
    c->trace_cap = t->max_iter;
    int rc = correct_batch_impl(c, b, t);
    c->trace_cap = 0;
    return rc;
}

static int correct_batch_impl(rc_ctx *c, rc_batch *b, rc_trace *t)
{
    if (!c || !b) return RC_ERR_ARG;
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    if (b->n == 0) return RC_OK;
    if (b->mode < 0 || b->mode > 2 || !b->seq || !b->qual || !b->off || !b->ret || !b->l || !b->m || !b->h ||
        (b->mode == 1 && (!b->seq2 || !b->qual2 || !b->off2))) {
        rc_set_error(ctx, "correct_batch: bad batch descriptor");
        return RC_ERR_ARG;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n1 = b->n;
    const size_t bytes1 = b->off[n1], bytes2 = b->mode == 1 ? b->off2[n1] : 0;
    const size_t total_reads = b->mode == 1 ? 2 * n1 : n1;
    const size_t nbytes = bytes1 + bytes2;
    if (nbytes >= (1ull << 32) || total_reads >= (1ull << 32)) {
        rc_set_error(ctx, "correct_batch: batch too large (split it)");
        return RC_ERR_ARG;
    }
    std::vector<uint32_t> off(total_reads + 1);
    int max_len = 0;
    for (size_t i = 0; i <= n1; ++i) off[i] = b->off[i];
    for (size_t i = 0; i < n1; ++i) max_len = std::max(max_len, (int)(b->off[i + 1] - b->off[i]) - 1);
    if (b->mode == 1) {
        for (size_t i = 0; i <= n1; ++i) off[n1 + i] = (uint32_t)bytes1 + b->off2[i];
        for (size_t i = 0; i < n1; ++i) max_len = std::max(max_len, (int)(b->off2[i + 1] - b->off2[i]) - 1);
    }
    int rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->h_seq, nbytes + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->h_qual, nbytes + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->h_off, (total_reads + 1) * 4))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &ctx->h_res, total_reads * 16))) return rc;
    uint8_t *d_seq = (uint8_t *)ctx->h_seq.p, *d_qual = (uint8_t *)ctx->h_qual.p;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(d_seq, b->seq, bytes1, hipMemcpyHostToDevice, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(d_qual, b->qual, bytes1, hipMemcpyHostToDevice, ctx->stream));
    if (b->mode == 1) {
        RC_CHECK_HIP(ctx, hipMemcpyAsync(d_seq + bytes1, b->seq2, bytes2, hipMemcpyHostToDevice, ctx->stream));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(d_qual + bytes1, b->qual2, bytes2, hipMemcpyHostToDevice, ctx->stream));
    }
    RC_CHECK_HIP(ctx, hipMemcpyAsync(ctx->h_off.p, off.data(), (total_reads + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    int32_t *d_res = (int32_t *)ctx->h_res.p;
    rc_device_batch db;
    db.mode = b->mode;
    db.n_reads = (uint32_t)total_reads;
    db.nbytes = nbytes;
    db.max_read_len = max_len;
    db.d_seq = d_seq;
    db.d_qual = d_qual;
    db.d_off = (const uint32_t *)ctx->h_off.p;
    db.d_ret = d_res;
    db.d_l = d_res + total_reads;
    db.d_m = d_res + 2 * total_reads;
    db.d_h = d_res + 3 * total_reads;
    if ((rc = rc_correct_device(ctx, &db))) return rc;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(b->seq, d_seq, bytes1, hipMemcpyDeviceToHost, ctx->stream));
    if (b->mode == 1) RC_CHECK_HIP(ctx, hipMemcpyAsync(b->seq2, d_seq + bytes1, bytes2, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(b->ret, db.d_ret, total_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(b->l, db.d_l, total_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(b->m, db.d_m, total_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(b->h, db.d_h, total_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (t) {
        // before: K1's output is still in ctx->counts; after: probe the corrected arena once more
        RC_CHECK_HIP(ctx, hipMemcpyAsync(t->counts_before, ctx->counts.p, nbytes * 4, hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = rc_launch_probe(ctx, d_seq, nbytes, (int32_t *)ctx->counts.p))) return rc;
        RC_CHECK_HIP(ctx, hipMemcpyAsync(t->counts_after, ctx->counts.p, nbytes * 4, hipMemcpyDeviceToHost, ctx->stream));
        const size_t rec = 2 + (size_t)t->max_iter * RC_TRACE_WORDS;
        std::vector<int32_t> raw(total_reads * rec);
        RC_CHECK_HIP(ctx, hipMemcpyAsync(raw.data(), ctx->trace.p, raw.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < total_reads; ++i) {
            const int32_t *r = raw.data() + i * rec;
            t->flags[i] = r[0];
            t->n_iter[i] = r[1];
            memcpy(t->iter + i * (size_t)t->max_iter * RC_TRACE_WORDS, r + 2, (size_t)t->max_iter * RC_TRACE_WORDS * 4);
        }
    }
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

// ---- asynchronous host-buffer path ---------------------------------------------------------------
// The reference overlaps the I/O of batch N+1 with the correction of batch N by handing batches to
// worker threads (main.cpp:479-516).  Here one context keeps up to RC_MAX_SLOTS batches in flight on
// three streams: H2D(N+1) || kernels(N) || D2H(N-1).  Scratch memory of the kernels is shared --
// they serialise on the compute stream -- only the arenas and result arrays exist per slot.
int rc_hbuf_reserve(rc_ctx *ctx, rc_hbuf *h, size_t bytes)
{
    if (bytes <= h->bytes) return RC_OK;
    if (h->p) (void)hipHostFree(h->p);
    h->p = nullptr;
    h->bytes = 0;
    const size_t want = bytes + bytes / 8 + 4096;
    RC_CHECK_HIP(ctx, hipHostMalloc(&h->p, want, hipHostMallocDefault));
    h->bytes = want;
    return RC_OK;
}

static bool is_pinned_at(const void *p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // pageable memory the runtime has never seen
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// the whole range [p, p + bytes) is page-locked: its first and last byte are (a registration or a
// hipHostMalloc block is one contiguous range, so a buffer that starts and ends inside pinned memory and was
// handed over as one array lies in it -- unless it straddles two separate registrations, which then both
// cover their part)
bool rc_is_pinned(const void *p, size_t bytes)
{
    if (!p) return false;
    if (!is_pinned_at(p)) return false;
    return bytes <= 1 || is_pinned_at(static_cast<const char *>(p) + bytes - 1);
}

int rc_host_alloc(rc_ctx *ctx, size_t bytes, void **out)
{
    if (!ctx || !out) return RC_ERR_ARG;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    RC_CHECK_HIP(ctx, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return RC_OK;
}

int rc_host_free(rc_ctx *ctx, void *p)
{
    if (!ctx) return RC_ERR_ARG;
    if (p) RC_CHECK_HIP(ctx, hipHostFree(p));
    return RC_OK;
}

// page-locks caller memory (any allocation, whole pages) so that rc_submit can DMA straight from / to it
int rc_host_register(void *p, size_t bytes)
{
    if (!p || !bytes) return RC_ERR_ARG;
    return hipHostRegister(p, bytes, hipHostRegisterPortable) == hipSuccess ? RC_OK : RC_ERR_HIP;
}

int rc_host_unregister(void *p)
{
    if (!p) return RC_ERR_ARG;
    return hipHostUnregister(p) == hipSuccess ? RC_OK : RC_ERR_HIP;
}

int rc_slots_init(rc_ctx *ctx)
{
    if (ctx->slots) return RC_OK;
    RC_CHECK_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_h2d, hipStreamNonBlocking));
    RC_CHECK_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_d2h, hipStreamNonBlocking));
    ctx->slots = new (std::nothrow) rc_slot[RC_MAX_SLOTS];
    if (!ctx->slots) return RC_ERR_NOMEM;
    for (int i = 0; i < RC_MAX_SLOTS; ++i) {
        rc_slot &sl = ctx->slots[i];
        RC_CHECK_HIP(ctx, hipEventCreateWithFlags(&sl.e_h2d, hipEventDisableTiming));
        RC_CHECK_HIP(ctx, hipEventCreateWithFlags(&sl.e_k, hipEventDisableTiming));
        RC_CHECK_HIP(ctx, hipEventCreateWithFlags(&sl.e_done, hipEventDisableTiming));
    }
    return RC_OK;
}

int rc_submit(rc_ctx *c, const rc_batch *b, int slot)
{
    if (!c || !b || slot < 0 || slot >= RC_MAX_SLOTS) return RC_ERR_ARG;
    if (rc_ctx *ln = rc_slot_lane(c, slot, true, true); ln != c) {  // (slot lanes, rc_internal.h: this slot runs in a context of its own)
        if (!ln) return RC_ERR_HIP;
        const int lrc = rc_submit(ln, b, 0);
        if (lrc) rc_lane_error(c, ln);
        return lrc;
    }
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    if (b->mode < 0 || b->mode > 2 || (b->n && (!b->seq || !b->qual || !b->off || !b->ret || !b->l || !b->m || !b->h)) ||
        (b->n && b->mode == 1 && (!b->seq2 || !b->qual2 || !b->off2))) {
        rc_set_error(ctx, "submit: bad batch descriptor");
        return RC_ERR_ARG;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = rc_slots_init(ctx);
    if (rc) return rc;
    rc_slot &sl = ctx->slots[slot];
    if (sl.busy) {
        rc_set_error(ctx, "submit: slot %d still holds a batch (rc_wait it first)", slot);
        return RC_ERR_STATE;
    }
    sl.b = *b;
    sl.pb = nullptr;
    sl.rb = nullptr;
    const size_t n1 = b->n;
    sl.total_reads = b->mode == 1 ? 2 * n1 : n1;
    sl.bytes1 = n1 ? b->off[n1] : 0;
    sl.bytes2 = (n1 && b->mode == 1) ? b->off2[n1] : 0;
    if (n1 == 0) {
        sl.busy = true;
        return RC_OK;
    }
    const size_t nbytes = sl.bytes1 + sl.bytes2, total = sl.total_reads;
    if (nbytes >= (1ull << 32) || total >= (1ull << 32)) {
        rc_set_error(ctx, "submit: batch too large (split it)");
        return RC_ERR_ARG;
    }
    if (b->mode == 2 && (n1 & 1)) {  // (before any copy is queued)
        rc_set_error(ctx, "submit: interleaved mode needs an even number of reads (got %zu)", n1);
        return RC_ERR_ARG;
    }
    if (!ctx->d_buckets) {
        rc_set_error(ctx, "correct: no k-mer table loaded");
        return RC_ERR_STATE;
    }
    // offsets of the device arena (arena 1 then arena 2) and the longest read, into pinned memory
    if ((rc = rc_hbuf_reserve(ctx, &sl.p_off, (total + 1) * 4))) return rc;
    uint32_t *off = (uint32_t *)sl.p_off.p;
    int max_len = 0;
    memcpy(off, b->off, (n1 + 1) * 4);
    for (size_t i = 0; i < n1; ++i) max_len = std::max(max_len, (int)(b->off[i + 1] - b->off[i]) - 1);
    if (b->mode == 1) {
        for (size_t i = 0; i <= n1; ++i) off[n1 + i] = (uint32_t)sl.bytes1 + b->off2[i];
        for (size_t i = 0; i < n1; ++i) max_len = std::max(max_len, (int)(b->off2[i + 1] - b->off2[i]) - 1);
    }
    // quality arenas: a byte per base, or (rc_set_quality_bits) a bit per arena byte, arena 2's bits in
    // a region of their own
    const bool qbits = ctx->qual_bits;
    const size_t q1 = qbits ? (sl.bytes1 + 7) / 8 : sl.bytes1, q2 = qbits ? (sl.bytes2 + 7) / 8 : sl.bytes2;
    const size_t qbase2 = qbits ? ((q1 + 15) & ~(size_t)15) : sl.bytes1;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_seq, nbytes + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_qual, qbase2 + q2 + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_off, (total + 1) * 4))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_res, total * 16))) return rc;
    sl.seq_pinned = rc_is_pinned(b->seq, sl.bytes1) && rc_is_pinned(b->qual, q1) &&
                    (b->mode != 1 || (rc_is_pinned(b->seq2, sl.bytes2) && rc_is_pinned(b->qual2, q2)));
    sl.res_pinned = rc_is_pinned(b->ret, total * 4) && rc_is_pinned(b->l, total * 4) && rc_is_pinned(b->m, total * 4) && rc_is_pinned(b->h, total * 4);
    const char *h_seq1 = b->seq, *h_qual1 = b->qual, *h_seq2 = b->seq2, *h_qual2 = b->qual2;
    if (!sl.seq_pinned) {  // pageable buffers: through the slot's pinned staging
        if ((rc = rc_hbuf_reserve(ctx, &sl.p_seq, nbytes))) return rc;
        if ((rc = rc_hbuf_reserve(ctx, &sl.p_qual, qbase2 + q2))) return rc;
        memcpy(sl.p_seq.p, b->seq, sl.bytes1);
        memcpy(sl.p_qual.p, b->qual, q1);
        if (b->mode == 1) {
            memcpy((char *)sl.p_seq.p + sl.bytes1, b->seq2, sl.bytes2);
            memcpy((char *)sl.p_qual.p + qbase2, b->qual2, q2);
        }
        h_seq1 = (const char *)sl.p_seq.p;
        h_qual1 = (const char *)sl.p_qual.p;
        h_seq2 = h_seq1 + sl.bytes1;
        h_qual2 = h_qual1 + qbase2;
    }
    if (!sl.res_pinned && (rc = rc_hbuf_reserve(ctx, &sl.p_res, total * 16))) return rc;
    uint8_t *d_seq = (uint8_t *)sl.d_seq.p, *d_qual = (uint8_t *)sl.d_qual.p;
    // one upload stream: bases and qualities on two streams measured 21 GB/s against 26.6 GB/s on one
    // (the link, not a DMA engine, is the bound)
    hipStream_t sq = ctx->s_h2d;
    // from here on copies are in flight from the caller's buffers (or the slot's staging): an error must not
    // return before they have drained, or the caller could free / the next submit could overwrite memory the
    // DMA engines still read
    struct drain_on_error {
        rc_ctx *c;
        bool armed = true;
        ~drain_on_error()
        {
            if (!armed) return;
            (void)hipStreamSynchronize(c->s_h2d);
            (void)hipStreamSynchronize(c->stream);
            (void)hipStreamSynchronize(c->s_d2h);
        }
    } guard{ctx};
    RC_CHECK_HIP(ctx, hipMemcpyAsync(d_seq, h_seq1, sl.bytes1, hipMemcpyHostToDevice, ctx->s_h2d));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(d_qual, h_qual1, q1, hipMemcpyHostToDevice, sq));
    if (b->mode == 1) {
        RC_CHECK_HIP(ctx, hipMemcpyAsync(d_seq + sl.bytes1, h_seq2, sl.bytes2, hipMemcpyHostToDevice, ctx->s_h2d));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(d_qual + qbase2, h_qual2, q2, hipMemcpyHostToDevice, sq));
    }
    RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.d_off.p, off, (total + 1) * 4, hipMemcpyHostToDevice, ctx->s_h2d));
    RC_CHECK_HIP(ctx, hipEventRecord(sl.e_h2d, ctx->s_h2d));
    // kernels
    RC_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->stream, sl.e_h2d, 0));
    int32_t *d_res = (int32_t *)sl.d_res.p;
    rc_device_batch db;
    db.mode = b->mode;
    db.n_reads = (uint32_t)total;
    db.nbytes = nbytes;
    db.max_read_len = max_len;
    db.d_seq = d_seq;
    db.d_qual = d_qual;
    db.d_off = (const uint32_t *)sl.d_off.p;
    db.d_ret = d_res;
    db.d_l = d_res + total;
    db.d_m = d_res + 2 * total;
    db.d_h = d_res + 3 * total;
    if ((rc = rc_correct_device_impl(ctx, &db, qbits && b->mode == 1 ? (uint32_t)sl.bytes1 : 0xFFFFFFFFu, (uint32_t)qbase2))) return rc;
    RC_CHECK_HIP(ctx, hipEventRecord(sl.e_k, ctx->stream));
    // results
    RC_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->s_d2h, sl.e_k, 0));
    char *o_seq1 = sl.seq_pinned ? b->seq : (char *)sl.p_seq.p;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(o_seq1, d_seq, sl.bytes1, hipMemcpyDeviceToHost, ctx->s_d2h));
    if (b->mode == 1) {
        char *o_seq2 = sl.seq_pinned ? b->seq2 : (char *)sl.p_seq.p + sl.bytes1;
        RC_CHECK_HIP(ctx, hipMemcpyAsync(o_seq2, d_seq + sl.bytes1, sl.bytes2, hipMemcpyDeviceToHost, ctx->s_d2h));
    }
    if (sl.res_pinned) {
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->ret, db.d_ret, total * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->l, db.d_l, total * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->m, db.d_m, total * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->h, db.d_h, total * 4, hipMemcpyDeviceToHost, ctx->s_d2h));
    } else {
        RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.p_res.p, d_res, total * 16, hipMemcpyDeviceToHost, ctx->s_d2h));
    }
    RC_CHECK_HIP(ctx, hipEventRecord(sl.e_done, ctx->s_d2h));
    guard.armed = false;
    sl.busy = true;
    return RC_OK;
}

int rc_wait(rc_ctx *c, int slot)
{
    if (!c || slot < 0 || slot >= RC_MAX_SLOTS) return RC_ERR_ARG;
    if (rc_ctx *ln = rc_slot_lane(c, slot, false, false); ln != c) {  // (slot lanes, rc_internal.h: this slot runs in a context of its own)
        if (!ln) return RC_ERR_HIP;
        const int lrc = rc_wait(ln, 0);
        if (lrc) rc_lane_error(c, ln);
        return lrc;
    }
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    if (!ctx->slots || !ctx->slots[slot].busy) {
        rc_set_error(ctx, "wait: slot %d holds no batch", slot);
        return RC_ERR_STATE;
    }
    rc_slot &sl = ctx->slots[slot];
    if (sl.pb || sl.rb) {
        rc_set_error(ctx, "wait: slot %d holds a packed batch (rc_wait_packed / rc_wait_resident)", slot);
        return RC_ERR_STATE;
    }
    sl.busy = false;
    if (sl.b.n == 0) return RC_OK;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    RC_CHECK_HIP(ctx, hipEventSynchronize(sl.e_done));
    const size_t total = sl.total_reads;
    if (!sl.seq_pinned) {
        memcpy(sl.b.seq, sl.p_seq.p, sl.bytes1);
        if (sl.b.mode == 1) memcpy(sl.b.seq2, (char *)sl.p_seq.p + sl.bytes1, sl.bytes2);
    }
    if (!sl.res_pinned) {
        const int32_t *r = (const int32_t *)sl.p_res.p;
        memcpy(sl.b.ret, r, total * 4);
        memcpy(sl.b.l, r + total, total * 4);
        memcpy(sl.b.m, r + 2 * total, total * 4);
        memcpy(sl.b.h, r + 3 * total, total * 4);
    }
    return RC_OK;
}

}  // extern "C"
