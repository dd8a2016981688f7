// rc_api_packed.hip -- C ABI, the packed and the resident boundary (include/rcorrector_amd.h: rc_packed_batch,
// rc_resident_batch; device side in rc_transport.hip): what crosses PCIe is 2-bit bases / quality bits / offsets down and
// ret / l / m / h + a fix list up, or -- for reads the counter kept in HBM -- offsets and quality bits only.
#include "rc_api_internal.h"

extern "C" {

// ---- the packed boundary (include/rcorrector_amd.h: rc_packed_batch; device side in rc_transport.hip) ------------
size_t rc_pack_bases(const char *seq, size_t begin, size_t end, uint32_t *bases, uint32_t *exc_pos, uint8_t *exc_chr, size_t exc_cap)
{
    // letter -> code: A0 C1 G2 T3, 4 = NUL, 5 = anything else
    static const struct lut {
        uint8_t v[256];
        lut()
        {
            for (int i = 0; i < 256; ++i) v[i] = 5;
            v[0] = 4;
            v[(int)'A'] = 0;
            v[(int)'C'] = 1;
            v[(int)'G'] = 2;
            v[(int)'T'] = 3;
        }
    } L;
    size_t n_exc = 0;
    const unsigned char *s = reinterpret_cast<const unsigned char *>(seq);
    for (size_t w = begin >> 4; (w << 4) < end; ++w) {
        const size_t p0 = w << 4, lo = p0 < begin ? begin : p0, hi = p0 + 16 > end ? end : p0 + 16;
        uint32_t word = 0;
        for (size_t p = lo; p < hi; ++p) {
            const uint8_t c = L.v[s[p]];
            if (c < 4) {
                word |= (uint32_t)c << (30 - 2 * (p & 15));
            } else if (c == 5) {
                if (n_exc < exc_cap) {
                    exc_pos[n_exc] = (uint32_t)p;
                    exc_chr[n_exc] = s[p];
                }
                ++n_exc;
            }
        }
        // a range that starts inside a word keeps the bits of the positions in front of it (the caller packed them first)
        if (lo > p0) word |= bases[w] & ~(0xFFFFFFFFu >> (2 * (lo - p0)));
        bases[w] = word;
    }
    return n_exc;
}

void rc_apply_fixes(char *seq, const uint32_t *fix_pos, const uint8_t *fix_chr, size_t n_fix)
{
    for (size_t j = 0; j < n_fix; ++j) seq[fix_pos[j]] = (char)fix_chr[j];
}

int rc_submit_packed(rc_ctx *c, rc_packed_batch *b, int slot)
{
    if (!c || !b || slot < 0 || slot >= RC_MAX_SLOTS) return RC_ERR_ARG;
    if (rc_ctx *ln = rc_slot_lane(c, slot, true, true); ln != c) {  // (slot lanes, rc_internal.h: this slot runs in a context of its own)
        if (!ln) return RC_ERR_HIP;
        const int lrc = rc_submit_packed(ln, b, 0);
        if (lrc) rc_lane_error(c, ln);
        return lrc;
    }
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    if (b->mode < 0 || b->mode > 2 || (b->n && (!b->off || !b->bases || !b->ret || !b->l || !b->m || !b->h)) ||
        (b->n_exc && (!b->exc_pos || !b->exc_chr)) || (b->fix_cap && (!b->fix_pos || !b->fix_chr))) {
        rc_set_error(ctx, "submit_packed: bad batch descriptor");
        return RC_ERR_ARG;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = rc_slots_init(ctx);
    if (rc) return rc;
    rc_slot &sl = ctx->slots[slot];
    if (sl.busy) {
        rc_set_error(ctx, "submit_packed: slot %d still holds a batch (rc_wait_packed it first)", slot);
        return RC_ERR_STATE;
    }
    const size_t total = b->mode == 1 ? 2 * b->n : b->n, nbytes = (size_t)b->nbytes;
    b->n_fix = 0;
    if (total == 0) {
        sl.pb = b;
        sl.rb = nullptr;
        sl.b.n = b->n;
        sl.total_reads = total;
        sl.busy = true;
        return RC_OK;
    }
    if (nbytes >= (1ull << 32) || total >= (1ull << 32) || b->n_exc >= (1ull << 32) || b->fix_cap >= (1ull << 32)) {
        rc_set_error(ctx, "submit_packed: batch too large (split it)");
        return RC_ERR_ARG;
    }
    if (b->mode != 0 && (total & 1)) {
        rc_set_error(ctx, "submit_packed: %s mode needs an even number of reads", b->mode == 1 ? "paired" : "interleaved");
        return RC_ERR_ARG;
    }
    if (b->off[0] != 0 || b->off[total] != nbytes) {
        rc_set_error(ctx, "submit_packed: off[0] = %u, off[%zu] = %u do not describe the arena's %zu bytes", b->off[0], total, b->off[total], nbytes);
        return RC_ERR_ARG;
    }
    if (!ctx->d_buckets) {
        rc_set_error(ctx, "correct: no k-mer table loaded");
        return RC_ERR_STATE;
    }
    // every read ends with its NUL: strictly ascending offsets (the terminator / exception kernels write seq[off[i+1]-1] and
    // seq[exc_pos[i]] unchecked)
    int max_len = 0;
    for (size_t i = 0; i < total; ++i) {
        if (b->off[i + 1] <= b->off[i]) {
            rc_set_error(ctx, "submit_packed: off[%zu] = %u, off[%zu] = %u: offsets must ascend (a read is its bases and a NUL)", i, b->off[i], i + 1, b->off[i + 1]);
            return RC_ERR_ARG;
        }
        max_len = std::max(max_len, (int)(b->off[i + 1] - b->off[i]) - 1);
    }
    for (size_t i = 0; i < b->n_exc; ++i)
        if (b->exc_pos[i] >= nbytes) {
            rc_set_error(ctx, "submit_packed: exc_pos[%zu] = %u lies outside the arena's %zu bytes", i, b->exc_pos[i], nbytes);
            return RC_ERR_ARG;
        }
    sl.pb = b;
    sl.rb = nullptr;
    sl.b.n = b->n;
    sl.total_reads = total;
    const size_t n_words = (nbytes + 15) / 16, qb = (nbytes + 7) / 8, n_exc = b->n_exc;
    const uint32_t cap = (uint32_t)b->fix_cap;
    // device memory: the packed arena, the byte arena it expands into, qualities, offsets, results, exceptions, fixes
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_packed, n_words * 4 + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_seq, n_words * 16 + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_qual, (b->qual_bits ? qb : nbytes) + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_off, (total + 1) * 4))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_res, total * 16))) return rc;
    const size_t exc_chr_off = ((size_t)n_exc * 4 + 15) & ~(size_t)15;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_exc, exc_chr_off + n_exc + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_fix, 64))) return rc;  // (the count; the list itself goes to host memory)
    if ((rc = rc_hbuf_reserve(ctx, &sl.p_nfix, 64))) return rc;
    // inputs that are not page-locked go through one staging block of the slot
    const bool in_pinned = rc_is_pinned(b->off, (total + 1) * 4) && rc_is_pinned(b->bases, n_words * 4) && (!b->qual_bits || rc_is_pinned(b->qual_bits, qb)) &&
                           (!n_exc || (rc_is_pinned(b->exc_pos, n_exc * 4) && rc_is_pinned(b->exc_chr, n_exc)));
    const uint32_t *h_off = b->off, *h_bases = b->bases, *h_exc_pos = b->exc_pos;
    const uint8_t *h_qb = b->qual_bits, *h_exc_chr = b->exc_chr;
    if (!in_pinned) {
        const size_t o_bases = ((total + 1) * 4 + 63) & ~(size_t)63, o_qb = (o_bases + n_words * 4 + 63) & ~(size_t)63,
                     o_ep = (o_qb + qb + 63) & ~(size_t)63, o_ec = o_ep + n_exc * 4;
        if ((rc = rc_hbuf_reserve(ctx, &sl.p_in, o_ec + n_exc + 64))) return rc;
        char *s = (char *)sl.p_in.p;
        memcpy(s, b->off, (total + 1) * 4);
        memcpy(s + o_bases, b->bases, n_words * 4);
        if (b->qual_bits) memcpy(s + o_qb, b->qual_bits, qb);
        if (n_exc) {
            memcpy(s + o_ep, b->exc_pos, n_exc * 4);
            memcpy(s + o_ec, b->exc_chr, n_exc);
        }
        h_off = (const uint32_t *)s;
        h_bases = (const uint32_t *)(s + o_bases);
        h_qb = b->qual_bits ? (const uint8_t *)(s + o_qb) : nullptr;
        h_exc_pos = (const uint32_t *)(s + o_ep);
        h_exc_chr = (const uint8_t *)(s + o_ec);
    }
    sl.res_pinned = rc_is_pinned(b->ret, total * 4) && rc_is_pinned(b->l, total * 4) && rc_is_pinned(b->m, total * 4) && rc_is_pinned(b->h, total * 4);
    sl.fix_pinned = !cap || (rc_is_pinned(b->fix_pos, (size_t)cap * 4) && rc_is_pinned(b->fix_chr, cap));
    sl.fix_room = cap;
    if (!sl.res_pinned && (rc = rc_hbuf_reserve(ctx, &sl.p_res, total * 16))) return rc;
    if (!sl.fix_pinned && (rc = rc_hbuf_reserve(ctx, &sl.p_fix, (size_t)cap * 5 + 64))) return rc;
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
    uint32_t *d_exc_pos = (uint32_t *)sl.d_exc.p;
    uint8_t *d_exc_chr = (uint8_t *)sl.d_exc.p + exc_chr_off;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.d_packed.p, h_bases, n_words * 4, hipMemcpyHostToDevice, ctx->s_h2d));
    if (h_qb) RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.d_qual.p, h_qb, qb, hipMemcpyHostToDevice, ctx->s_h2d));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.d_off.p, h_off, (total + 1) * 4, hipMemcpyHostToDevice, ctx->s_h2d));
    if (n_exc) {
        RC_CHECK_HIP(ctx, hipMemcpyAsync(d_exc_pos, h_exc_pos, n_exc * 4, hipMemcpyHostToDevice, ctx->s_h2d));
        RC_CHECK_HIP(ctx, hipMemcpyAsync(d_exc_chr, h_exc_chr, n_exc, hipMemcpyHostToDevice, ctx->s_h2d));
    }
    RC_CHECK_HIP(ctx, hipEventRecord(sl.e_h2d, ctx->s_h2d));
    // kernels: expand, correct, list the substitutions
    RC_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->stream, sl.e_h2d, 0));
    uint8_t *d_seq = (uint8_t *)sl.d_seq.p;
    if ((rc = rc_launch_unpack(ctx, (const uint32_t *)sl.d_packed.p, nbytes, (const uint32_t *)sl.d_off.p, (uint32_t)total, d_exc_pos, d_exc_chr,
                               (uint32_t)n_exc, d_seq)))
        return rc;
    if (!h_qb) RC_CHECK_HIP(ctx, hipMemsetAsync(sl.d_qual.p, 0, nbytes, ctx->stream));  // FASTA: qual[0] == 0 (Reads.h:224-266)
    int32_t *d_res = (int32_t *)sl.d_res.p;
    rc_device_batch db;
    db.mode = b->mode;
    db.n_reads = (uint32_t)total;
    db.nbytes = nbytes;
    db.max_read_len = max_len;
    db.d_seq = d_seq;
    db.d_qual = (const uint8_t *)sl.d_qual.p;
    db.d_off = (const uint32_t *)sl.d_off.p;
    db.d_ret = d_res;
    db.d_l = d_res + total;
    db.d_m = d_res + 2 * total;
    db.d_h = d_res + 3 * total;
    if ((rc = rc_correct_device_impl(ctx, &db, 0xFFFFFFFFu, 0, h_qb ? 1 : 0))) return rc;
    // The fix list is written by the kernel straight into page-locked host memory (the caller's arrays, or the slot's
    // staging where those are pageable): a few bytes per read, consecutive entries from consecutive lanes.  A copy after
    // the kernels would have to wait for the count first -- a second round trip per batch on a stream of its own, which
    // on this runtime shares a hardware queue with one of the other four and stalls behind it.
    void *dp = nullptr, *dc = nullptr;
    if (cap) {
        RC_CHECK_HIP(ctx, hipHostGetDevicePointer(&dp, sl.fix_pinned ? (void *)b->fix_pos : sl.p_fix.p, 0));
        RC_CHECK_HIP(ctx, hipHostGetDevicePointer(&dc, sl.fix_pinned ? (void *)b->fix_chr : (void *)((char *)sl.p_fix.p + (size_t)cap * 4), 0));
    }
    uint32_t *d_fix_pos = (uint32_t *)dp;
    uint8_t *d_fix_chr = (uint8_t *)dc;
    uint32_t *d_nfix = (uint32_t *)sl.d_fix.p;
    if ((rc = rc_launch_fix_list(ctx, (const uint32_t *)sl.d_packed.p, nbytes, d_seq, d_exc_pos, (uint32_t)n_exc, d_nfix, cap, d_fix_pos, d_fix_chr))) return rc;
    RC_CHECK_HIP(ctx, hipEventRecord(sl.e_k, ctx->stream));
    // results; the fix list follows in rc_wait_packed, once its length is known
    RC_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->s_d2h, sl.e_k, 0));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.p_nfix.p, d_nfix, 4, hipMemcpyDeviceToHost, ctx->s_d2h));
    if (sl.res_pinned && b->l == b->ret + total && b->m == b->l + total && b->h == b->m + total) {
        // (the caller's four arrays are one block, as the device's are: one copy instead of four)
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->ret, d_res, total * 16, hipMemcpyDeviceToHost, ctx->s_d2h));
    } else if (sl.res_pinned) {
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

int rc_wait_packed(rc_ctx *c, int slot)
{
    if (!c || slot < 0 || slot >= RC_MAX_SLOTS) return RC_ERR_ARG;
    if (rc_ctx *ln = rc_slot_lane(c, slot, false, false); ln != c) {  // (slot lanes, rc_internal.h: this slot runs in a context of its own)
        if (!ln) return RC_ERR_HIP;
        const int lrc = rc_wait_packed(ln, 0);
        if (lrc) rc_lane_error(c, ln);
        return lrc;
    }
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    if (!ctx->slots || !ctx->slots[slot].busy || !ctx->slots[slot].pb) {
        rc_set_error(ctx, "wait_packed: slot %d holds no packed batch", slot);
        return RC_ERR_STATE;
    }
    rc_slot &sl = ctx->slots[slot];
    rc_packed_batch *b = sl.pb;
    sl.busy = false;
    sl.pb = nullptr;
    const size_t total = sl.total_reads;
    if (total == 0) return RC_OK;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    RC_CHECK_HIP(ctx, hipEventSynchronize(sl.e_done));  // the results and the fix count have landed; the list was written by the kernel
    const uint32_t n_fix = *(const volatile uint32_t *)sl.p_nfix.p, cap = sl.fix_room;
    if (n_fix > cap) {  // (the kernel stopped writing at cap; the results are complete, the list is not)
        b->n_fix = n_fix;
        rc_set_error(ctx, "wait_packed: %u substitutions, room for %u (fix_cap)", n_fix, cap);
        return RC_ERR_NOSPACE;
    }
    const uint32_t *o_pos = (const uint32_t *)sl.p_fix.p;
    const uint8_t *o_chr = (const uint8_t *)sl.p_fix.p + (size_t)cap * 4;
    if (!sl.res_pinned) {
        const int32_t *r = (const int32_t *)sl.p_res.p;
        memcpy(b->ret, r, total * 4);
        memcpy(b->l, r + total, total * 4);
        memcpy(b->m, r + 2 * total, total * 4);
        memcpy(b->h, r + 3 * total, total * 4);
    }
    if (!sl.fix_pinned && n_fix) {
        memcpy(b->fix_pos, o_pos, (size_t)n_fix * 4);
        memcpy(b->fix_chr, o_chr, n_fix);
    }
    b->n_fix = n_fix;
    return RC_OK;
}

int rc_submit_resident(rc_ctx *c, rc_resident_batch *b, int slot)
{
    if (!c || !b || slot < 0 || slot >= RC_MAX_SLOTS) return RC_ERR_ARG;
    if (rc_ctx *ln = rc_slot_lane(c, slot, true, true); ln != c) {  // (slot lanes, rc_internal.h: this slot runs in a context of its own)
        if (!ln) return RC_ERR_HIP;
        const int lrc = rc_submit_resident(ln, b, 0);
        if (lrc) rc_lane_error(c, ln);
        return lrc;
    }
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    if (b->mode < 0 || b->mode > 2 || (b->n && (!b->off || !b->ret || !b->l || !b->m || !b->h)) || (b->fix_cap && (!b->fix_pos || !b->fix_chr))) {
        rc_set_error(ctx, "submit_resident: bad batch descriptor");
        return RC_ERR_ARG;
    }
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = rc_slots_init(ctx);
    if (rc) return rc;
    rc_slot &sl = ctx->slots[slot];
    if (sl.busy) {
        rc_set_error(ctx, "submit_resident: slot %d still holds a batch (wait for it first)", slot);
        return RC_ERR_STATE;
    }
    const size_t total = b->mode == 1 ? 2 * b->n : b->n;
    const uint64_t bytes_b = b->mode == 1 ? b->bytes_b : 0;
    const size_t nbytes = (size_t)(b->bytes_a + bytes_b);
    b->n_fix = 0;
    if (total == 0) {
        sl.pb = nullptr;
        sl.rb = b;
        sl.b.n = b->n;
        sl.total_reads = total;
        sl.busy = true;
        return RC_OK;
    }
    if (b->bytes_a + bytes_b >= (1ull << 32) || total >= (1ull << 32) || b->fix_cap >= (1ull << 32)) {
        rc_set_error(ctx, "submit_resident: batch too large (split it)");
        return RC_ERR_ARG;
    }
    if (b->mode != 0 && (total & 1)) {
        rc_set_error(ctx, "submit_resident: %s mode needs an even number of reads", b->mode == 1 ? "paired" : "interleaved");
        return RC_ERR_ARG;
    }
    const size_t n_kept = ctx->kept_arenas.size();
    auto in_range = [&](int idx, uint64_t begin, uint64_t bytes) {
        return idx >= 0 && (size_t)idx < n_kept && begin <= ctx->kept_arenas[(size_t)idx].bytes && bytes <= ctx->kept_arenas[(size_t)idx].bytes - begin;
    };
    if (!in_range(b->arena_a, b->begin_a, b->bytes_a) || (b->mode == 1 && !in_range(b->arena_b, b->begin_b, b->bytes_b))) {
        rc_set_error(ctx, "submit_resident: no such range of a kept arena (%zu kept; rc_table_count_keep before counting)", n_kept);
        return RC_ERR_ARG;
    }
    if (b->off[0] != 0 || b->off[total] != nbytes || (b->mode == 1 && b->off[b->n] != b->bytes_a)) {
        rc_set_error(ctx, "submit_resident: the offsets do not describe the ranges (off[0] = %u, off[%zu] = %u, %zu bytes)", b->off[0], total, b->off[total], nbytes);
        return RC_ERR_ARG;
    }
    if (!ctx->d_buckets) {
        rc_set_error(ctx, "correct: no k-mer table loaded");
        return RC_ERR_STATE;
    }
    int max_len = 0;
    for (size_t i = 0; i < total; ++i) {
        if (b->off[i + 1] <= b->off[i]) {
            rc_set_error(ctx, "submit_resident: off[%zu] = %u, off[%zu] = %u: offsets must ascend (a read is its bases and a NUL)", i, b->off[i], i + 1, b->off[i + 1]);
            return RC_ERR_ARG;
        }
        max_len = std::max(max_len, (int)(b->off[i + 1] - b->off[i]) - 1);
    }
    sl.pb = nullptr;
    sl.rb = b;
    sl.b.n = b->n;
    sl.total_reads = total;
    const size_t qb = (nbytes + 7) / 8;
    const uint32_t cap = (uint32_t)b->fix_cap;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_seq, ((nbytes + 15) & ~(size_t)15) + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_qual, (b->qual_bits ? qb : nbytes) + 64))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_off, (total + 1) * 4))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_res, total * 16))) return rc;
    if ((rc = rc_dbuf_reserve(ctx, &sl.d_fix, 64))) return rc;
    if ((rc = rc_hbuf_reserve(ctx, &sl.p_nfix, 64))) return rc;
    const bool in_pinned = rc_is_pinned(b->off, (total + 1) * 4) && (!b->qual_bits || rc_is_pinned(b->qual_bits, qb));
    const uint32_t *h_off = b->off;
    const uint8_t *h_qb = b->qual_bits;
    if (!in_pinned) {
        const size_t o_qb = ((total + 1) * 4 + 63) & ~(size_t)63;
        if ((rc = rc_hbuf_reserve(ctx, &sl.p_in, o_qb + qb + 64))) return rc;
        char *s = (char *)sl.p_in.p;
        memcpy(s, b->off, (total + 1) * 4);
        if (b->qual_bits) memcpy(s + o_qb, b->qual_bits, qb);
        h_off = (const uint32_t *)s;
        h_qb = b->qual_bits ? (const uint8_t *)(s + o_qb) : nullptr;
    }
    sl.res_pinned = rc_is_pinned(b->ret, total * 4) && rc_is_pinned(b->l, total * 4) && rc_is_pinned(b->m, total * 4) && rc_is_pinned(b->h, total * 4);
    sl.fix_pinned = !cap || (rc_is_pinned(b->fix_pos, (size_t)cap * 4) && rc_is_pinned(b->fix_chr, cap));
    sl.fix_room = cap;
    if (!sl.res_pinned && (rc = rc_hbuf_reserve(ctx, &sl.p_res, total * 16))) return rc;
    if (!sl.fix_pinned && (rc = rc_hbuf_reserve(ctx, &sl.p_fix, (size_t)cap * 5 + 64))) return rc;
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
    if (h_qb) RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.d_qual.p, h_qb, qb, hipMemcpyHostToDevice, ctx->s_h2d));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.d_off.p, h_off, (total + 1) * 4, hipMemcpyHostToDevice, ctx->s_h2d));
    RC_CHECK_HIP(ctx, hipEventRecord(sl.e_h2d, ctx->s_h2d));
    RC_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->stream, sl.e_h2d, 0));
    // the batch's own arena: its ranges of the kept arenas, side by side
    uint8_t *d_seq = (uint8_t *)sl.d_seq.p;
    const uint8_t *orig_a = (const uint8_t *)ctx->kept_arenas[(size_t)b->arena_a].p + b->begin_a;
    const uint8_t *orig_b = bytes_b ? (const uint8_t *)ctx->kept_arenas[(size_t)b->arena_b].p + b->begin_b : nullptr;
    if (b->bytes_a) RC_CHECK_HIP(ctx, hipMemcpyAsync(d_seq, orig_a, b->bytes_a, hipMemcpyDeviceToDevice, ctx->stream));
    if (bytes_b) RC_CHECK_HIP(ctx, hipMemcpyAsync(d_seq + b->bytes_a, orig_b, bytes_b, hipMemcpyDeviceToDevice, ctx->stream));
    if (!h_qb) RC_CHECK_HIP(ctx, hipMemsetAsync(sl.d_qual.p, 0, nbytes, ctx->stream));  // FASTA: qual[0] == 0 (Reads.h:224-266)
    int32_t *d_res = (int32_t *)sl.d_res.p;
    rc_device_batch db;
    db.mode = b->mode;
    db.n_reads = (uint32_t)total;
    db.nbytes = nbytes;
    db.max_read_len = max_len;
    db.d_seq = d_seq;
    db.d_qual = (const uint8_t *)sl.d_qual.p;
    db.d_off = (const uint32_t *)sl.d_off.p;
    db.d_ret = d_res;
    db.d_l = d_res + total;
    db.d_m = d_res + 2 * total;
    db.d_h = d_res + 3 * total;
    if ((rc = rc_correct_device_impl(ctx, &db, 0xFFFFFFFFu, 0, h_qb ? 1 : 0))) return rc;
    void *dp = nullptr, *dc = nullptr;  // (the fix list goes straight into page-locked host memory, as in rc_submit_packed)
    if (cap) {
        RC_CHECK_HIP(ctx, hipHostGetDevicePointer(&dp, sl.fix_pinned ? (void *)b->fix_pos : sl.p_fix.p, 0));
        RC_CHECK_HIP(ctx, hipHostGetDevicePointer(&dc, sl.fix_pinned ? (void *)b->fix_chr : (void *)((char *)sl.p_fix.p + (size_t)cap * 4), 0));
    }
    uint32_t *d_nfix = (uint32_t *)sl.d_fix.p;
    if ((rc = rc_launch_fix_list_bytes(ctx, orig_a, (size_t)b->bytes_a, orig_b, (size_t)bytes_b, d_seq, d_nfix, cap, (uint32_t *)dp, (uint8_t *)dc))) return rc;
    RC_CHECK_HIP(ctx, hipEventRecord(sl.e_k, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->s_d2h, sl.e_k, 0));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(sl.p_nfix.p, d_nfix, 4, hipMemcpyDeviceToHost, ctx->s_d2h));
    if (sl.res_pinned && b->l == b->ret + total && b->m == b->l + total && b->h == b->m + total) {
        // (the caller's four arrays are one block, as the device's are: one copy instead of four)
        RC_CHECK_HIP(ctx, hipMemcpyAsync(b->ret, d_res, total * 16, hipMemcpyDeviceToHost, ctx->s_d2h));
    } else if (sl.res_pinned) {
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

int rc_wait_resident(rc_ctx *c, int slot)
{
    if (!c || slot < 0 || slot >= RC_MAX_SLOTS) return RC_ERR_ARG;
    if (rc_ctx *ln = rc_slot_lane(c, slot, false, false); ln != c) {  // (slot lanes, rc_internal.h: this slot runs in a context of its own)
        if (!ln) return RC_ERR_HIP;
        const int lrc = rc_wait_resident(ln, 0);
        if (lrc) rc_lane_error(c, ln);
        return lrc;
    }
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    if (!ctx->slots || !ctx->slots[slot].busy || !ctx->slots[slot].rb) {
        rc_set_error(ctx, "wait_resident: slot %d holds no resident batch", slot);
        return RC_ERR_STATE;
    }
    rc_slot &sl = ctx->slots[slot];
    rc_resident_batch *b = sl.rb;
    sl.busy = false;
    sl.rb = nullptr;
    const size_t total = sl.total_reads;
    if (total == 0) return RC_OK;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    RC_CHECK_HIP(ctx, hipEventSynchronize(sl.e_done));
    const uint32_t n_fix = *(const volatile uint32_t *)sl.p_nfix.p, cap = sl.fix_room;
    if (n_fix > cap) {  // (the kernel stopped writing at cap; the results are complete, the list is not)
        b->n_fix = n_fix;
        rc_set_error(ctx, "wait_resident: %u substitutions, room for %u (fix_cap)", n_fix, cap);
        return RC_ERR_NOSPACE;
    }
    if (!sl.res_pinned) {
        const int32_t *r = (const int32_t *)sl.p_res.p;
        memcpy(b->ret, r, total * 4);
        memcpy(b->l, r + total, total * 4);
        memcpy(b->m, r + 2 * total, total * 4);
        memcpy(b->h, r + 3 * total, total * 4);
    }
    if (!sl.fix_pinned && n_fix) {
        memcpy(b->fix_pos, sl.p_fix.p, (size_t)n_fix * 4);
        memcpy(b->fix_chr, (const uint8_t *)sl.p_fix.p + (size_t)cap * 4, n_fix);
    }
    b->n_fix = n_fix;
    return RC_OK;
}

}  // extern "C"
