// rc_api_internal.h -- what the translation units of the C ABI (rc_api*.hip) share: the context as the ABI layer sees it,
// the slots of the asynchronous entry points, and the helpers that cross the units.
#pragma once
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <new>
#include <unistd.h>
#include <thread>
#include <vector>

#include "../../include/rcorrector_amd.h"
#include "rc_internal.h"

// parsed dump kept between rc_table_load_jfdump() and rc_estimate_error_rate()
struct rc_dump_cache {
    // forward code of every entry (file order), main.cpp:326-328, and a flag "holds a non-ACGT
    // letter before its last base" -- kept in the chunks the parser threads produced (concatenating
    // a few hundred MB on one thread cost more than parsing them on thirty-two)
    std::vector<std::vector<uint64_t>> codes;
    std::vector<std::vector<int8_t>> inv_mid;
    size_t n = 0;
    int load_state_invalid = 0;   // validity of the KmerCode object the load pass leaves behind
    bool valid = false;
};

struct rc_ctx_full : rc_ctx {
    rc_dump_cache dump;
};

// pinned host buffer, grow-only
struct rc_hbuf {
    void *p = nullptr;
    size_t bytes = 0;
};

// one batch in flight on the asynchronous host-buffer path
struct rc_slot {
    bool busy = false;
    rc_batch b;                      // the caller's descriptor (its buffers stay valid until rc_wait)
    size_t total_reads = 0, bytes1 = 0, bytes2 = 0;
    bool seq_pinned = false, res_pinned = false;  // the caller's buffers are page-locked: DMA straight from / to them
    rc_hbuf p_seq, p_qual, p_off, p_res;           // pinned staging (seq/qual only when the caller's are pageable)
    rc_dbuf d_seq, d_qual, d_off, d_res;
    hipEvent_t e_h2d = nullptr, e_k = nullptr, e_done = nullptr;
    // the packed boundary (rc_submit_packed): the caller's descriptor, the packed arena / exceptions / fix list in HBM,
    // pinned staging for descriptor arrays that are not page-locked, and the fix count's landing place
    rc_packed_batch *pb = nullptr;
    rc_resident_batch *rb = nullptr;  // rc_submit_resident: same slot state, the arena copied from the counter's kept arenas
    rc_dbuf d_packed, d_exc, d_fix;
    rc_hbuf p_in, p_fix, p_nfix;
    uint32_t fix_room = 0;
    bool fix_pinned = false;
};


extern "C" {  // (defined inside the units' extern "C" blocks)
// rc_api_batch.hip
// qual_split / qual_base2 (quality-bit mode only): arena bytes from qual_split on have their bits at byte qual_base2 of d_qual;
// qual_bits: -1 = as rc_set_quality_bits says, 0 / 1 = this batch's quality arena holds bytes / bits (the packed boundary)
int rc_correct_device_impl(rc_ctx *ctx, const rc_device_batch *b, uint32_t qual_split, uint32_t qual_base2, int qual_bits = -1);
int rc_hbuf_reserve(rc_ctx *ctx, rc_hbuf *h, size_t bytes);
bool rc_is_pinned(const void *p, size_t bytes);
int rc_slots_init(rc_ctx *ctx);
// the context slot `slot` runs in: ctx itself (slot 0, a lane, or RC_SLOT_LANES=0), else its lane -- created if `create`, and
// brought up to date with ctx's table / parameters / kept arenas if `refresh` (submits); nullptr + error text on failure
rc_ctx *rc_slot_lane(rc_ctx *ctx, int slot, bool create, bool refresh);
// the lane's error text and summary counters seen through the parent
void rc_lane_error(rc_ctx *ctx, const rc_ctx *lane);
}
