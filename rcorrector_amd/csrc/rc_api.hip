// rc_api.hip -- the C ABI of librcorrector_amd.so (include/rcorrector_amd.h): context, table
// load, run-parameter estimation and the batch entry points.  Host code only drives HIP; every
// per-read computation happens in the kernels of rc_table.hip / rc_correct.hip.  There is no CPU
// fallback anywhere in this library.
#include "rc_api_internal.h"
#include <mutex>

static thread_local char g_create_err[512];

void rc_set_error(rc_ctx *ctx, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx ? ctx->err : g_create_err, 512, fmt, ap);
    va_end(ap);
}

int rc_dbuf_reserve(rc_ctx *ctx, rc_dbuf *b, size_t bytes)
{
    if (bytes <= b->bytes) return RC_OK;
    if (b->p) {
        RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(b->p);
        b->p = nullptr;
        b->bytes = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    RC_CHECK_HIP(ctx, hipMalloc(&b->p, want));
    b->bytes = want;
    return RC_OK;
}

void rc_table_release(rc_ctx *ctx)
{
    // a slot lane may still be probing the table it borrowed (streams of its own): wait for it and take the loan back before
    // the buckets go -- the next batch of that slot gets the new table with its refresh (rc_slot_lane)
    if (ctx->d_buckets && !ctx->buckets_borrowed) {
        for (rc_ctx *ln : ctx->lane) {
            if (!ln || ln->d_buckets != ctx->d_buckets) continue;
            (void)hipStreamSynchronize(ln->stream);
            ln->d_buckets = nullptr;
            ln->buckets_borrowed = false;
            ln->n_entries = 0;
        }
    }
    if (ctx->d_buckets && !ctx->buckets_borrowed) (void)hipFree(reinterpret_cast<char *>(ctx->d_buckets) - RC_TABLE_PREFIX_BYTES);
    ctx->d_buckets = nullptr;
    ctx->buckets_borrowed = false;
    if (ctx->counted_codes) (void)hipFree(ctx->counted_codes);
    ctx->counted_codes = nullptr;
    ctx->counted_n = 0;
    ctx->filter_words = 0;
    ctx->table_bytes = 0;
    ctx->n_entries = 0;
}

rc_table_view rc_view(const rc_ctx *ctx)
{
    rc_table_view v;
    v.buckets = ctx->d_buckets;
    v.nb_home = ctx->nb_home;
    v.nbuckets_alloc = ctx->nb_alloc;
    v.layout = ctx->layout;
    v.ext = ctx->ext;
    v.k = ctx->k;
    v.filter_words = ctx->filter_words;
    v.filter = ctx->filter_words ? ctx->d_buckets + ctx->table_bytes / 4 : nullptr;
    v.filter_kind = ctx->filter_kind;
    v.filter_all = ctx->filter_all;
    return v;
}

void rc_timer_begin(rc_ctx *ctx)
{
    if (ctx->profile) (void)hipEventRecord(ctx->ev0, ctx->stream);
}

void rc_timer_end(rc_ctx *ctx, int which)
{
    if (!ctx->profile) return;
    (void)hipEventRecord(ctx->ev1, ctx->stream);
    (void)hipEventSynchronize(ctx->ev1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->timers[which].ms += ms;
    ctx->timers[which].launches += 1;
}

void rc_lane_error(rc_ctx *ctx, const rc_ctx *lane)
{
    if (ctx != lane) snprintf(ctx->err, sizeof ctx->err, "%s", lane->err);
}

extern "C" {

rc_ctx *rc_slot_lane(rc_ctx *ctx, int slot, bool create, bool refresh)
{
    if (slot <= 0 || slot >= RC_MAX_SLOTS || ctx->is_lane) return ctx;
    if (!create) return ctx->slot_home[slot] ? ctx->slot_home[slot] : ctx;  // a wait: where that slot's batch went
    if (!ctx->env_slot_lanes) {
        ctx->slot_home[slot] = ctx;
        return ctx;
    }
    rc_ctx *&ln = ctx->lane[slot];
    if (!ln) {
        // lanes only overlap when their streams land on different hardware queues: say so, once, if the runtime has few
        static std::once_flag warned;
        std::call_once(warned, [] {
            const char *q = getenv("GPU_MAX_HW_QUEUES");
            if ((!q || atoi(q) < 8) && !getenv("RC_QUIET"))
                fprintf(stderr, "[rcorrector_amd] note: batches in slots > 0 run in lanes with streams of their own, but GPU_MAX_HW_QUEUES is %s: the HIP "
                                "runtime will put them on 4 hardware queues and some will run one after the other. Export GPU_MAX_HW_QUEUES=16 (or call "
                                "rc_runtime_prepare) before the process first touches HIP, or switch the lanes off (rc_set_slot_lanes / RC_SLOT_LANES=0).\n",
                        q ? q : "not set");
        });
        rc_config cfg = {ctx->device, ctx->k, ctx->P.max_fix_per_k};
        char err[256];
        ln = rc_create(&cfg, err, sizeof err);
        if (!ln) {
            rc_set_error(ctx, "slot %d: %s", slot, err);
            return nullptr;
        }
        ln->is_lane = true;
    }
    if (refresh) {  // plain assignments: the table (borrowed, as rc_table_share lends it), the parameters, the mode, the kept arenas
        ln->d_buckets = ctx->d_buckets;
        ln->buckets_borrowed = true;
        ln->nb_home = ctx->nb_home;
        ln->layout = ctx->layout;
        ln->ext = ctx->ext;
        ln->nb_alloc = ctx->nb_alloc;
        ln->n_entries = ctx->n_entries;
        ln->table_bytes = ctx->table_bytes;
        ln->filter_words = ctx->filter_words;
        ln->filter_kind = ctx->filter_kind;
        ln->filter_all = ctx->filter_all;
        ln->P = ctx->P;
        ln->params_set = ctx->params_set;
        ln->qual_bits = ctx->qual_bits;
        ln->kept_arenas = ctx->kept_arenas;  // (descriptors only: the chunks stay the parent's)
        ln->profile = ctx->profile;          // measurement follows the batch into its lane (rc_profile_get adds the lanes up)
        ln->phase_prof = ctx->phase_prof;
        ln->phase_prof_print = ctx->phase_prof_print;
        ln->rounds_out = ctx->rounds_out;
    }
    ctx->slot_home[slot] = ln;
    return ln;
}

int rc_runtime_prepare(int hw_queues)
{
    if (hw_queues < 1) return -1;
    if (getenv("GPU_MAX_HW_QUEUES")) return 0;
    char v[16];
    snprintf(v, sizeof v, "%d", hw_queues);
    return setenv("GPU_MAX_HW_QUEUES", v, 0) == 0 ? 1 : -1;
}

rc_ctx *rc_create(const rc_config *cfg, char *errbuf, size_t errbuf_len)
{
    auto fail = [&](const char *msg) -> rc_ctx * {
        if (errbuf && errbuf_len) snprintf(errbuf, errbuf_len, "%s", msg);
        return nullptr;
    };
    if (!cfg) return fail("rc_create: null config");
    // (Slot lanes want their streams on hardware queues of their own -- GPU_MAX_HW_QUEUES, the host's to export before HIP
    // starts: rc_runtime_prepare / rcorrector_amd.h.  The library does not touch its host's environment from here.)
    if (cfg->k < 1 || cfg->k > 32) return fail("rc_create: k must be in 1..32 (run_rcorrector.pl:225-228)");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail("rc_create: no HIP device available (this library has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail("rc_create: device ordinal out of range");
    if (hipSetDevice(cfg->device) != hipSuccess) return fail("rc_create: hipSetDevice failed");
    rc_ctx_full *ctx = new (std::nothrow) rc_ctx_full();
    if (!ctx) return fail("rc_create: out of memory");
    ctx->err[0] = 0;
    ctx->device = cfg->device;
    ctx->k = cfg->k;
    ctx->P.k = cfg->k;
    ctx->P.max_fix_per_k = cfg->max_fix_per_k > 0 ? cfg->max_fix_per_k : 4;
    ctx->P.error_rate = 0.01;
    ctx->P.bad_qual = 0;
    memset(ctx->P.bs, 0, sizeof ctx->P.bs);
    ctx->P.bs_ext = nullptr;
    ctx->P.bound_small = nullptr;
    ctx->P.flags = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipMalloc(&ctx->work.p, RC_WORK_BYTES) != hipSuccess) {
        delete ctx;
        return fail("rc_create: could not create stream/events");
    }
    ctx->work.bytes = RC_WORK_BYTES;
    (void)hipMemset(ctx->work.p, 0, RC_WORK_BYTES);
    if (const char *e = getenv("RC_PHASE_PROF")) ctx->phase_prof = ctx->phase_prof_print = atoi(e) != 0;
    if (const char *e = getenv("RC_TABLE_LOAD")) {
        ctx->table_load = ctx->table_load_packed = atof(e);
        ctx->table_load_set = true;
    }  // tuning knob
    if (const char *e = getenv("RC_TABLE_LAYOUT")) ctx->layout_pref = strcmp(e, "wide") != 0;         // dev: A/B the slot layouts
    ctx->env_k2_wave_per_read = getenv("RC_K2_WAVE_PER_READ") != nullptr;  // dev: force the wave-per-read threshold kernel
    ctx->env_no_classify = getenv("RC_NO_CLASSIFY") != nullptr;            // dev: every read goes through k_correct
    ctx->env_timing = getenv("RC_TIMING") != nullptr;
    ctx->env_no_fuse = getenv("RC_NO_FUSE") != nullptr;
    if (const char *dd = getenv("RC_FUSED_DEDUP")) ctx->env_dedup = atoi(dd) != 0 ? 1 : 0;
    if (const char *xo = getenv("RC_FUSED_XCD")) ctx->env_fused_xcd = atoi(xo) != 0;
    if (const char *qd = getenv("RC_PROBE_QUAD")) ctx->env_quad = atoi(qd) != 0 ? 1 : 0;
    if (const char *kl = getenv("RC_K3_LOCAL")) ctx->env_k3_local = atoi(kl) != 0;
    if (const char *wt = getenv("RC_FUSED_WAVE_TILES")) ctx->env_wave_tiles = atoi(wt) != 0;
    ctx->env_no_tier = getenv("RC_NO_TIER") != nullptr;
    if (const char *e = getenv("RC_FORCE_EC")) {
        const int v = atoi(e);
        if (v == 9 || v == 10) ctx->env_force_ec = v;
    }
    ctx->env_k3_generic = getenv("RC_K3_GENERIC") != nullptr;
    ctx->env_no_single = getenv("RC_NO_SINGLE") != nullptr;
    ctx->env_no_alt = getenv("RC_NO_ALT") != nullptr;  // dev / tests: no alternative chains in the search's speculation rounds
    if (const char *e = getenv("RC_LOCALITY")) ctx->locality_mode = !strcmp(e, "force") ? 1 : (!strcmp(e, "off") ? -1 : 0);  // tests / A-B
    if (const char *e = getenv("RC_K3_GRID_WAVES")) ctx->env_k3_grid_waves = atoi(e);
    if (const char *e = getenv("RC_SLOT_LANES")) ctx->env_slot_lanes = atoi(e) != 0;
    return ctx;
}

void rc_destroy(rc_ctx *c)
{
    if (!c) return;
    rc_ctx_full *ctx = static_cast<rc_ctx_full *>(c);
    for (rc_ctx *&ln : ctx->lane) {  // (they borrow this context's table and arenas: they go first)
        if (ln) {
            ln->kept_arenas.clear();
            rc_destroy(ln);
        }
        ln = nullptr;
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    rc_dbuf *bufs[] = {&ctx->counts, &ctx->strong, &ctx->info, &ctx->stack, &ctx->work,
                       &ctx->h_seq, &ctx->h_qual, &ctx->h_off, &ctx->h_res, &ctx->trace, &ctx->cls, &ctx->worklist, &ctx->sel_tmp,
                       &ctx->loc_a, &ctx->loc_list, &ctx->loc_span, &ctx->tier_flag, &ctx->tier_list, &ctx->cand, &ctx->single_list, &ctx->runs, &ctx->bs_dev};
    for (rc_dbuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    if (ctx->slots) {
        for (int i = 0; i < RC_MAX_SLOTS; ++i) {
            rc_slot &sl = ctx->slots[i];
            if (sl.e_done) (void)hipEventSynchronize(sl.e_done);
            rc_hbuf *hb[] = {&sl.p_seq, &sl.p_qual, &sl.p_off, &sl.p_res, &sl.p_in, &sl.p_fix, &sl.p_nfix};
            for (rc_hbuf *h : hb)
                if (h->p) (void)hipHostFree(h->p);
            rc_dbuf *db[] = {&sl.d_seq, &sl.d_qual, &sl.d_off, &sl.d_res, &sl.d_packed, &sl.d_exc, &sl.d_fix};
            for (rc_dbuf *d : db)
                if (d->p) (void)hipFree(d->p);
            hipEvent_t ev[] = {sl.e_h2d, sl.e_k, sl.e_done};
            for (hipEvent_t e : ev)
                if (e) (void)hipEventDestroy(e);
        }
        delete[] ctx->slots;
    }
    if (ctx->s_h2d) (void)hipStreamDestroy(ctx->s_h2d);
    if (ctx->s_d2h) (void)hipStreamDestroy(ctx->s_d2h);
    for (auto &a : ctx->cnt_chunks)
        if (a.p) (void)hipFree(a.p);
    rc_kept_release(ctx);
    rc_table_release(ctx);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *rc_last_error(const rc_ctx *ctx) { return ctx ? ctx->err : g_create_err; }

int rc_set_slot_lanes(rc_ctx *ctx, int on)
{
    if (!ctx) return RC_ERR_ARG;
    ctx->env_slot_lanes = on != 0;
    return RC_OK;
}

int rc_device_numa_node(const rc_ctx *ctx)
{
    if (!ctx) return -1;
    char bus[64];
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, ctx->device) != hipSuccess) return -1;
    for (char *p = bus; *p; ++p)
        if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');  // sysfs spells the address in lower case
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *fp = fopen(path, "r");
    if (!fp) return -1;
    int node = -1;
    if (fscanf(fp, "%d", &node) != 1) node = -1;
    fclose(fp);
    return node;
}

int rc_device_memory(rc_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes)
{
    if (!ctx) return RC_ERR_ARG;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    size_t f = 0, t = 0;
    RC_CHECK_HIP(ctx, hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (uint64_t)f;
    if (total_bytes) *total_bytes = (uint64_t)t;
    return RC_OK;
}

// ---- measurement -----------------------------------------------------------------------------
int rc_profile_enable(rc_ctx *ctx, int on)
{
    if (!ctx) return RC_ERR_ARG;
    ctx->profile = on != 0;
    if (!ctx->phase_prof_print) ctx->phase_prof = on == 2;
    for (rc_ctx *ln : ctx->lane) {  // (and a lane that exists already; new ones copy the flags when they take a batch)
        if (!ln) continue;
        ln->profile = ctx->profile;
        ln->phase_prof = ctx->phase_prof;
    }
    return RC_OK;
}

int rc_profile_get(rc_ctx *ctx, int kernel, double *total_ms, uint64_t *launches)
{
    if (!ctx || kernel < 0 || kernel >= RC_T_COUNT) return RC_ERR_ARG;
    double ms = ctx->timers[kernel].ms;
    uint64_t n = ctx->timers[kernel].launches;
    for (rc_ctx *ln : ctx->lane) {  // (the batches of slots > 0 ran in lane contexts)
        if (!ln) continue;
        ms += ln->timers[kernel].ms;
        n += ln->timers[kernel].launches;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    return RC_OK;
}

int rc_profile_reset(rc_ctx *ctx)
{
    if (!ctx) return RC_ERR_ARG;
    for (auto &t : ctx->timers) t = rc_kernel_timer();
    ctx->k3_listed = ctx->k3_rounds = ctx->k3_requests = 0;
    for (rc_ctx *ln : ctx->lane) {
        if (!ln) continue;
        for (auto &t : ln->timers) t = rc_kernel_timer();
        ln->k3_listed = ln->k3_rounds = ln->k3_requests = 0;
    }
    return RC_OK;
}

int rc_profile_correct_counters(rc_ctx *ctx, uint64_t *reads_listed, uint64_t *gather_rounds, uint64_t *bucket_requests)
{
    if (!ctx) return RC_ERR_ARG;
    uint64_t a = ctx->k3_listed, b = ctx->k3_rounds, c = ctx->k3_requests;
    for (rc_ctx *ln : ctx->lane) {
        if (!ln) continue;
        a += ln->k3_listed;
        b += ln->k3_rounds;
        c += ln->k3_requests;
    }
    if (reads_listed) *reads_listed = a;
    if (gather_rounds) *gather_rounds = b;
    if (bucket_requests) *bucket_requests = c;
    return RC_OK;
}

int rc_profile_read_rounds(rc_ctx *ctx, int32_t *d_rounds)
{
    if (!ctx) return RC_ERR_ARG;
    ctx->rounds_out = d_rounds;
    for (rc_ctx *ln : ctx->lane)
        if (ln) ln->rounds_out = d_rounds;
    return RC_OK;
}

int rc_selftest_get_bound(rc_ctx *ctx, const int32_t *c, size_t n, double error_rate, int32_t *out_int, double *out_dbl)
{
    if (!ctx || (n && (!c || !out_int || !out_dbl))) return RC_ERR_ARG;
    if (n == 0) return RC_OK;
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    rc_dev_tmp b_c, b_i, b_d;
    RC_CHECK_HIP(ctx, b_c.alloc(n * 4));
    RC_CHECK_HIP(ctx, b_i.alloc(n * 4));
    RC_CHECK_HIP(ctx, b_d.alloc(n * 8));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(b_c.p, c, n * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = rc_launch_selftest_bound(ctx, b_c.as<int32_t>(), n, error_rate, b_i.as<int32_t>(), b_d.as<double>());
    if (rc) return rc;
    RC_CHECK_HIP(ctx, hipMemcpyAsync(out_int, b_i.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipMemcpyAsync(out_dbl, b_d.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

int rc_summary(const rc_ctx *c, uint64_t *total_reads, uint64_t *total_corrections)
{
    if (!c) return RC_ERR_ARG;
    rc_ctx *ctx = const_cast<rc_ctx *>(c);  // (reads device memory; the counters themselves do not change)
    unsigned long long v[2] = {0, 0};
    RC_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    RC_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RC_CHECK_HIP(ctx, hipMemcpy(v, (char *)ctx->work.p + RC_WORK_SUMMARY_OFF, sizeof v, hipMemcpyDeviceToHost));
    for (rc_ctx *ln : ctx->lane) {  // (the batches its slot lanes ran)
        if (!ln) continue;
        unsigned long long w[2] = {0, 0};
        RC_CHECK_HIP(ctx, hipStreamSynchronize(ln->stream));
        RC_CHECK_HIP(ctx, hipMemcpy(w, (char *)ln->work.p + RC_WORK_SUMMARY_OFF, sizeof w, hipMemcpyDeviceToHost));
        v[0] += w[0];
        v[1] += w[1];
    }
    if (total_reads) *total_reads = v[0];
    if (total_corrections) *total_corrections = v[1];
    return RC_OK;
}

}  // extern "C"
