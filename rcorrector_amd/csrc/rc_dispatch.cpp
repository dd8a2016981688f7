// rc_dispatch.cpp -- see rc_dispatch.h
#include "rc_dispatch.h"

#include <sys/stat.h>
#include <unistd.h>

// room for the substitutions of a batch of nbytes arena bytes: one per sixteen bases (0.5 % errors fill a twelfth of it, 5 %
// two thirds; the reference's bound, MAX_FIX_PER_K per k-mer window and more with -maxcorK, is about a third of the bases: a
// batch that needs more comes back with RC_STATUS_NOSPACE and is run again through the byte path).  The list is page-locked,
// and page-locking touches every page whether a fix ever lands there or not: at one entry per four bases it was 190 MB per
// job of a million 150-base reads, most of what a run locks and unlocks.  RC_FIX_CAP=<entries>: tests.
static size_t fix_list_room(size_t nbytes)
{
    static const char *e = getenv("RC_FIX_CAP");
    return e ? (size_t)atoll(e) : nbytes / 16 + 64;
}

// the reader of the one-pass ingest: host memory only (text blocks and their line index), so that it can run while the GPU
// runtime is still starting up
void Ingest::start()
{
    reader = std::thread([this]() {
        std::vector<ReadFile> &files = R.files, &mates = R.mates;
        for (size_t fi = 0; fi < files.size() && !stop; ++fi) {
            ReadFile &f = files[fi];
            Source own_a, own_b;
            if (!keep) {
                own_a.open(f.path);
                if (f.paired) own_b.open(mates[fi].path);
            }
            Source &src_a = keep ? f.src : own_a, &src_b = keep ? mates[fi].src : own_b;
            while (!stop) {
                std::unique_ptr<Retained> B;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (!spare.empty()) {
                        B = std::move(spare.back());
                        spare.pop_back();
                    }
                }
                if (!B) B.reset(new Retained);
                B->file = (int)fi;
                B->mode = f.paired ? 1 : (f.interleaved ? 2 : 0);
                B->fastq = f.fastq;
                B->lpr_a = f.fastq ? 4 : 2;
                B->lpr_b = f.paired ? (mates[fi].fastq ? 4 : 2) : B->lpr_a;
                const double tr0 = now_s();
                B->b.records = 0;
                if (f.paired) {
                    std::thread mate([&]() { take_records(src_b, batch_reads, B->lpr_b, B->b); });
                    take_records(src_a, batch_reads, B->lpr_a, B->a);
                    mate.join();
                    // (two passes: files that are not paired are the correction loop's to refuse, with the reference's message
                    // in the reference's place on stderr; the counter takes whatever reads there are)
                    if (keep && B->b.records != B->a.records) die("ERROR: The files are not paired!\n");
                } else {
                    take_records(src_a, batch_reads, B->lpr_a, B->a);
                }
                if (B->a.records == 0 && B->b.records == 0) break;
                if (keep && B->mode == 2 && (B->a.records & 1)) die("ERROR: interleaved file %s holds an odd number of reads\n", f.path.c_str());
                g_t_read += now_s() - tr0;
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || q.size() < depth; });
                q.emplace_back(std::move(B));
                cv.notify_all();
            }
            if (!keep) {
                own_a.close();
                if (f.paired) own_b.close();
            }
        }
        std::lock_guard<std::mutex> lk(mu);
        done = true;
        cv.notify_all();
    });
}

// a reader that ran ahead of a decision that then went the other way (Ingest::depth blocks at most): its blocks are dropped; the
// caller rewinds the sources
void Ingest::abort()
{
    {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
    }
    cv.notify_all();
    reader.join();
    q.clear();
}

void Ingest::consume(int64_t *stored)
{
    Run &R_ = R;
    rc_ctx *ctx = R_.ctx[0];
    std::vector<ReadFile> &files = R_.files, &mates = R_.mates;
    std::vector<std::unique_ptr<Retained>> &kept = R_.kept;
    {
        std::lock_guard<std::mutex> lk(mu);
        depth = 3;  // (a reader that ran ahead may hold more)
    }
    if (rc_table_count_keep(ctx, keep ? 1 : 0) || rc_table_count_begin(ctx)) die("rcorrector: %s\n", rc_last_error(ctx));
    // Several GPUs, one pass: the batches are dealt round-robin, each arena uploaded to the GPU that will correct it, and the
    // GPUs count together (rc_table_count_finish_sharded: every GPU scans its own reads, the key space is shared out; one
    // Store for all workers, main.cpp:294-308): the files are read once, and a batch is corrected where its bases already
    // are.  RC_COUNT_SHARDED=0: every arena goes to GPU 0 as well, which counts alone (the others rc_table_count_park).
    const int n_gpus = keep ? R_.gpus : 1;
    const bool sharded = n_gpus > 1 && !(getenv("RC_COUNT_SHARDED") && !strcmp(getenv("RC_COUNT_SHARDED"), "0"));
    for (int g = 1; g < n_gpus; ++g)
        if (rc_table_count_begin(R_.ctx[(size_t)g])) die("rcorrector: %s\n", rc_last_error(R_.ctx[(size_t)g]));
    PinBuf stage;  // the sequences of one file's share of a batch on their way to HBM
    std::vector<int> next_arena((size_t)n_gpus, 0);
    size_t n_batches = 0;
    for (;;) {
        std::unique_ptr<Retained> R;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return done || !q.empty(); });
            if (q.empty()) break;
            R = std::move(q.front());
            q.pop_front();
            cv.notify_all();
        }
        const double tp0 = now_s();
        R->gpu = (int)(n_batches++ % (size_t)n_gpus);
        for (int sd = 0; sd < (R->mode == 1 ? 2 : 1); ++sd) {
            if ((sd ? R->b : R->a).records == 0) continue;  // (keep = false: one mate's file ended before the other's)
            Arena A;  // (a view for index_arena / pack_sequences: the block is swapped in and out)
            A.lpr = sd ? R->lpr_b : R->lpr_a;
            A.blk.swap(sd ? R->b : R->a);
            A.off.swap(sd ? R->off_b : R->off_a);  // (its capacity, when the block is a recycled one)
            const uint64_t total = index_arena(A, sd ? mates[(size_t)R->file].path : files[(size_t)R->file].path);
            stage.need(total + 64);
            pack_sequences(A, stage.data());
            // (an arena without a byte is not kept: cannot happen, every record has at least its NUL)
            int idx;
            if (sharded) {  // to the GPU that will correct it, and nowhere else
                rc_ctx *cg = R_.ctx[(size_t)R->gpu];
                if (rc_table_count_add(cg, stage.data(), total)) die("rcorrector: %s\n", rc_last_error(cg));
                idx = next_arena[(size_t)R->gpu]++;
            } else {
                if (rc_table_count_add(ctx, stage.data(), total)) die("rcorrector: %s\n", rc_last_error(ctx));
                idx = next_arena[0]++;
                if (R->gpu != 0) {
                    rc_ctx *cg = R_.ctx[(size_t)R->gpu];
                    if (rc_table_count_add(cg, stage.data(), total)) die("rcorrector: %s\n", rc_last_error(cg));
                    idx = next_arena[(size_t)R->gpu]++;
                }
            }
            (sd ? R->arena_b : R->arena_a) = idx;
            (sd ? R->off_b : R->off_a).swap(A.off);
            A.blk.swap(sd ? R->b : R->a);
        }
        g_t_pack += now_s() - tp0;
        if (keep) {
            kept.emplace_back(std::move(R));
        } else {
            std::lock_guard<std::mutex> lk(mu);
            spare.emplace_back(std::move(R));
        }
    }
    reader.join();
    stamp(keep ? "inputs read, indexed and uploaded" : "inputs read and uploaded for the k-mer count");
    if (sharded) {
        if (rc_table_count_finish_sharded(R_.ctx.data(), n_gpus, 2, stored)) die("rcorrector: %s\n", rc_last_error(ctx));
    } else {
        for (int g = 1; g < n_gpus; ++g)
            if (rc_table_count_park(R_.ctx[(size_t)g])) die("rcorrector: %s\n", rc_last_error(R_.ctx[(size_t)g]));
        if (rc_table_count_finish(ctx, 2, stored)) die("rcorrector: %s\n", rc_last_error(ctx));
    }
    stamp("k-mers counted, table built");
}

void ingest_resident(Run &R, size_t batch_reads, int64_t *stored, bool keep)
{
    Ingest I(R, batch_reads, keep);
    I.start();
    I.consume(stored);
}

HeadStats head_stats(const Run &R)
{
    const std::vector<ReadFile> &files = R.files;
    size_t head_nl = 0, head_last = 0, head_seq_len = 0;
    if (!files.empty() && !g_verbose && !files[0].src.is_gz && files[0].src.seekable) {
        const ReadFile &f = files[0];
        const int lpr = f.fastq ? 4 : 2;
        const char *h = f.src.left.p;
        size_t l1 = 0;
        for (size_t i = 0; i < f.src.left_len; ++i)
            if (h[i] == '\n') {
                ++head_nl;
                if (head_nl == 1) l1 = i;
                if (head_nl == 2) head_seq_len = i - l1 - 1;
                if (head_nl % (size_t)lpr == 0) head_last = i + 1;
            }
    }
    HeadStats H;
    H.nl = head_nl;
    H.last = head_last;
    H.seq_len = head_seq_len;
    return H;
}

void warm_buffers(Run &R, const HeadStats &H)
{
    const std::vector<ReadFile> &files = R.files;
    const size_t batch_reads = R.batch_reads, max_in_flight = R.max_in_flight;
    const bool resident = R.resident;
    std::vector<std::shared_ptr<Job>> &warm_jobs = R.warm_jobs;
    const size_t head_nl = H.nl, head_last = H.last, head_seq_len = H.seq_len;
    if (files.empty() || g_verbose || files[0].src.is_gz || !files[0].src.seekable) return;
    const ReadFile &f = files[0];
    const int lpr = f.fastq ? 4 : 2;
    const size_t nl = head_nl, last = head_last, seq_len = head_seq_len;
    if (last == 0 || seq_len == 0) return;
    const double rec_bytes = (double)last / (double)(nl / (size_t)lpr);
    struct stat st;
    if (stat(f.path.c_str(), &st) != 0) return;
    const double file_recs = (double)st.st_size / rec_bytes;
    size_t recs = batch_reads;
    if (f.interleaved) recs = batch_reads;  // (a batch of an interleaved file holds batch_reads records as well)
    if ((double)recs > file_recs * 1.02 + 16) recs = (size_t)(file_recs * 1.02) + 16;
    size_t njobs = (size_t)(file_recs / (double)recs) + 1;
    if (njobs > max_in_flight) njobs = max_in_flight;
    const size_t text_bytes = (size_t)((double)recs * rec_bytes * 1.04) + ((size_t)1 << 20);
    const size_t arena_bytes = (size_t)((double)recs * (double)(seq_len + 1) * 1.02) + 4096;
    const size_t S = std::max<size_t>(1, std::min<size_t>((size_t)g_threads, (recs + 8191) / 8192));
    const size_t out_slice = (size_t)(((double)recs / (double)S + 1.0) * (rec_bytes + 48.0));
    for (size_t jn = 0; jn < njobs; ++jn) {
        auto j = std::make_shared<Job>();
        const int sides = f.paired ? 2 : 1;
        for (int sd = 0; sd < sides; ++sd) {
            Arena &A = sd ? j->b : j->a;
            if (!resident) {
                A.blk.text.need(text_bytes);
                A.blk.line.reserve(recs * (size_t)lpr + 8);
                A.off.reserve(recs + 1);
                A.seq.need(arena_bytes);   // (page-locked here: rc_host_register)
                A.qual.need(arena_bytes);
            }
            std::vector<OutBuf> &o = sd ? j->o2 : j->o1;
            o.resize(S);
            for (auto &v : o) v.reserve(out_slice);
        }
        // touch what malloc handed out untouched (the arenas were touched by the registration)
        g_pool.run(16, [&](size_t t) {
            for (int sd = 0; sd < sides; ++sd) {
                Arena &A = sd ? j->b : j->a;
                const size_t lo = text_bytes * t / 16, hi = text_bytes * (t + 1) / 16;
                if (!resident) memset(A.blk.text.p + lo, 0, hi - lo);
                std::vector<OutBuf> &o = sd ? j->o2 : j->o1;
                for (size_t s2 = t; s2 < S; s2 += 16) {
                    o[s2].resize(out_slice);
                    memset(o[s2].data(), 0, out_slice);
                    o[s2].clear();
                }
            }
        });
        const size_t total = (size_t)sides * recs;
        if (resident) {  // what a resident batch sends and receives (page-locked)
            const size_t nb = (size_t)sides * arena_bytes;
            (void)j->pk_carve(total, nb, fix_list_room(nb));
        }
        j->ret.reserve(total);
        j->l.reserve(total);
        j->m.reserve(total);
        j->h.reserve(total);
        warm_jobs.push_back(j);
    }
}

// the output records of a finished batch, formatted (and deflated for .gz outputs) in slices by
// the worker that ran it; the writer thread only writes
static void format_job(Run &R, Job &J)
{
    std::vector<ReadFile> &files = R.files, &mates = R.mates;
    const Job *j = &J;
    const size_t n = j->a.n();
    ReadFile &f = files[(size_t)j->file];
    const bool alternate = j->mode == 1 && g_stdout;  // main.cpp:487-495
    // compression is a property of each output file (Reads::AddReadFile picks it per input name):
    // `-p a.fq.gz b.fq` writes a gzip stream for the first mates and plain text for the second
    const bool gz1 = f.out_gz && !g_stdout, gz2 = j->mode == 1 && mates[(size_t)j->file].out_gz && !g_stdout;
    // (slices of plain output are copied by at most g_threads threads -- memory-bound, more get in each other's way --
    // slices that are deflated by as many as the pool has: that is arithmetic)
    const size_t width = (gz1 || gz2) ? std::max<size_t>((size_t)g_threads, g_deflate_threads) : (size_t)g_threads;
    const size_t S = std::max<size_t>(1, std::min<size_t>(width, (n + 8191) / 8192));
    std::vector<OutBuf> &o1 = J.o1, &o2 = J.o2;
    o1.resize(S);
    o2.resize(S);
    for (auto &v : o1) v.clear();
    for (auto &v : o2) v.clear();
    std::vector<uint64_t> cor(S, 0);  // (the writer thread is the pipeline's narrow place: it only writes)
    auto fmt = [&](size_t lo, size_t hi) {
        for (size_t s = lo; s < hi; ++s) {
            const size_t r0 = n * s / S, r1 = n * (s + 1) / S;
            uint64_t c = 0;
            for (size_t r = r0; r < r1; ++r) {
                if (j->ret[r] > 0) c += (uint64_t)j->ret[r];
                if (j->mode == 1 && j->ret[n + r] > 0) c += (uint64_t)j->ret[n + r];
            }
            cor[s] = c;
            o1[s].reserve((r1 - r0) * 300);
            for (size_t r = r0; r < r1; ++r) {
                put_record(o1[s], j->a, r, j->fastq, j->ret[r], j->l[r], j->m[r], j->h[r]);
                if (alternate) put_record(o1[s], j->b, r, j->fastq, j->ret[n + r], j->l[n + r], j->m[n + r], j->h[n + r]);
            }
            if (j->mode == 1 && !alternate) {
                o2[s].reserve((r1 - r0) * 300);
                for (size_t r = r0; r < r1; ++r)
                    put_record(o2[s], j->b, r, j->fastq, j->ret[n + r], j->l[n + r], j->m[n + r], j->h[n + r]);
            }
        }
    };
    g_pool.run(S, [&](size_t s) { fmt(s, s + 1); });
    J.cor_bases = 0;
    for (uint64_t c : cor) J.cor_bases += c;
    if (gz1 || gz2) {  // deflate every slice into its own gzip member, in parallel
        std::vector<OutBuf> z1(S), z2(S);
        g_pool.run(S, [&](size_t s) {
            if (gz1 && !o1[s].empty()) gzip_member(o1[s], z1[s]);
            if (gz2 && !o2[s].empty()) gzip_member(o2[s], z2[s]);
        });
        if (gz1) o1.swap(z1);
        if (gz2) o2.swap(z2);
    }
}

// (Run::lane_limit; called with R.mu held) the GPU is what this run waits for: as many batches in flight as there are workers, each in its slot lane
static void raise_lane_limit(Run &R, const char *why)
{
    if (!R.adaptive || R.lane_limit >= R.inflight) return;
    R.lane_limit = R.inflight;  // (all the way: 25 M reads of the stress preset, loop 3.3-3.4 s one step per three batches, 2.3 s with four from the start)
    if (!getenv("RC_SLOT_LANES"))
        for (size_t g = 0; g < R.ctx.size(); ++g) {  // (between two submits of that GPU)
            std::lock_guard<std::mutex> sk(R.submit_mu[g]);
            rc_set_slot_lanes(R.ctx[g], 1);
        }
    if (g_timing) fprintf(stderr, "[rc timing] %s: %d batches in flight per GPU from now on\n", why, R.lane_limit);
}

static void worker_body(Run &R, int wk)
{
    std::vector<ReadFile> &files = R.files, &mates = R.mates;
    std::vector<rc_ctx *> &ctx = R.ctx;
    std::mutex *submit_mu = R.submit_mu.get();
    std::mutex &mu = R.mu;
    std::condition_variable &cv = R.cv;
    std::deque<std::shared_ptr<Job>> &q = R.q;
    const bool &closing = R.closing;
    const int gpus = R.gpus;
    const bool numa_on = R.numa_on, shared_gpu = R.shared_gpu;
    const char bad_q = R.bad_q;
    const int g = wk % gpus, slot = wk / gpus;
    if (numa_on && gpus > 1 && !shared_gpu) {
        const int node = rc_device_numa_node(ctx[g]);
        if (node >= 0) bind_to_numa_node(node);
    }
    for (;;) {
        std::shared_ptr<Job> j;
        {
            std::unique_lock<std::mutex> lk(mu);
            const double tw = now_s();
            // (a resident batch belongs to the GPU that holds its bases; the others go to whichever context is free)
            auto mine = [&]() {
                for (auto it = q.begin(); it != q.end(); ++it)
                    if ((*it)->gpu < 0 || (*it)->gpu == g) return it;
                return q.end();
            };
            cv.wait(lk, [&] { return closing || (slot < R.lane_limit && R.active[(size_t)g] < R.lane_limit && mine() != q.end()); });
            g_w_worker += now_s() - tw;
            auto it = mine();
            if (it == q.end() || slot >= R.lane_limit || R.active[(size_t)g] >= R.lane_limit) {
                if (closing) return;
                continue;
            }
            j = *it;
            q.erase(it);
            ++R.active[(size_t)g];
        }
        const double tp0 = now_s();
        const size_t n = j->a.n();
        const size_t total = j->mode == 1 ? 2 * n : n;
        j->ret.assign(total, 0);
        j->l.assign(total, 0);
        j->m.assign(total, 0);
        j->h.assign(total, 0);
        bool resident_done = false;
        int rrc = 0;
        double tq1 = tp0;
        if (j->resident) {
            // the reads are in HBM since they were counted: offsets and quality bits go down, the results and the
            // substitutions come back and are applied to the sequence lines of the text
            Job &J = *j;
            const size_t bytes1 = J.a.off[n], bytes2 = J.mode == 1 ? J.b.off[n] : 0, nbytes = bytes1 + bytes2;
            const size_t cap = fix_list_room(nbytes);
            const Job::PkView pk = J.pk_carve(total, nbytes, cap);
            uint32_t *off = pk.off;
            memcpy(off, J.a.off.data(), (n + 1) * 4);
            if (J.mode == 1)
                for (size_t r = 0; r <= n; ++r) off[n + r] = (uint32_t)bytes1 + J.b.off[r];
            bool bits_ok = true;
            if (J.fastq) {
                QualView V{{&J.a, J.mode == 1 ? &J.b : &J.a}, J.mode == 1 ? bytes1 : nbytes, nbytes};
                const size_t Q = std::max<size_t>(1, std::min<size_t>((size_t)g_threads, nbytes / 65536 + 1));
                std::vector<char> okv(Q, 1);
                g_pool.run(Q, [&](size_t t) {
                    const size_t lo = (nbytes * t / Q) & ~(size_t)7, hi = t + 1 == Q ? nbytes : ((nbytes * (t + 1) / Q) & ~(size_t)7);
                    if (lo < hi) okv[t] = pack_quality_bits_from_text(V, bad_q, lo, hi, pk.qbits) ? 1 : 0;
                });
                for (char c : okv) bits_ok = bits_ok && c;
            }
            tq1 = now_s();
            if (bits_ok) {
                rc_resident_batch rb;
                memset(&rb, 0, sizeof rb);
                rb.mode = J.mode;
                rb.n = n;
                rb.arena_a = J.arena_a;
                rb.bytes_a = bytes1;
                rb.arena_b = J.arena_b;
                rb.bytes_b = bytes2;
                rb.off = off;
                rb.qual_bits = J.fastq ? (const uint8_t *)pk.qbits : nullptr;
                rb.ret = J.ret.data();
                rb.l = J.l.data();
                rb.m = J.m.data();
                rb.h = J.h.data();
                rb.fix_pos = pk.fix_pos;
                rb.fix_chr = pk.fix_chr;
                rb.fix_cap = cap;
                {
                    std::lock_guard<std::mutex> lk(submit_mu[(size_t)g]);
                    rrc = rc_submit_resident(ctx[g], &rb, slot);
                }
                if (!rrc) rrc = rc_wait_resident(ctx[g], slot);
                // more substitutions than the list has room for (a heavily corrected tail batch: the list is sized for one fix
                // per four bases, -maxcorK allows more): the text is untouched, the batch goes through the byte path below
                const bool overflow = rrc == RC_STATUS_NOSPACE;
                if (overflow) rrc = 0;
                if (!rrc && !overflow && rb.n_fix) {  // positions are distinct: any number of threads
                    const size_t F = std::max<size_t>(1, std::min<size_t>((size_t)g_threads, rb.n_fix / 16384 + 1));
                    g_pool.run(F, [&](size_t t) {
                        apply_fixes_to_text(J.a, J.mode == 1 ? &J.b : nullptr, bytes1, rb.fix_pos, rb.fix_chr, rb.n_fix * t / F, rb.n_fix * (t + 1) / F);
                    });
                }
                resident_done = !overflow;
            }
        }
        if (!resident_done) {
            pack_arena(j->a, files[(size_t)j->file].path);
            if (j->mode == 1) pack_arena(j->b, mates[(size_t)j->file].path);
        }
        const double tp1 = resident_done ? tq1 : now_s();
        rc_batch rb;
        memset(&rb, 0, sizeof rb);
        rb.mode = j->mode;
        rb.n = n;
        rb.seq = j->a.seq.data();
        rb.qual = j->a.qual.data();
        rb.off = j->a.off.data();
        if (j->mode == 1) {
            rb.seq2 = j->b.seq.data();
            rb.qual2 = j->b.qual.data();
            rb.off2 = j->b.off.data();
        }
        rb.ret = j->ret.data();
        rb.l = j->l.data();
        rb.m = j->m.data();
        rb.h = j->h.data();
        int rc;
        const double tg0 = resident_done ? tq1 : now_s();
        if (resident_done) {
            rc = rrc;
        } else if (g_verbose) {
            const size_t nbytes = (size_t)j->a.off[n] + (j->mode == 1 ? (size_t)j->b.off[n] : 0);
            j->tr_before.assign(nbytes, 0);
            j->tr_after.assign(nbytes, 0);
            j->tr_flags.assign(total, 0);
            j->tr_niter.assign(total, 0);
            j->tr_iter.assign(total * (size_t)g_trace_iter * RC_TRACE_ITER_WORDS, 0);
            rc_trace tr;
            tr.max_iter = g_trace_iter;
            tr.counts_before = j->tr_before.data();
            tr.counts_after = j->tr_after.data();
            tr.flags = j->tr_flags.data();
            tr.n_iter = j->tr_niter.data();
            tr.iter = j->tr_iter.data();
            std::lock_guard<std::mutex> lk(submit_mu[(size_t)g]);
            rc = rc_correct_batch_traced(ctx[g], &rb, &tr);
        } else if (g_packed && [&]() {
                       // One bit per quality cannot say "this read has no quality string" (qual[0] == 0: an empty
                       // quality line in a FASTQ file; ErrorCorrection.cpp:1316 asks): such a batch takes the bytes
                       if (!j->fastq) return true;
                       for (int sd = 0; sd < (j->mode == 1 ? 2 : 1); ++sd) {
                           const Arena &A = sd ? j->b : j->a;
                           for (size_t r = 0; r < A.n(); ++r)
                               if (A.off[r + 1] - A.off[r] > 1 && A.qual.data()[A.off[r]] == 0) return false;
                       }
                       return true;
                   }()) {
            // the packed boundary: the arenas stay here; 2-bit codes, quality bits and the letters outside ACGT go
            // down, the substitutions come back as a list and are applied to the arenas in front of the formatter
            Job &J = *j;
            const size_t bytes1 = J.a.off[n], bytes2 = J.mode == 1 ? J.b.off[n] : 0, nbytes = bytes1 + bytes2;
            const size_t n_words = (nbytes + 15) / 16, cap = fix_list_room(nbytes);
            const Job::PkView pk = J.pk_carve(total, nbytes, cap);
            J.pk_bases.need(n_words * 4 + 64);
            uint32_t *off = pk.off;
            memcpy(off, J.a.off.data(), (n + 1) * 4);
            if (J.mode == 1)
                for (size_t r = 0; r <= n; ++r) off[n + r] = (uint32_t)bytes1 + J.b.off[r];
            // bases: 16-byte-aligned pieces of the combined arena side by side, the exceptions of each piece after it
            const size_t P = std::max<size_t>(1, std::min<size_t>((size_t)g_threads, n_words / 4096 + 1));
            std::vector<std::vector<uint32_t>> ep(P);
            std::vector<std::vector<uint8_t>> ec(P);
            auto piece = [&](size_t t) {
                const size_t w0 = n_words * t / P, w1 = n_words * (t + 1) / P;
                size_t lo = w0 * 16, hi = std::min(w1 * 16, nbytes);
                uint32_t *bases = (uint32_t *)J.pk_bases.data();
                for (int pass = 0; pass < 2; ++pass) {  // (first pass counts the exceptions, second stores them)
                    size_t cnt = 0;
                    uint32_t *pp = pass ? ep[t].data() : nullptr;
                    uint8_t *pc = pass ? ec[t].data() : nullptr;
                    const size_t room = pass ? ep[t].size() : 0;
                    size_t got = 0;
                    if (lo < bytes1) cnt += (got = rc_pack_bases(J.a.seq.data(), lo, std::min(hi, bytes1), bases, pp, pc, room));
                    if (hi > bytes1) {
                        const size_t b0 = std::max(lo, bytes1);
                        cnt += rc_pack_bases(J.b.seq.data() - bytes1, b0, hi, bases, pp ? pp + std::min(got, room) : nullptr,
                                             pc ? pc + std::min(got, room) : nullptr, room > got ? room - got : 0);
                    }
                    if (pass == 0) {
                        if (cnt == 0) break;
                        ep[t].resize(cnt);
                        ec[t].resize(cnt);
                    }
                }
            };
            g_pool.run(P, piece);
            size_t n_exc = 0;
            for (size_t t = 0; t < P; ++t) n_exc += ep[t].size();
            J.pk_exc_pos.need(n_exc * 4 + 64);
            J.pk_exc_chr.need(n_exc + 64);
            {
                size_t at = 0;
                for (size_t t = 0; t < P; ++t) {
                    if (ep[t].empty()) continue;
                    memcpy(J.pk_exc_pos.data() + at * 4, ep[t].data(), ep[t].size() * 4);
                    memcpy(J.pk_exc_chr.data() + at, ec[t].data(), ec[t].size());
                    at += ep[t].size();
                }
            }
            // quality bits (FASTQ) over the combined arena; byte-aligned pieces
            const bool fq = J.fastq;
            if (fq) {
                uint8_t *qb = pk.qbits;
                const size_t Q = std::max<size_t>(1, std::min<size_t>((size_t)g_threads, nbytes / 65536 + 1));
                // (arena 2's bits start at bit bytes1 of the same array: pack the two arenas' bytes through one view)
                g_pool.run(Q, [&](size_t t) {
                    const size_t lo = (nbytes * t / Q) & ~(size_t)7, hi = t + 1 == Q ? nbytes : ((nbytes * (t + 1) / Q) & ~(size_t)7);
                    for (size_t p8 = lo; p8 < hi; p8 += 8) {
                        unsigned v = 0;
                        for (size_t q = p8; q < std::min(p8 + 8, hi); ++q) {
                            const signed char c = q < bytes1 ? (signed char)J.a.qual.data()[q] : (signed char)J.b.qual.data()[q - bytes1];
                            v |= (unsigned)(c > (signed char)bad_q) << (q - p8);
                        }
                        qb[p8 >> 3] = (uint8_t)v;
                    }
                });
            }
            rc_packed_batch pb;
            memset(&pb, 0, sizeof pb);
            pb.mode = J.mode;
            pb.n = n;
            pb.nbytes = nbytes;
            pb.off = off;
            pb.bases = (const uint32_t *)J.pk_bases.data();
            pb.qual_bits = fq ? (const uint8_t *)pk.qbits : nullptr;
            pb.exc_pos = n_exc ? (const uint32_t *)J.pk_exc_pos.data() : nullptr;
            pb.exc_chr = n_exc ? (const uint8_t *)J.pk_exc_chr.data() : nullptr;
            pb.n_exc = n_exc;
            pb.ret = J.ret.data();
            pb.l = J.l.data();
            pb.m = J.m.data();
            pb.h = J.h.data();
            pb.fix_pos = pk.fix_pos;
            pb.fix_chr = pk.fix_chr;
            pb.fix_cap = cap;
            {
                std::lock_guard<std::mutex> lk(submit_mu[(size_t)g]);
                rc = rc_submit_packed(ctx[g], &pb, slot);
            }
            if (!rc) rc = rc_wait_packed(ctx[g], slot);
            if (rc == RC_STATUS_NOSPACE) {  // (see the resident path) the arenas are untouched: once more, as bytes
                {
                    std::lock_guard<std::mutex> lk(submit_mu[(size_t)g]);
                    rc = rc_submit(ctx[g], &rb, slot);
                }
                if (!rc) rc = rc_wait(ctx[g], slot);
                pb.n_fix = 0;
            }
            if (!rc && pb.n_fix) {  // positions are distinct: any number of threads
                const size_t F = std::max<size_t>(1, std::min<size_t>((size_t)g_threads, pb.n_fix / 16384 + 1));
                g_pool.run(F, [&](size_t t) {
                    for (size_t q = pb.n_fix * t / F; q < pb.n_fix * (t + 1) / F; ++q) {
                        const size_t pos = pb.fix_pos[q];
                        if (pos < bytes1)
                            J.a.seq.data()[pos] = (char)pb.fix_chr[q];
                        else
                            J.b.seq.data()[pos - bytes1] = (char)pb.fix_chr[q];
                    }
                });
            }
        } else {
            {   // upload + kernels + download are queued here; the wait below overlaps with the other
                // workers' packing, submitting and formatting
                std::lock_guard<std::mutex> lk(submit_mu[(size_t)g]);
                rc = rc_submit(ctx[g], &rb, slot);
            }
            if (!rc) rc = rc_wait(ctx[g], slot);
        }
        const double tf0 = now_s();
        if (!rc) format_job(R, *j);
        const double tf1 = now_s();
        {
            std::lock_guard<std::mutex> lk(mu);
            g_t_gpu += tf0 - tg0;
            g_t_format += tf1 - tf0;
            g_t_pack += tp1 - tp0;
            j->rc = rc;
            if (rc) j->err = rc_last_error(ctx[g]);
            j->done = true;
            --R.active[(size_t)g];
            // a batch that took the GPU longer than twice what its output takes to write (15 GB/s, 2.2 output bytes per base):
            // no need to wait for the writer to find out
            // (not the first batches: theirs is the time the kernels' code takes to load -- 10 M x 100 bp reads, ten batches: the first
            // two took 0.1 s each, the lanes they asked for 0.15 s more of a 0.28 s loop)
            const double out_bytes = 2.2 * ((double)j->a.off[n] + (j->mode == 1 ? (double)j->b.off[n] : 0.0));
            if (!rc && R.batches_done++ >= 2 && tf0 - tg0 > 2.0 * out_bytes / 15e9 + 0.005) raise_lane_limit(R, "a batch takes the GPU longer than the writer");
        }
        cv.notify_all();
    }
}

static void writer_body(Run &R)
{
    std::vector<ReadFile> &files = R.files, &mates = R.mates;
    std::mutex &mu = R.mu;
    std::condition_variable &cv = R.cv;
    std::deque<std::shared_ptr<Job>> &order = R.order;
    std::vector<std::shared_ptr<Job>> &pool = R.pool;
    const bool &reader_done = R.reader_done;
    uint64_t &total_reads = R.total_reads, &total_cor = R.total_cor;
    const int k = R.k;
    double waited = 0;
    int starved = 0;
for (;;) {
    std::shared_ptr<Job> j;
    {
        std::unique_lock<std::mutex> lk(mu);
        const double tw = now_s();
        cv.wait(lk, [&] { return (!order.empty() && order.front()->done) || (reader_done && order.empty()); });
        waited = now_s() - tw;
        g_w_writer += waited;
        if (order.empty()) return;
        j = order.front();
    }
    if (j->rc) die("rcorrector: %s\n", j->err.c_str());
    const size_t n = j->a.n();
    ReadFile &f = files[(size_t)j->file], &g2 = mates[(size_t)j->file];
    const bool alternate = j->mode == 1 && g_stdout;  // main.cpp:487-495
    if (g_verbose) {
        // the transcript in the order of the reference's -t 1 loop (main.cpp:368-438): per
        // unit, mate 1's trace [and record, under -stdout], then mate 2's
        std::vector<char> vt;
        const size_t bytes1 = j->a.off[n];
        auto flush = [&]() {
            fwrite(vt.data(), 1, vt.size(), stdout);
            vt.clear();
        };
        for (size_t r = 0; r < n; ++r) {
            put_transcript(vt, *j, j->a, r, r, 0, k);
            if (g_stdout) put_record(vt, j->a, r, j->fastq, j->ret[r], j->l[r], j->m[r], j->h[r]);
            if (j->mode == 1) {
                put_transcript(vt, *j, j->b, r, n + r, bytes1, k);
                if (g_stdout) put_record(vt, j->b, r, j->fastq, j->ret[n + r], j->l[n + r], j->m[n + r], j->h[n + r]);
            }
            if (vt.size() > (1u << 20)) flush();
        }
        flush();
        fflush(stdout);
    }
    const double tw0 = now_s();
    if (j->mode == 1 && !alternate && !g_stdout) {  // two output files: written side by side
        std::thread second([&]() { emit_slices(g2, j->o2); });
        emit_slices(f, j->o1);
        second.join();
    } else {
        if (!(g_verbose && g_stdout)) emit_slices(f, j->o1);
        if (j->mode == 1 && !alternate) emit_slices(g2, j->o2);
    }
    const double wrote = now_s() - tw0;
    g_t_write += wrote;
    if (R.adaptive) {  // (see Run::lane_limit) the writer waited for this batch longer than it then took to write it: twice in a row
        starved = waited > wrote ? starved + 1 : 0;
        if (starved >= 2) {
            starved = 0;
            std::lock_guard<std::mutex> lk(mu);
            raise_lane_limit(R, "the writer waits for the GPU");
            cv.notify_all();
        }
    }
    total_reads += j->ret.size();  // UpdateSummary, main.cpp:73-79
    total_cor += j->cor_bases;
    bool retire = false;
    {
        std::lock_guard<std::mutex> lk(mu);
        order.pop_front();
        j->done = false;
        j->rc = 0;
        // once the reader has handed out its last batch no job is needed again: its buffers -- a GB of text, arenas
        // and output slices each -- are unmapped now, beside the batches still in flight, instead of after _exit
        // where the parent waits for it (0.25 s of a 2 s run)
        if (reader_done)
            retire = true;
        else
            pool.push_back(j);
    }
    cv.notify_all();
    if (retire) {  // (the last reference, as a rule: text, line index, page-locked slab, result arrays and output slices go with it)
        std::shared_ptr<Job> *last = new std::shared_ptr<Job>(std::move(j));
        std::thread([last]() { delete last; }).detach();
    }
}
}

void run_pipeline(Run &R)
{
    std::vector<ReadFile> &files = R.files, &mates = R.mates;
    std::mutex &mu = R.mu;
    std::condition_variable &cv = R.cv;
    std::deque<std::shared_ptr<Job>> &order = R.order, &q = R.q;
    std::vector<std::shared_ptr<Job>> &pool = R.pool;
    bool &closing = R.closing, &reader_done = R.reader_done;
    std::vector<std::unique_ptr<Retained>> &kept = R.kept;
    const bool resident = R.resident;
    const size_t batch_reads = R.batch_reads;
    pool.swap(R.warm_jobs);  // (the pool holds the only reference to a job at rest: see the writer's retirement)
    R.warm_jobs.clear();
    R.active.assign((size_t)R.gpus, 0);
    auto in_flight_cap = [&R]() { return (size_t)(R.gpus * R.lane_limit + 2); };  // (called under mu)
    std::vector<std::thread> workers;
    for (int wk = 0; wk < R.nworkers; ++wk) workers.emplace_back([&R, wk]() { worker_body(R, wk); });
    std::thread writer([&R]() { writer_body(R); });

    // reader
    if (resident) {  // the batches are here already: a pooled job takes over the next one's text, line index and offsets
        for (auto &R : kept) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                const double tw = now_s();
                cv.wait(lk, [&] { return order.size() < in_flight_cap(); });
                g_w_reader += now_s() - tw;
                if (!pool.empty()) {
                    j = pool.back();
                    pool.pop_back();
                }
            }
            if (!j) j = std::make_shared<Job>();
            j->file = R->file;
            j->mode = R->mode;
            j->fastq = R->fastq;
            j->resident = true;
            j->gpu = R->gpu;
            j->arena_a = R->arena_a;
            j->arena_b = R->arena_b;
            j->a.lpr = R->lpr_a;
            j->b.lpr = R->lpr_b;
            j->a.blk.swap(R->a);
            j->a.off.swap(R->off_a);
            j->a.seq_in_text = true;
            if (R->mode == 1) {
                j->b.blk.swap(R->b);
                j->b.off.swap(R->off_b);
                j->b.seq_in_text = true;
            }
            R.reset();  // (the text of the batch this job carried before: written, no longer needed)
            {
                std::lock_guard<std::mutex> lk(mu);
                order.push_back(j);
                q.push_back(j);
            }
            cv.notify_all();
        }
    } else {
        int ramp = 0;
        for (size_t fi = 0; fi < files.size(); ++fi) {
            ReadFile &f = files[fi];
            const int lpr = f.fastq ? 4 : 2;
            for (;;) {
                std::shared_ptr<Job> j;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (!pool.empty()) {
                        j = pool.back();
                        pool.pop_back();
                    }
                }
                if (!j) j = std::make_shared<Job>();
                j->file = (int)fi;
                j->mode = f.paired ? 1 : (f.interleaved ? 2 : 0);
                j->fastq = f.fastq;
                j->resident = false;
                j->gpu = -1;
                j->a.lpr = lpr;
                j->b.lpr = f.paired ? (mates[fi].fastq ? 4 : 2) : lpr;
                const double tr0 = now_s();
                // the first batches of a run are small, so that the stages behind the reader start early: an eighth,
                // a quarter, a half of -batch (whole pairs; a read's result does not depend on its batch)
                size_t want_reads = batch_reads;
                if (ramp < 3 && batch_reads >= ((size_t)1 << 19)) want_reads = (batch_reads >> (3 - ramp)) & ~(size_t)1;
                ++ramp;
                if (f.paired) {  // both mates' files at once (two inflate streams run side by side for .gz pairs)
                    std::thread mate([&]() { take_records(mates[fi].src, want_reads, j->b.lpr, j->b.blk); });
                    take_records(f.src, want_reads, lpr, j->a.blk);
                    mate.join();
                    if (j->b.blk.records != j->a.blk.records) die("ERROR: The files are not paired!\n");
                } else {
                    take_records(f.src, want_reads, lpr, j->a.blk);
                }
                if (j->a.blk.records == 0) break;
                if (j->mode == 2 && (j->a.blk.records & 1)) die("ERROR: interleaved file %s holds an odd number of reads\n", f.path.c_str());
                g_t_read += now_s() - tr0;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    const double tw = now_s();
                    cv.wait(lk, [&] { return order.size() < in_flight_cap(); });
                    g_w_reader += now_s() - tw;
                    order.push_back(j);
                    q.push_back(j);
                }
                cv.notify_all();
            }
        }
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        reader_done = true;
        if (!pool.empty()) {  // jobs at rest are not needed again either
            auto *idle = new std::vector<std::shared_ptr<Job>>();
            idle->swap(pool);
            std::thread([idle]() { delete idle; }).detach();
        }
    }
    cv.notify_all();
    writer.join();
    stamp("last batch written");
    {
        std::lock_guard<std::mutex> lk(mu);
        closing = true;
    }
    cv.notify_all();
    for (auto &t : workers) t.join();
}
