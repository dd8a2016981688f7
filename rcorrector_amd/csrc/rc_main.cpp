// rc_main.cpp -- `rcorrector`: stage 3 of run_rcorrector.pl on MI355X.  Same command line, stderr
// lines, output file names and *.cor.fq bytes as the reference's main.cpp (paths below are relative
// to /root/reference), with the per-read work done by librcorrector_amd.so (HIP) through its C ABI.
//
//   flags        main.cpp:50-71,165-268     -r/-p/-i/-c/-k/-od/-t/-maxcor/-maxcorK/-wk/-stdout/-verbose
//   file typing  Reads.h:108-162            first byte '>' FASTA, '@' FASTQ; ".gz" by the last two chars
//   output name  Reads.h:39-75,140-157      <od>/<name minus last extension>.cor.f[aq][.gz]
//   record       Reads.h:224-266,360-421    4 lines in; "<id> l:%d m:%d h:%d[ cor| unfixable_error]" out
//   batching     main.cpp:439-523           batches never span files; mates travel together
//
// Host pipeline: one reader thread cuts the input into batches of whole records
// (block reads, memchr line index, parallel packing into the SoA arenas of the C ABI), one worker
// threads per GPU run rc_submit / rc_wait on page-locked arenas, one writer thread formats in parallel and writes in input order.
//
// Extra flags (not in the reference): -gpus N shards batches over N GPUs (table replicated),
// -batch N sets the reads per batch, -inflight N the batches in flight per GPU (worker threads sharing
// the GPU's context through rc_submit / rc_wait slots: upload, kernels and download of consecutive
// batches overlap, and the GPU stays busy while a batch's slowest reads finish).  -t sets the host threads used
// for packing / formatting.
// -verbose prints the reference's per-read transcript (the -t 1 order) from rc_correct_batch_traced;
// -write-dump FILE keeps the k-mer table as jellyfish-dump text; without -c the k-mers are counted here.
#include <fcntl.h>
#include <malloc.h>
#include <sys/stat.h>
#include <unistd.h>

#include "rc_dispatch.h"

static void print_help()
{
    fprintf(stderr,
            "Usage: ./rcorrector [OPTIONS]\n"
            "OPTIONS:\n"
            "Required parameters:\n"
            "\t-r seq_file: seq_file is the path to the sequence file. Can use multiple -r to specifiy multiple sequence files\n"
            "\t-p seq_file_left seq_file_right: the paths to the paired-end data set. Can use multiple -p to specifiy multiple sequence files\n"
            "\t-i seq_file: seq_file is the path to the interleaved mate-pair sequence file. Can use multiple -i\n"
            "\t-c jf_dump: the kmer counts dumped by JellyFish (without -c the k-mers of the input files are counted on the GPU)\n"
            "\t-k kmer_length\n"
            "Other parameters:\n"
            "\t-od output_file_directory (default: ./)\n"
            "\t-t number of threads to use (default: 1)\n"
            "\t-maxcor INT: the maximum number of correction every 100bp (default: 8)\n"
            "\t-maxcorK INT: the maximum number of correction within k-bp window (default: 4)\n"
            "\t-wk FLOAT: the proportion of kmers that are used to estimate weak kmer count threshold (default: 0.95)\n"
            "\t-stdout: output the corrected sequences to stdout (default: not used)\n"
            "\t-verbose: output some correction information to stdout (default: not used)\n"
            "MI355X build only:\n"
            "\t(-t: host threads that read, pack, format and write around the GPU; without it, or with -t 1: 16)\n"
            "\t-gpus INT: number of GPUs to shard the reads over, k-mer table replicated (default: 1)\n"
            "\twithout -c the k-mers are counted here (exact counts >= 2); ERROR_RATE is then estimated over this program's own\n"
            "\t\tdump order, not Jellyfish's: self-consistent, not byte-comparable with a jellyfish + reference run on large inputs\n"
            "\t-batch INT: reads per GPU batch (default: 1048576)\n"
            "\t-inflight INT: batches in flight per GPU, 1-4 (default: 2, raised to 4 while the writer waits for the GPU)\n"
            "\t-write-dump STRING: also write the k-mer table as a jellyfish-dump text file\n"
            "\t-verbose-iter INT: threshold iterations recorded per read for -verbose (default: 64)\n");
}

int main(int argc, char **argv)
{
    Run run;
    int &k = run.k, &gpus = run.gpus, &inflight = run.inflight;
    bool inflight_given = false;
    size_t &batch_reads = run.batch_reads;
    std::vector<ReadFile> &files = run.files, &mates = run.mates;
    int max_fix_per_k = 4, i;
    double wk = 0.95;
    const char *dump = nullptr, *write_dump = nullptr;
    std::string od = "./";
    bool verbose = false;
    int t_flag = 0;
    if (argc == 1) {
        print_help();
        return 0;
    }
    g_timing = getenv("RC_TIMING") != nullptr;
    stamp("main() entered");
    for (i = 1; i < argc; ++i) {  // main.cpp:165-247
        if (!strcmp("-r", argv[i]) || !strcmp("-i", argv[i]))
            ++i;
        else if (!strcmp("-p", argv[i]))
            i += 2;
        else if (!strcmp("-od", argv[i])) {
            mkdir(argv[i + 1], 0700);
            od = argv[++i];
        } else if (!strcmp("-c", argv[i])) {
            dump = argv[++i];
            FILE *fp = fopen(dump, "r");
            if (!fp) die("Could not open file %s\n", dump);
            fclose(fp);
        } else if (!strcmp("-k", argv[i]))
            k = atoi(argv[++i]);
        else if (!strcmp("-t", argv[i]))
            t_flag = atoi(argv[++i]);
        else if (!strcmp("-maxcor", argv[i]))
            ++i;
        else if (!strcmp("-maxcorK", argv[i]))
            max_fix_per_k = atoi(argv[++i]);
        else if (!strcmp("-wk", argv[i]))
            wk = atof(argv[++i]);
        else if (!strcmp("-stdout", argv[i]))
            g_stdout = true;
        else if (!strcmp("-verbose", argv[i]))
            verbose = true;
        else if (!strcmp("-verbose-iter", argv[i]))
            g_trace_iter = atoi(argv[++i]);
        else if (!strcmp("-gpus", argv[i]))
            gpus = atoi(argv[++i]);
        else if (!strcmp("-batch", argv[i]))
            batch_reads = (size_t)atol(argv[++i]);
        else if (!strcmp("-inflight", argv[i])) {
            inflight = atoi(argv[++i]);
            inflight_given = true;
        }
        else if (!strcmp("-write-dump", argv[i]))
            write_dump = argv[++i];
        else if (!strcmp("-packed", argv[i]))
            g_packed = true;
        else if (!strcmp("-h", argv[i])) {
            print_help();
            return 0;
        } else {
            fprintf(stderr, "Unknown argument: %s\n", argv[i]);
            return 0;
        }
    }
    g_verbose = verbose;
    if (g_trace_iter < 1) g_trace_iter = 1;
    // -verbose carries RC_TRACE_ITER_WORDS x trace-iter words per read through host and device
    // (9 KB per read at the default 64 iterations): small batches, or a real data set needs tens of GB
    if (verbose && batch_reads > (1u << 16)) batch_reads = 1u << 16;
    if (gpus < 1) gpus = 1;
    if (inflight < 1) inflight = 1;
    if (inflight > 8) inflight = 8;
    if (batch_reads < 2) batch_reads = 2;
    batch_reads &= ~(size_t)1;
    {
        unsigned hc = std::thread::hardware_concurrency();
        // (default: 16 -- the loops these threads share are memory copies and page-cache calls, and beyond that
        // they get in each other's way: 16 M x 150 bp pairs, files to files, 0.53 / 0.50 / 0.49 / 0.51 / 0.52 / 0.62 s
        // at 8 / 12 / 16 / 20 / 24 / 32 threads on a 2 x 64-core host)
        g_threads = t_flag > 1 ? t_flag : (int)std::min<unsigned>(hc ? hc : 8, 16);
        if (g_threads < 1) g_threads = 1;
    }
    g_timing = getenv("RC_TIMING") != nullptr;
    if (const char *e = getenv("RC_TRANSPORT")) g_packed = g_packed || !strcmp(e, "packed");
    if (verbose) g_packed = false;  // (the transcript needs the traced entry point)
    // batches recycle buffers of hundreds of MB: keep freed memory in the heap instead of handing it
    // back to the kernel and faulting it in again page by page (with dozens of threads every
    // mmap/munmap/page fault also serialises on the process's memory-map lock)
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, -1);

    files.reserve(MAX_READ_FILE);
    mates.reserve(MAX_READ_FILE);
    for (i = 1; i < argc; ++i) {  // main.cpp:250-268
        const bool is_in = !strcmp("-r", argv[i]) || !strcmp("-p", argv[i]) || !strcmp("-i", argv[i]);
        if (!is_in) continue;
        if (files.size() >= MAX_READ_FILE) die("The number of read files exceeds the limit %d.\n", MAX_READ_FILE);
        files.emplace_back();
        mates.emplace_back();
        if (!strcmp("-r", argv[i])) {
            open_file(files.back(), argv[i + 1], false, false, od);
            ++i;
        } else if (!strcmp("-p", argv[i])) {
            open_file(files.back(), argv[i + 1], true, false, od);
            open_file(mates.back(), argv[i + 2], true, false, od);
            if (files.back().fastq != mates.back().fastq)
                die("rcorrector: %s and %s are a FASTQ and a FASTA file: the mates of a pair must have the same format\n", argv[i + 1], argv[i + 2]);
            i += 2;
        } else {
            open_file(files.back(), argv[i + 1], false, true, od);
            ++i;
        }
    }

    // one context per GPU (the table is replicated across GPUs); `inflight` worker threads per GPU keep
    // that many batches in flight in it through rc_submit / rc_wait, one slot each
    if (inflight > RC_MAX_SLOTS) inflight = RC_MAX_SLOTS;
    run.lane_limit = inflight;
    if (!inflight_given && !verbose) {  // (Run::lane_limit) two at work, up to four where the GPU is what the writer waits for
        run.adaptive = true;
        inflight = RC_MAX_SLOTS;
    }
    const int nctx = gpus, nworkers = run.nworkers = gpus * inflight;
    run.submit_mu.reset(new std::mutex[(size_t)gpus]);
    std::vector<rc_ctx *> &ctx = run.ctx;
    ctx.assign((size_t)nctx, nullptr);
    // RC_SHARED_GPU=1 (tests): every "GPU" is device 0, so that the -gpus N path -- one table replica
    // per GPU, batches dealt to whichever context is free -- runs on a one-GPU box
    const bool shared_gpu = run.shared_gpu = getenv("RC_SHARED_GPU") != nullptr;
    const bool numa_on = run.numa_on = !(getenv("RC_NUMA") && !strcmp(getenv("RC_NUMA"), "0"));
    const bool lanes_env = getenv("RC_SLOT_LANES") != nullptr;
    rc_runtime_prepare(16);  // hardware queues for the slot lanes' streams: no thread exists yet, HIP has not started
    // The GPU runtime takes 0.08-0.25 s to come up (tools/mb/hipinit.hip): the contexts are created on a thread of their own,
    // and what needs no GPU goes ahead -- the test whether this is a one-pass run, as far as the host can say, and its reader.
    std::string ctx_err;
    std::thread ctx_thread([&]() {
        char err[512];
        for (int c = 0; c < nctx; ++c) {
            rc_config cfg = {shared_gpu ? 0 : c, k, max_fix_per_k};
            ctx[c] = rc_create(&cfg, err, sizeof err);
            if (!ctx[c]) {
                ctx_err = err;
                return;
            }
            // slot lanes (rcorrector_amd.h: rc_submit): off while the run is bound by its writer, which wants its batches back one
            // after the other (two batches side by side on the GPU each take twice as long: 25 M x 150 bp pairs, loop 0.60 against
            // 0.52 s); on from the moment the writer waits for the GPU (rc_dispatch: Run::lane_limit).  RC_SLOT_LANES decides if set.
            if (!lanes_env) rc_set_slot_lanes(ctx[c], run.adaptive ? 0 : 1);
        }
    });
    for (size_t fi = 0; fi < files.size(); ++fi)
        if ((files[fi].out_gz || (files[fi].paired && mates[fi].out_gz)) && !g_stdout) {
            const unsigned hc = std::thread::hardware_concurrency();
            g_deflate_threads = t_flag > 1 ? (size_t)t_flag : std::min<size_t>(hc ? hc / 2 : 8, 96);
        }
    // One pass (see ingest_resident): no dump, any number of GPUs (the batches are dealt to them as they are read), regular files (plain or .gz: one inflate pass instead of two) whose text
    // fits a third of the host memory that is available and whose bases, count scratch and table fit the HBM that is free.  RC_RESIDENT=0 keeps the two passes, =1 skips the size test.
    bool &resident = run.resident;
    std::vector<std::unique_ptr<Retained>> &kept = run.kept;
    const bool one_pass_shape = !dump && !verbose && !files.empty();
    uint64_t text_bytes = 0;
    bool plain = true, any_gz = false, host_fits = false;
    const char *res_env = getenv("RC_RESIDENT");
    if (one_pass_shape) {
        for (size_t fi = 0; fi < files.size(); ++fi)
            for (const ReadFile *f : {(const ReadFile *)&files[fi], files[fi].paired ? (const ReadFile *)&mates[fi] : (const ReadFile *)nullptr}) {
                if (!f) continue;
                struct stat st;
                if (f->src.is_gz) {  // (its text is taken as eight times the file: FASTQ deflates to a fifth or a quarter)
                    any_gz = true;
                    if (stat(f->path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) {
                        plain = false;
                        continue;
                    }
                    text_bytes += (uint64_t)st.st_size * 8;
                    continue;
                }
                if (!f->src.seekable || fstat(f->src.fd, &st) != 0) {
                    plain = false;
                    continue;
                }
                text_bytes += (uint64_t)st.st_size;
            }
        uint64_t avail = 0;
        if (FILE *mi = fopen("/proc/meminfo", "r")) {
            char ln[256];
            while (fgets(ln, sizeof ln, mi))
                if (!strncmp(ln, "MemAvailable:", 13)) avail = (uint64_t)atoll(ln + 13) << 10;
            fclose(mi);
        }
        for (const char *lim : {"/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"})  // a container's own limit
            if (FILE *cg = fopen(lim, "r")) {
                char ln[64];
                if (fgets(ln, sizeof ln, cg) && ln[0] >= '0' && ln[0] <= '9') avail = std::min<uint64_t>(avail, strtoull(ln, nullptr, 10));
                fclose(cg);
            }
        host_fits = plain && (res_env ? atoi(res_env) != 0 : text_bytes <= avail / 3);
        if (res_env && atoi(res_env) > 1) batch_reads = std::max<size_t>(2, (size_t)atoi(res_env)) & ~(size_t)1;  // (tests: RC_RESIDENT=<batch size>)
    }
    uint64_t count_mem = (uint64_t)24 << 30;
    if (const char *cm = getenv("RC_COUNT_MEM_MB")) count_mem = (uint64_t)atoll(cm) << 20;
    // The reader goes ahead where the answer is all but certain -- plain files (a .gz source cannot be rewound, and how it is
    // inflated depends on the answer), a text that leaves most of an MI355X's HBM free -- and where its buffers land on the right
    // NUMA node: one GPU means the host side is bound to that GPU's node, which sysfs can tell before HIP can.
    int node_guess = -2;
    bool ahead = host_fits && !any_gz && text_bytes + count_mem + ((uint64_t)1 << 30) <= ((uint64_t)160 << 30) && !getenv("RC_NO_READ_AHEAD");
    if (ahead && numa_on && gpus == 1) {
        node_guess = first_gpu_numa_node();
        if (node_guess == -2) ahead = false;
    }
    bool pool_started = false;
    HeadStats head;
    std::unique_ptr<Ingest> ingest;
    if (ahead) {
        if (numa_on && gpus == 1 && node_guess >= 0 && bind_to_numa_node(node_guess) && g_timing)
            fprintf(stderr, "[rc timing] host threads bound to NUMA node %d (the first GPU's, by sysfs)\n", node_guess);
        g_pool.start(std::max<size_t>((size_t)g_threads * 2, g_deflate_threads));
        pool_started = true;
        if (g_timing) fprintf(stderr, "[rc timing] the reader starts before the GPU runtime is up\n");
        head = head_stats(run);  // (before the reader takes the head of the first file out of its source)
        ingest.reset(new Ingest(run, batch_reads, true));
        ingest->depth = 8;
        ingest->start();
    }
    ctx_thread.join();
    if (!ctx_err.empty()) die("rcorrector: %s\n", ctx_err.c_str());
    // One GPU: the whole host pipeline -- reader, packers, formatters, writers and their buffers -- lives on the NUMA
    // node that GPU hangs off (every byte of a read crosses host memory a dozen times on its way through; across the
    // socket link each crossing costs more).  Several GPUs: each GPU's worker threads bind themselves (below).
    if (numa_on && gpus == 1) {
        const int node = rc_device_numa_node(ctx[0]);
        if (node >= 0 && node != node_guess && bind_to_numa_node(node) && g_timing)
            fprintf(stderr, "[rc timing] host threads bound to NUMA node %d%s\n", node, ahead ? " (sysfs had said another: the helper threads and the first blocks stay where they are)" : "");
    }
    // (the reader, the mate's reader and the workers call the pool side by side)
    if (!pool_started) g_pool.start(std::max<size_t>((size_t)g_threads * 2, g_deflate_threads));
    stamp("contexts created (HIP initialised, scratch allocated)");
    const double t_start = now_s();
    // While the table loads: the batch buffers of the pipeline -- text blocks, page-locked arenas, output slices --
    // are allocated, sized from the head of the first input, touched and registered with the GPU runtime here, so
    // that the first batches do not pay for a few GB of page faults and hipHostRegister calls one after the other
    run.max_in_flight = (size_t)(gpus * run.lane_limit + 2);
    if (one_pass_shape) {
        // HBM: the bases stay with the counter through the table build (about half of a FASTQ file's bytes), next to its
        // sort scratch (RC_COUNT_MEM_MB, 24 GiB by default) and the table itself, which the bases bound from above for
        // anything but a tiny input -- against what the device has free right now (another process may share it)
        // Several GPUs: every one of them holds its share of the bases, a sort scratch of its own and the staging of the
        // sharded count (rc_table_count_finish_sharded) -- priced as if each held everything, against the GPU with the
        // least free memory (a GPU another process shares must not run out in the middle of the count)
        uint64_t hbm_free = ~(uint64_t)0;
        for (int g = 0; g < gpus; ++g) {
            uint64_t f = 0;
            if (rc_device_memory(ctx[g], &f, nullptr)) f = 0;
            hbm_free = std::min(hbm_free, f);
        }
        if (const char *hf = getenv("RC_HBM_FREE_MB")) hbm_free = (uint64_t)atoll(hf) << 20;  // tests: as if this much were free
        const bool fits_hbm = text_bytes + count_mem + ((uint64_t)1 << 30) <= hbm_free;
        resident = plain && (res_env ? atoi(res_env) != 0 : (host_fits && fits_hbm));
        g_gz_whole = resident;
    }
    if (ahead && !resident) {  // the HBM is not free after all: two passes, from the start of the files
        ingest->abort();
        ingest.reset();
        for (size_t fi = 0; fi < files.size(); ++fi)
            for (ReadFile *f : {&files[fi], files[fi].paired ? &mates[fi] : (ReadFile *)nullptr}) {
                if (!f) continue;
                f->src.rewind();
                f->src.left.need(4096);
                f->src.left_len = f->src.fill(f->src.left.p, 4096);
            }
        if (g_timing) fprintf(stderr, "[rc timing] the reader had gone ahead for one pass; the device memory says two: started over\n");
        ahead = false;
    }
    // (the head of the first file is looked at here, not in the thread: the one-pass reader takes it out of the source)
    if (!ahead) head = head_stats(run);
    std::thread warm([&]() { warm_buffers(run, head); });
    int64_t stored = 0;
    if (dump) {  // main.cpp:294-308: ONE Store, loaded once
        if (rc_table_load_jfdump(ctx[0], dump, &stored)) die("rcorrector: %s\n", rc_last_error(ctx[0]));
    } else {
        // no -c: stages 0-2 of run_rcorrector.pl:262-281 on the GPU -- count the canonical k-mers of
        // every input file (mates included), keep count >= 2, build the table
        std::vector<std::pair<std::string, bool>> inputs;  // path, fastq
        for (size_t fi = 0; fi < files.size(); ++fi) {
            inputs.emplace_back(files[fi].path, files[fi].fastq);
            if (files[fi].paired) inputs.emplace_back(mates[fi].path, mates[fi].fastq);
        }
        if (ahead)
            ingest->consume(&stored);
        else
            ingest_resident(run, resident ? batch_reads : std::max<size_t>(batch_reads, (size_t)1 << 20), &stored, resident);
        if (g_timing)
            fprintf(stderr, "[rc timing] k-mer counting pass over %zu file(s): %.2f s%s\n", inputs.size(), now_s() - t_start,
                    resident ? " (one pass: the text stays in host memory, the bases in HBM)" : "");
    }
    if (gpus > 1) {  // replicate the bucket array device to device (xGMI) and make sure the replicas agree
        uint64_t d0 = 0;
        if (rc_table_digest(ctx[0], &d0)) die("rcorrector: %s\n", rc_last_error(ctx[0]));
        for (int g = 1; g < gpus; ++g)  // every copy is queued before the first one is waited for: one per xGMI link
            if (rc_table_replicate_async(ctx[g], ctx[0])) die("rcorrector: %s\n", rc_last_error(ctx[g]));
        for (int g = 1; g < gpus; ++g) {
            uint64_t dg = 0;
            if (rc_sync(ctx[g]) || rc_table_digest(ctx[g], &dg)) die("rcorrector: %s\n", rc_last_error(ctx[g]));
            if (dg != d0) die("rcorrector: the k-mer table replica on GPU %d differs from the original\n", g);
        }
    }
    if (write_dump && rc_table_write_jfdump(ctx[0], write_dump, nullptr)) die("rcorrector: %s\n", rc_last_error(ctx[0]));
    fprintf(stderr, "Stored %d kmers\n", (int)stored);
    double rate = 0.01;
    if (rc_estimate_error_rate(ctx[0], wk, &rate)) die("rcorrector: %s\n", rc_last_error(ctx[0]));
    fprintf(stderr, "Weak kmer threshold rate: %lf (estimated from %.3lf/1 of the chosen kmers)\n", rate, wk);
    stamp("ERROR_RATE known");

    // GetBadQuality, main.cpp:88-128: first <= 1M records of the primary files, in order
    char &bad_q = run.bad_q;
    if (!files.empty() && files[0].fastq) {
        std::vector<int32_t> fh(300, 0), lh(300, 0);
        int total = 0;
        if (resident) {  // the same records, from the blocks the counting pass kept (primary files, in order)
            for (const auto &R : kept) {
                if (total >= 1000000) break;
                quality_histograms(R->a, R->lpr_a, (size_t)(1000000 - total), fh, lh, &total);
            }
        } else {
            for (size_t fi = 0; fi < files.size() && total < 1000000; ++fi) {
                Source s;
                s.open(files[fi].path);
                Block b;
                const int lpr = files[fi].fastq ? 4 : 2;
                while (total < 1000000) {
                    take_records(s, std::min<size_t>((size_t)(1000000 - total), (size_t)1 << 18), lpr, b);
                    if (b.records == 0) break;
                    quality_histograms(b, lpr, b.records, fh, lh, &total);
                }
                s.close();
            }
        }
        bad_q = rc_bad_quality_from_hist(fh.data(), lh.data(), total);
    }
    fprintf(stderr, "Bad quality threshold is '%c'\n", bad_q);
    stamp("ERROR_RATE and bad quality known");
    for (int c = 0; c < nctx; ++c)
        if (rc_set_run_params(ctx[c], rate, bad_q)) die("rcorrector: %s\n", rc_last_error(ctx[c]));
    const double t_setup = now_s();
    stamp("start-up done");

    // pipeline: reader (this thread) -> `inflight` workers per GPU -> writer thread (input order)
    warm.join();
    stamp("batch buffers ready");
    run_pipeline(run);
    const double t_loop_end = now_s();

    for (size_t fi = 0; fi < files.size(); ++fi) {
        for (ReadFile *f : {&files[fi], &mates[fi]}) {
            if (f->out && f->out_gz && !f->wrote) {  // an empty .gz is still one (empty) gzip member
                OutBuf none, z;
                gzip_member(none, z);
                std::vector<OutBuf> one(1);
                one[0].swap(z);
                emit_slices(*f, one);
            }
            if (f->out && f->out != stdout && f->preallocated && ftruncate(fileno(f->out), f->out_off) != 0)
                die("ERROR: could not set the length of the output of %s\n", f->path.c_str());
            if (f->out && f->out != stdout) fclose(f->out);
            f->src.close();
        }
    }
    // (the contexts are not torn down: the process ends here, and releasing gigabytes of device and
    // host memory piece by piece costs more than everything else after the last write)
    if (g_timing)
        fprintf(stderr, "[rc timing] start-up (dump load, table build, ERROR_RATE, bad quality) %.2f s; correction loop (read, correct, write) %.2f s; %d host threads\n",
                t_setup - t_start, t_loop_end - t_setup, g_threads);
    if (g_timing)
        fprintf(stderr, "[rc timing] stage totals: read+index %.2f s (reader thread); pack %.2f s + correct_batch %.2f s + format %.2f s (sum over %d worker threads); write %.2f s (writer thread)\n",
                g_t_read, g_t_pack, g_t_gpu, g_t_format, nworkers, g_t_write);
    if (g_timing) fprintf(stderr, "[rc timing] inside read+index (all files, thread-seconds): pread %.2f s, newline scan %.2f s, line index %.2f s\n", g_t_fill, g_t_nl, g_t_idx);
    if (g_timing)
        fprintf(stderr, "[rc timing] blocked: reader %.2f s (no free slot), workers %.2f s (no batch), writer %.2f s (next batch not done)\n", g_w_reader, g_w_worker, g_w_writer);
    fprintf(stderr, "Processed %llu reads\n\tCorrected %llu bases.\n", (unsigned long long)run.total_reads, (unsigned long long)run.total_cor);
    stamp("outputs closed, leaving");
    if (getenv("RC_TEARDOWN")) {  // dev: where the time between _exit and the parent's wait goes
        run.pool.clear();
        run.warm_jobs.clear();
        run.order.clear();
        run.q.clear();
        stamp("teardown: job buffers unregistered and freed");
        for (rc_ctx *c : ctx) rc_destroy(c);
        stamp("teardown: contexts destroyed");
        if (atoi(getenv("RC_TEARDOWN")) >= 2) {  // ... and what is left once the helper threads, the heap and the HIP runtime are gone too
            g_pool.stop_and_join();
            stamp("teardown: helper threads joined");
            malloc_trim(0);
            stamp("teardown: heap trimmed");
            if (void *h = dlopen("libamdhip64.so", RTLD_NOW | RTLD_NOLOAD)) {
                typedef int (*reset_fn)();
                if (reset_fn f = (reset_fn)dlsym(h, "hipDeviceReset")) f();
                stamp("teardown: hipDeviceReset");
            }
            if (FILE *sm = fopen("/proc/self/smaps", "r")) {  // the largest resident mappings that are left
                std::vector<std::pair<long, std::string>> maps;
                char ln[512];
                std::string head;
                while (fgets(ln, sizeof ln, sm)) {
                    if (strchr(ln, '-') && strchr(ln, '-') < ln + 20 && !strstr(ln, "kB")) head = ln;
                    if (!strncmp(ln, "Rss:", 4)) maps.emplace_back(atol(ln + 4), head);
                }
                fclose(sm);
                std::sort(maps.begin(), maps.end(), [](const std::pair<long, std::string> &a, const std::pair<long, std::string> &b) { return a.first > b.first; });
                for (size_t i2 = 0; i2 < maps.size() && i2 < 8; ++i2) fprintf(stderr, "[rc timing] at exit %8ld kB resident: %s", maps[i2].first, maps[i2].second.c_str());
            }
            if (FILE *st = fopen("/proc/self/status", "r")) {
                char ln[256];
                while (fgets(ln, sizeof ln, st))
                    if (!strncmp(ln, "VmRSS:", 6) || !strncmp(ln, "RssAnon:", 8) || !strncmp(ln, "RssFile:", 8) || !strncmp(ln, "RssShmem:", 9) || !strncmp(ln, "Threads:", 8) || !strncmp(ln, "VmPTE:", 6))
                        fprintf(stderr, "[rc timing] at exit %s", ln);
                fclose(st);
            }
        }
    }
    fflush(NULL);
    _exit(0);  // every output is closed: skip unmapping gigabytes of buffers one by one
}
