// rc_main.cpp -- `rcorrector`: stage 3 of run_rcorrector.pl on MI355X.  Same command line, stderr
// lines, output file names and *.cor.fq bytes as the reference's main.cpp (paths below are relative
// to /root/reference), with the per-read work done by librcorrector_amd.so (HIP) through its C ABI.
//
//   flags        main.cpp:50-71,165-268     -r/-p/-i/-c/-k/-od/-t/-maxcor/-maxcorK/-wk/-stdout/-verbose
//   file typing  Reads.h:108-162            first byte '>' FASTA, '@' FASTQ; ".gz" by the last two chars
//   output name  Reads.h:39-75,140-157      <od>/<name minus last extension>.cor.f[aq][.gz]
//   record       Reads.h:360-421            "<id> l:%d m:%d h:%d[ cor| unfixable_error]"
//   batching     main.cpp:439-523           batches never span files; mates travel together
//
// Extra (not in the reference): -gpus N shards batches over N GPUs (table replicated, ordered
// writer), and -batch N sets the reads per batch.  -t is accepted; the GPU path does not need it.
// -verbose (per-read trace on stdout) is not provided by the GPU path and is refused loudly.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <zlib.h>

#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rcorrector_amd.h"

#define MAX_READ_FILE 100    // Reads.h:11
#define MAX_READ_LENGTH 1024 // utils.h:7
#define MAX_ID_LENGTH 2048   // utils.h:8

static bool g_stdout = false;

struct ReadFile {
    std::string path;
    bool paired = false, interleaved = false, fastq = true, out_gz = false;
    gzFile in = nullptr;
    FILE *out = nullptr;
    gzFile outz = nullptr;
};

static void die(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    exit(1);
}

// Reads.h:39-75
static std::string base_name(const std::string &path)
{
    std::string in = path;
    int len = (int)in.size(), i, j;
    for (i = len; i >= 0 && in.c_str()[i] != '.' && in.c_str()[i] != '/'; --i)
        ;
    if (i >= 0 && !strcmp(in.c_str() + i, ".gz")) {
        int tmp = i;
        for (i = i - 1; i >= 0 && in[i] != '.' && in[i] != '/'; --i)
            ;
        std::string stem = in.substr(0, tmp);
        const char *e = i >= 0 ? stem.c_str() + i : "";
        if (!(i >= 0 && (!strcmp(e, ".fastq") || !strcmp(e, ".fasta") || !strcmp(e, ".fq") || !strcmp(e, ".fa")))) i = tmp;
    }
    for (j = len; j >= 0 && in.c_str()[j] != '/'; --j)
        ;
    if (i >= 0 && in.c_str()[i] == '.') return in.substr(j + 1, i - (j + 1));
    return in.substr(j + 1);
}

static void open_file(ReadFile &f, const char *path, bool paired, bool interleaved, const std::string &od)
{
    f.path = path;
    f.paired = paired;
    f.interleaved = interleaved;
    f.in = gzopen(path, "r");
    if (!f.in) die("ERROR: Could not access file %s\n", path);
    char first[2048];
    first[0] = 0;
    gzgets(f.in, first, sizeof first);
    if (first[0] == '>')
        f.fastq = false;
    else if (first[0] == '@')
        f.fastq = true;
    else
        die("\"%s\"'s format is wrong: %s\n", path, first);
    gzrewind(f.in);
    size_t len = strlen(path);
    f.out_gz = len >= 2 && path[len - 2] == 'g' && path[len - 1] == 'z';
    std::string outp = od + "/" + base_name(path) + (f.fastq ? ".cor.fq" : ".cor.fa") + (f.out_gz ? ".gz" : "");
    if (g_stdout) {
        f.out = stdout;
        f.out_gz = false;
    } else if (f.out_gz) {
        f.outz = gzopen(outp.c_str(), "w1");  // compressLevel 1, Reads.h:84, File.h:62-66
        if (!f.outz) die("ERROR: Could not access file %s\n", outp.c_str());
    } else {
        f.out = fopen(outp.c_str(), "w");
        if (!f.out) die("ERROR: Could not access file %s\n", outp.c_str());
    }
}

// one batch of reads of one file (plus its mate file), SoA for the C ABI
struct Batch {
    int file = 0;
    int mode = 0;
    std::vector<char> seq, qual, seq2, qual2;
    std::vector<uint32_t> off, off2;
    std::vector<std::string> id, id2;
    std::vector<int32_t> ret, l, m, h;
    size_t n() const { return id.size(); }
};

struct Reader {
    std::vector<ReadFile> files, mates;
};

static void emit(ReadFile &f, const char *s, size_t n)
{
    if (f.out_gz)
        gzwrite(f.outz, s, (unsigned)n);
    else
        fwrite(s, 1, n, f.out);
}

// Reads.h:360-421
static void write_record(ReadFile &f, const std::string &id, const char *seq, const char *qual, int cor, int l, int m, int h,
                         std::string &line)
{
    char info[96];
    snprintf(info, sizeof info, " l:%d m:%d h:%d", l, m, h);
    line.clear();
    line += id;
    line += info;
    if (cor == -1)
        line += " unfixable_error";
    else if (cor > 0)
        line += " cor";
    line += '\n';
    line += seq;
    line += '\n';
    if (f.fastq) {
        line += "+\n";
        line += qual;
        line += '\n';
    }
    emit(f, line.data(), line.size());
}

static void print_help()
{
    fprintf(stderr,
            "Usage: ./rcorrector [OPTIONS]\n"
            "OPTIONS:\n"
            "Required parameters:\n"
            "\t-r seq_file: seq_file is the path to the sequence file. Can use multiple -r to specifiy multiple sequence files\n"
            "\t-p seq_file_left seq_file_right: the paths to the paired-end data set. Can use multiple -p to specifiy multiple sequence files\n"
            "\t-i seq_file: seq_file is the path to the interleaved mate-pair sequence file. Can use multiple -i\n"
            "\t-c jf_dump: the kmer counts dumped by JellyFish\n"
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
            "\t-gpus INT: number of GPUs to shard the reads over, k-mer table replicated (default: 1)\n"
            "\t-batch INT: reads per GPU batch (default: 1048576)\n");
}

// a batch travelling through the pipeline, with the verbatim quality strings for the writer
struct Job {
    Batch b;
    std::vector<std::string> q1, q2;
    bool done = false;
    int rc = 0;
    std::string err;
};

int main(int argc, char **argv)
{
    int k = 23, max_fix_per_k = 4, gpus = 1, i;
    double wk = 0.95;
    const char *dump = nullptr;
    std::string od = "./";
    size_t batch_reads = 1 << 20;
    bool verbose = false;
    if (argc == 1) {
        print_help();
        return 0;
    }
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
            ++i;
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
        else if (!strcmp("-gpus", argv[i]))
            gpus = atoi(argv[++i]);
        else if (!strcmp("-batch", argv[i]))
            batch_reads = (size_t)atol(argv[++i]);
        else if (!strcmp("-h", argv[i])) {
            print_help();
            return 0;
        } else {
            fprintf(stderr, "Unknown argument: %s\n", argv[i]);
            return 0;
        }
    }
    if (verbose) die("-verbose (per-read trace) is not available on the GPU path; use the CPU reference for traces\n");
    if (!dump) die("Could not open file %s\n", "(no -c given)");
    if (gpus < 1) gpus = 1;
    if (batch_reads < 2) batch_reads = 2;

    Reader rd;
    for (i = 1; i < argc; ++i) {  // main.cpp:250-268
        if (rd.files.size() >= MAX_READ_FILE && (!strcmp("-r", argv[i]) || !strcmp("-p", argv[i]) || !strcmp("-i", argv[i])))
            die("The number of read files exceeds the limit %d.\n", MAX_READ_FILE);
        if (!strcmp("-r", argv[i])) {
            rd.files.emplace_back();
            rd.mates.emplace_back();
            open_file(rd.files.back(), argv[i + 1], false, false, od);
            ++i;
        } else if (!strcmp("-p", argv[i])) {
            rd.files.emplace_back();
            rd.mates.emplace_back();
            open_file(rd.files.back(), argv[i + 1], true, false, od);
            open_file(rd.mates.back(), argv[i + 2], true, false, od);
            i += 2;
        } else if (!strcmp("-i", argv[i])) {
            rd.files.emplace_back();
            rd.mates.emplace_back();
            open_file(rd.files.back(), argv[i + 1], false, true, od);
            ++i;
        }
    }

    // contexts: one per GPU, table replicated
    std::vector<rc_ctx *> ctx((size_t)gpus, nullptr);
    char err[512];
    for (int g = 0; g < gpus; ++g) {
        rc_config cfg = {g, k, max_fix_per_k};
        ctx[g] = rc_create(&cfg, err, sizeof err);
        if (!ctx[g]) die("rcorrector: %s\n", err);
    }
    int64_t stored = 0;
    for (int g = 0; g < gpus; ++g)
        if (rc_table_load_jfdump(ctx[g], dump, &stored)) die("rcorrector: %s\n", rc_last_error(ctx[g]));
    fprintf(stderr, "Stored %d kmers\n", (int)stored);
    double rate = 0.01;
    if (rc_estimate_error_rate(ctx[0], wk, &rate)) die("rcorrector: %s\n", rc_last_error(ctx[0]));
    fprintf(stderr, "Weak kmer threshold rate: %lf (estimated from %.3lf/1 of the chosen kmers)\n", rate, wk);

    // GetBadQuality, main.cpp:88-128: first <= 1M records of the primary files, in order
    char bad_q = 0;
    if (!rd.files.empty() && rd.files[0].fastq) {
        std::vector<int32_t> fh(300, 0), lh(300, 0);
        int total = 0;
        std::vector<char> s, q;
        std::string id;
        static char idb[MAX_ID_LENGTH], sb[MAX_READ_LENGTH], qb[MAX_READ_LENGTH], plus[2048];
        for (size_t fi = 0; fi < rd.files.size() && total < 1000000; ++fi) {
            ReadFile &f = rd.files[fi];
            while (total < 1000000 && gzgets(f.in, idb, MAX_ID_LENGTH)) {
                sb[0] = qb[0] = 0;
                gzgets(f.in, sb, MAX_READ_LENGTH);
                if (f.fastq) {
                    gzgets(f.in, plus, sizeof plus);
                    gzgets(f.in, qb, MAX_READ_LENGTH);
                }
                size_t len = strlen(sb);
                if (len && sb[len - 1] == '\n') sb[len - 1] = 0;
                if (f.fastq && len && qb[len - 1] == '\n') qb[len - 1] = 0;
                size_t sl = strlen(sb);
                if (sl == 0) continue;
                ++lh[(int)(unsigned char)qb[sl - 1]];
                ++fh[(int)(unsigned char)qb[0]];
                ++total;
            }
            gzrewind(f.in);
        }
        bad_q = rc_bad_quality_from_hist(fh.data(), lh.data(), total);
    }
    fprintf(stderr, "Bad quality threshold is '%c'\n", bad_q);
    for (int g = 0; g < gpus; ++g) rc_set_run_params(ctx[g], rate, bad_q);

    // pipeline: reader (this thread) -> one worker per GPU -> ordered writer (this thread)
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<Job>> order;                   // submission order, for the writer
    std::vector<std::deque<std::shared_ptr<Job>>> q((size_t)gpus);
    bool closing = false;
    std::vector<std::thread> workers;
    for (int g = 0; g < gpus; ++g) {
        workers.emplace_back([&, g]() {
            for (;;) {
                std::shared_ptr<Job> j;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return closing || !q[g].empty(); });
                    if (q[g].empty()) return;
                    j = q[g].front();
                    q[g].pop_front();
                }
                Batch &b = j->b;
                const size_t total = b.mode == 1 ? 2 * b.n() : b.n();
                b.ret.assign(total, 0);
                b.l.assign(total, 0);
                b.m.assign(total, 0);
                b.h.assign(total, 0);
                rc_batch rb;
                memset(&rb, 0, sizeof rb);
                rb.mode = b.mode;
                rb.n = b.n();
                rb.seq = b.seq.data();
                rb.qual = b.qual.data();
                rb.off = b.off.data();
                if (b.mode == 1) {
                    rb.seq2 = b.seq2.data();
                    rb.qual2 = b.qual2.data();
                    rb.off2 = b.off2.data();
                }
                rb.ret = b.ret.data();
                rb.l = b.l.data();
                rb.m = b.m.data();
                rb.h = b.h.data();
                int rc = rc_correct_batch(ctx[g], &rb);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    j->rc = rc;
                    if (rc) j->err = rc_last_error(ctx[g]);
                    j->done = true;
                }
                cv.notify_all();
            }
        });
    }

    uint64_t total_reads = 0, total_cor = 0;
    std::string line;
    auto drain = [&](bool all) {
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                if (order.empty()) return;
                if (!all && order.size() < (size_t)(2 * gpus) && !order.front()->done) return;
                cv.wait(lk, [&] { return order.front()->done; });
                j = order.front();
                order.pop_front();
            }
            if (j->rc) die("rcorrector: %s\n", j->err.c_str());
            Batch &b = j->b;
            ReadFile &f = rd.files[b.file], &g2 = rd.mates[b.file];
            const size_t n = b.n();
            auto upd = [&](int c) {  // UpdateSummary, main.cpp:73-79
                ++total_reads;
                if (c > 0) total_cor += (uint64_t)c;
            };
            if (b.mode == 1 && g_stdout) {  // main.cpp:487-495
                for (size_t u = 0; u < n; ++u) {
                    write_record(f, b.id[u], b.seq.data() + b.off[u], j->q1[u].c_str(), b.ret[u], b.l[u], b.m[u], b.h[u], line);
                    upd(b.ret[u]);
                    write_record(g2, b.id2[u], b.seq2.data() + b.off2[u], j->q2[u].c_str(), b.ret[n + u], b.l[n + u], b.m[n + u], b.h[n + u], line);
                    upd(b.ret[n + u]);
                }
            } else {
                for (size_t u = 0; u < n; ++u) {
                    write_record(f, b.id[u], b.seq.data() + b.off[u], j->q1[u].c_str(), b.ret[u], b.l[u], b.m[u], b.h[u], line);
                    upd(b.ret[u]);
                }
                if (b.mode == 1)
                    for (size_t u = 0; u < n; ++u) {
                        write_record(g2, b.id2[u], b.seq2.data() + b.off2[u], j->q2[u].c_str(), b.ret[n + u], b.l[n + u], b.m[n + u], b.h[n + u], line);
                        upd(b.ret[n + u]);
                    }
            }
        }
    };

    // reader loop.  The verbatim quality line of each record is kept next to the batch because the
    // output prints it unchanged even when it is longer or shorter than the sequence.
    {
        size_t seqno = 0;
        for (size_t fi = 0; fi < rd.files.size(); ++fi) {
            ReadFile &f = rd.files[fi];
            for (;;) {
                auto j = std::make_shared<Job>();
                Batch &b = j->b;
                b.file = (int)fi;
                b.mode = f.paired ? 1 : (f.interleaved ? 2 : 0);
                b.off.push_back(0);
                b.off2.push_back(0);
                static char idb[MAX_ID_LENGTH], sb[MAX_READ_LENGTH], qb[MAX_READ_LENGTH], plus[2048];
                auto read_one = [&](ReadFile &rf, std::vector<char> &sa, std::vector<char> &qa, std::vector<uint32_t> &off,
                                    std::vector<std::string> &ids, std::vector<std::string> &quals) -> bool {
                    if (!gzgets(rf.in, idb, MAX_ID_LENGTH)) return false;
                    sb[0] = qb[0] = 0;
                    gzgets(rf.in, sb, MAX_READ_LENGTH);
                    if (rf.fastq) {
                        gzgets(rf.in, plus, sizeof plus);
                        gzgets(rf.in, qb, MAX_READ_LENGTH);
                    }
                    size_t il = strlen(idb);
                    if (il && idb[il - 1] == '\n') idb[il - 1] = 0;
                    size_t len = strlen(sb);
                    if (len && sb[len - 1] == '\n') sb[len - 1] = 0;
                    if (rf.fastq && len && qb[len - 1] == '\n') qb[len - 1] = 0;
                    size_t sl = strlen(sb), ql = strlen(qb);
                    ids.emplace_back(idb);
                    quals.emplace_back(qb);
                    sa.insert(sa.end(), sb, sb + sl + 1);
                    size_t at = qa.size();
                    qa.resize(at + sl + 1, 0);
                    memcpy(qa.data() + at, qb, ql < sl ? ql : sl);
                    off.push_back((uint32_t)sa.size());
                    return true;
                };
                const size_t cap = batch_reads & ~(size_t)1;
                while (b.id.size() < cap) {
                    if (!read_one(f, b.seq, b.qual, b.off, b.id, j->q1)) break;
                    if (f.paired && !read_one(rd.mates[fi], b.seq2, b.qual2, b.off2, b.id2, j->q2))
                        die("ERROR: The files are not paired!\n");
                }
                if (b.id.empty()) {
                    if (f.paired && gzgets(rd.mates[fi].in, idb, MAX_ID_LENGTH)) die("ERROR: The files are not paired!\n");
                    break;
                }
                if (b.mode == 2 && (b.id.size() & 1)) die("ERROR: interleaved file %s holds an odd number of reads\n", f.path.c_str());
                {
                    std::lock_guard<std::mutex> lk(mu);
                    order.push_back(j);
                    q[seqno % (size_t)gpus].push_back(j);
                }
                ++seqno;
                cv.notify_all();
                drain(false);
            }
        }
    }
    drain(true);
    {
        std::lock_guard<std::mutex> lk(mu);
        closing = true;
    }
    cv.notify_all();
    for (auto &t : workers) t.join();

    for (size_t fi = 0; fi < rd.files.size(); ++fi) {
        for (ReadFile *f : {&rd.files[fi], &rd.mates[fi]}) {
            if (f->outz) gzclose(f->outz);
            if (f->out && f->out != stdout) fclose(f->out);
            if (f->in) gzclose(f->in);
        }
    }
    for (int g = 0; g < gpus; ++g) rc_destroy(ctx[g]);
    fprintf(stderr, "Processed %llu reads\n\tCorrected %llu bases.\n", (unsigned long long)total_reads, (unsigned long long)total_cor);
    return 0;
}
