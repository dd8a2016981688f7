/*
 * rc_oracle_cli.c -- command-line front end of the CPU oracle, same flags / stderr lines /
 * output naming as the reference's stage-3 binary (main.cpp:50-71,165-268; Reads.h:39-75,
 * 140-157,360-421) so the two can be diffed file-for-file.  TEST INFRASTRUCTURE.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <zlib.h>

#include "rc_oracle.h"

#define MAX_FILES 100

typedef struct {
    char path[1024];
    int paired, interleaved;
    int fastq;
    gzFile in;
    int out_gz;
    FILE *out;
    gzFile outz;
} read_file;

static int use_stdout = 0;

static void die(const char *msg, const char *arg)
{
    fprintf(stderr, msg, arg);
    exit(1);
}

/* Reads.h:39-75 */
static void base_name(const char *in_c, char *out)
{
    char in[1024];
    int i, j, len;
    strcpy(in, in_c);
    len = (int)strlen(in);
    for (i = len; i >= 0 && in[i] != '.' && in[i] != '/'; --i)
        ;
    if (i >= 0 && !strcmp(&in[i], ".gz")) {
        int tmp = i;
        for (i = i - 1; i >= 0 && in[i] != '.' && in[i] != '/'; --i)
            ;
        in[tmp] = '\0';
        if (!(i >= 0 && (!strcmp(&in[i], ".fastq") || !strcmp(&in[i], ".fasta") ||
                         !strcmp(&in[i], ".fq") || !strcmp(&in[i], ".fa"))))
            i = tmp;
        in[tmp] = '.';
    }
    for (j = len; j >= 0 && in[j] != '/'; --j)
        ;
    if (i >= 0 && in[i] == '.') {
        in[i] = '\0';
        strcpy(out, in + j + 1);
    } else
        strcpy(out, in + j + 1);
}

static void open_file(read_file *f, const char *path, int paired, int interleaved, const char *od)
{
    char first[2048], name[1024], outp[4096];
    int len = (int)strlen(path);
    memset(f, 0, sizeof *f);
    strcpy(f->path, path);
    f->paired = paired;
    f->interleaved = interleaved;
    f->in = gzopen(path, "r");
    if (!f->in) die("ERROR: Could not access file %s\n", path);
    first[0] = 0;
    gzgets(f->in, first, sizeof first);
    if (first[0] == '>')
        f->fastq = 0;
    else if (first[0] == '@')
        f->fastq = 1;
    else {
        fprintf(stderr, "\"%s\"'s format is wrong: %s\n", path, first);
        exit(1);
    }
    gzrewind(f->in);
    base_name(path, name);
    f->out_gz = (len >= 2 && path[len - 2] == 'g' && path[len - 1] == 'z');
    snprintf(outp, sizeof outp, "%s/%s.cor.%s%s", od, name, f->fastq ? "fq" : "fa", f->out_gz ? ".gz" : "");
    if (use_stdout) {
        f->out = stdout;
        f->out_gz = 0;
    } else if (f->out_gz) {
        f->outz = gzopen(outp, "w1");
        if (!f->outz) die("ERROR: Could not access file %s\n", outp);
    } else {
        f->out = fopen(outp, "w");
        if (!f->out) die("ERROR: Could not access file %s\n", outp);
    }
}

static void emit(read_file *f, const char *s)
{
    if (f->out_gz)
        gzwrite(f->outz, s, (unsigned)strlen(s));
    else
        fputs(s, f->out);
}

static void strip_nl(char *s)
{
    size_t n = strlen(s);
    if (n && s[n - 1] == '\n') s[n - 1] = 0;
}

/* Reads.h:224-266 */
static int next_record(read_file *f, char *id, char *seq, char *qual)
{
    char plus[2048];
    if (!gzgets(f->in, id, RCO_MAX_ID_LENGTH)) return 0;
    seq[0] = qual[0] = 0;
    gzgets(f->in, seq, RCO_MAX_READ_LENGTH);
    if (f->fastq) {
        gzgets(f->in, plus, sizeof plus);
        gzgets(f->in, qual, RCO_MAX_READ_LENGTH);
    }
    strip_nl(id);
    {
        size_t len = strlen(seq);
        if (len && seq[len - 1] == '\n') seq[len - 1] = 0;
        if (f->fastq && len && qual[len - 1] == '\n') qual[len - 1] = 0;
    }
    return 1;
}

/* Reads.h:360-421 */
static void write_record(read_file *f, const char *id, const char *seq, const char *qual, int cor,
                         int l, int m, int h)
{
    static char line[3 * RCO_MAX_READ_LENGTH + 3 * RCO_MAX_ID_LENGTH];
    const char *tag = cor == 0 ? "" : (cor == -1 ? " unfixable_error" : (cor > 0 ? " cor" : ""));
    snprintf(line, sizeof line, "%s l:%d m:%d h:%d%s\n%s\n", id, l, m, h, tag, seq);
    emit(f, line);
    if (f->fastq) {
        snprintf(line, sizeof line, "+\n%s\n", qual);
        emit(f, line);
    }
}

typedef struct {
    char id[RCO_MAX_ID_LENGTH], seq[RCO_MAX_READ_LENGTH], qual[RCO_MAX_READ_LENGTH];
    int cor, l, m, h;
} rec;

int main(int argc, char **argv)
{
    static read_file rf[MAX_FILES], pf[MAX_FILES];
    int nrf = 0, i;
    rco_params P;
    const char *dump = NULL, *od = "./";
    double wk = 0.95;
    int threads = 1, verbose = 0;
    unsigned long long total_reads = 0, total_cor = 0;

    memset(&P, 0, sizeof P);
    P.k = 23;
    P.max_fix_per_k = 4;
    if (argc == 1) {
        fprintf(stderr, "Usage: oracle_cli [-r f|-p f1 f2|-i f] -c jf_dump -k K [-od DIR -t T -maxcorK N -wk F -stdout -verbose]\n");
        return 0;
    }
    for (i = 1; i < argc; ++i) { /* main.cpp:165-247 */
        if (!strcmp("-r", argv[i]) || !strcmp("-i", argv[i]))
            ++i;
        else if (!strcmp("-p", argv[i]))
            i += 2;
        else if (!strcmp("-od", argv[i])) {
            mkdir(argv[i + 1], 0700);
            od = argv[++i];
        } else if (!strcmp("-c", argv[i]))
            dump = argv[++i];
        else if (!strcmp("-k", argv[i]))
            P.k = atoi(argv[++i]);
        else if (!strcmp("-t", argv[i]))
            threads = atoi(argv[++i]);
        else if (!strcmp("-maxcor", argv[i]))
            ++i;
        else if (!strcmp("-maxcorK", argv[i]))
            P.max_fix_per_k = atoi(argv[++i]);
        else if (!strcmp("-wk", argv[i]))
            wk = atof(argv[++i]);
        else if (!strcmp("-stdout", argv[i]))
            use_stdout = 1;
        else if (!strcmp("-verbose", argv[i]))
            verbose = 1;
        else {
            fprintf(stderr, "Unknown argument: %s\n", argv[i]);
            return 0;
        }
    }
    for (i = 1; i < argc; ++i) { /* main.cpp:250-268 */
        if (nrf >= MAX_FILES) die("The number of read files exceeds the limit %s.\n", "100");
        if (!strcmp("-r", argv[i])) {
            open_file(&rf[nrf], argv[i + 1], 0, 0, od);
            ++nrf;
            ++i;
        } else if (!strcmp("-p", argv[i])) {
            open_file(&rf[nrf], argv[i + 1], 1, 0, od);
            open_file(&pf[nrf], argv[i + 2], 1, 0, od);
            ++nrf;
            i += 2;
        } else if (!strcmp("-i", argv[i])) {
            open_file(&rf[nrf], argv[i + 1], 0, 1, od);
            ++nrf;
            ++i;
        }
    }
    if (!dump) die("Could not open file %s\n", "(no -c given)");

    rco_table *T = rco_table_new(P.k, 1 << 20);
    long stored = rco_load_dump(T, P.k, dump);
    if (stored < 0) die("Could not open file %s\n", dump);
    fprintf(stderr, "Stored %d kmers\n", (int)stored);
    P.error_rate = rco_estimate_error_rate(T, P.k, dump, wk);
    fprintf(stderr, "Weak kmer threshold rate: %lf (estimated from %.3lf/1 of the chosen kmers)\n",
            P.error_rate, wk);

    { /* GetBadQuality, main.cpp:88-128: first <=1M records over the primary files in order */
        int fh[300] = {0}, lh[300] = {0}, total = 0;
        static rec r;
        if (nrf > 0 && rf[0].fastq) {
            for (int fi = 0; fi < nrf && total < 1000000; ++fi) {
                while (total < 1000000 && next_record(&rf[fi], r.id, r.seq, r.qual)) {
                    ++lh[(int)r.qual[strlen(r.seq) - 1]];
                    ++fh[(int)r.qual[0]];
                    ++total;
                }
                gzrewind(rf[fi].in);
            }
            P.bad_qual = rco_bad_quality_from_hist(fh, lh, total);
        } else
            P.bad_qual = 0;
        fprintf(stderr, "Bad quality threshold is '%c'\n", P.bad_qual);
    }
    if (verbose) {
        P.verbose_fp = stdout;
        threads = 1;
    }

    /* correction, unit by unit in input order (main.cpp:368-438; results do not depend on -t) */
    {
        const int CH = 65536;
        rec *a = (rec *)malloc(sizeof(rec) * (size_t)CH), *b = (rec *)malloc(sizeof(rec) * (size_t)CH);
        for (int fi = 0; fi < nrf; ++fi) {
            read_file *f = &rf[fi], *g = &pf[fi];
            for (;;) {
                int n = 0;
                while (n < CH && next_record(f, a[n].id, a[n].seq, a[n].qual)) ++n;
                if (f->paired) {
                    int n2 = 0;
                    while (n2 < n && next_record(g, b[n2].id, b[n2].seq, b[n2].qual)) ++n2;
                    if (n2 != n) die("ERROR: The files are not paired!%s\n", "");
                }
                if (n == 0) break;
                int step = f->interleaved ? 2 : 1;
#pragma omp parallel for schedule(dynamic, 64) if (threads > 1 && !verbose) num_threads(threads)
                for (int u = 0; u < n; u += step) {
                    rec *m1 = &a[u], *m2 = f->paired ? &b[u] : (f->interleaved && u + 1 < n ? &a[u + 1] : NULL);
                    int t = -1;
                    if (m2) {
                        int t1 = rco_strong_trusted_threshold(&P, T, m1->seq);
                        int t2 = rco_strong_trusted_threshold(&P, T, m2->seq);
                        t = t1 < t2 ? t1 : t2;
                    }
                    m1->cor = rco_error_correction(&P, T, m1->id, m1->seq, m1->qual, t);
                    rco_kmer_information(&P, T, m1->seq, &m1->l, &m1->m, &m1->h);
                    /* -verbose -stdout: the reference's -t 1 loop writes every record right after
                     * its own trace (main.cpp:401-432), trace and record share stdout */
                    if (verbose && use_stdout) {
                        fflush(stdout);
                        write_record(f, m1->id, m1->seq, m1->qual, m1->cor, m1->l, m1->m, m1->h);
                        if (f->out == stdout) fflush(stdout);
                    }
                    if (m2) {
                        m2->cor = rco_error_correction(&P, T, m2->id, m2->seq, m2->qual, t);
                        rco_kmer_information(&P, T, m2->seq, &m2->l, &m2->m, &m2->h);
                        if (verbose && use_stdout) {
                            fflush(stdout);
                            write_record(f->paired ? g : f, m2->id, m2->seq, m2->qual, m2->cor, m2->l, m2->m, m2->h);
                        }
                    }
                }
                if (verbose && use_stdout) {  /* already written; only the summary is left */
                    for (int u = 0; u < n; ++u) {
                        ++total_reads;
                        if (a[u].cor > 0) total_cor += (unsigned)a[u].cor;
                        if (f->paired) {
                            ++total_reads;
                            if (b[u].cor > 0) total_cor += (unsigned)b[u].cor;
                        }
                    }
                    continue;
                }
                for (int u = 0; u < n; ++u) {
                    write_record(f, a[u].id, a[u].seq, a[u].qual, a[u].cor, a[u].l, a[u].m, a[u].h);
                    ++total_reads;
                    if (a[u].cor > 0) total_cor += (unsigned)a[u].cor;
                    if (f->paired && use_stdout) {
                        write_record(g, b[u].id, b[u].seq, b[u].qual, b[u].cor, b[u].l, b[u].m, b[u].h);
                        ++total_reads;
                        if (b[u].cor > 0) total_cor += (unsigned)b[u].cor;
                    }
                }
                if (f->paired && !use_stdout)
                    for (int u = 0; u < n; ++u) {
                        write_record(g, b[u].id, b[u].seq, b[u].qual, b[u].cor, b[u].l, b[u].m, b[u].h);
                        ++total_reads;
                        if (b[u].cor > 0) total_cor += (unsigned)b[u].cor;
                    }
            }
        }
        free(a);
        free(b);
    }
    for (i = 0; i < nrf; ++i) {
        if (rf[i].outz) gzclose(rf[i].outz);
        if (rf[i].out && rf[i].out != stdout) fclose(rf[i].out);
        if (pf[i].outz) gzclose(pf[i].outz);
        if (pf[i].out && pf[i].out != stdout) fclose(pf[i].out);
    }
    fprintf(stderr, "Processed %llu reads\n\tCorrected %llu bases.\n", total_reads, total_cor);
    rco_table_free(T);
    return 0;
}
