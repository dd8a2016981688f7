/*
 * rc_oracle_io.c -- start-up path of the reference restated: jf_dump -> table, ERROR_RATE,
 * bad-quality threshold.  TEST INFRASTRUCTURE (see rc_oracle.h).
 */
#include "rc_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* whitespace-separated token reader with fscanf("%s") semantics (main.cpp:295,298,319,322) */
typedef struct {
    FILE *fp;
    char tok[4096];
} tokenizer;

static int next_token(tokenizer *z)
{
    int c, n = 0;
    do {
        c = fgetc(z->fp);
    } while (c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\f' || c == '\v');
    if (c == EOF) return 0;
    while (c != EOF && !(c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\f' || c == '\v')) {
        if (n < (int)sizeof(z->tok) - 1) z->tok[n++] = (char)c;
        c = fgetc(z->fp);
    }
    z->tok[n] = 0;
    return 1;
}

/* main.cpp:294-308 */
long rco_load_dump(rco_table *t, int k, const char *path)
{
    tokenizer z;
    long stored = 0;
    z.fp = fopen(path, "r");
    if (!z.fp) return -1;
    while (next_token(&z)) {
        int count = atoi(z.tok + 1);
        rco_kmer kc;
        if (!next_token(&z)) z.tok[0] = 0; /* fscanf failure leaves the old buffer; dumps are well formed */
        if (count <= 1) continue;
        rco_kmer_restart(&kc);
        for (int i = 0; z.tok[i]; ++i) rco_kmer_append(&kc, k, z.tok[i]);
        rco_table_put(t, &kc, count);
        ++stored;
    }
    fclose(z.fp);
    return stored;
}

static int cmp_double(const void *a, const void *b)
{
    double d = *(const double *)a - *(const double *)b; /* CompDouble, main.cpp:39-48 */
    return d > 0 ? 1 : (d < 0 ? -1 : 0);
}

/* main.cpp:310-358.  The IsValid() test at :323 looks at the k-mer object left over from the
 * PREVIOUS entry (initially: the last accepted entry of the load pass), restated literally. */
double rco_estimate_error_rate(const rco_table *t, int k, const char *path, double wk)
{
    const int rate_size = 100000;
    static const char nuc[4] = {'A', 'C', 'G', 'T'};
    tokenizer z;
    rco_kmer kc;
    double *rates, rate;
    int n = 0;

    rco_kmer_restart(&kc);
    z.fp = fopen(path, "r");
    if (!z.fp) return 0.01;
    while (next_token(&z)) { /* state the load pass leaves behind */
        int count = atoi(z.tok + 1);
        if (!next_token(&z)) z.tok[0] = 0;
        if (count <= 1) continue;
        rco_kmer_restart(&kc);
        for (int i = 0; z.tok[i]; ++i) rco_kmer_append(&kc, k, z.tok[i]);
    }
    rewind(z.fp);

    rates = (double *)malloc(sizeof(double) * (size_t)(rate_size + 2));
    rates[0] = 0;
    rates[1] = 0;
    double *r = rates + 1; /* r[-1] is readable, as rates[-1] happens to be in the reference */
    while (next_token(&z) && n < rate_size) {
        int max = 0, second = 0;
        if (!next_token(&z)) z.tok[0] = 0;
        if (kc.inv != -1) continue;
        rco_kmer_restart(&kc);
        for (int i = 0; z.tok[i]; ++i) rco_kmer_append(&kc, k, z.tok[i]);
        for (int i = 0; i < 4; ++i) {
            rco_kmer_shift_right(&kc, k, 1);
            rco_kmer_append(&kc, k, nuc[i]);
            int c = rco_table_get(t, &kc);
            if (c > max) {
                second = max;
                max = c;
            } else if (c > second)
                second = c;
        }
        if (max < 1000) continue;
        r[n++] = (double)second / (double)max;
    }
    fclose(z.fp);
    qsort(r, (size_t)n, sizeof(double), cmp_double);
    r[n] = r[n - 1];
    rate = r[(int)(n * wk)];
    if (rate == 0 || n < 100) rate = 0.01;
    free(rates);
    return rate;
}

/* GetBadQuality, main.cpp:108-127 */
char rco_bad_quality_from_hist(const int first_hist[300], const int last_hist[300], int total)
{
    int i, cnt = 0, t1, t2;
    for (i = 0; i < 300; ++i) {
        cnt += first_hist[i];
        if (cnt > total * 0.05) break;
    }
    t1 = i - 1;
    cnt = 0;
    for (i = 0; i < 300; ++i) {
        cnt += last_hist[i];
        if (cnt > total * 0.05) break;
    }
    t2 = i;
    return (char)(t2 < t1 ? t2 : t1);
}
