/*
 * rc_oracle.c -- CPU restatement of Rcorrector's k-mer table, rolling k-mer code and per-read
 * error correction.  TEST INFRASTRUCTURE (see rc_oracle.h).  Written from the behavioural
 * spec in SURVEY.md §9; each function names the reference lines it restates so a reviewer
 * can check parity.  Reference paths are relative to /root/reference (v1.0.7).
 */
#define _GNU_SOURCE /* posix_memalign, madvise, syscall (placement of the table's memory: big_alloc, rco_set_interleave) */
#include "rc_oracle.h"

#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * base coding, main.cpp:17-22: A0 C1 G2 T3, every other upper-case letter -1.  Characters
 * outside 'A'..'Z' index out of bounds in the reference (undefined); we define them as
 * invalid bases too.
 * ---------------------------------------------------------------------------------------- */
static inline int nuc_to_num(char c)
{
    switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
    }
}
static const char NUM_TO_NUC[4] = {'A', 'C', 'G', 'T'};

/* ------------------------------------------------------------------------------------------
 * KmerCode.h:14-27 (mask), :38 (Restart), KmerCode.cpp:7-42 (Append/Prepend/ShiftRight),
 * KmerCode.h:58-71 (canonical)
 * ---------------------------------------------------------------------------------------- */
uint64_t rco_kmer_mask(int k) { return k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull); }

void rco_kmer_restart(rco_kmer *km)
{
    km->code = 0;
    km->inv = -1;
}

void rco_kmer_append(rco_kmer *km, int k, char c)
{
    int n = nuc_to_num(c);
    if (km->inv != -1) ++km->inv;
    km->code = ((km->code << 2) & rco_kmer_mask(k)) | (uint64_t)(n & 3);
    if (n == -1) km->inv = 0;
    if (km->inv >= k) km->inv = -1;
}

void rco_kmer_shift_right(rco_kmer *km, int k, int n)
{
    if (km->inv != -1) km->inv -= n;
    km->code = (km->code >> (2 * n)) & (rco_kmer_mask(k) >> (2 * n));
    if (km->inv < 0) km->inv = -1;
}

void rco_kmer_prepend(rco_kmer *km, int k, char c)
{
    int n = nuc_to_num(c);
    rco_kmer_shift_right(km, k, 1);
    if (n == -1) km->inv = k - 1;
    km->code = (km->code | ((uint64_t)(n & 3) << (2 * (k - 1)))) & rco_kmer_mask(k);
}

uint64_t rco_kmer_canonical(const rco_kmer *km, int k)
{
    uint64_t rc = 0, c = km->code;
    for (int i = 0; i < k; ++i) {
        rc = (rc << 2) | (3ull - ((c >> (2 * i)) & 3ull));
    }
    return rc < c ? rc : c;
}

/* ------------------------------------------------------------------------------------------
 * Store.h:17-88 -- the reference uses std::unordered_map<uint64_t,int>; any exact map is
 * equivalent.  Open addressing, linear probing, slot empty <=> used[i]==0.
 * ---------------------------------------------------------------------------------------- */
struct rco_table {
    int k;
    size_t cap; /* power of two */
    size_t n;
    uint64_t *keys;
    int32_t *vals;
    uint8_t *used;
};

static inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

/* The table's three arrays are gigabytes that every lookup touches at random: with 4 KB pages each touch is a TLB miss and a page
 * walk on top of the cache miss.  Large arrays are therefore 2 MB-aligned and offered to the kernel for transparent huge pages
 * (a hint: a host that has them off loses nothing).  Placement only -- no effect on any result. */
#include <sys/mman.h>
static void *big_alloc(size_t bytes, int zero)
{
    void *p = NULL;
    if (bytes >= ((size_t)8 << 20)) {
        size_t rounded = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        if (posix_memalign(&p, (size_t)2 << 20, rounded) != 0) return NULL;
#ifdef MADV_HUGEPAGE
        (void)madvise(p, rounded, MADV_HUGEPAGE);
#endif
        if (zero) memset(p, 0, bytes);
        return p;
    }
    return zero ? calloc(bytes, 1) : malloc(bytes);
}

static void table_alloc(rco_table *t, size_t cap)
{
    t->cap = cap;
    t->keys = (uint64_t *)big_alloc(cap * sizeof(uint64_t), 0);
    t->vals = (int32_t *)big_alloc(cap * sizeof(int32_t), 0);
    t->used = (uint8_t *)big_alloc(cap, 1);
    if (!t->keys || !t->vals || !t->used) {
        fprintf(stderr, "rc_oracle: out of memory\n");
        abort();
    }
}

rco_table *rco_table_new(int k, size_t expected)
{
    rco_table *t = (rco_table *)calloc(1, sizeof(*t));
    size_t cap = 1024;
    while (cap < expected * 2) cap <<= 1;
    t->k = k;
    table_alloc(t, cap);
    return t;
}

void rco_table_free(rco_table *t)
{
    if (!t) return;
    free(t->keys);
    free(t->vals);
    free(t->used);
    free(t);
}

static void table_insert_raw(rco_table *t, uint64_t key, int val)
{
    size_t m = t->cap - 1, i = (size_t)mix64(key) & m;
    while (t->used[i]) {
        if (t->keys[i] == key) {
            t->vals[i] = val; /* Store.h:55, later Put overwrites */
            return;
        }
        i = (i + 1) & m;
    }
    t->used[i] = 1;
    t->keys[i] = key;
    t->vals[i] = val;
    ++t->n;
}

static void table_grow(rco_table *t)
{
    rco_table old = *t;
    table_alloc(t, old.cap * 2);
    t->n = 0;
    for (size_t i = 0; i < old.cap; ++i)
        if (old.used[i]) table_insert_raw(t, old.keys[i], old.vals[i]);
    free(old.keys);
    free(old.vals);
    free(old.used);
}

void rco_table_put_canon(rco_table *t, uint64_t canon, int count)
{
    if ((t->n + 1) * 2 > t->cap) table_grow(t);
    table_insert_raw(t, canon, count);
}

void rco_table_put(rco_table *t, const rco_kmer *km, int count)
{
    if (km->inv != -1) return; /* Store.h:53-54 */
    rco_table_put_canon(t, rco_kmer_canonical(km, t->k), count);
}

static inline int table_get_canon(const rco_table *t, uint64_t key)
{
    size_t m = t->cap - 1, i = (size_t)mix64(key) & m;
    while (t->used[i]) {
        if (t->keys[i] == key) return t->vals[i];
        i = (i + 1) & m;
    }
    return 0;
}

int rco_table_get(const rco_table *t, const rco_kmer *km)
{
    if (km->inv != -1) return 0; /* Store.h:61-62 */
    return table_get_canon(t, rco_kmer_canonical(km, t->k));
}

void rco_table_put_many(rco_table *t, const uint64_t *canon, const int32_t *counts, size_t n)
{
    for (size_t i = 0; i < n; ++i) rco_table_put_canon(t, canon[i], counts[i]);
}

size_t rco_table_size(const rco_table *t) { return t->n; }

size_t rco_table_export(const rco_table *t, uint64_t *codes, int32_t *counts, size_t cap)
{
    size_t w = 0;
    for (size_t i = 0; i < t->cap && w < cap; ++i)
        if (t->used[i]) {
            codes[w] = t->keys[i];
            counts[w] = t->vals[i];
            ++w;
        }
    return w;
}

/* ------------------------------------------------------------------------------------------
 * GetBound, ErrorCorrection.cpp:139-142.  The reference is x86-64 SSE2: separate mul/add,
 * correctly rounded sqrt, and cvttsd2si for the implicit double->int conversions at
 * :164,798,815,821,1282 -- which yields INT_MIN for NaN (c = -1 => sqrt(negative)) and for
 * out-of-range values.  Spelled out here so the result does not depend on the C compiler.
 * ---------------------------------------------------------------------------------------- */
double rco_get_bound(const rco_params *p, int c)
{
    volatile double ce = (double)c * p->error_rate; /* volatile: forbid FMA contraction */
    volatile double s = 6.0 * sqrt(ce);
    volatile double r = ce + s;
    return r + 1.0;
}

static inline int dbl_to_int_x86(double x)
{
    if (isnan(x) || x >= 2147483648.0 || x <= -2147483649.0) return INT_MIN;
    return (int)x;
}

int rco_get_bound_int(const rco_params *p, int c) { return dbl_to_int_x86(rco_get_bound(p, c)); }

/* IsPolyA, ErrorCorrection.cpp:53-71 */
static int is_poly_a(const char *buf, int length, int threshold)
{
    int cnt = 0;
    for (int i = 0; i < length; ++i)
        if (buf[i] == 'A') ++cnt;
    if (cnt >= length - threshold) return 1;
    cnt = 0;
    for (int i = 0; i < length; ++i)
        if (buf[i] == 'T') ++cnt;
    if (cnt >= length - threshold) return 1;
    return 0;
}

static int cmp_int(const void *a, const void *b)
{
    /* CompInt, ErrorCorrection.cpp:37-40, without its overflow for huge counts */
    int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

/* InferPosThreshold, ErrorCorrection.cpp:144-173.  direction -1 = left (Prepend), 1 = right */
static int infer_pos_threshold(const rco_params *p, const rco_table *t, const rco_kmer *kc,
                               int direction, int upper)
{
    int max_cnt = 0;
    for (int i = 0; i < 4; ++i) {
        rco_kmer tmp = *kc;
        if (direction == -1)
            rco_kmer_prepend(&tmp, p->k, NUM_TO_NUC[i]);
        else
            rco_kmer_append(&tmp, p->k, NUM_TO_NUC[i]);
        int c = rco_table_get(t, &tmp);
        if (c > max_cnt) max_cnt = c;
    }
    int ret = rco_get_bound_int(p, max_cnt);
    if (ret < 1) ret = 1;
    if (upper > ret || upper <= 0) return ret;
    return upper;
}

/* state shared by one SearchPaths_* call tree (the reference passes these by reference) */
typedef struct {
    const rco_params *p;
    const rco_table *t;
    const char *seq;
    int start, to;
    int *fix;       /* scratch path, the reference's iBuffer */
    int *best_fix;  /* the reference's fix[] */
    int max_fix_cnt;
    int best_fix_cnt; /* carried across segments, ErrorCorrection.cpp:1120 */
    int best_bottleneck;
    int *top2;
    const unsigned char *strong; /* isStrongTrusted */
    const unsigned char *polya;  /* isPolyAKmer */
    int trial_cnt;
} search_ctx;

/* terminal bookkeeping shared by both directions, ErrorCorrection.cpp:243-284 / :483-523 */
static void search_terminal(search_ctx *s, int pos, int t, int fix_cnt, int bottleneck, int right)
{
    if (bottleneck < t) ++fix_cnt;
    if (fix_cnt < s->max_fix_cnt) {
        s->top2[0] = bottleneck;
        s->top2[1] = -1;
    } else if (fix_cnt == s->max_fix_cnt) {
        if (bottleneck > s->top2[0]) {
            s->top2[1] = s->top2[0];
            s->top2[0] = bottleneck;
        } else if (bottleneck > s->top2[1]) {
            s->top2[1] = bottleneck;
        }
    }
    if (fix_cnt < s->max_fix_cnt || (fix_cnt == s->max_fix_cnt && bottleneck > s->best_bottleneck)) {
        if (fix_cnt < s->max_fix_cnt) s->trial_cnt = -(s->max_fix_cnt - fix_cnt + 1) * RCO_MAX_TRIAL;
        if (right) {
            for (int i = s->start; i < pos; ++i) s->best_fix[i] = s->fix[i];
        } else {
            for (int i = s->start; i > pos; --i) s->best_fix[i] = s->fix[i];
        }
        s->max_fix_cnt = fix_cnt;
        s->best_bottleneck = bottleneck;
        s->best_fix_cnt = 1;
    } else if (fix_cnt == s->max_fix_cnt && bottleneck == s->best_bottleneck) {
        s->best_fix_cnt += 1;
    }
}

/* the shared entry test, ErrorCorrection.cpp:211-224 / :453-467; returns 1 = give up this node */
static int search_entry_gate(search_ctx *s, int fix_cnt)
{
    if (s->trial_cnt > RCO_MAX_TRIAL) {
        if (s->max_fix_cnt > 2) {
            --s->max_fix_cnt;
            s->best_fix_cnt = 0;
            s->best_bottleneck = -1;
            s->trial_cnt = 0;
        } else
            return 1;
    }
    if (fix_cnt > s->max_fix_cnt) return 1;
    return 0;
}

/* SearchPaths_Right, ErrorCorrection.cpp:201-442 */
static void search_right(search_ctx *s, int pos, int t, int fix_cnt, int bottleneck, rco_kmer kc)
{
    const int k = s->p->k;
    const char *seq = s->seq;
    int extension = 0, cnt, i;

    if (search_entry_gate(s, fix_cnt)) return;
    if (pos >= s->to) {
        search_terminal(s, pos, t, fix_cnt, bottleneck, 1);
        return;
    }
    int threshold = infer_pos_threshold(s->p, s->t, &kc, 1, t); /* :287 */
    rco_kmer tmp = kc;

    if (nuc_to_num(seq[pos]) != -1) { /* :294 */
        rco_kmer_append(&tmp, k, seq[pos]);
        cnt = rco_table_get(s->t, &tmp);
        if (cnt >= threshold) { /* :302-312 keep the base */
            s->fix[pos] = -1;
            ++extension;
            search_right(s, pos + 1, t, fix_cnt, cnt < bottleneck ? cnt : bottleneck, tmp);
        } else if (threshold == 1 && t <= 2) { /* :313-338 accidental gap */
            int steps = 0;
            for (i = pos; cnt < threshold && steps < k; ++steps) {
                ++i;
                if (i >= s->to) break;
                rco_kmer_append(&tmp, k, seq[i]);
                cnt = rco_table_get(s->t, &tmp);
            }
            if (steps < k && i < s->to) {
                for (int j = pos; j <= i; ++j) s->fix[j] = -1;
                ++extension;
                search_right(s, i + 1, t, fix_cnt + 1, bottleneck, tmp);
            }
        }
    }

    if (!s->strong[pos] && !s->polya[pos - k + 1]) { /* :343-372 substitutions */
        for (i = 0; i < 4; ++i) {
            if (NUM_TO_NUC[i] == seq[pos]) continue;
            tmp = kc;
            rco_kmer_append(&tmp, k, NUM_TO_NUC[i]);
            cnt = rco_table_get(s->t, &tmp);
            if (cnt >= threshold) {
                int fc = fix_cnt;
                s->fix[pos] = i;
                ++s->trial_cnt;
                ++extension;
                if (nuc_to_num(seq[pos]) != -1) ++fc;
                search_right(s, pos + 1, t, fc, cnt < bottleneck ? cnt : bottleneck, tmp);
            }
        }
    }

    if (extension == 0) { /* :393-441 jump over an unfixable stretch */
        int pen;
        tmp = kc;
        for (i = pos; i < s->to; ++i) {
            threshold = infer_pos_threshold(s->p, s->t, &tmp, 1, t);
            rco_kmer_append(&tmp, k, seq[i]);
            cnt = rco_table_get(s->t, &tmp);
            s->fix[i] = -1;
            if (cnt >= threshold) break;
        }
        if (seq[i])
            pen = i - pos - k + 1;
        else
            pen = (i - pos) / 2;
        if (pen <= 0) pen = 1;
        fix_cnt += pen;
        if (i >= s->to) i -= 1;
        search_right(s, i + 1, t, fix_cnt, bottleneck, tmp);
    }
}

/* SearchPaths_Left, ErrorCorrection.cpp:444-678 */
static void search_left(search_ctx *s, int pos, int t, int fix_cnt, int bottleneck, rco_kmer kc)
{
    const int k = s->p->k;
    const char *seq = s->seq;
    int extension = 0, cnt, i;

    if (search_entry_gate(s, fix_cnt)) return;
    if (pos < s->to) {
        search_terminal(s, pos, t, fix_cnt, bottleneck, 0);
        return;
    }
    int threshold = infer_pos_threshold(s->p, s->t, &kc, -1, t); /* :525 */
    rco_kmer tmp = kc;

    if (nuc_to_num(seq[pos]) != -1) { /* :532 */
        rco_kmer_prepend(&tmp, k, seq[pos]);
        cnt = rco_table_get(s->t, &tmp);
        if (cnt >= threshold) { /* :539-549; NB passes `threshold` down as the new t */
            s->fix[pos] = -1;
            ++extension;
            search_left(s, pos - 1, threshold, fix_cnt, cnt < bottleneck ? cnt : bottleneck, tmp);
        } else if (threshold == 1 && t <= 2) { /* :550-573 */
            int steps = 0;
            for (i = pos; cnt < threshold && steps < k; ++steps) {
                --i;
                if (i < s->to) break;
                rco_kmer_prepend(&tmp, k, seq[i]);
                cnt = rco_table_get(s->t, &tmp);
            }
            if (steps < k && i >= s->to) {
                for (int j = i; j <= pos; ++j) s->fix[j] = -1;
                ++extension;
                search_left(s, i - 1, threshold, fix_cnt + 1, bottleneck, tmp);
            }
        }
    }

    if (!s->strong[pos] && !s->polya[pos]) { /* :577-608 */
        for (i = 0; i < 4; ++i) {
            if (NUM_TO_NUC[i] == seq[pos]) continue;
            tmp = kc;
            rco_kmer_prepend(&tmp, k, NUM_TO_NUC[i]);
            cnt = rco_table_get(s->t, &tmp);
            if (cnt >= threshold) {
                int fc = fix_cnt;
                s->fix[pos] = i;
                ++s->trial_cnt;
                if (nuc_to_num(seq[pos]) != -1) ++fc;
                ++extension;
                search_left(s, pos - 1, threshold, fc, cnt < bottleneck ? cnt : bottleneck, tmp);
            }
        }
    }

    if (extension == 0) { /* :629-677 */
        int pen;
        tmp = kc;
        for (i = pos; i >= s->to; --i) {
            threshold = infer_pos_threshold(s->p, s->t, &tmp, -1, t);
            rco_kmer_prepend(&tmp, k, seq[i]);
            cnt = rco_table_get(s->t, &tmp);
            s->fix[i] = -1;
            if (cnt >= threshold) break;
        }
        if (i >= 0)
            pen = pos - i - k + 1;
        else
            pen = pos - 1;
        if (pen <= 0) pen = 1;
        fix_cnt += pen;
        if (i <= s->to) ++i; /* :672 -- re-visits `to` when the recovery is exactly there */
        search_left(s, i - 1, threshold, fix_cnt, bottleneck, tmp);
    }
}

/* counts[], ErrorCorrection.cpp:716-723 */
int rco_kmer_counts(const rco_params *p, const rco_table *t, const char *seq, int *counts)
{
    const int k = p->k;
    int len = (int)strlen(seq), kcnt = 0, i;
    rco_kmer kc;
    if (len < k) return 0;
    rco_kmer_restart(&kc);
    for (i = 0; i < k - 1; ++i) rco_kmer_append(&kc, k, seq[i]);
    for (; seq[i]; ++i, ++kcnt) {
        rco_kmer_append(&kc, k, seq[i]);
        counts[kcnt] = rco_table_get(t, &kc);
    }
    return kcnt;
}

/* the read screens, ErrorCorrection.cpp:735-755 / :1507-1527; 1 = screened out */
static int read_screened(const char *seq, int len, int k)
{
    int n = 0, a = 0, tt = 0;
    for (int i = 0; i < len; ++i) {
        if (seq[i] == 'N') ++n;
        if (seq[i] == 'A') ++a;
        if (seq[i] == 'T') ++tt;
    }
    return n > 5 || a > len - k || tt > len - k;
}

/* poly-A masked, sorted copy of counts, ErrorCorrection.cpp:774-784 / :1247-1257 / :1530-1540 */
static void masked_sorted(const char *seq, int k, const int *counts, int kcnt, int *out)
{
    int thr = 7;
    if (k / 2 > thr) thr = k / 2;
    for (int i = 0; i < kcnt; ++i) out[i] = is_poly_a(seq + i, k, thr) ? -1 : counts[i];
    qsort(out, kcnt, sizeof(int), cmp_int);
}

/* GetStrongTrustedThreshold, ErrorCorrection.cpp:1482-1565 */
int rco_strong_trusted_threshold(const rco_params *p, const rco_table *t, const char *seq)
{
    int counts[RCO_MAX_READ_LENGTH], buf[RCO_MAX_READ_LENGTH];
    const int k = p->k;
    int len = (int)strlen(seq), i;
    if (len < k) return -1;
    int kcnt = rco_kmer_counts(p, t, seq, counts);
    if (read_screened(seq, len, k)) return -1;
    masked_sorted(seq, k, counts, kcnt, buf);
    for (i = kcnt - 1; i >= 1; --i)
        if (buf[i] > 2 * buf[i - 1] && buf[i] > 10) break;
    if (i >= 1) return buf[i];
    for (i = 0; i < kcnt; ++i)
        if (buf[i] > 0) break;
    return buf[(i + kcnt - 1) / 2];
}

typedef struct {
    int from, to, lanchor, ranchor;
    int top2[2];
} segment;
typedef struct {
    int from, to;
} island;

/* ErrorCorrection, ErrorCorrection.cpp:682-1480 */
int rco_error_correction(const rco_params *p, const rco_table *t, const char *id, char *seq,
                         const char *qual, int pair_t)
{
    const int k = p->k;
    int counts[RCO_MAX_READ_LENGTH], ibuf[RCO_MAX_READ_LENGTH], fix[RCO_MAX_READ_LENGTH];
    int dbuf2[RCO_MAX_READ_LENGTH + 1]; /* dBuffer x2 (the reference holds multiples of 0.5) */
    unsigned char strong_base[RCO_MAX_READ_LENGTH], polya[RCO_MAX_READ_LENGTH];
    static __thread segment seg[RCO_MAX_READ_LENGTH];
    static __thread island isl[RCO_MAX_READ_LENGTH];
    int seg_cnt, isl_cnt;
    int i, j, kcnt, len;
    int tstart = 0, tend = 0, longest, trust, strong;
    int total_fix = 0, allowed_fix, bad_segment_cnt = 0, best_bottleneck;
    int unfixable, force_next, flag;
    FILE *vf = (FILE *)p->verbose_fp;

    if (vf) fprintf(vf, "%s\n", id ? id : ""); /* :686-689 */
    len = (int)strlen(seq);
    if (len < k) return -1; /* :713 */
    kcnt = rco_kmer_counts(p, t, seq, counts);
    if (read_screened(seq, len, k)) return -1; /* :735-755 */

    if (vf) { /* :759-770 */
        fprintf(vf, "Before correction:\n%s\n", seq);
        for (i = 0; i < kcnt; ++i) fprintf(vf, "%d ", counts[i] != 0 ? counts[i] : 1);
        fprintf(vf, "\n");
    }

    /* initial thresholds, :772-842 */
    masked_sorted(seq, k, counts, kcnt, ibuf);
    for (i = kcnt - 1; i >= 1; --i)
        if (ibuf[i] > 2 * ibuf[i - 1] && ibuf[i] > 10) break;
    flag = 0;
    if (i >= 1) {
        trust = rco_get_bound_int(p, ibuf[i]);
        strong = ibuf[i];
        if (strong >= 20 && ibuf[i - 1] == 2 && trust < 3) {
            flag = 1;
            trust = 3;
        }
    } else {
        for (i = 0; i < kcnt; ++i)
            if (ibuf[i] > 0) break;
        strong = ibuf[(i + kcnt - 1) / 2];
        trust = rco_get_bound_int(p, strong);
    }
    if (pair_t >= 1 && strong > pair_t) { /* :818-823 */
        if (!flag || pair_t < 20) trust = rco_get_bound_int(p, pair_t);
        strong = pair_t;
    }
    for (i = 0; i < kcnt; ++i) polya[i] = (unsigned char)is_poly_a(seq + i, k, 2); /* :824-830 */
    if (trust < 2) trust = 2; /* :841-842 */

    int iter = 0;
    for (;;) { /* :854-1291 */
        if (vf) fprintf(vf, "strong trust threshold=%d threshold=%d\n", strong, trust);
        allowed_fix = len;
        total_fix = 0;
        unfixable = 0;
        force_next = 0;
        isl_cnt = 0;
        longest = -1;
        j = 0;
        memset(strong_base, 0, (size_t)len);
        for (i = 0; i < len; ++i) ibuf[i] = -1;

        /* trusted k-mer islands, :870-931 */
        for (i = 0; i < kcnt; ++i) {
            if (counts[i] >= strong && !polya[i]) {
                ++j;
            } else {
                if (j > longest) {
                    longest = j;
                    tstart = i - longest;
                    tend = i - 1;
                }
                if (j >= 2) {
                    isl[isl_cnt].from = i - j;
                    isl[isl_cnt].to = i - 1;
                    ++isl_cnt;
                }
                j = 0;
            }
        }
        if (j > longest) {
            longest = j;
            tstart = i - longest;
            tend = i - 1;
        }
        if (j >= 2) {
            isl[isl_cnt].from = i - j;
            isl[isl_cnt].to = i - 1;
            ++isl_cnt;
        }

        /* boundary adjustment, :934-965 */
        for (i = 1; i < isl_cnt; ++i) {
            if (isl[i].from <= isl[i - 1].to + k) {
                int len1 = isl[i - 1].to - isl[i - 1].from;
                int len2 = isl[i].to - isl[i].from;
                int overlap = isl[i - 1].to + k - isl[i].from;
                for (j = isl[i - 1].to + 1; j < isl[i].from; ++j)
                    if (counts[j] <= 2 && counts[j] < trust) break;
                if (j >= isl[i].from) continue;
                if (overlap > 3) continue;
                if (len1 < len2)
                    isl[i - 1].to -= (overlap + 1);
                else
                    isl[i].from += (overlap + 1);
            }
        }

        /* to base space, :968-1007 */
        for (i = 0; i < isl_cnt; ++i) {
            if (isl[i].from > isl[i].to) continue;
            for (j = isl[i].from; j <= isl[i].to + k - 1; ++j) strong_base[j] = 1;
        }
        isl_cnt = 0;
        j = -1;
        for (i = 0; i < len; ++i) {
            if (j == -1 && strong_base[i]) j = i;
            if (j != -1 && !strong_base[i] && strong_base[i - 1]) {
                isl[isl_cnt].from = j;
                isl[isl_cnt].to = i - 1;
                ++isl_cnt;
                j = -1;
            }
        }
        if (j != -1) {
            isl[isl_cnt].from = j;
            isl[isl_cnt].to = i - 1;
            ++isl_cnt;
        }
        if (isl_cnt == 0) {
            isl[0].from = tstart;
            isl[0].to = tend + k - 1;
            isl_cnt = 1;
        }

        /* segments, :1009-1046 */
        seg_cnt = 0;
        if (isl[0].from > 0) {
            seg[seg_cnt].from = 0;
            seg[seg_cnt].to = isl[0].from - 1;
            seg[seg_cnt].lanchor = 0;
            seg[seg_cnt].ranchor = isl[0].to - isl[0].from + 1;
            ++seg_cnt;
        }
        for (i = 0; i < isl_cnt - 1; ++i) {
            seg[seg_cnt].from = isl[i].to + 1;
            seg[seg_cnt].to = isl[i + 1].from - 1;
            seg[seg_cnt].lanchor = isl[i].to - isl[i].from + 1;
            seg[seg_cnt].ranchor = isl[i + 1].to - isl[i + 1].from + 1;
            ++seg_cnt;
        }
        if (isl[i].to < len - 1) {
            seg[seg_cnt].from = isl[i].to + 1;
            seg[seg_cnt].to = len; /* [from,to) here */
            seg[seg_cnt].lanchor = isl[i].to - isl[i].from + 1;
            seg[seg_cnt].ranchor = 0;
            ++seg_cnt;
        }
        for (i = 0; i < seg_cnt; ++i) seg[i].top2[0] = seg[i].top2[1] = -1;

        if (vf) { /* :1088-1094 */
            fprintf(vf, "Is corresponding base strong trusted?\n");
            for (i = 0; i < len; ++i) fprintf(vf, "%d", strong_base[i]);
            fprintf(vf, "\n");
        }

        if (longest == -1) return -1; /* :1107 (unreachable for kcnt>=1) */
        if (longest == kcnt) return 0; /* :1110 */

        for (i = 0; i < len; ++i) fix[i] = -1;
        bad_segment_cnt = 0;
        if (seg_cnt > 0) { /* :1118-1230 */
            search_ctx s;
            int best_fix_cnt;
            memset(&s, 0, sizeof s);
            s.p = p;
            s.t = t;
            s.seq = seq;
            s.fix = ibuf;
            s.best_fix = fix;
            s.best_fix_cnt = -1;
            s.strong = strong_base;
            s.polya = polya;
            s.max_fix_cnt = allowed_fix;
            best_bottleneck = RCO_INF;
            for (int si = 0; si < seg_cnt; ++si) {
                rco_kmer kc;
                s.trial_cnt = 0;
                s.max_fix_cnt = (seg[si].to - seg[si].from + 1) * p->max_fix_per_k / k * 2 + 1;
                if (s.max_fix_cnt < p->max_fix_per_k) s.max_fix_cnt = p->max_fix_per_k;
                s.best_bottleneck = -1;
                s.top2 = seg[si].top2;
                rco_kmer_restart(&kc);
                if (seg[si].lanchor >= seg[si].ranchor) {
                    int extend = (seg[si].to == len) ? 0 : (k - 1);
                    int a = seg[si].from - k;
                    if (a < 0) return -1; /* reference reads seq[-1] here: undefined */
                    for (i = a; i < a + k; ++i) rco_kmer_append(&kc, k, seq[i]);
                    s.start = a + k;
                    s.to = seg[si].to + extend;
                    search_right(&s, a + k, trust, 0, 1000000000, kc);
                } else {
                    int extend = (seg[si].from == 0) ? 0 : (k - 1);
                    int a = seg[si].to + 1;
                    if (a + k > len) return -1; /* undefined in the reference (reads past NUL) */
                    for (i = a; i < a + k; ++i) rco_kmer_append(&kc, k, seq[i]);
                    s.start = a - 1;
                    s.to = seg[si].from - extend;
                    search_left(&s, a - 1, trust, 0, 1000000000, kc);
                }
                if (s.best_bottleneck == -1) {
                    ++bad_segment_cnt;
                    continue;
                }
                if (s.best_bottleneck < best_bottleneck) best_bottleneck = s.best_bottleneck;
                if (best_bottleneck == -1) break;
                if (s.trial_cnt > RCO_MAX_TRIAL) return -1;
                total_fix += s.max_fix_cnt;
            }
            best_fix_cnt = s.best_fix_cnt;
            if (best_bottleneck != -1) { /* :1178-1192 */
                best_fix_cnt = 1;
                for (i = 0; i < seg_cnt; ++i)
                    if (seg[i].top2[1] >= best_bottleneck) best_fix_cnt *= 2;
            }
            if (best_bottleneck != -1 && iter == 0 &&
                (double)best_bottleneck < rco_get_bound(p, strong)) /* :1195, double compare */
                force_next = 1;
            if (best_fix_cnt >= 2)
                return -1;
            else if (best_fix_cnt <= 0)
                unfixable = 1;
        }
        if (total_fix == 0 && force_next) return 0; /* :1231 */
        if (total_fix > allowed_fix) unfixable = 1;
        if (!unfixable && !force_next) break;
        if (trust < 10 && !force_next) return -1; /* :1241 */

        /* lower the thresholds, :1247-1289 */
        masked_sorted(seq, k, counts, kcnt, ibuf);
        int has_drop = 0;
        for (i = kcnt - 1; i >= 1; --i) {
            if (ibuf[i] > strong) continue;
            if (ibuf[i] > 2 * ibuf[i - 1] && ibuf[i] > 10) {
                has_drop = 1;
                if (ibuf[i] < strong) break;
            } else if (ibuf[i - 1] == 0 && ibuf[i] >= 5) {
                has_drop = 1;
                if (ibuf[i] < strong) break;
            }
        }
        if (has_drop) {
            ++iter;
            trust = rco_get_bound_int(p, ibuf[i]);
            strong = ibuf[i];
        } else
            break;
    }

    /* pairwise veto, :1296-1398 */
    int cnt = 0;
    for (i = 0; i < len; ++i) {
        if (seq[i] == 'N' || fix[i] == -1) continue;
        ibuf[cnt++] = i;
    }
    for (i = 1; i < cnt; ++i) {
        if (qual[0] != '\0' && (qual[ibuf[i]] <= p->bad_qual && qual[ibuf[i - 1]] <= p->bad_qual))
            continue;
        if (ibuf[i] - ibuf[i - 1] + 1 <= k) {
            int min_single = RCO_INF, min_double = RCO_INF, taga = -1, tagb = -1;
            j = ibuf[i - 1] - k + 1;
            if (j < 0) j = 0;
            for (; j < kcnt; ++j) {
                if (i >= 2 && j <= ibuf[i - 2]) continue;
                if (i < cnt - 1 && j + k - 1 >= ibuf[i + 1]) break;
                if (j + k - 1 >= ibuf[i]) break;
                if (counts[j] < min_single) {
                    min_single = counts[j];
                    taga = j;
                }
            }
            for (; j < kcnt; ++j) {
                if (i < cnt - 1 && j + k - 1 >= ibuf[i + 1]) break;
                if (j > ibuf[i - 1]) break;
                if (counts[j] < min_double) {
                    min_double = counts[j];
                    tagb = j;
                }
            }
            for (; j < kcnt; ++j) {
                if (i < cnt - 1 && j + k - 1 >= ibuf[i + 1]) break;
                if (j > ibuf[i]) break;
                if (counts[j] < min_single) {
                    min_single = counts[j];
                    taga = j;
                }
            }
            if (min_single != RCO_INF && min_double != RCO_INF && counts[taga] > 1 &&
                counts[tagb] > 1 && counts[taga] > counts[tagb] / 2 &&
                counts[taga] < 2 * counts[tagb]) {
                fix[ibuf[i]] = -1;
                fix[ibuf[i - 1]] = -1;
                j = i - 2;
                while (j >= 0 && ibuf[j + 1] - ibuf[j] + 1 <= k) {
                    fix[ibuf[j]] = -1;
                    --j;
                }
                while (i + 1 < cnt && ibuf[i + 1] - ibuf[i] + 1 <= k) {
                    fix[ibuf[i + 1]] = -1;
                    ++i;
                }
            }
        }
    }

    /* end-of-read veto, :1407-1430 */
    if (total_fix > 3 && len > 10) {
        int tmp = 0;
        for (i = 0; i < 10; ++i)
            if (fix[i] != -1 && seq[i] != 'N' && qual[i] > p->bad_qual) ++tmp;
        if (tmp >= 2)
            for (i = 0; i < 10; ++i)
                if (seq[i] != 'N') fix[i] = -1;
        tmp = 0;
        for (i = len - 10; i < len; ++i)
            if (fix[i] != -1 && seq[i] != 'N' && qual[i] > p->bad_qual) ++tmp;
        if (tmp >= 3)
            for (i = len - 10; i < len; ++i)
                if (seq[i] != 'N') fix[i] = -1;
    }

    /* density veto, :1432-1466 (weights doubled to stay in integers) */
    if (total_fix >= p->max_fix_per_k) {
        dbuf2[0] = 0;
        for (i = 0; i < len; ++i) {
            if (seq[i] != 'N' && fix[i] != -1)
                dbuf2[i + 1] = dbuf2[i] + (qual[i] > p->bad_qual ? 2 : 1);
            else
                dbuf2[i + 1] = dbuf2[i];
        }
        for (i = 0; i < kcnt; ++i)
            if (dbuf2[i + k] - dbuf2[i] > 2 * p->max_fix_per_k) return -1;
    }

    /* apply, :1468-1479 */
    int ret = 0;
    for (i = 0; i < len; ++i) {
        if (fix[i] != -1) {
            seq[i] = NUM_TO_NUC[fix[i]];
            ++ret;
        }
    }
    if (ret == 0 && bad_segment_cnt > 0) return -1;
    return ret;
}

/* GetKmerInformation, ErrorCorrection.cpp:1567-1602 */
void rco_kmer_information(const rco_params *p, const rco_table *t, const char *seq, int *l, int *m,
                          int *h)
{
    int kc_cnt[RCO_MAX_READ_LENGTH];
    const int k = p->k;
    int i, n = 0;
    rco_kmer kc;
    FILE *vf = (FILE *)p->verbose_fp;
    rco_kmer_restart(&kc);
    for (i = 0; seq[i] && i < k - 1; ++i) rco_kmer_append(&kc, k, seq[i]);
    *l = *m = *h = 0;
    for (; seq[i]; ++i) {
        rco_kmer_append(&kc, k, seq[i]);
        if (kc.inv == -1) {
            int c = rco_table_get(t, &kc);
            kc_cnt[n++] = c == 0 ? 1 : c;
        }
    }
    if (n == 0) return;
    if (vf) { /* :1590-1597 */
        fprintf(vf, "After coorrection:\n%s\n", seq);
        for (i = 0; i < n; ++i) fprintf(vf, "%d ", kc_cnt[i]);
        fprintf(vf, "\n");
    }
    qsort(kc_cnt, n, sizeof(int), cmp_int);
    *l = kc_cnt[0];
    *m = kc_cnt[n / 2];
    *h = kc_cnt[n - 1];
}

/* ------------------------------------------------------------------------------------------
 * ErrorCorrection_Thread, ErrorCorrection.cpp:73-136: T workers pull unit indices from a
 * mutex-protected counter; a unit is one read (single), one index in both arenas (paired) or
 * two adjacent reads (interleaved).  Both mates get the same t = min(t1,t2) computed on the
 * uncorrected mates (:97-106 precede :108).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const rco_params *p;
    const rco_table *t;
    rco_batch *b;
    size_t used;
    pthread_mutex_t lock;
} worker_arg;

static void correct_one(const rco_params *p, const rco_table *t, char *seq, const char *qual,
                        int pair_t, int32_t *ret, int32_t *l, int32_t *m, int32_t *h)
{
    int ll, mm, hh;
    *ret = rco_error_correction(p, t, NULL, seq, qual, pair_t);
    rco_kmer_information(p, t, seq, &ll, &mm, &hh);
    *l = ll;
    *m = mm;
    *h = hh;
}

/* Units a worker takes per visit to the counter.  The reference takes one (:87-90); the results do not depend on who
 * corrects which unit, and with one unit per visit 64 threads on a two-socket host write neighbouring reads' results -- 4-byte
 * ret / l / m / h entries and 151-byte reads of the same cache lines -- at the same time.  rco_set_chunk(1) is the
 * reference's schedule. */
static int g_chunk = 64;
void rco_set_chunk(int units) { g_chunk = units > 0 ? units : 1; }

/* MPOL_INTERLEAVE over every memory node for the allocations that follow (on = 1; the table of a 78 M-k-mer run is 1.6 GB
 * that one thread fills, i.e. one node's memory, probed by the threads of both sockets), MPOL_DEFAULT again with on = 0.
 * Raw system call (no libnuma in the image); a host that refuses it keeps its policy.  Returns 0 on success. */
#include <sys/syscall.h>
#include <unistd.h>
int rco_set_interleave(int on)
{
#ifdef SYS_set_mempolicy
    unsigned long mask[16];
    memset(mask, 0, sizeof mask);
    if (!on) return (int)syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, NULL, 0);
    FILE *f = fopen("/sys/devices/system/node/online", "r");
    int lo = 0, hi = 0;
    if (f) {
        if (fscanf(f, "%d-%d", &lo, &hi) < 2) hi = lo;
        fclose(f);
    }
    for (int n = lo; n <= hi && n < 1024; ++n) mask[n / (8 * sizeof(long))] |= 1ul << (n % (8 * sizeof(long)));
    return (int)syscall(SYS_set_mempolicy, 3 /* MPOL_INTERLEAVE */, mask, 1024 + 1);
#else
    (void)on;
    return -1;
#endif
}

static void *worker_main(void *argp)
{
    worker_arg *a = (worker_arg *)argp;
    rco_batch *b = a->b;
    size_t inc = b->mode == 2 ? 2 : 1;
    for (;;) {
        size_t ind0, ind1;
        pthread_mutex_lock(&a->lock);
        ind0 = a->used;
        a->used += inc * (size_t)g_chunk;
        pthread_mutex_unlock(&a->lock);
        if (ind0 >= b->n) break;
        ind1 = ind0 + inc * (size_t)g_chunk;
        if (ind1 > b->n) ind1 = b->n;
      for (size_t ind = ind0; ind < ind1; ind += inc) {
        int tt = -1;
        if (b->mode == 1) {
            int t1 = rco_strong_trusted_threshold(a->p, a->t, b->seq + b->off[ind]);
            int t2 = rco_strong_trusted_threshold(a->p, a->t, b->seq2 + b->off2[ind]);
            tt = t1 < t2 ? t1 : t2;
        } else if (b->mode == 2) {
            int t1 = rco_strong_trusted_threshold(a->p, a->t, b->seq + b->off[ind]);
            int t2 = rco_strong_trusted_threshold(a->p, a->t, b->seq + b->off[ind + 1]);
            tt = t1 < t2 ? t1 : t2;
        }
        correct_one(a->p, a->t, b->seq + b->off[ind], b->qual + b->off[ind], tt, &b->ret[ind],
                    &b->l[ind], &b->m[ind], &b->h[ind]);
        if (b->mode == 1) {
            size_t o = b->n + ind;
            correct_one(a->p, a->t, b->seq2 + b->off2[ind], b->qual2 + b->off2[ind], tt, &b->ret[o],
                        &b->l[o], &b->m[o], &b->h[o]);
        } else if (b->mode == 2) {
            correct_one(a->p, a->t, b->seq + b->off[ind + 1], b->qual + b->off[ind + 1], tt,
                        &b->ret[ind + 1], &b->l[ind + 1], &b->m[ind + 1], &b->h[ind + 1]);
        }
      }
    }
    return NULL;
}

void rco_correct_batch(const rco_params *p, const rco_table *t, rco_batch *b, int threads)
{
    worker_arg a;
    a.p = p;
    a.t = t;
    a.b = b;
    a.used = 0;
    pthread_mutex_init(&a.lock, NULL);
    if (threads <= 1) {
        worker_main(&a);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
        for (int i = 0; i < threads; ++i) pthread_create(&th[i], NULL, worker_main, &a);
        for (int i = 0; i < threads; ++i) pthread_join(th[i], NULL);
        free(th);
    }
    pthread_mutex_destroy(&a.lock);
}
