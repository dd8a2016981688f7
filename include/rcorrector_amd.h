/*
 * rcorrector_amd.h -- C ABI of librcorrector_amd.so: the MI355X-native drop-in for Rcorrector's
 * per-read correction path (stage 3 of run_rcorrector.pl).
 *
 * Plain pointers and sizes only; no C++/torch types.  Each entry point names the reference
 * interface (/root/reference, v1.0.7) it replaces.  INTEGRATION.md shows the binding a maintainer
 * of the reference would add in main.cpp.
 *
 * Conventions
 *   - every function returns 0 on success or a negative rc_status; rc_last_error() has the text.
 *     The library never calls exit() (the reference exits on I/O errors, File.h:69-73).
 *   - k-mer codes are the reference's KmerCode::GetCode() values (KmerCode.h:39): 2 bits per base,
 *     A0 C1 G2 T3, first base in the most significant position, k <= 32.
 *   - a context owns one GPU, its stream, the k-mer table in HBM and all scratch memory.  One
 *     context per GPU / per host thread; reads shard across contexts with no communication
 *     (the table is replicated), SURVEY.md §8(e).
 *   - there is NO CPU fallback: without a usable HIP device rc_create() fails.
 */
#ifndef RCORRECTOR_AMD_H
#define RCORRECTOR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rc_ctx rc_ctx;

typedef enum {
    RC_STATUS_OK = 0,
    RC_STATUS_ARG = -1,   /* bad argument */
    RC_STATUS_HIP = -2,   /* HIP runtime error */
    RC_STATUS_IO = -3,    /* file could not be read */
    RC_STATUS_STATE = -4, /* call sequence error (no table, no run parameters, ...) */
    RC_STATUS_NOMEM = -5,
    RC_STATUS_NOSPACE = -6 /* rc_wait_packed / rc_wait_resident: more substitutions than fix_cap (n_fix = how many): the batch's
                              ret / l / m / h are complete, its fix list is not -- resubmit with more room, or through rc_submit */
} rc_status;

typedef struct {
    int device;        /* HIP device ordinal */
    int k;             /* kmerLength, main.cpp:133,200-204 (1..32) */
    int max_fix_per_k; /* MAX_FIX_PER_K / -maxcorK, main.cpp:159,215-219 */
} rc_config;

/* replaces: KmerCode kcode(kmerLength); Store kmers;  (main.cpp:140,270) */
rc_ctx *rc_create(const rc_config *cfg, char *errbuf, size_t errbuf_len);
void rc_destroy(rc_ctx *ctx);
const char *rc_last_error(const rc_ctx *ctx);
/* the NUMA node of the host the context's GPU hangs off (its PCI device's numa_node), or -1 if the system does
 * not say: a host that feeds the GPU from page-locked buffers wants its threads and those buffers there
 * (the reference has no counterpart: its workers are plain pthreads, main.cpp:479-483) */
int rc_device_numa_node(const rc_ctx *ctx);
/* free and total bytes of the context's GPU memory right now (hipMemGetInfo): what a host checks before it asks the
 * k-mer counter to keep a data set's bases in HBM (rc_table_count_keep).  No reference counterpart. */
int rc_device_memory(rc_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);

/* ---- k-mer table (Store.h:17-88) ------------------------------------------------------------ */
/* replaces the load loop main.cpp:294-308 when the caller has already parsed the dump:
 * n x Store::Put(code, count) in index order (a later duplicate overrides an earlier one,
 * Store.h:55).  codes may be forward or canonical; counts <= 1 must already be filtered. */
int rc_table_build(rc_ctx *ctx, const uint64_t *codes, const int32_t *counts, size_t n);
/* same with the arrays already in HBM (d_codes is canonicalised in place) */
int rc_table_build_device(rc_ctx *ctx, uint64_t *d_codes, const int32_t *d_counts, size_t n);
/* replaces main.cpp:294-308 including the parse: reads a `jellyfish dump` text file
 * (">COUNT\nKMER\n"), drops count <= 1, builds the table.  *stored = the "Stored %d kmers" value. */
int rc_table_load_jfdump(rc_ctx *ctx, const char *path, int64_t *stored);
/* replace stages 0-2 (run_rcorrector.pl:262-281: jellyfish bc / count -C / dump -L 2): exact
 * canonical k-mer counts over any number of arenas of reads (reads separated by NUL bytes, each
 * arena < 2^32 bytes; k-mers holding a letter outside ACGT are skipped), then the table from the
 * entries with count >= min_count.  count_add takes a host arena, count_add_device one already in
 * HBM.  *n_kmers = entries kept (the "Stored %d kmers" value).  The kept entries' codes (8 bytes each) stay in HBM behind
 * the table for rc_estimate_error_rate, which reads them instead of decoding the buckets and frees them; a caller that
 * never estimates gets the memory back at rc_set_run_params (or with the table). */
int rc_table_count_begin(rc_ctx *ctx);
int rc_table_count_add(rc_ctx *ctx, const char *seq, size_t nbytes);
int rc_table_count_add_device(rc_ctx *ctx, const uint8_t *d_seq, size_t nbytes);
int rc_table_count_finish(rc_ctx *ctx, int min_count, int64_t *n_kmers);
/* on != 0: rc_table_count_finish leaves the arenas it was given in HBM (in the order of the non-empty count_add calls)
 * instead of releasing them, for rc_submit_resident below: a data set whose k-mers were counted on this GPU is corrected
 * where it lies and crosses PCIe once.  rc_table_count_arenas reports how many there are (and the bytes of the first
 * `cap`); they are released by the next rc_table_count_begin, by rc_table_count_release, or with the context. */
int rc_table_count_keep(rc_ctx *ctx, int on);
int rc_table_count_arenas(const rc_ctx *ctx, size_t *n_arenas, uint64_t *bytes, size_t cap);
int rc_table_count_release(rc_ctx *ctx);
/* ends a session opened by rc_table_count_begin WITHOUT counting: the arenas added since stay in HBM as kept arenas (in the
 * order of the non-empty count_add calls), no table is built and the context's table, if any, stays.  For a GPU that will
 * correct reads whose k-mers another GPU counts (the table arrives by rc_table_replicate): one Store, T workers,
 * main.cpp:294-308,451 -- each worker's reads uploaded once, to the GPU that corrects them. */
int rc_table_count_park(rc_ctx *ctx);
/* rc_table_count_finish for reads that are spread over n contexts, one per GPU -- each with a session of its own
 * (rc_table_count_begin, count_add of the reads THAT GPU will correct): the key space is cut into the slices one GPU would
 * use, slice p belongs to ctxs[p % n]; every GPU emits a slice's keys from its own arenas and sends them to the owner
 * (peer to peer where the GPUs can, else through the host), which sorts and reduces them; the entries with count >=
 * min_count are put end to end in slice order on ctxs[0], where the table is built -- the same entries in the same order
 * as one GPU holding all the reads would produce (ERROR_RATE and rc_table_write_jfdump depend on the order); the other
 * contexts get the table by rc_table_replicate.  ctxs[0]'s rc_table_count_keep setting applies to every context (the
 * arenas stay where they are for rc_submit_resident).  One Store for T workers, main.cpp:294-308, counted by all of them. */
int rc_table_count_finish_sharded(rc_ctx **ctxs, int n, int min_count, int64_t *n_kmers);
/* begin + add_device + finish for one arena */
int rc_table_count_reads_device(rc_ctx *ctx, const uint8_t *d_seq, size_t nbytes, int min_count,
                                int64_t *n_kmers);
/* the table as the text `jellyfish dump` writes (">COUNT\nKMER\n", canonical k-mers) -- the file
 * main.cpp:295-307 parses -- so a table counted here can be handed to the reference binary.
 * Entries are written in "dump order" (ascending splitmix64 of the code: pseudo-random like
 * Jellyfish's hash order, but reproducible). */
int rc_table_write_jfdump(rc_ctx *ctx, const char *path, int64_t *n_written);
/* dst uses src's table (same device; src must outlive dst and must not rebuild its table meanwhile).
 * The reference shares one Store between all worker threads (main.cpp:451); this lets several
 * contexts -- several batches in flight on one GPU -- do the same instead of replicating it. */
int rc_table_share(rc_ctx *dst, const rc_ctx *src);
/* dst gets its own copy of src's table: the bucket array goes device to device (over xGMI when the
 * contexts sit on different GPUs) -- the replication step of a multi-GPU run; the dump is parsed and the
 * table built once (main.cpp:294-308 loads one Store for all workers).  rc_estimate_error_rate() keeps
 * working on src only (dst has no dump order of its own until asked: it falls back to the table's). */
int rc_table_replicate(rc_ctx *dst, const rc_ctx *src);
/* the same with the copy left in flight on dst's stream (rc_sync(dst) waits for it): a host replicating to the other
 * GPUs of a node queues all its copies first -- they travel over different xGMI links at the same time.  GPUs without
 * peer access to the source get their copy staged through page-locked host memory (then complete on return). */
int rc_table_replicate_async(rc_ctx *dst, const rc_ctx *src);
/* Store::GetCount (Store.h:59-66) for n valid k-mer codes (host arrays) */
int rc_table_lookup(rc_ctx *ctx, const uint64_t *codes, size_t n, int32_t *counts_out);
/* every stored (canonical code, count) pair, unspecified order -- what `jellyfish dump` would
 * print (run_rcorrector.pl:280); *n_out = number stored (may exceed cap: call again) */
int rc_table_export(rc_ctx *ctx, uint64_t *codes, int32_t *counts, size_t cap, size_t *n_out);
/* 64-bit digest of the table's content (every stored canonical code with its count; independent of
 * the bucket layout): equal digests = replicas that answer every Store::GetCount alike.  The
 * multi-GPU callers compare it across devices after replicating the table (main.cpp:294-308 loads ONE
 * Store for all workers). */
int rc_table_digest(rc_ctx *ctx, uint64_t *digest);
/* slot layout the last build chose: 0 = WIDE (5 x 12-byte {code, count} slots per 64-byte bucket, any k
 * and count), 1 = PACKED (8 x 8-byte {remainder, count} slots: the code is implied by the bucket it
 * hashes to; the count field has 27 bits, less up to 8 where a large k over a small table needs more
 * remainder bits, larger counts live in a side array of the same allocation; taken when at most 4000
 * counts overflow and the placement allows: a third less HBM per k-mer).
 * Same answers either way (rc_table_digest is layout independent).  < 0: no table. */
int rc_table_layout(const rc_ctx *ctx);
/* bytes of HBM held by the table, number of buckets, number of stored entries */
int rc_table_stats(const rc_ctx *ctx, uint64_t *bytes, uint64_t *buckets, uint64_t *entries);

/* ---- run parameters (globals of main.cpp:17-30) ----------------------------------------------- */
/* replaces main.cpp:310-358 (ERROR_RATE estimation).  Uses the entries parsed by the last
 * rc_table_load_jfdump() in file order -- or, for a table that was counted here or built from
 * arrays, the table's entries in dump order (what the reference would scan if given
 * rc_table_write_jfdump's file); the probes run on the GPU, the <=100000 divisions and the sort
 * on the host in IEEE double exactly as the reference does. */
int rc_estimate_error_rate(rc_ctx *ctx, double wk, double *rate_out);
/* replaces GetBadQuality's arithmetic (main.cpp:108-127) given the two histograms gathered over
 * the first <= 1,000,000 records (first_hist[q] = #reads whose first quality char is q,
 * last_hist likewise for the last base) */
char rc_bad_quality_from_hist(const int32_t first_hist[300], const int32_t last_hist[300], int32_t total);
/* sets ERROR_RATE and badQualityThreshold (the globals of main.cpp:24-27) for subsequent corrections; also builds
 * the inverse of GetBound (ErrorCorrection.cpp:139-142) at this rate on the host -- the reference's own arithmetic,
 * ~20 ms -- and leaves it in device memory for the kernels: call it once per run, with no batch in flight */
int rc_set_run_params(rc_ctx *ctx, double error_rate, char bad_quality);

/* Quality as one bit per base.  The correction only ever compares a quality with badQualityThreshold
 * (the vetoes, ErrorCorrection.cpp:1313-1466) and tests qual[0] != 0, so a host that is bound by the
 * upload can ship bits instead of bytes: with on != 0 every quality arena handed to this context
 * (rc_batch.qual / qual2, rc_device_batch.d_qual) is a bit array over the arena, bit (p & 7) of byte
 * p >> 3 = the quality character at arena byte p is greater than the threshold -- 19 instead of 151
 * bytes per 150-base read over PCIe.  rc_pack_quality_bits() makes such an array from a byte arena
 * (nbytes = the arena's size; bits has (nbytes + 7) / 8 bytes).  FASTQ input only (the FASTA marker
 * qual[0] == 0 cannot be expressed); same results as with bytes. */
int rc_set_quality_bits(rc_ctx *ctx, int on);
void rc_pack_quality_bits(const char *qual, size_t nbytes, char bad_quality, uint8_t *bits);

/* ---- correction (ErrorCorrection.h:12-28) ------------------------------------------------------ */
/* Batch in host memory.  Replaces struct _ErrorCorrectionThreadArg + the pthread fan-out of
 * main.cpp:439-523 / the inline loop main.cpp:368-438: one call = one batch through
 * ErrorCorrection_Thread (ErrorCorrection.cpp:73-136).
 *   read i of an arena is the NUL-terminated string at seq + off[i]; off has n+1 entries and
 *   off[i+1]-off[i] = strlen+1; qual uses the same offsets.
 *   mode 0: single-end.  mode 1: paired, mate i of arena 1 pairs with mate i of arena 2
 *   (readBatch/readBatch2).  mode 2: interleaved, reads 2j and 2j+1 are mates.
 *   Outputs: seq/seq2 corrected in place; ret = ErrorCorrection()'s return value
 *   (_Read::correction), l/m/h = GetKmerInformation().  In mode 1 entries [n, 2n) of
 *   ret/l/m/h belong to arena 2. */
typedef struct {
    int mode;
    size_t n;
    char *seq;
    const char *qual;
    const uint32_t *off;
    char *seq2;
    const char *qual2;
    const uint32_t *off2;
    int32_t *ret, *l, *m, *h;
} rc_batch;
int rc_correct_batch(rc_ctx *ctx, rc_batch *b);

/* The same, asynchronous: up to RC_MAX_SLOTS batches in flight in ONE context, so that the upload
 * of batch N+1, the kernels of batch N and the download of batch N-1 overlap -- what the reference
 * gets from filling the next batch while its worker threads correct the current one
 * (main.cpp:479-516).  rc_submit(slot) starts a batch and returns; the descriptor is copied, the
 * buffers it points to must stay valid and untouched until rc_wait(slot) returns, after which they
 * hold the results (and rc_summary() includes the batch).  Slot 0 runs in the context itself; every other slot is a LANE with
 * streams, events and scratch memory of its own (created at its first use; the table, the run parameters and the kept arenas
 * are the context's, lent), so that the kernels of batches in different slots overlap on the GPU -- a batch's last waves, on
 * its slowest reads, run under the next batch's probe kernel instead of in front of it (the reference's workers pick up the
 * next read while another one is still searching: ErrorCorrection.cpp:87-90).  Batches in different slots may therefore
 * complete in any order; rc_wait(slot) is what orders a caller.  RC_SLOT_LANES=0 in the environment keeps every slot in the
 * one context (one compute stream: batches complete in submission order), and rc_set_slot_lanes() switches at run time (it
 * applies to the batches submitted after it): lanes pay where a batch's slowest reads leave the GPU idle -- small batches, error-
 * laden data: 1 M-read batches of 150-base pairs at 0.5 % errors 174 -> 223 M reads/s, at 5 % errors and k = 31 3.5 -> 9.2 M --
 * and cost a few per cent where the caller is bound elsewhere and wants its batches back one at a time (`rcorrector` starts
 * without them and turns them on when its writer starts waiting for the GPU).
 * Buffers obtained from rc_host_alloc() are page-locked: the DMA engines read and write them
 * directly; any other buffer is staged through pinned memory the slot owns (one extra copy each
 * way).  rc_correct_batch(b) == rc_submit(b, 0); rc_wait(0).  Calls on one context come from one
 * thread at a time, with one exception: rc_wait(slot) only touches its slot and may run on another
 * thread while the next rc_submit (a different slot) is issued -- how `rcorrector` keeps several
 * worker threads busy on one context. */
#define RC_MAX_SLOTS 4
int rc_set_slot_lanes(rc_ctx *ctx, int on);
/* The lanes' streams overlap only when the HIP runtime gives them hardware queues of their own: it multiplexes a process's
 * streams onto GPU_MAX_HW_QUEUES queues, four by default, and two compute streams that share one run one after the other.
 * That variable is read when the runtime starts, so it is the HOST's to set: export GPU_MAX_HW_QUEUES=16, or call
 * rc_runtime_prepare(16) -- from one thread, before any thread exists that reads the environment and before the process
 * first touches HIP (its own use included; torch counts) -- which sets it unless the process has it already.  Returns 1 if it
 * set the variable, 0 if it was set already (left alone), -1 on a bad argument.  rc_create does not touch the environment
 * (it did until round 5); a lane created while the variable is unset or below 8 prints one note on stderr (RC_QUIET=1: none).
 * No reference counterpart: the reference's workers are pthreads on one Store (main.cpp:439-523). */
int rc_runtime_prepare(int hw_queues);
int rc_submit(rc_ctx *ctx, const rc_batch *b, int slot);
int rc_wait(rc_ctx *ctx, int slot);
int rc_host_alloc(rc_ctx *ctx, size_t bytes, void **out);
int rc_host_free(rc_ctx *ctx, void *p);
/* ... or page-lock memory the caller already owns (page-aligned, whole pages) */
int rc_host_register(void *p, size_t bytes);
int rc_host_unregister(void *p);

/* The packed boundary: what SURVEY.md section 3 gives as the device boundary of a batch -- "only packed reads go down
 * and (fix list, ret, l, m, h) come back" -- for callers bound by the PCIe link (the byte path above moves 306 bytes up
 * and 167 down per 150-base read, this one about 61 and 20).  The reads of a batch are described as ONE arena of
 * `nbytes` bytes, read i the NUL-terminated string at off[i] (mode 1: the n first mates, then the n second mates;
 * off has total + 1 entries, total = n reads of mode 0 / 2 or 2 n of mode 1); the arena itself stays with the caller:
 *   bases     2 bits per arena byte, 16 per word: byte p at bits 30 - 2 (p & 15) of word p >> 4, A0 C1 G2 T3
 *             (KmerCode.h:7-89's code); NULs and letters outside ACGT are 0.  (nbytes + 15) / 16 words.
 *   exc_pos / exc_chr   the n_exc letters outside ACGT: arena position (ascending) and the letter
 *   qual_bits one bit per arena byte as rc_pack_quality_bits() makes them, or NULL: no qualities (FASTA input,
 *             Reads.h:224-266: the reference then sees qual[0] == 0)
 * rc_pack_bases() makes bases / exc_* from a byte arena (callers with several threads pack ranges that start at multiples
 * of 16 bytes side by side: range [begin, end) writes words [begin / 16, (end + 15) / 16) and its own exception list,
 * whose positions are arena positions; a range that starts inside a word keeps the bits of the positions in front of it,
 * so the second arena of a pair is packed behind the first one: seq = arena2 - bytes1, begin = bytes1).  rc_submit_packed() starts the batch, rc_wait_packed() completes it: ret / l / m / h
 * as for rc_batch, and the substitutions the correction made (ErrorCorrection.cpp:1468-1479) as n_fix pairs
 * (fix_pos[j] = arena position, fix_chr[j] = the new letter), in no particular order -- positions are distinct, so the
 * caller may apply them from several threads (rc_apply_fixes() is the loop).  fix_cap = room in the caller's arrays;
 * more fixes than that is an error (a batch of N bases never has more than N).  All arrays of the descriptor should be
 * page-locked (rc_host_alloc); others are staged through the slot's own pinned memory.
 * The results equal rc_submit()'s on the same reads: ret / l / m / h identical, arena + fixes = the corrected arena. */
typedef struct {
    int mode;
    size_t n;               /* reads per arena as in rc_batch (mode 1: pairs) */
    uint64_t nbytes;        /* bytes of the arena (mode 1: both mates' arenas together) */
    const uint32_t *off;    /* [total + 1] */
    const uint32_t *bases;  /* [(nbytes + 15) / 16] */
    const uint8_t *qual_bits; /* [(nbytes + 7) / 8] or NULL */
    const uint32_t *exc_pos;
    const uint8_t *exc_chr;
    size_t n_exc;
    int32_t *ret, *l, *m, *h; /* [total] */
    uint32_t *fix_pos;      /* [fix_cap] */
    uint8_t *fix_chr;       /* [fix_cap] */
    size_t fix_cap;
    size_t n_fix;           /* out, valid after rc_wait_packed */
} rc_packed_batch;
/* returns the number of exceptions found in [begin, end) (all of them are counted, the first exc_cap are stored) */
size_t rc_pack_bases(const char *seq, size_t begin, size_t end, uint32_t *bases, uint32_t *exc_pos, uint8_t *exc_chr, size_t exc_cap);
int rc_submit_packed(rc_ctx *ctx, rc_packed_batch *b, int slot);
int rc_wait_packed(rc_ctx *ctx, int slot);
void rc_apply_fixes(char *seq, const uint32_t *fix_pos, const uint8_t *fix_chr, size_t n_fix);

/* The packed boundary for reads that are already in HBM -- the arenas the k-mer counter kept (rc_table_count_keep; without
 * a jellyfish dump the reads have been uploaded once to be counted, run_rcorrector.pl:262-281 reads them a second time
 * for stage 3): nothing but the offsets and the quality bits goes down, the same results come back.  A batch is a byte
 * range of one kept arena (mode 0 / 2), or of two (mode 1: first mates from arena_a, second mates from arena_b); the ranges
 * hold whole NUL-terminated reads.  off / qual_bits / ret .. fix_* / n_fix as in rc_packed_batch, over the batch's own
 * arena of bytes_a + bytes_b bytes (range a, then range b); the kept arenas themselves are never modified, so a batch
 * may be submitted again. */
typedef struct {
    int mode;
    size_t n;                  /* reads per arena as in rc_batch (mode 1: pairs) */
    int arena_a, arena_b;      /* indices of kept arenas (arena_b: mode 1 only) */
    uint64_t begin_a, bytes_a; /* the batch's byte range of arena_a */
    uint64_t begin_b, bytes_b; /* ... of arena_b (mode 1), else 0 */
    const uint32_t *off;       /* [total + 1], off[total] = bytes_a + bytes_b */
    const uint8_t *qual_bits;  /* [(bytes_a + bytes_b + 7) / 8] or NULL (FASTA) */
    int32_t *ret, *l, *m, *h;  /* [total] */
    uint32_t *fix_pos;         /* [fix_cap] positions in the batch's arena */
    uint8_t *fix_chr;          /* [fix_cap] */
    size_t fix_cap;
    size_t n_fix;              /* out, valid after rc_wait_resident */
} rc_resident_batch;
int rc_submit_resident(rc_ctx *ctx, rc_resident_batch *b, int slot);
int rc_wait_resident(rc_ctx *ctx, int slot);

/* rc_correct_batch plus everything the reference prints per read under -verbose (VERBOSE,
 * ErrorCorrection.cpp:15,686-689,759-770,856-857,1088-1094,1590-1597), as data; the caller formats
 * it (rc_main.cpp does, byte for byte).  Reads are indexed like ret/l/m/h (mode 1: arena 2's reads
 * at [n, 2n)); arena bytes are arena 1's followed by arena 2's.
 *   counts_before/after[a] = GetCount of the k-mer starting at arena byte a, before / after the
 *     correction (0 where the window holds a non-ACGT letter or runs past the read);
 *   flags[r] bit 0: read r passed the screens, i.e. "Before correction" is printed;
 *   n_iter[r]: threshold iterations of read r (may exceed max_iter: only the first max_iter are
 *     recorded);  iter + (r*max_iter + i)*RC_TRACE_ITER_WORDS: iteration i = {strong trust
 *     threshold, threshold, 1 if the bitmap was reached, 0, 32 words of the "Is corresponding base
 *     strong trusted?" bitmap (bit b of word w = base 32w+b)}. */
#define RC_TRACE_ITER_WORDS 36
typedef struct {
    int32_t max_iter;
    int32_t *counts_before, *counts_after; /* [arena bytes] */
    int32_t *flags, *n_iter;               /* [reads] */
    int32_t *iter;                         /* [reads * max_iter * RC_TRACE_ITER_WORDS] */
} rc_trace;
int rc_correct_batch_traced(rc_ctx *ctx, rc_batch *b, rc_trace *t);

/* Batch already resident in HBM (asynchronous on the context's stream; rc_sync() to wait).
 * In mode 1 the arena holds the n/2 first mates followed by the n/2 second mates. */
typedef struct {
    int mode;
    uint32_t n_reads;
    uint64_t nbytes;        /* bytes of the arena (sum of strlen+1) */
    int32_t max_read_len;   /* longest read, bases */
    uint8_t *d_seq;         /* corrected in place */
    const uint8_t *d_qual;
    const uint32_t *d_off;  /* n_reads + 1 */
    int32_t *d_ret, *d_l, *d_m, *d_h;
} rc_device_batch;
int rc_correct_device(rc_ctx *ctx, const rc_device_batch *b);
/* replaces GetStrongTrustedThreshold (ErrorCorrection.h:26, ErrorCorrection.cpp:1482-1565) for every
 * read of an arena in HBM (asynchronous; the run parameters must be set): d_strong[r] = the
 * function's return value for read r (-1 for reads it screens out). */
int rc_strong_threshold_device(rc_ctx *ctx, const uint8_t *d_seq, const uint32_t *d_off, uint32_t n_reads,
                               uint64_t nbytes, int32_t max_read_len, int32_t *d_strong);
/* The three functions of ErrorCorrection.h:26-28 at the reference's own granularity, one NUL-terminated read per call
 * (each a batch of one: a kernel launch and two small copies -- for bindings that work read by read and for spot checks;
 * throughput lives in the batch calls above).  The table and the run parameters must be set.
 *   rc_strong_threshold_read  = GetStrongTrustedThreshold(seq, qual, kcode, kmers), ErrorCorrection.cpp:1482-1565
 *       (the function never reads qual);
 *   rc_correct_read           = ErrorCorrection(id, seq, qual, pairStrongTrustThreshold, kcode, kmers), :682-1480: seq is
 *       corrected in place, *ret = the return value; pair_strong_threshold = min of the two mates' strong thresholds, or
 *       -1 for a read without a mate (:96-106); qual == NULL stands for a FASTA record (qual[0] == 0, Reads.h:241);
 *       counted by rc_summary like any batch;
 *   rc_kmer_info_read         = GetKmerInformation(seq, kmerLength, kmers, l, m, h), :1567-1602, of the read as given. */
int rc_strong_threshold_read(rc_ctx *ctx, const char *seq, int32_t *strong);
int rc_correct_read(rc_ctx *ctx, char *seq, const char *qual, int32_t pair_strong_threshold, int32_t *ret);
int rc_kmer_info_read(rc_ctx *ctx, const char *seq, int32_t *l, int32_t *m, int32_t *h);
/* the hash-probe kernel alone: d_counts[a] = count of the k-mer starting at arena byte a
 * (ErrorCorrection.cpp:716-723 for every read of the arena) */
int rc_probe_device(rc_ctx *ctx, const uint8_t *d_seq, uint64_t nbytes, int32_t *d_counts);
int rc_sync(rc_ctx *ctx);

/* ---- measurement ----------------------------------------------------------------------------- */
/* with profiling on, every kernel launch is bracketed by HIP events on the context's stream */
int rc_profile_enable(rc_ctx *ctx, int on);
/* on = 2 additionally runs the instrumented build of the correction kernel (slower; same results),
 * which counts, over the launches since the last reset: the reads the threshold kernel could not
 * finish and handed to the correction kernel, their gather rounds (one round = up to 64 table
 * probes issued together) and the table buckets they read -- the denominators of the kernel's
 * request-rate figures. */
int rc_profile_correct_counters(rc_ctx *ctx, uint64_t *reads_listed, uint64_t *gather_rounds, uint64_t *bucket_requests);
/* d_rounds != NULL: the instrumented build (rc_profile_enable(ctx, 2)) also leaves the gather rounds of every read the
 * correction kernel processes in d_rounds[read index] (device memory, one int32 per read of the batch, zeroed by the caller:
 * reads finished before that kernel are not written) -- which reads of a batch the search works hardest on
 * (MAX_TRIAL, ErrorCorrection.cpp:7; the straggler of tools/find_straggler.py).  NULL switches it off. */
int rc_profile_read_rounds(rc_ctx *ctx, int32_t *d_rounds);
/* kernel 0 = probe, 1 = threshold, 2 = correct; accumulated since the last reset */
int rc_profile_get(rc_ctx *ctx, int kernel, double *total_ms, uint64_t *launches);
int rc_profile_reset(rc_ctx *ctx);

/* diagnostic: GetBound(c) (ErrorCorrection.cpp:139-142) exactly as the kernels evaluate it, for n
 * counts and one error rate: out_int = the implicit double->int conversion (cvttsd2si semantics),
 * out_dbl = the double itself (NaN for c < 0).  Lets a test compare the device arithmetic with the
 * host's bit for bit. */
int rc_selftest_get_bound(rc_ctx *ctx, const int32_t *c, size_t n, double error_rate, int32_t *out_int, double *out_dbl);

/* summary counters, struct _summary main.cpp:32-36,73-79: reads and corrected bases of every batch
 * this context has corrected through any entry point (accumulated on the device; waits for the
 * context's kernels) */
int rc_summary(const rc_ctx *ctx, uint64_t *total_reads, uint64_t *total_corrections);

#ifdef __cplusplus
}
#endif
#endif
