// rc_format.h -- a batch between the reader and the writer of the `rcorrector` CLI: the SoA arenas of the C ABI packed from
// the text, quality bits and fixes of the one-pass path, and the output records.
//   record       Reads.h:224-266,360-421    4 lines in; "<id> l:%d m:%d h:%d[ cor| unfixable_error]" out
//   transcript   ErrorCorrection.cpp:686-689,759-770,856-857,1088-1094,1590-1597 (-verbose)
#pragma once
#include <emmintrin.h>

#include "rc_reader.h"

// ---- one batch travelling through the pipeline -------------------------------------------------
struct Arena {  // one file's share of a batch
    Block blk;
    int lpr = 4;  // lines per record
    PinBuf seq, qual;
    std::vector<uint32_t> off;
    // resident batches (the reads are in HBM since they were counted): there is no byte arena here, the fixes are applied
    // to the sequence lines of the text itself
    bool seq_in_text = false;
    size_t n() const { return blk.records; }
    const char *sequence(size_t r) const { return seq_in_text ? blk.text.data() + blk.line[r * (size_t)lpr + 1] : seq.data() + off[r]; }
    const char *line(size_t rec, int which, uint32_t *len) const
    {
        const size_t li = rec * (size_t)lpr + (size_t)which;
        *len = blk.line[li + 1] - blk.line[li] - 1;
        return blk.text.data() + blk.line[li];
    }
};

struct Job {
    int file = 0;
    int mode = 0;
    bool fastq = true;
    Arena a, b;
    std::vector<int32_t> ret, l, m, h;
    std::vector<int32_t> tr_before, tr_after, tr_flags, tr_niter, tr_iter;  // -verbose only
    bool resident = false;        // the batch's reads are arenas the k-mer counter kept in HBM (rc_submit_resident)
    int arena_a = 0, arena_b = 0;
    int gpu = -1;                 // the GPU that holds them (-1: any GPU may take the batch)
    // -packed: the batch as rc_packed_batch wants it (one offset array over both arenas, 2-bit codes, quality bits, the
    // letters outside ACGT) and the room for the fix list
    PinBuf pk_bases, pk_exc_pos, pk_exc_chr;
    // offsets, quality bits and the room for the fix list: ONE page-locked allocation per job -- a registration costs
    // milliseconds to make and to undo (at exit, too) whatever its size, and four of them per job were 0.1 s of a 1.5 s run
    PinBuf pk_slab;
    struct PkView {
        uint32_t *off;
        uint8_t *qbits;
        uint32_t *fix_pos;
        uint8_t *fix_chr;
    };
    PkView pk_carve(size_t total_reads, size_t nbytes, size_t fix_cap)
    {
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t o_q = up((total_reads + 1) * 4), o_fp = o_q + up((nbytes + 7) / 8 + 64), o_fc = o_fp + up(fix_cap * 4), end = o_fc + up(fix_cap + 1);
        pk_slab.need(end);
        char *p = pk_slab.data();
        return PkView{(uint32_t *)p, (uint8_t *)(p + o_q), (uint32_t *)(p + o_fp), (uint8_t *)(p + o_fc)};
    }
    std::vector<OutBuf> o1, o2;  // the formatted (and, for .gz, deflated) output records, in slices
    uint64_t cor_bases = 0;  // sum of the positive return values (UpdateSummary, main.cpp:73-79), added up by the formatter
    bool done = false;
    int rc = 0;
    std::string err;
};

// Reads.h:224-266 for a whole block: sequence -> NUL-terminated arena, quality cut / padded to the
// sequence length for the kernels (the output prints the quality line verbatim, see put_record)
uint64_t index_arena(Arena &A, const std::string &path);
// the sequences alone, NUL-terminated, at dst (A.off must be set): what the k-mer counter is given
void pack_sequences(const Arena &A, char *dst);
void pack_arena(Arena &A, const std::string &path);

// Quality bits of a resident batch straight from the text: bit p of the batch's arena (arena 1's bytes, then arena 2's)
// = the quality character of that base is above the threshold; positions without one (the NUL behind a read, a
// quality line shorter than its sequence: pack_arena pads with 0) compare as 0.  [lo, hi) is a range of arena positions
// that starts and ends at multiples of 8 (or at the arena's end): ranges are packed side by side by different threads.
// Returns false if a read with bases has no first quality character (qual[0] == 0 asks for the byte path,
// ErrorCorrection.cpp:1316).
struct QualView {
    const Arena *A[2];
    size_t bytes1;
    size_t nbytes;
};
bool pack_quality_bits_from_text(const QualView &V, char bad_q, size_t lo, size_t hi, uint8_t *bits);

// the substitutions of a resident batch, applied to the sequence lines of the text
void apply_fixes_to_text(Arena &A1, Arena *A2, size_t bytes1, const uint32_t *fix_pos, const uint8_t *fix_chr, size_t lo, size_t hi);

static inline char *put_int(char *p, int v)
{
    char tmp[16];
    int n = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do {
        tmp[n++] = (char)('0' + u % 10);
        u /= 10;
    } while (u);
    if (v < 0) *p++ = '-';
    while (n) *p++ = tmp[--n];
    return p;
}

// Reads.h:360-421: one record.  The quality line is printed as fgets left it in the reference:
// stripped of its newline only when it is exactly as long as the sequence line (Reads.h:255-262).
template <class B>
static inline void put_record(B &out, const Arena &A, size_t r, bool fastq, int cor, int l, int m, int h)
{
    uint32_t il, ql = 0;
    const char *id = A.line(r, 0, &il);
    const char *seq = A.sequence(r);
    const uint32_t sl = A.off[r + 1] - A.off[r] - 1;
    const char *q = fastq ? A.line(r, 3, &ql) : nullptr;
    const size_t need = (size_t)il + sl + ql + 96;
    const size_t at = out.size();
    out.resize(at + need);
    char *p = out.data() + at;
    memcpy(p, id, il);
    p += il;
    memcpy(p, " l:", 3);
    p = put_int(p + 3, l);
    memcpy(p, " m:", 3);
    p = put_int(p + 3, m);
    memcpy(p, " h:", 3);
    p = put_int(p + 3, h);
    if (cor == -1) {
        memcpy(p, " unfixable_error", 16);
        p += 16;
    } else if (cor > 0) {
        memcpy(p, " cor", 4);
        p += 4;
    }
    *p++ = '\n';
    memcpy(p, seq, sl);
    p += sl;
    *p++ = '\n';
    if (fastq) {
        *p++ = '+';
        *p++ = '\n';
        memcpy(p, q, ql);
        p += ql;
        // fgets kept the quality line's own newline (Reads.h strips it only at index strlen(seq)) --
        // unless this is the last line of a file that does not end with one
        if (ql != sl && !(A.blk.unterminated_last && r + 1 == A.n())) *p++ = '\n';
        *p++ = '\n';
    }
    out.resize((size_t)(p - out.data()));
}

// what the reference prints to stdout for one read under -verbose (ErrorCorrection.cpp:686-689,
// 759-770,856-857,1088-1094 and GetKmerInformation :1590-1597), from the data
// rc_correct_batch_traced returns.  gi = the read's index in ret/l/m/h order, ab = offset of its
// arena in the batch's device arena (0, or the size of arena 1 for second mates)
void put_transcript(std::vector<char> &out, const Job &J, const Arena &A, size_t r, size_t gi, size_t ab, int k);

// GetBadQuality's two histograms over the records of one block (main.cpp:88-128), at most `room` of them
void quality_histograms(const Block &b, int lpr, size_t room, std::vector<int32_t> &fh, std::vector<int32_t> &lh, int *total);

// one gzip member (RFC 1952) holding `in`, deflate level 1
void gzip_member(const OutBuf &in, OutBuf &out);
