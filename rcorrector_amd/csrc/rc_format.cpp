// rc_format.cpp -- see rc_format.h
#include "rc_format.h"

#include <zlib.h>

// Reads.h:224-266 for a whole block: sequence -> NUL-terminated arena, quality cut / padded to the
// sequence length for the kernels (the output prints the quality line verbatim, see put_record)
uint64_t index_arena(Arena &A, const std::string &path)
{
    const size_t n = A.n();
    A.off.resize(n + 1);
    A.off[0] = 0;
    uint64_t total = 0;
    for (size_t r = 0; r < n; ++r) {
        uint32_t sl, il;
        A.line(r, 1, &sl);
        A.line(r, 0, &il);
        if (sl > MAX_READ_LENGTH - 1)
            die("ERROR: %s: a read of %u bases exceeds the limit of %d (utils.h:7)\n", path.c_str(), sl, MAX_READ_LENGTH - 1);
        if (il > MAX_ID_LENGTH - 1) die("ERROR: %s: a header line longer than %d characters\n", path.c_str(), MAX_ID_LENGTH - 1);
        total += sl + 1;
        A.off[r + 1] = (uint32_t)total;
    }
    if (total >= (1ull << 32)) die("ERROR: batch too large; lower -batch\n");
    return total;
}

// the sequences alone, NUL-terminated, at dst (A.off must be set): what the k-mer counter is given
void pack_sequences(const Arena &A, char *dst)
{
    parallel_for(A.n(), [&](size_t lo, size_t hi) {
        for (size_t r = lo; r < hi; ++r) {
            uint32_t sl;
            const char *s = A.line(r, 1, &sl);
            char *d = dst + A.off[r];
            memcpy(d, s, sl);
            d[sl] = 0;
        }
    });
}

void pack_arena(Arena &A, const std::string &path)
{
    const size_t n = A.n();
    const uint64_t total = index_arena(A, path);
    A.seq_in_text = false;
    A.seq.need(total);
    A.qual.need(total);
    parallel_for(n, [&](size_t lo, size_t hi) {
        for (size_t r = lo; r < hi; ++r) {
            uint32_t sl, ql = 0;
            const char *s = A.line(r, 1, &sl);
            char *d = A.seq.data() + A.off[r];
            memcpy(d, s, sl);
            d[sl] = 0;
            char *dq = A.qual.data() + A.off[r];
            uint32_t qc = 0;
            if (A.lpr == 4) {
                const char *q = A.line(r, 3, &ql);
                qc = std::min(ql, sl);
                memcpy(dq, q, qc);
            }
            memset(dq + qc, 0, sl + 1 - qc);
        }
    });
}

bool pack_quality_bits_from_text(const QualView &V, char bad_q, size_t lo, size_t hi, uint8_t *bits)
{
    bool ok = true;
    uint64_t acc = 0;
    int nacc = 0;
    uint8_t *out = bits + (lo >> 3);
    auto put = [&](uint64_t v, int nb) {  // nb <= 16 bits at a time; four bytes leave at once (they all belong to this range)
        acc |= v << nacc;
        nacc += nb;
        if (nacc >= 32) {
            const uint32_t w = (uint32_t)acc;
            memcpy(out, &w, 4);
            out += 4;
            acc >>= 32;
            nacc -= 32;
        }
    };
    const __m128i thr = _mm_set1_epi8(bad_q);
    const bool zero_above = (signed char)0 > (signed char)bad_q;  // (a negative threshold: the padding compares as "good")
    size_t pos = lo;
    while (pos < hi) {
        const int sd = pos >= V.bytes1 ? 1 : 0;
        const Arena &A = *V.A[sd];
        const size_t base = sd ? V.bytes1 : 0, p = pos - base;
        const size_t r = (size_t)(std::upper_bound(A.off.begin(), A.off.begin() + (ptrdiff_t)A.n() + 1, (uint32_t)p) - A.off.begin()) - 1;
        const size_t end_side = std::min(hi, sd ? V.nbytes : V.bytes1);
        for (size_t rr = r; rr < A.n() && base + A.off[rr] < end_side; ++rr) {
            const uint32_t sl = A.off[rr + 1] - A.off[rr] - 1;
            uint32_t ql = 0;
            const char *q = A.line(rr, 3, &ql);
            const uint32_t qc = std::min(ql, sl);
            if (sl && (qc == 0 || q[0] == 0)) ok = false;
            // this read's positions inside [lo, hi): characters j0 .. j1 - 1 of its sl + 1 bytes
            const size_t r0 = base + A.off[rr];
            const uint32_t j0 = r0 < pos ? (uint32_t)(pos - r0) : 0;
            const uint32_t j1 = (uint32_t)std::min<size_t>(sl + 1, end_side - r0);
            uint32_t j = j0;
            while (j < j1) {
                const uint32_t nb = std::min<uint32_t>(16, j1 - j);
                uint32_t m = 0;
                if (j < qc) {  // (the text buffer carries 64 bytes of slack behind its last line)
                    m = (uint32_t)_mm_movemask_epi8(_mm_cmpgt_epi8(_mm_loadu_si128((const __m128i *)(q + j)), thr));
                    if (qc - j < 16) {
                        const uint32_t keep = (1u << (qc - j)) - 1u;
                        m = (m & keep) | (zero_above ? (0xffffu & ~keep) : 0u);
                    }
                } else if (zero_above) {
                    m = 0xffffu;
                }
                put(m & ((1u << nb) - 1u), (int)nb);
                j += nb;
            }
            pos = r0 + j1;
        }
        if (pos < end_side) pos = end_side;  // (cannot happen: the reads tile the arena)
    }
    for (; nacc > 0; nacc -= 8) {  // (the last byte is a partial one only at the arena's end)
        *out++ = (uint8_t)acc;
        acc >>= 8;
    }
    return ok;
}

void apply_fixes_to_text(Arena &A1, Arena *A2, size_t bytes1, const uint32_t *fix_pos, const uint8_t *fix_chr, size_t lo, size_t hi)
{
    for (size_t q = lo; q < hi; ++q) {
        size_t p = fix_pos[q];
        Arena &A = (A2 && p >= bytes1) ? *A2 : A1;
        if (&A == A2) p -= bytes1;
        const size_t n = A.n();
        size_t r;
        const uint32_t stride = A.off[1];
        if ((uint64_t)stride * n == A.off[n] && A.off[p / stride] == (p / stride) * (size_t)stride && A.off[p / stride + 1] == (p / stride + 1) * (size_t)stride)
            r = p / stride;  // reads of one length
        else
            r = (size_t)(std::upper_bound(A.off.begin(), A.off.begin() + (ptrdiff_t)n + 1, (uint32_t)p) - A.off.begin()) - 1;
        A.blk.text.p[A.blk.line[r * (size_t)A.lpr + 1] + (p - A.off[r])] = (char)fix_chr[q];
    }
}

void quality_histograms(const Block &b, int lpr, size_t room, std::vector<int32_t> &fh, std::vector<int32_t> &lh, int *total)
{
    static char qbuf[MAX_READ_LENGTH];  // Reads::qual, reused from record to record
    for (size_t r = 0; r < b.records && r < room; ++r) {
        const uint32_t *L = b.line.data() + r * (size_t)lpr;
        const uint32_t sl = L[2] - L[1] - 1, ql = lpr == 4 ? L[4] - L[3] - 1 : 0;
        ++*total;
        if (lpr != 4) continue;
        const char *q = b.text.data() + L[3];
        // qual[strlen(seq)-1] and qual[0] as GetBadQuality sees them: Reads::Next reads every
        // quality line into ONE reused buffer (bytes behind a short line keep what earlier
        // records left there) and strips a newline only at index strlen(seq)
        // (Reads.h:204-219); qual[-1], for an empty sequence, is the last byte of the
        // sequence buffer in front of it, 0.
        const uint32_t qn = std::min<uint32_t>(ql, MAX_READ_LENGTH - 1);
        memcpy(qbuf, q, qn);
        if (qn + 1 < MAX_READ_LENGTH) {
            qbuf[qn] = '\n';
            qbuf[qn + 1] = 0;
        } else {
            qbuf[qn] = 0;
        }
        if (sl < MAX_READ_LENGTH && qbuf[sl] == '\n') qbuf[sl] = 0;
        const unsigned char lastq = sl ? (unsigned char)qbuf[sl - 1] : 0;
        const unsigned char firstq = (unsigned char)qbuf[0];
        ++lh[lastq];
        ++fh[firstq];
    }
}

void gzip_member(const OutBuf &in, OutBuf &out)
{
    const LibDeflate &LD = libdeflate();
    if (LD.ok) {
        static thread_local void *c = nullptr;  // (a compressor per thread: they are not shareable, and cost a few hundred KB)
        if (!c) c = LD.alloc_c(1);
        if (c) {
            out.resize(LD.gzip_bound(c, in.size()) + 64);
            const size_t n = LD.gzip(c, in.data(), in.size(), out.data(), out.size());
            if (n) {
                out.resize(n);
                return;
            }
        }
    }
    z_stream z;
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, 1, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) die("ERROR: zlib deflateInit2 failed\n");
    out.resize(deflateBound(&z, (uLong)in.size()) + 64);
    z.next_in = (Bytef *)in.data();
    z.avail_in = (uInt)in.size();
    z.next_out = (Bytef *)out.data();
    z.avail_out = (uInt)out.size();
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) die("ERROR: zlib deflate failed\n");
    out.resize(z.total_out);
    deflateEnd(&z);
}

void put_transcript(std::vector<char> &out, const Job &J, const Arena &A, size_t r, size_t gi, size_t ab, int k)
{
    uint32_t il, ol;
    const char *id = A.line(r, 0, &il);
    const char *orig = A.line(r, 1, &ol);
    const char *seq = A.seq.data() + A.off[r];
    const int len = (int)(A.off[r + 1] - A.off[r] - 1);
    const int kcnt = len >= k ? len - k + 1 : 0;
    const size_t a0 = ab + A.off[r];
    auto put = [&](const char *p, size_t n) { out.insert(out.end(), p, p + n); };
    auto puts_ = [&](const char *p) { put(p, strlen(p)); };
    auto puti = [&](int v) {
        char tmp[16];
        char *e = put_int(tmp, v);
        put(tmp, (size_t)(e - tmp));
    };
    put(id, il);
    puts_("\n");
    if (J.tr_flags[gi] & 1) {
        puts_("Before correction:\n");
        put(orig, (size_t)len);
        puts_("\n");
        for (int i = 0; i < kcnt; ++i) {
            const int c = J.tr_before[a0 + (size_t)i];
            puti(c != 0 ? c : 1);
            puts_(" ");
        }
        puts_("\n");
        const int n_it = J.tr_niter[gi];
        if (n_it > g_trace_iter)
            die("rcorrector: -verbose: read %.*s went through %d threshold iterations, more than the %d recorded; raise -verbose-iter\n",
                (int)il, id, n_it, g_trace_iter);
        for (int it = 0; it < n_it; ++it) {
            const int32_t *e = J.tr_iter.data() + (gi * (size_t)g_trace_iter + (size_t)it) * RC_TRACE_ITER_WORDS;
            puts_("strong trust threshold=");
            puti(e[0]);
            puts_(" threshold=");
            puti(e[1]);
            puts_("\n");
            if (e[2]) {
                puts_("Is corresponding base strong trusted?\n");
                for (int b = 0; b < len; ++b) out.push_back((char)('0' + (((uint32_t)e[4 + (b >> 5)] >> (b & 31)) & 1u)));
                puts_("\n");
            }
        }
    }
    // GetKmerInformation: the counts of the k-mers without a non-ACGT letter, 0 shown as 1
    int bad = 0, n_valid = 0;
    std::vector<int> cnt;
    for (int i = 0; i < len; ++i) {
        const char ch = seq[i];
        const bool ok = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
        bad = ok ? (bad > 0 ? bad - 1 : 0) : k;  // windows ending at i are invalid while bad > 0
        if (i >= k - 1 && bad == 0) {
            const int c = J.tr_after[a0 + (size_t)(i - k + 1)];
            cnt.push_back(c != 0 ? c : 1);
            ++n_valid;
        }
    }
    if (n_valid > 0) {
        puts_("After coorrection:\n");
        put(seq, (size_t)len);
        puts_("\n");
        for (int c : cnt) {
            puti(c);
            puts_(" ");
        }
        puts_("\n");
    }
}
