// The host-side arithmetic of rcorrector's one-pass path (rc_format.cpp, rc_reader.cpp) checked without a GPU: quality bits packed straight
// from the FASTQ text (pack_quality_bits_from_text: SSE compares, bit streams cut at byte-aligned piece boundaries, ragged
// and empty quality lines, two arenas side by side) against rc_pack_quality_bits over the byte arenas pack_arena makes, and
// fixes applied to the text's sequence lines (apply_fixes_to_text) against fixes applied to the byte arena.
// Test infrastructure: links the CLI's host units (rc_pool, rc_reader, rc_format) through their headers.
#include <unistd.h>

#include "../../rcorrector_amd/csrc/rc_dispatch.h"
#include <random>

static int fail(const char *what, unsigned seed)
{
    printf("FAIL %s (seed %u)\n", what, seed);
    return 1;
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    g_threads = 4;
    g_pool.start(8);
    for (unsigned seed = 1; seed <= 60; ++seed) {
        std::mt19937 rng(seed);
        const bool uniform = seed % 3 == 0;
        const int n = 1 + (int)(rng() % 700), L0 = 1 + (int)(rng() % 200);
        std::string paths[2] = {dir + "/hm_1.fq", dir + "/hm_2.fq"};
        for (int sd = 0; sd < 2; ++sd) {
            FILE *f = fopen(paths[sd].c_str(), "wb");
            for (int r = 0; r < n; ++r) {
                const int sl = uniform ? L0 : (int)(rng() % (unsigned)(L0 + 1));
                int ql = sl;
                const unsigned u = rng() % 10;
                if (!uniform && u == 0) ql = sl ? (int)(rng() % (unsigned)sl) : 0;  // short (or empty) quality line
                if (!uniform && u == 1) ql = sl + 1 + (int)(rng() % 5);            // long one
                std::string s(sl, 'A'), q(ql, 'I');
                for (auto &c : s) c = "ACGTN"[rng() % 5];
                for (auto &c : q) c = (char)(33 + rng() % 60);
                fprintf(f, "@r%d\n%s\n+\n%s\n", r, s.c_str(), q.c_str());
            }
            fclose(f);
        }
        Arena A[2];
        for (int sd = 0; sd < 2; ++sd) {
            Source s;
            s.open(paths[sd]);
            A[sd].lpr = 4;
            take_records(s, (size_t)n, 4, A[sd].blk);
            s.close();
            if ((int)A[sd].blk.records != n) return fail("records", seed);
        }
        const bool paired = seed % 2 == 0;
        Arena B[2];  // the same blocks through pack_arena: byte arenas (the reference for both checks)
        for (int sd = 0; sd < 2; ++sd) {
            B[sd].lpr = 4;
            B[sd].blk.text.need(A[sd].blk.text.cap);
            memcpy(B[sd].blk.text.p, A[sd].blk.text.p, A[sd].blk.text.cap);
            B[sd].blk.line = A[sd].blk.line;
            B[sd].blk.records = A[sd].blk.records;
            pack_arena(B[sd], paths[sd]);
            index_arena(A[sd], paths[sd]);
            A[sd].seq_in_text = true;
        }
        const size_t bytes1 = A[0].off[n], bytes2 = paired ? A[1].off[n] : 0, nbytes = bytes1 + bytes2;
        const char bad_q = (char)(33 + rng() % 60);
        // reference bits: rc_pack_quality_bits over the concatenated byte arenas
        std::vector<char> qcat(nbytes);
        memcpy(qcat.data(), B[0].qual.data(), bytes1);
        if (paired) memcpy(qcat.data() + bytes1, B[1].qual.data(), bytes2);
        std::vector<uint8_t> want((nbytes + 7) / 8 + 8, 0), got((nbytes + 7) / 8 + 8, 0);
        rc_pack_quality_bits(qcat.data(), nbytes, bad_q, want.data());
        QualView V{{&A[0], paired ? &A[1] : &A[0]}, paired ? bytes1 : nbytes, nbytes};
        const size_t Q = 1 + rng() % 7;
        bool ok = true, expect_ok = true;
        for (size_t t = 0; t < Q; ++t) {
            const size_t lo = (nbytes * t / Q) & ~(size_t)7, hi = t + 1 == Q ? nbytes : ((nbytes * (t + 1) / Q) & ~(size_t)7);
            if (lo < hi) ok = pack_quality_bits_from_text(V, bad_q, lo, hi, got.data()) && ok;
        }
        for (int sd = 0; sd < (paired ? 2 : 1); ++sd)
            for (int r = 0; r < n; ++r)
                if (B[sd].off[r + 1] - B[sd].off[r] > 1 && B[sd].qual.data()[B[sd].off[r]] == 0) expect_ok = false;
        if (ok != expect_ok) return fail("empty-quality flag", seed);
        if (memcmp(want.data(), got.data(), (nbytes + 7) / 8) != 0) return fail("quality bits", seed);
        // fixes: random positions that hold a base, new letters
        std::vector<uint32_t> pos;
        std::vector<uint8_t> chr;
        for (size_t p = 0; p < nbytes; ++p) {
            const char c = p < bytes1 ? B[0].seq.data()[p] : B[1].seq.data()[p - bytes1];
            if (c != 0 && rng() % 50 == 0) {
                pos.push_back((uint32_t)p);
                chr.push_back((uint8_t)"ACGT"[rng() % 4]);
            }
        }
        for (size_t i = 0; i < pos.size(); ++i) {
            if (pos[i] < bytes1)
                B[0].seq.data()[pos[i]] = (char)chr[i];
            else
                B[1].seq.data()[pos[i] - bytes1] = (char)chr[i];
        }
        const size_t F = 1 + rng() % 3;
        for (size_t t = 0; t < F; ++t)
            apply_fixes_to_text(A[0], paired ? &A[1] : nullptr, bytes1, pos.data(), chr.data(), pos.size() * t / F, pos.size() * (t + 1) / F);
        for (int sd = 0; sd < (paired ? 2 : 1); ++sd)
            for (int r = 0; r < n; ++r) {
                const uint32_t sl = A[sd].off[r + 1] - A[sd].off[r] - 1;
                if (memcmp(A[sd].sequence((size_t)r), B[sd].sequence((size_t)r), sl) != 0) return fail("fixes applied to the text", seed);
            }
    }
    {   // a file of several read slices (the reader cuts a request into 8 MB slices read side by side, each a megabyte at a
        // time with its newlines found on the way): the blocks, put end to end, are the file, and the line index is exact
        const std::string big = dir + "/hm_big.fq";
        std::mt19937 rng(99);
        std::string content;
        const int n = 260000;
        for (int r = 0; r < n; ++r) {
            const int sl = 40 + (int)(rng() % 120);
            std::string sq(sl, 'A'), q(sl, 'I');
            for (auto &c : sq) c = "ACGT"[rng() % 4];
            content += "@big" + std::to_string(r) + "\n" + sq + "\n+\n" + q + "\n";
        }
        FILE *f = fopen(big.c_str(), "wb");
        fwrite(content.data(), 1, content.size(), f);
        fclose(f);
        Source s;
        s.open(big);
        Block b;
        size_t at = 0, recs = 0;
        for (int round = 0;; ++round) {
            take_records(s, round % 2 ? 90000 : 50001, 4, b);
            if (b.records == 0) break;
            const size_t nl = b.records * 4, end = b.line[nl];
            if (at + end > content.size() || memcmp(b.text.p, content.data() + at, end) != 0) return fail("big file: block bytes", (unsigned)round);
            for (size_t i = 0; i < nl; ++i) {
                if (b.text.p[b.line[i + 1] - 1] != '\n') return fail("big file: line end", (unsigned)round);
                if (memchr(b.text.p + b.line[i], '\n', b.line[i + 1] - 1 - b.line[i])) return fail("big file: newline inside a line", (unsigned)round);
            }
            at += end;
            recs += b.records;
        }
        s.close();
        if (at != content.size() || recs != (size_t)n) return fail("big file: size", 0);
        unlink(big.c_str());
    }
    {   // the one-pass reader that runs ahead of the GPU runtime (Ingest::start): stopped with blocks in its queue (abort), the
        // sources rewound as rcorrector's main does, started again and drained by hand -- both mates' blocks, put end to end,
        // are the files, whole pairs per block, in order (no GPU: consume() is not called)
        Run run;
        std::string content[2];
        const int n = 70001;
        std::mt19937 rng(7);
        for (int sd = 0; sd < 2; ++sd) {
            for (int r = 0; r < n; ++r) {
                const int sl = 30 + (int)(rng() % 100);
                std::string sq(sl, 'A'), q(sl, 'I');
                for (auto &c : sq) c = "ACGT"[rng() % 4];
                content[sd] += "@p" + std::to_string(r) + "/" + std::to_string(sd + 1) + "\n" + sq + "\n+\n" + q + "\n";
            }
            FILE *f = fopen((dir + (sd ? "/ra_2.fq" : "/ra_1.fq")).c_str(), "wb");
            fwrite(content[sd].data(), 1, content[sd].size(), f);
            fclose(f);
        }
        run.files.emplace_back();
        run.mates.emplace_back();
        open_file(run.files.back(), (dir + "/ra_1.fq").c_str(), true, false, dir);
        open_file(run.mates.back(), (dir + "/ra_2.fq").c_str(), true, false, dir);
        for (int attempt = 0; attempt < 2; ++attempt) {
            Ingest I(run, 9000, true);
            I.depth = attempt ? 3 : 4;
            I.start();
            if (attempt == 0) {
                for (;;) {  // let it get ahead, then change our mind
                    std::unique_lock<std::mutex> lk(I.mu);
                    if (I.q.size() >= 3 || I.done) break;
                    lk.unlock();
                    usleep(1000);
                }
                I.abort();
                if (!I.q.empty()) return fail("read-ahead: blocks left after abort", 0);
                for (ReadFile *f : {&run.files[0], &run.mates[0]}) {
                    f->src.rewind();
                    f->src.left.need(4096);
                    f->src.left_len = f->src.fill(f->src.left.p, 4096);
                }
                continue;
            }
            size_t at[2] = {0, 0}, recs = 0;
            for (;;) {
                std::unique_ptr<Retained> B;
                {
                    std::unique_lock<std::mutex> lk(I.mu);
                    I.cv.wait(lk, [&] { return I.done || !I.q.empty(); });
                    if (I.q.empty()) break;
                    B = std::move(I.q.front());
                    I.q.pop_front();
                    I.cv.notify_all();
                }
                if (B->a.records != B->b.records || B->mode != 1) return fail("read-ahead: pairs", (unsigned)recs);
                for (int sd = 0; sd < 2; ++sd) {
                    const Block &b = sd ? B->b : B->a;
                    const size_t end = b.line[b.records * 4];
                    if (at[sd] + end > content[sd].size() || memcmp(b.text.p, content[sd].data() + at[sd], end) != 0) return fail("read-ahead: block bytes", (unsigned)recs);
                    at[sd] += end;
                }
                recs += B->a.records;
            }
            I.reader.join();
            if (recs != (size_t)n || at[0] != content[0].size() || at[1] != content[1].size()) return fail("read-ahead: size", 0);
        }
        for (const char *f : {"/ra_1.fq", "/ra_2.fq", "/ra_1.cor.fq", "/ra_2.cor.fq"}) unlink((dir + f).c_str());
    }
    printf("ok 60 cases\n");
    fflush(stdout);
    _exit(0);
}
