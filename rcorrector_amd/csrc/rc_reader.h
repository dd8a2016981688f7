// rc_reader.h -- input side of the `rcorrector` CLI: a file as a stream of bytes cut into blocks of whole records
// (parallel block reads, libdeflate / zlib for .gz, a line index per block), file typing and output naming.
//   file typing  Reads.h:108-162            first byte '>' FASTA, '@' FASTQ; ".gz" by the last two chars (File.h:51-55)
//   output name  Reads.h:39-75,140-157      <od>/<name minus last extension>.cor.f[aq][.gz]
//   record       Reads.h:224-266            4 (FASTQ) or 2 (FASTA) lines, fgets semantics at the end of a file
#pragma once
#include <zlib.h>

#include "rc_pool.h"

// where Source::fill leaves the positions (relative to `base`) of the newlines it reads, in ascending order, when it can
// find them on the way (done = it did: regular files)
struct NlSink {
    const char *base;
    std::vector<uint32_t> *nl;
    bool done;
};

// ---- input: a stream of bytes cut into blocks of whole records --------------------------------
struct Source {
    std::string path;
    bool is_gz = false, seekable = false;
    int fd = -1;
    gzFile gz = nullptr;
    Buf left;  // bytes read from the file but not handed out yet (the tail behind the last block)
    size_t left_len = 0;
    off_t pos = 0;  // file offset of the next unread byte (seekable files)
    bool eof = false;
    double per_line = 0;  // bytes per line of the last block: sizes the next block's buffer in one go
    // .gz, one-pass runs with libdeflate: the file's whole text, inflated at the first large request (the text of such a run
    // stays in memory anyway); `served` = bytes handed out so far, by zlib before that (the peek at the head of the file)
    size_t served = 0;
    bool whole_tried = false, whole = false;
    Buf dec;
    size_t dec_len = 0;

    bool inflate_bgzf(const LibDeflate &LD, const unsigned char *c, size_t csize);
    // The whole file through libdeflate: every member, into `dec` (false: read on through zlib)
    bool inflate_whole();
    void open(const std::string &p);
    void close();
    // back to the first byte (plain seekable files only: what a reader that ran ahead of a decision took is read again)
    void rewind();
    // appends up to `want` bytes of the file at dst; sets eof when the file ends first.  Regular
    // files are read by several threads at once (pread into disjoint slices: the copy out of the
    // page cache is what limits a single reader), streams and .gz by this thread alone.
    size_t fill(char *dst, size_t want, NlSink *sink = nullptr);
};

// a batch of raw records: the text plus the start of every line (lines_per_record per record)
struct Block {
    Buf text;
    std::vector<uint32_t> line;  // line i = text[line[i] .. line[i+1]-1), without its '\n'
    size_t records = 0;
    bool unterminated_last = false;  // the file ended without a newline: the last line got one here
    void swap(Block &o)
    {
        text.swap(o.text);
        line.swap(o.line);
        std::swap(records, o.records);
        std::swap(unterminated_last, o.unterminated_last);
    }
};

// up to max_records whole records from the source (fewer only at end of file), read straight into
// the block's own buffer
void take_records(Source &s, size_t max_records, int lines_per_record, Block &b);

struct ReadFile {
    std::string path;
    bool paired = false, interleaved = false, fastq = true, out_gz = false;
    Source src;
    FILE *out = nullptr;  // only its descriptor is used, with pwrite (stdout: fwrite)
    off_t out_off = 0;
    bool wrote = false;
    bool preallocated = false;  // the output's blocks were reserved beyond its final size (open_file)
};

// Reads.h:39-75
std::string base_name(const std::string &path);
// Reads.h:108-162: type the file by its first byte, open (truncate) the output
void open_file(ReadFile &f, const char *path, bool paired, bool interleaved, const std::string &od);
