// .gz inputs of a one-pass run are inflated whole by libdeflate where that is possible (Source::inflate_whole, BGZF blocks in
// parallel) and read through zlib's gzread otherwise: both ways must hand out the same bytes for every kind of file -- one
// member, several, BGZF, bytes behind the last member, a truncated file, a plain file called .gz, an empty one.
// Test infrastructure: links the CLI's host units (rc_pool, rc_reader) through their headers.  usage: gz_test <file>...
#include <unistd.h>

#include "../../rcorrector_amd/csrc/rc_reader.h"

static std::string slurp(const std::string &path, bool whole, bool *used_whole)
{
    g_gz_whole = whole;
    Source s;
    s.open(path);
    std::string out;
    std::vector<char> buf((size_t)1 << 20);
    // (the CLI peeks at 4096 bytes first, through zlib, then asks for megabytes: the same here)
    size_t n = s.fill(buf.data(), 4096);
    out.append(buf.data(), n);
    while (!s.eof) {
        n = s.fill(buf.data(), buf.size());
        out.append(buf.data(), n);
        if (n == 0 && !s.eof) break;
    }
    if (used_whole) *used_whole = s.whole;
    s.close();
    return out;
}

int main(int argc, char **argv)
{
    g_threads = 4;
    g_pool.start(8);
    for (int i = 1; i < argc; ++i) {
        bool w = false;
        const std::string a = slurp(argv[i], false, nullptr), b = slurp(argv[i], true, &w);
        printf("%s: %zu bytes via zlib, %zu via %s\n", argv[i], a.size(), b.size(), w ? "libdeflate" : "zlib (libdeflate declined)");
        if (a != b) {
            printf("FAIL %s\n", argv[i]);
            return 1;
        }
    }
    printf("ok\n");
    fflush(stdout);
    _exit(0);
}
