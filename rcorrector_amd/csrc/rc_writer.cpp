// rc_writer.cpp -- see rc_writer.h
#include "rc_writer.h"

#include <unistd.h>

// the slices of a batch, in order.  One thread per output file: buffered writes to one file are serialised by the
// kernel (the inode's lock), so more writers only add contention -- measured on the GPU box's host: 9.1 GB/s from
// one thread, 8.4 from 32 (tools/mb/iob.cpp); two files written side by side get 14.5 GB/s
void emit_slices(ReadFile &f, const std::vector<OutBuf> &sl)
{
    size_t total = 0;
    for (const auto &v : sl) total += v.size();
    if (total == 0) return;
    f.wrote = true;
    if (f.out == stdout) {
        for (const auto &v : sl)
            if (!v.empty()) fwrite(v.data(), 1, v.size(), stdout);
        return;
    }
    const int fd = fileno(f.out);
    for (const auto &v : sl) {
        size_t done = 0;
        while (done < v.size()) {
            const ssize_t n = ::pwrite(fd, v.data() + done, v.size() - done, f.out_off + (off_t)done);
            if (n <= 0) die("ERROR: write failed on %s\n", f.path.c_str());
            done += (size_t)n;
        }
        f.out_off += (off_t)v.size();
    }
}
