// host I/O microbenchmark 3 (round 5): can the output of one file be filled in parallel?  The two mechanisms the round-4
// review proposed for the CLI's write path -- formatting into a MAP_SHARED mapping of the pre-sized output, and N pwrite
// calls on disjoint page-aligned ranges -- plus what a second write over pages that are already in the page cache costs
// (would a prefill of the output during start-up pay?), for one file and for the two files of a pair.
// usage: iob3 <dir> <GB>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const char *dir = argc > 1 ? argv[1] : "/tmp";
    const size_t GB = argc > 2 ? (size_t)atol(argv[2]) : 4, N = GB << 30;
    char path[2][256];
    snprintf(path[0], 256, "%s/iob3_a.dat", dir);
    snprintf(path[1], 256, "%s/iob3_b.dat", dir);
    char *src = (char *)aligned_alloc(4096, N);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < 32; ++t)
            th.emplace_back([&, t] {
                for (size_t i = N * t / 32; i < N * (t + 1) / 32; i += 8) *(size_t *)(src + i) = i * 0x9E3779B97F4A7C15ull;
            });
        for (auto &x : th) x.join();
    }
    // files = 1 or 2 (N bytes in all), T threads per file on disjoint 2 MB-aligned ranges
    auto pw = [&](const char *name, int files, int T, bool falloc, bool second_pass) {
        int fd[2];
        const size_t per = N / files;
        for (int f = 0; f < files; ++f) {
            unlink(path[f]);
            fd[f] = open(path[f], O_CREAT | O_RDWR | O_TRUNC, 0644);
            if (falloc && fallocate(fd[f], 0, 0, (off_t)per) != 0) printf("(fallocate failed) ");
        }
        double best = 0;
        for (int pass = 0; pass < (second_pass ? 2 : 1); ++pass) {
            const double t0 = now();
            std::vector<std::thread> th;
            for (int f = 0; f < files; ++f)
                for (int t = 0; t < T; ++t)
                    th.emplace_back([&, f, t] {
                        const size_t A = (size_t)2 << 20;
                        size_t lo = (per * t / T) & ~(A - 1), hi = t + 1 == T ? per : ((per * (t + 1) / T) & ~(A - 1));
                        const size_t CH = (size_t)8 << 20;
                        for (size_t at = lo; at < hi; at += CH) {
                            const size_t n = std::min(CH, hi - at);
                            if (pwrite(fd[f], src + f * per + at, n, (off_t)at) != (ssize_t)n) abort();
                        }
                    });
            for (auto &x : th) x.join();
            best = N / (now() - t0) / 1e9;
            if (second_pass && pass == 0) printf("%-46s files %d T=%2d: first pass %.2f GB/s, ", name, files, T, best);
        }
        if (second_pass)
            printf("second pass over cached pages %.2f GB/s\n", best);
        else
            printf("%-46s files %d T=%2d: %.2f GB/s\n", name, files, T, best);
        fflush(stdout);
        for (int f = 0; f < files; ++f) close(fd[f]);
    };
    auto mm = [&](const char *name, int files, int T, bool falloc, bool populate) {
        int fd[2];
        char *m[2];
        const size_t per = N / files;
        const double t0 = now();
        for (int f = 0; f < files; ++f) {
            unlink(path[f]);
            fd[f] = open(path[f], O_CREAT | O_RDWR | O_TRUNC, 0644);
            if (falloc ? fallocate(fd[f], 0, 0, (off_t)per) != 0 : ftruncate(fd[f], (off_t)per) != 0) printf("(sizing failed) ");
            m[f] = (char *)mmap(nullptr, per, PROT_READ | PROT_WRITE, MAP_SHARED | (populate ? MAP_POPULATE : 0), fd[f], 0);
            if (m[f] == MAP_FAILED) abort();
        }
        const double t1 = now();
        std::vector<std::thread> th;
        for (int f = 0; f < files; ++f)
            for (int t = 0; t < T; ++t) th.emplace_back([&, f, t] { memcpy(m[f] + per * t / T, src + f * per + per * t / T, per * (t + 1) / T - per * t / T); });
        for (auto &x : th) x.join();
        const double t2 = now();
        for (int f = 0; f < files; ++f) {
            munmap(m[f], per);
            close(fd[f]);
        }
        const double t3 = now();
        printf("%-46s files %d T=%2d: %.2f GB/s (map %.2f s, copy %.2f s, unmap %.2f s)\n", name, files, T, N / (t3 - t0) / 1e9, t1 - t0, t2 - t1, t3 - t2);
        fflush(stdout);
    };
    for (int files : {1, 2}) {
        pw("pwrite, one writer per file", files, 1, false, false);
        pw("pwrite + fallocate, one writer per file", files, 1, true, false);
        pw("pwrite + fallocate, disjoint aligned ranges", files, 4, true, false);
        pw("pwrite + fallocate, disjoint aligned ranges", files, 16, true, false);
        pw("pwrite twice (second pass over cached pages)", files, 1, true, true);
        pw("pwrite twice (second pass over cached pages)", files, 8, true, true);
        mm("MAP_SHARED of the pre-sized file (ftruncate)", files, 1, false, false);
        mm("MAP_SHARED of the pre-sized file (ftruncate)", files, 8, false, false);
        mm("MAP_SHARED of the pre-sized file (fallocate)", files, 8, true, false);
        mm("MAP_SHARED of the pre-sized file (fallocate)", files, 32, true, false);
        mm("MAP_SHARED + MAP_POPULATE (fallocate)", files, 8, true, true);
    }
    unlink(path[0]);
    unlink(path[1]);
    FILE *fp = popen("uname -r; df -T /tmp | tail -1; grep -E ' /tmp | / ' /proc/mounts | head -3 | cut -c1-160; grep -E 'MemTotal|Dirty:|Cached:' /proc/meminfo; "
                     "cat /sys/kernel/mm/transparent_hugepage/enabled /proc/sys/vm/dirty_ratio /proc/sys/vm/dirty_background_ratio 2>/dev/null", "r");
    char b[512];
    while (fp && fgets(b, 512, fp)) fputs(b, stdout);
    if (fp) pclose(fp);
    return 0;
}
