// rc_pool.cpp -- see rc_pool.h
#include "rc_pool.h"

#include <dlfcn.h>
#include <errno.h>
#include <sched.h>
#include <stdarg.h>
#include <unistd.h>

#include <chrono>

bool g_stdout = false;
bool g_verbose = false;
int g_trace_iter = 64;
bool g_timing = false;
int g_threads = 8;
size_t g_deflate_threads = 0;
bool g_packed = false;
bool g_gz_whole = false;

double g_w_reader = 0, g_w_writer = 0, g_w_worker = 0;
double g_t_read = 0, g_t_pack = 0, g_t_gpu = 0, g_t_format = 0, g_t_write = 0;
static std::mutex g_t_mu;
double g_t_fill = 0, g_t_nl = 0, g_t_idx = 0;
void timing_add(double &acc, double dt)
{
    if (!g_timing) return;
    std::lock_guard<std::mutex> lk(g_t_mu);
    acc += dt;
}

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// RC_TIMING with RC_T0=<seconds since the epoch at which the caller started this process>: where the process is on the
// caller's clock (process start, HIP initialisation and the exit are outside the phases the other lines time)
void stamp(const char *what)
{
    static const char *e = getenv("RC_T0");
    if (!g_timing || !e) return;
    const double t = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    fprintf(stderr, "[rc timing] +%.3f s %s\n", t - atof(e), what);
}

void die(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fflush(NULL);
    _exit(1);  // (not exit(): it would join the helper threads from whichever thread failed)
}

Pool g_pool;

int first_gpu_numa_node()
{
    // GPUs this process can use: KFD nodes with SIMDs whose properties it may read and whose render node it may open (a
    // container is shown all of the host's nodes, but not their properties, and gets the render nodes of its own GPUs only)
    bool reordered = false;
    for (const char *v : {"HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"})
        if (getenv(v)) reordered = true;
    int found = 0, node_of_first = -2;
    for (int n = 0; n < 1024; ++n) {
        char path[160];
        snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/properties", n);
        FILE *fp = fopen(path, "r");
        if (!fp) {
            if (errno == ENOENT) break;  // past the last node
            continue;                    // someone else's
        }
        long simd = 0, minor = -1, domain = 0, loc = 0;
        char name[64];
        long long val;
        while (fscanf(fp, "%63s %lld", name, &val) == 2) {
            if (!strcmp(name, "simd_count")) simd = (long)val;
            if (!strcmp(name, "drm_render_minor")) minor = (long)val;
            if (!strcmp(name, "domain")) domain = (long)val;
            if (!strcmp(name, "location_id")) loc = (long)val;
        }
        fclose(fp);
        if (simd <= 0) continue;  // a CPU node
        if (minor >= 0) {
            snprintf(path, sizeof path, "/dev/dri/renderD%ld", minor);
            if (access(path, R_OK | W_OK) != 0) continue;  // not ours: the runtime skips it too
        }
        if (found++ == 0) {
            snprintf(path, sizeof path, "/sys/bus/pci/devices/%04lx:%02lx:%02lx.%lx/numa_node", domain, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7);
            if ((fp = fopen(path, "r"))) {
                if (fscanf(fp, "%d", &node_of_first) != 1) node_of_first = -2;
                fclose(fp);
            }
        }
    }
    // one GPU: it is device 0 whatever the variables say; several: the first one, unless a variable picks or reorders
    if (found == 0 || (found > 1 && reordered)) return -2;
    return node_of_first;
}

bool bind_to_numa_node(int node)
{
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *fp = fopen(path, "r");
    if (!fp) return false;
    char buf[4096];
    const bool ok = fgets(buf, sizeof buf, fp) != nullptr;
    fclose(fp);
    if (!ok) return false;
    cpu_set_t set, cur;
    CPU_ZERO(&set);
    CPU_ZERO(&cur);
    const bool have_cur = sched_getaffinity(0, sizeof cur, &cur) == 0;
    int n_cpu = 0;
    for (char *p = buf; *p;) {  // "0-63,128-191"
        char *e;
        long a = strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        if (*e == '-') b = strtol(e + 1, &e, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) {
            if (have_cur && !CPU_ISSET((int)c, &cur)) continue;
            CPU_SET((int)c, &set);
            ++n_cpu;
        }
        p = *e == ',' ? e + 1 : e;
        if (*e != ',') break;
    }
    return n_cpu > 0 && sched_setaffinity(0, sizeof set, &set) == 0;
}

void *big_alloc(size_t n, size_t *cap)
{
    const size_t c = (n + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    void *p = mmap(nullptr, c, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) {
        fprintf(stderr, "rcorrector: out of memory (%zu bytes)\n", c);
        exit(1);
    }
    (void)madvise(p, c, MADV_HUGEPAGE);
    *cap = c;
    return p;
}

const LibDeflate &libdeflate()
{
    static LibDeflate L;
    return L;
}
