// what the first HIP calls of a process cost (round 5: rc_create is 0.085 s of a 1.5 s files-to-files run)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/mb/hipinit tools/mb/hipinit.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <unistd.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_nop(int *p) { if (p) *p = 1; }
int main()
{
    double t0 = now(), t = t0;
    auto lap = [&](const char *what) {
        const double n = now();
        printf("%-44s %7.1f ms (at %6.1f)\n", what, (n - t) * 1e3, (n - t0) * 1e3);
        t = n;
    };
    int nd = 0;
    hipGetDeviceCount(&nd); lap("hipGetDeviceCount (runtime initialised)");
    hipSetDevice(0); lap("hipSetDevice");
    int cu = 0;
    hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, 0); lap("hipDeviceGetAttribute(CU count)");
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0); lap("hipGetDeviceProperties");
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking); lap("hipStreamCreateWithFlags");
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1); lap("2 x hipEventCreate");
    void *p = nullptr;
    hipMalloc(&p, 5120); lap("hipMalloc(5 KB), the first");
    hipMemset(p, 0, 5120); lap("hipMemset (null stream, synchronous)");
    hipMemsetAsync(p, 0, 5120, s); hipStreamSynchronize(s); lap("hipMemsetAsync + sync on the stream");
    size_t f = 0, tt = 0;
    hipMemGetInfo(&f, &tt); lap("hipMemGetInfo");
    char bus[64];
    hipDeviceGetPCIBusId(bus, 64, 0); lap("hipDeviceGetPCIBusId");
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, (int *)p); hipStreamSynchronize(s); lap("first kernel launch + sync");
    void *big = nullptr;
    hipMalloc(&big, (size_t)24 << 30); lap("hipMalloc(24 GiB)");
    void *h = nullptr;
    hipHostMalloc(&h, (size_t)256 << 20, 0); lap("hipHostMalloc(256 MB)");
    printf("devices %d, CUs %d / %d, free %.1f GB\n", nd, cu, prop.multiProcessorCount, f / 1e9);
    fflush(stdout);
    _exit(0);
}
