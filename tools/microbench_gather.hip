// microbench_gather.hip -- what can the MI355X memory system do for random 64-byte (and 128-byte)
// gathers out of a multi-GB array?  This calibrates the roofline of the hash-probe kernel: the
// 8 TB/s nominal figure is a streaming number, a k-mer table probe is one random sector per lane.
//   build: hipcc --offload-arch=gfx950 -O3 -o microbench_gather microbench_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ __forceinline__ uint32_t mix(uint32_t h)
{
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}

template <int BYTES, int UNROLL>
__global__ __launch_bounds__(256) void gather(const uint4 *__restrict__ buf, uint32_t line_mask, size_t n, uint32_t *__restrict__ out)
{
    size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * UNROLL;
    uint32_t acc = 0;
    if (i0 >= n) return;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        uint32_t line = mix((uint32_t)(i0 + u) * 0x9E3779B1u) & line_mask;
        const uint4 *p = buf + (size_t)line * (BYTES / 16);
#pragma unroll
        for (int j = 0; j < BYTES / 16; ++j) {
            uint4 v = p[j];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int BYTES, int UNROLL>
static void run(const uint4 *buf, size_t bytes, size_t n, uint32_t *out, const char *tag)
{
    uint32_t lines = (uint32_t)(bytes / BYTES);
    uint32_t mask = 1; while ((mask << 1) <= lines) mask <<= 1; mask -= 1;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    unsigned grid = (unsigned)((n / UNROLL + 255) / 256);
    hipLaunchKernelGGL((gather<BYTES, UNROLL>), dim3(grid), dim3(256), 0, 0, buf, mask, n, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((gather<BYTES, UNROLL>), dim3(grid), dim3(256), 0, 0, buf, mask, n, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
    printf("%-28s table=%5.2f GiB  %6.2f G gathers/s  %7.1f GB/s useful\n", tag, (double)(mask + 1.0) * BYTES / (1 << 30),
           n / ms / 1e6, (double)n * BYTES / ms / 1e6);
}

int main(int argc, char **argv)
{
    size_t gib = argc > 1 ? atoi(argv[1]) : 2;
    size_t bytes = gib << 30, n = 1ull << 30;
    uint4 *buf; uint32_t *out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 64);
    hipMemset(buf, 1, bytes);
    run<64, 1>(buf, bytes, n, out, "64B gathers, 1/lane");
    run<64, 2>(buf, bytes, n, out, "64B gathers, 2/lane");
    run<64, 4>(buf, bytes, n, out, "64B gathers, 4/lane");
    run<128, 1>(buf, bytes, n, out, "128B gathers, 1/lane");
    run<128, 2>(buf, bytes, n, out, "128B gathers, 2/lane");
    run<32, 2>(buf, bytes, n, out, "32B gathers, 2/lane");
    run<16, 4>(buf, bytes, n, out, "16B gathers, 4/lane");
    return 0;
}
