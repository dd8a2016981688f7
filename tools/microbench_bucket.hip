// microbench_bucket.hip -- two ways for a wavefront to read 64 random 64-byte buckets (round 6):
//   lane:  every lane reads its own bucket with four 16-byte loads (what rc_table_lookup does): four load instructions, each of
//          which touches 64 different lines;
//   quad:  the four lanes of a quad read one bucket, 16 bytes each, in four rounds (round r: the bucket of the quad's lane r, its
//          index passed round with a DPP quad_perm): four load instructions again, each of which touches 16 lines, 64 contiguous
//          bytes a quad -- and the per-lane compare of two slots is followed by a combine across the quad.
// Both variants do the PACKED slot compare of rc_device.h on what they read (rem / displacement match, first slot in probe order
// wins), so the instruction overhead of the quad variant is part of the comparison.  Tables from L2-resident to beyond the caches.
//   build: hipcc --offload-arch=gfx950 -O3 -o microbench_bucket microbench_bucket.hip ; run: ./microbench_bucket
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ __forceinline__ uint32_t mix(uint32_t h)
{
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t match2(uint32_t lo, uint32_t hi, uint32_t rem, uint32_t whi, uint32_t mhi)
{
    return __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(hi, whi, mhi, 0x28), lo, rem, 0xF6);
}

template <int UNROLL>
__global__ __launch_bounds__(256) void probe_lane(const uint4 *__restrict__ buf, uint32_t bmask, size_t n, uint32_t *__restrict__ out)
{
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * UNROLL;
    if (i0 >= n) return;
    uint32_t acc = 0;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const uint32_t h = mix((uint32_t)(i0 + u) * 0x9E3779B1u), b = h & bmask, rem = mix(h);
        const uint4 *p = buf + (size_t)b * 4;
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 v = p[q];
            lo[2 * q] = v.x; hi[2 * q] = v.y; lo[2 * q + 1] = v.z; hi[2 * q + 1] = v.w;
        }
        uint32_t rh = 0;
#pragma unroll
        for (int s = 7; s >= 0; --s) rh = match2(lo[s], hi[s], rem, 0, 0x78000000u) == 0 ? hi[s] : rh;
        acc += rh & 0x07FFFFFFu;
    }
    if (acc == 0x01234567u) out[0] = acc;
}

template <int UNROLL>
__global__ __launch_bounds__(256) void probe_quad(const uint4 *__restrict__ buf, uint32_t bmask, size_t n, uint32_t *__restrict__ out)
{
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * UNROLL;
    if (i0 >= n) return;
    const int ql = threadIdx.x & 3;
    uint32_t acc = 0;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const uint32_t h = mix((uint32_t)(i0 + u) * 0x9E3779B1u), b = h & bmask, rem = mix(h);
        uint32_t mine = 0;
        auto round = [&](uint32_t bb, uint32_t rr, bool me) {
            const uint4 v = buf[(size_t)bb * 4 + ql];
            const uint32_t t0 = match2(v.x, v.y, rr, 0, 0x78000000u), t1 = match2(v.z, v.w, rr, 0, 0x78000000u);
            uint32_t x = t1 == 0 ? v.w : 0u;
            x = t0 == 0 ? v.y : x;
            // first non-zero in lane order across the quad
            const uint32_t y = dpp<0xB1>(x);                       // quad_perm [1,0,3,2]
            const uint32_t p01 = (ql & 1) ? (y ? y : x) : (x ? x : y);
            const uint32_t z = dpp<0x4E>(p01);                     // quad_perm [2,3,0,1]
            const uint32_t r = (ql & 2) ? (z ? z : p01) : (p01 ? p01 : z);
            mine = me ? r : mine;
        };
        round(dpp<0x00>(b), dpp<0x00>(rem), ql == 0);
        round(dpp<0x55>(b), dpp<0x55>(rem), ql == 1);
        round(dpp<0xAA>(b), dpp<0xAA>(rem), ql == 2);
        round(dpp<0xFF>(b), dpp<0xFF>(rem), ql == 3);
        acc += mine & 0x07FFFFFFu;
    }
    if (acc == 0x01234567u) out[0] = acc;
}

template <class K>
static void run(K kern, int unroll, const uint4 *buf, size_t bytes, size_t n, uint32_t *out, const char *tag)
{
    uint32_t buckets = (uint32_t)(bytes / 64);
    uint32_t mask = 1; while ((mask << 1) <= buckets) mask <<= 1; mask -= 1;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    unsigned grid = (unsigned)((n / unroll + 255) / 256);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, buf, mask, n, out);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess || (e = hipGetLastError()) != hipSuccess) printf("!! %s: %s\n", tag, hipGetErrorString(e));
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, buf, mask, n, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
    printf("%-34s table=%8.1f MiB  %7.2f G buckets/s  %6.2f ms\n", tag, (double)(mask + 1.0) * 64 / (1 << 20), n / ms / 1e6, ms);
}

int main()
{
    const size_t n = 1ull << 30;
    uint4 *buf; uint32_t *out;
    const size_t maxb = (size_t)4 << 30;
    if (hipMalloc(&buf, maxb) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    if (hipMemset(buf, 1, maxb) != hipSuccess) { printf("hipMemset failed\n"); return 1; }
    const size_t sizes[] = {(size_t)2 << 20, (size_t)128 << 20, (size_t)1536 << 20, (size_t)4 << 30};
    for (size_t bytes : sizes) {
        run(probe_lane<1>, 1, buf, bytes, n, out, "lane: 4 x 16 B per lane, 1 / lane");
        run(probe_lane<2>, 2, buf, bytes, n, out, "lane: 4 x 16 B per lane, 2 / lane");
        run(probe_quad<1>, 1, buf, bytes, n, out, "quad: 16 B per lane x 4 rounds, 1");
        run(probe_quad<2>, 2, buf, bytes, n, out, "quad: 16 B per lane x 4 rounds, 2");
    }
    return 0;
}
