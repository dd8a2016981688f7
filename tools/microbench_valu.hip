// microbench_valu.hip -- issue cost of the vector instructions the probe / threshold kernels are made of, on gfx950:
// cycles per wave-instruction with 8 waves per SIMD resident (the kernels are bound by VALU issue, so what an
// instruction costs relative to a plain add is what an "instruction diet" has to be priced in).
//   build: hipcc --offload-arch=gfx950 -O3 -o microbench_valu microbench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define N_ITER 2048
#define UNROLL 16
// 4 independent chains so that latency is not what is measured
#define BODY(ASM)                                                                 \
    for (int i = 0; i < N_ITER; ++i) {                                            \
        _Pragma("unroll") for (int u = 0; u < UNROLL / 4; ++u)                    \
        {                                                                         \
            asm volatile(ASM : "+v"(a) : "v"(x), "v"(y));                         \
            asm volatile(ASM : "+v"(b) : "v"(x), "v"(y));                         \
            asm volatile(ASM : "+v"(c) : "v"(x), "v"(y));                         \
            asm volatile(ASM : "+v"(d) : "v"(x), "v"(y));                         \
        }                                                                         \
    }

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t x, uint32_t y, unsigned long long *cyc)
{
    uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    uint64_t a64 = a, b64 = b, c64 = c, d64 = d;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (OP == 0) { BODY("v_add_u32 %0, %0, %1") }
    if (OP == 1) { BODY("v_mul_lo_u32 %0, %0, %1") }
    if (OP == 2) { BODY("v_mul_hi_u32 %0, %0, %1") }
    if (OP == 3) { BODY("v_mad_u32_u24 %0, %0, %1, %2") }
    if (OP == 4) { BODY("v_xor_b32 %0, %0, %1") }
    if (OP == 5) { BODY("v_bfrev_b32 %0, %0") }
    if (OP == 6) { BODY("v_and_or_b32 %0, %0, %1, %2") }
    if (OP == 7) { BODY("v_med3_i32 %0, %0, %1, %2") }
    if (OP == 8) { BODY("v_alignbit_b32 %0, %0, %1, %2") }
    if (OP == 9) { BODY("v_bcnt_u32_b32 %0, %0, %1") }
    if (OP == 10) { BODY("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") }
    if (OP == 11) { BODY("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf") }
    if (OP == 12) { BODY("v_min_i32 %0, %0, %1") }
    if (OP == 13) { BODY("v_pk_min_u16 %0, %0, %1") }
    if (OP == 14) { BODY("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x78") }
    if (OP == 15) { BODY("v_cndmask_b32 %0, %0, %1, vcc") }
    if (OP == 16) { BODY("v_perm_b32 %0, %0, %1, %2") }
    if (OP == 17) { BODY("v_lshl_add_u32 %0, %0, 2, %1") }
    if (OP == 18) { BODY("v_add3_u32 %0, %0, %1, %2") }
    if (OP == 19) {  // 64-bit shift
        for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
            for (int u = 0; u < UNROLL / 4; ++u) {
                asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a64) : "v"(x));
                asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(b64) : "v"(x));
                asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(c64) : "v"(x));
                asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(d64) : "v"(x));
            }
        }
    }
    if (OP == 20) {  // 64-bit compare
        for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) asm volatile("v_cmp_eq_u64 vcc, %0, %1" ::"v"(a64), "v"(b64) : "vcc");
        }
    }
    if (OP == 21) {  // 32-bit compare
        for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) asm volatile("v_cmp_eq_u32 vcc, %0, %1" ::"v"(a), "v"(b) : "vcc");
        }
    }
    if (OP == 22) { BODY("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf") }
    if (OP == 23) {  // ds_bpermute
        for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
            for (int u = 0; u < UNROLL / 4; ++u) {
                asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(x));
                asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(b) : "v"(x));
                asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(c) : "v"(x));
                asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(d) : "v"(x));
            }
        }
    }
    if (OP == 24) {  // v_readlane + s-op
        for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                uint32_t s;
                asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(a));
                asm volatile("" ::"s"(s));
            }
        }
    }
    if (OP == 25) { BODY("v_mul_u32_u24 %0, %0, %1") }
    if (OP == 30) { BODY("v_and_b32 %0, %0, %1") }
    if (OP == 31) { BODY("v_or_b32 %0, %0, %1") }
    if (OP == 32) { BODY("v_lshlrev_b32 %0, %1, %0") }
    if (OP == 33) { BODY("v_lshrrev_b32 %0, %1, %0") }
    if (OP == 34) { BODY("v_mov_b32 %0, %1") }
    if (OP == 35) { BODY("v_sub_u32 %0, %0, %1") }
    if (OP == 36) { BODY("v_not_b32 %0, %0") }
    if (OP == 37) { BODY("v_cndmask_b32_e64 %0, %0, %1, s[4:5]") }
    if (OP == 38) { BODY("v_min_u32 %0, %0, %1") }
    if (OP == 39) { BODY("v_add_u32_e64 %0, %0, %1") }
    if (OP == 41) { BODY("v_or3_b32 %0, %0, %1, %2") }
    if (OP == 42) { BODY("v_lshl_or_b32 %0, %0, 2, %1") }
    if (OP == 43) { BODY("v_bfe_u32 %0, %0, 3, 5") }
    if (OP == 44) { BODY("v_add_co_u32 %0, vcc, %0, %1") }
    if (OP == 45) { BODY("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %1, %2, vcc") }
    if (OP == 46) { BODY("v_and_b32 %0, 0xaaaaaaaa, %0") }
    if (OP == 47) { BODY("v_add_u32 %0, 0x12345, %0") }
    if (OP == 48) { BODY("v_ashrrev_i32 %0, %1, %0") }
    if (OP == 49) { BODY("v_max_i32 %0, %0, %1") }
    if (OP == 50) { BODY("v_min_i16 %0, %0, %1") }
    if (OP == 51) { BODY("v_pk_add_u16 %0, %0, %1") }
    if (OP == 52) { BODY("v_ffbl_b32 %0, %0") }
    if (OP == 53) { BODY("v_mbcnt_lo_u32_b32 %0, %1, %0") }
    if (OP == 54) { BODY("v_sad_u32 %0, %0, %1, %2") }
    if (OP == 26) { BODY("v_pk_max_u16 %0, %0, %1") }
    if (OP == 27) { BODY("v_max_u32 %0, %0, %1") }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (uint32_t)(a64 + b64 + c64 + d64);
    if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0);
}

template <int OP>
static void run(const char *name, uint32_t *out, unsigned long long *cyc)
{
    const int blocks = 256 * 8;  // 8 waves per SIMD on every CU
    hipMemset(cyc, 0, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u, cyc);  // warm
    hipDeviceSynchronize();
    hipMemset(cyc, 0, 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions issued per SIMD: blocks * 4 waves * N_ITER * UNROLL / 1024 SIMDs
    const double per_simd = (double)blocks * 4 * N_ITER * UNROLL / 1024.0;
    printf("%-28s %8.3f ms  %6.2f ns per wave-instruction per SIMD = %5.2f cycles at 2.4 GHz\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}

int main()
{
    uint32_t *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMalloc(&cyc, 8);
    run<0>("v_add_u32", out, cyc);
    run<1>("v_mul_lo_u32", out, cyc);
    run<2>("v_mul_hi_u32", out, cyc);
    run<3>("v_mad_u32_u24", out, cyc);
    run<25>("v_mul_u32_u24", out, cyc);
    run<4>("v_xor_b32", out, cyc);
    run<5>("v_bfrev_b32", out, cyc);
    run<6>("v_and_or_b32", out, cyc);
    run<14>("v_bitop3_b32", out, cyc);
    run<7>("v_med3_i32", out, cyc);
    run<12>("v_min_i32", out, cyc);
    run<27>("v_max_u32", out, cyc);
    run<13>("v_pk_min_u16", out, cyc);
    run<26>("v_pk_max_u16", out, cyc);
    run<8>("v_alignbit_b32", out, cyc);
    run<9>("v_bcnt_u32_b32", out, cyc);
    run<16>("v_perm_b32", out, cyc);
    run<17>("v_lshl_add_u32", out, cyc);
    run<18>("v_add3_u32", out, cyc);
    run<15>("v_cndmask_b32 (vcc)", out, cyc);
    run<10>("v_mov_b32 dpp quad_perm", out, cyc);
    run<11>("v_mov_b32 dpp row_ror:8", out, cyc);
    run<22>("v_mov_b32 dpp row_mirror", out, cyc);
    run<19>("v_lshlrev_b64", out, cyc);
    run<20>("v_cmp_eq_u64", out, cyc);
    run<21>("v_cmp_eq_u32", out, cyc);
    run<23>("ds_bpermute_b32 + wait", out, cyc);
    run<24>("v_readlane_b32", out, cyc);
    run<30>("v_and_b32", out, cyc);
    run<31>("v_or_b32", out, cyc);
    run<32>("v_lshlrev_b32", out, cyc);
    run<33>("v_lshrrev_b32", out, cyc);
    run<48>("v_ashrrev_i32", out, cyc);
    run<34>("v_mov_b32", out, cyc);
    run<35>("v_sub_u32", out, cyc);
    run<36>("v_not_b32", out, cyc);
    run<37>("v_cndmask_b32_e64 (sgpr)", out, cyc);
    run<45>("v_cmp_lt_i32 + v_cndmask", out, cyc);
    run<38>("v_min_u32", out, cyc);
    run<49>("v_max_i32", out, cyc);
    run<50>("v_min_i16", out, cyc);
    run<51>("v_pk_add_u16", out, cyc);
    run<39>("v_add_u32_e64", out, cyc);
    run<41>("v_or3_b32", out, cyc);
    run<42>("v_lshl_or_b32", out, cyc);
    run<43>("v_bfe_u32", out, cyc);
    run<44>("v_add_co_u32", out, cyc);
    run<46>("v_and_b32 literal", out, cyc);
    run<47>("v_add_u32 literal", out, cyc);
    run<52>("v_ffbl_b32", out, cyc);
    run<53>("v_mbcnt_lo_u32_b32", out, cyc);
    run<54>("v_sad_u32", out, cyc);
    return 0;
}
