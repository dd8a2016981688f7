// Host check of the PACKED table addressing (rc_common.h): (home, rem, xrem) <-> canonical code.
// Test infrastructure; built and run by tests/test_packed_math.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "rc_common.h"

static uint64_t rng_state = 0x243F6A8885A308D3ull;
static uint64_t rnd()
{
    rng_state += 0x9E3779B97F4A7C15ull;
    uint64_t z = rng_state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main()
{
    long checked = 0;
    for (int k = 5; k <= 32; ++k) {
        const int kb = 2 * k;
        const uint64_t mask = rc_kmer_mask(k);
        // the bijection and its inverse
        for (int i = 0; i < 2000; ++i) {
            const uint64_t c = rnd() & mask;
            const uint64_t m = rc_mix2k(c, k);
            if ((m & ~mask) != 0 || rc_unmix2k(m, k) != c) {
                printf("mix round trip failed: k=%d c=%llx\n", k, (unsigned long long)c);
                return 1;
            }
        }
        // address <-> key for several table sizes, with the smallest ext that makes the address unique
        const uint32_t sizes[] = {64u, 1000u, 65537u, 19660801u, 50287231u, 0x7FFFFFFFu, 0xFFFFFFF7u};
        for (uint32_t nb : sizes) {
            int ext = 0;
            while (kb > 32 && kb - 32 - ext > 0 && ((uint64_t)nb << ext) < (1ull << (kb - 32))) ++ext;
            if (ext > 27) continue;
            for (int i = 0; i < 3000; ++i) {
                uint64_t c = rnd() & mask;
                if (i < 8) c = i < 4 ? (uint64_t)i : mask - (uint64_t)(i - 4);  // the ends of the key space
                uint32_t h, rem, xr;
                rc_packed_addr(c, k, nb, ext, &h, &rem, &xr);
                if (h >= nb || (ext == 0 && xr != 0) || (ext > 0 && (xr >> ext) != 0)) {
                    printf("address out of range: k=%d nb=%u ext=%d c=%llx -> %u %u %u\n", k, nb, ext, (unsigned long long)c, h, rem, xr);
                    return 1;
                }
                const uint64_t back = rc_packed_key(h, rem, xr, ext, k, nb);
                if (back != c) {
                    printf("key round trip failed: k=%d nb=%u ext=%d c=%llx -> (%u,%u,%u) -> %llx\n", k, nb, ext, (unsigned long long)c, h, rem,
                           xr, (unsigned long long)back);
                    return 1;
                }
                // neighbours in the mixed space must not share an address (uniqueness at the finest grain)
                const uint64_t m = rc_mix2k(c, k);
                if (m < mask) {
                    const uint64_t c2 = rc_unmix2k(m + 1, k);
                    uint32_t h2, rem2, xr2;
                    rc_packed_addr(c2, k, nb, ext, &h2, &rem2, &xr2);
                    if (h2 == h && rem2 == rem && xr2 == xr) {
                        printf("two codes share an address: k=%d nb=%u ext=%d\n", k, nb, ext);
                        return 1;
                    }
                }
                ++checked;
            }
        }
    }
    printf("ok %ld\n", checked);
    return 0;
}
