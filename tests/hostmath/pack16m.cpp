// rc_pack16m (rc_common.h) against its definition: every byte value in every one of the 16 positions, in front of
// several backgrounds (all letters, all NULs, mixed), and random 16-byte blocks.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "rc_common.h"

static int check(const unsigned char *b)
{
    uint32_t w[4];
    memcpy(w, b, 16);
    uint32_t code, am, tm, bad;
    rc_pack16m(w, code, am, tm, bad);
    uint32_t ecode = 0, eam = 0, etm = 0, ebad = 0;
    for (int j = 0; j < 16; ++j) {
        const unsigned c = b[j];
        unsigned v = 3, ok = 0;
        if (c == 'A') { v = 0; ok = 1; }
        if (c == 'C') { v = 1; ok = 1; }
        if (c == 'G') { v = 2; ok = 1; }
        if (c == 'T') { v = 3; ok = 1; }
        ecode |= v << (30 - 2 * j);
        eam |= (c == 'A' ? 1u : 0u) << j;
        etm |= (c == 'T' ? 1u : 0u) << j;
        ebad |= (ok ^ 1u) << j;
    }
    if (code != ecode || am != eam || tm != etm || bad != ebad) {
        printf("mismatch:");
        for (int j = 0; j < 16; ++j) printf(" %02x", b[j]);
        printf("\n got code %08x am %04x tm %04x bad %04x\nwant code %08x am %04x tm %04x bad %04x\n", code, am, tm, bad, ecode, eam, etm, ebad);
        return 1;
    }
    return 0;
}

int main()
{
    const char *bg[] = {"ACGTACGTACGTACGT", "TTTTTTTTTTTTTTTT", "AAAAAAAAAAAAAAAA", "GGGGCCCCGGGGCCCC", "NNNNNNNNNNNNNNNN", "acgtacgtacgtacgt"};
    long n = 0;
    unsigned char b[16];
    for (int g = 0; g < 7; ++g)
        for (int pos = 0; pos < 16; ++pos)
            for (int c = 0; c < 256; ++c) {
                if (g < 6) memcpy(b, bg[g], 16); else memset(b, 0, 16);
                b[pos] = (unsigned char)c;
                if (check(b)) return 1;
                ++n;
            }
    uint64_t s = 0x9E3779B97F4A7C15ull;
    const char alpha[] = {'A', 'C', 'G', 'T', 'N', 0, 'a', 'U', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T'};
    for (long it = 0; it < 2000000; ++it) {
        for (int j = 0; j < 16; ++j) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            b[j] = (it & 1) ? (unsigned char)alpha[s & 15] : (unsigned char)(s >> 32);
        }
        if (check(b)) return 1;
        ++n;
    }
    printf("ok %ld blocks\n", n);
    return 0;
}
