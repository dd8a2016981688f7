// rc_device.h -- device-side table probe shared by the kernels (rc_table.hip, rc_correct.hip).
#pragma once
#include "rc_common.h"

// Store::GetCount on a canonical code (Store.h:59-66): one 64-byte bucket = four 16-byte loads
// of the same sector; the five {key,count} slots and the meta dword are picked out of registers.
// Of two equal keys the first in probe order wins (the build places the later Put first).
// n_req (optional): incremented once per bucket read (profiling builds of k_correct)
__device__ __forceinline__ int rc_table_lookup(const rc_table_view &T, uint64_t canon, uint32_t *n_req = nullptr)
{
    uint32_t b = rc_home(canon, T.nb_home);
    const uint32_t klo = (uint32_t)canon, khi = (uint32_t)(canon >> 32);
    for (;;) {
        const uint4 *p = reinterpret_cast<const uint4 *>(T.buckets + (size_t)b * RC_BUCKET_DWORDS);
        uint32_t d[RC_BUCKET_DWORDS];
        if (n_req) ++*n_req;
#pragma unroll
        for (int q = 0; q < RC_BUCKET_DWORDS / 4; ++q) {
            const uint4 v = p[q];
            d[4 * q + 0] = v.x;
            d[4 * q + 1] = v.y;
            d[4 * q + 2] = v.z;
            d[4 * q + 3] = v.w;
        }
        int r = 0;
#pragma unroll
        for (int s2 = 0; s2 < RC_BUCKET_SLOTS; ++s2)
            r = (d[3 * s2] == klo && d[3 * s2 + 1] == khi && r == 0) ? (int)d[3 * s2 + 2] : r;
        if (r != 0 || !(d[RC_BUCKET_DWORDS - 1] & 1u)) return r;
        ++b;
    }
}

