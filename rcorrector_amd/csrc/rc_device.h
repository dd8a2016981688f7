// rc_device.h -- device-side table probe shared by the kernels (rc_table.hip, rc_correct.hip).
#pragma once
#include "rc_common.h"

// Store::GetCount on a canonical code (Store.h:59-66): one 64-byte bucket = four 16-byte loads
// of the same sector; the five {key,count} slots and the meta dword are picked out of registers.
// Of two equal keys the first in probe order wins (the build places the later Put first).
__device__ __forceinline__ int rc_table_lookup(const rc_table_view &T, uint64_t canon)
{
    uint32_t b = rc_home(canon, T.nb_home);
    const uint32_t klo = (uint32_t)canon, khi = (uint32_t)(canon >> 32);
    for (;;) {
        const uint4 *p = reinterpret_cast<const uint4 *>(T.buckets + (size_t)b * RC_BUCKET_DWORDS);
        const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
        int r = 0;
        r = (q0.x == klo && q0.y == khi) ? (int)q0.z : r;
        r = (q0.w == klo && q1.x == khi && r == 0) ? (int)q1.y : r;
        r = (q1.z == klo && q1.w == khi && r == 0) ? (int)q2.x : r;
        r = (q2.y == klo && q2.z == khi && r == 0) ? (int)q2.w : r;
        r = (q3.x == klo && q3.y == khi && r == 0) ? (int)q3.z : r;
        if (r != 0 || !(q3.w & 1u)) return r;
        ++b;
    }
}

