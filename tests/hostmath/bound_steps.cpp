// The integer steps of GetBound (rc_common.h: rc_bound_steps_build, ErrorCorrection.cpp:139-142) built from a closed-form first guess must be the
// table the plain bisection builds, for every error rate a run can come up with.  Test infrastructure.
#include "rc_common.h"
#include <stdio.h>
#include <vector>
#include <chrono>
static void ref_build(double e, uint32_t *B) {
    for (int v = 0; v < RC_BOUND_STEPS; ++v) B[v] = RC_BOUND_NEVER;
    B[0] = 0; B[1] = 0;
    const double top = rc_bound_d(2147483647, e);
    if (!(e >= 0.0) || !(top < 2147483648.0)) return;
    for (int v = 2; v < RC_BOUND_STEPS; ++v) {
        if (rc_bound_i(2147483647, e) < v) break;
        long long lo = 0, hi = 2147483647;
        while (lo < hi) { const long long mid = (lo + hi) >> 1; if (rc_bound_i((int)mid, e) >= v) hi = mid; else lo = mid + 1; }
        B[v] = (uint32_t)lo;
    }
    B[0] = RC_BOUND_STEPS;
}
int main() {
    std::vector<uint32_t> a(RC_BOUND_STEPS), b(RC_BOUND_STEPS);
    const double rates[] = {0.004094631483166515, 0.01, 0.0, 1e-9, 1e-5, 0.05, 0.5, 1.0, 3.0, 1e3, 0.0123456789, 2.5e-4, -1.0, 1e12};
    for (double e : rates) {
        auto t0 = std::chrono::steady_clock::now();
        rc_bound_steps_build(e, a.data());
        auto t1 = std::chrono::steady_clock::now();
        ref_build(e, b.data());
        auto t2 = std::chrono::steady_clock::now();
        size_t bad = 0; for (int v = 0; v < RC_BOUND_STEPS; ++v) bad += a[v] != b[v];
        printf("e=%g: %zu differences; %.2f ms vs %.2f ms\n", e, bad, std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
        if (bad) return 1;
    }
    printf("ok\n");
    return 0;
}
