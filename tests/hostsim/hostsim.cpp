// hostsim.cpp -- lane-serial back end for rc_correct_core.h (TEST HARNESS, CPU only).
//
// The build container has no GPU, so the wave-uniform control flow of the correction kernel
// (rcorrector_amd/csrc/rc_correct_core.h) is instantiated here with STRIDE = 1 and the oracle's
// table as the probe target, and diffed against the oracle by tests/test_hostsim.py.  Nothing in
// the product path links this file.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../oracle/rc_oracle.h"
#include "../../rcorrector_amd/csrc/rc_correct_core.h"

namespace {

struct HostWave {
    static const int STRIDE = 1;
    static const int KT = 0;
    int lane = 0;
    const rco_table *tab;
    int k;
    const char *qualp;
    std::vector<rc_frame> stack;
    long max_sp = 0, pushes = 0, pops = 0;

    void sync() {}
    void phase(int) {}
    void trace_passed() {}
    void trace_iter(int, int) {}
    void trace_strong(const unsigned char *, int) {}
    int uni(int x) { return x; }
    uint64_t uni64(uint64_t x) { return x; }
    long stats[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    void stat(int i, int v) { stats[i] += v; }
    template <class F>
    uint64_t ballot64(int base, int n, F pred)
    {
        uint64_t m = 0;
        for (int l = 0; l < 64; ++l)
            if (base + l < n && base + l >= 0 && pred(base + l)) m |= 1ull << l;
        return m;
    }
    template <class F>
    void for_lanes64(int base, int n, F body)
    {
        for (int l = 0; l < 64; ++l)
            if (base + l < n) body(base + l, l);
    }
    int reduce_add(int x) { return x; }
    uint32_t wave_min_u32(uint32_t x) { return x; }
    uint32_t wave_max_u32(uint32_t x) { return x; }
    void prefix_min(const int *in, int *out, int j0, int n)
    {
        int m = 2147483647;
        for (int j = j0; j < n; ++j) {
            m = in[j] < m ? in[j] : m;
            out[j] = m;
        }
    }
    int min_range(const int *a, int lo, int hi)
    {
        int m = 2147483647;
        for (int i = lo; i < hi; ++i) m = a[i] < m ? a[i] : m;
        return m;
    }
    int get(rc_kmer km, int /*dir*/ = 0)
    {
        ++gets;
        rco_kmer q;
        q.code = km.code;
        q.inv = km.inv;
        return rco_table_get(tab, &q);
    }
    long gets = 0;
    int lookup(uint64_t code)
    {
        rc_kmer km;
        km.code = code;
        km.inv = -1;
        return get(km);
    }
    void sort(int *a, int n) { std::sort(a, a + n); }
    void stack_push(int sp, const rc_frame &f)
    {
        if ((int)stack.size() <= sp) stack.resize(sp + 1);
        stack[sp] = f;
        ++pushes;
        if (sp + 1 > max_sp) max_sp = sp + 1;
    }
    void stack_top(int idx, rc_frame &f) { f = stack[idx]; ++pops; }
    void stack_set_mask(int idx, int mask) { stack[idx].mask = mask; }
};

struct Buffers {
    std::vector<unsigned char> base, strongb, polya;
    std::vector<int> counts, v;
    std::vector<signed char> path, best, qualv;
    std::vector<rc_island> isl;
    std::vector<rc_segment> seg;
    std::vector<uint64_t> ma, mt, mn, mi, mx, scode;
    std::vector<uint32_t> pk;
    std::vector<int> scnt, sinv, sret, skeep, sthr, smask, smeta, memo;
    rc_read_state S;
    explicit Buffers(int cap)
    {
        int cap2 = 1;
        while (cap2 < cap) cap2 <<= 1;
        base.resize(cap);
        strongb.resize(cap);
        polya.resize(cap);
        counts.resize(cap);
        v.resize(cap2);
        path.resize(cap);
        best.resize(cap);
        qualv.resize(cap);
        S.qual = qualv.data();
        isl.resize(cap / 2 + 2);
        seg.resize(cap / 2 + 2);
        ma.resize(cap / 64 + 2);
        mt.resize(cap / 64 + 2);
        mn.resize(cap / 64 + 2);
        mi.resize(cap / 64 + 2);
        mx.resize(cap / 64 + 2);
        scode.resize(RC_SPEC);
        pk.resize(cap / 16 + 3);
        S.pk = pk.data();
        scnt.resize(RC_SPEC_ENTRIES);
        smeta.resize(4);
        memo.resize(4 + RC_MEMO_MAX);
        S.spec_meta = smeta.data();
        S.memo = memo.data();
        sinv.resize(RC_SPEC);
        sret.resize(RC_SPEC);
        skeep.resize(RC_SPEC);
        sthr.resize(RC_SPEC);
        smask.resize(RC_SPEC);
        S.base = base.data();
        S.strongb = strongb.data();
        S.polya = polya.data();
        S.counts = counts.data();
        S.v = v.data();
        S.path = path.data();
        S.best = best.data();
        S.isl = isl.data();
        S.seg = seg.data();
        S.m_a = ma.data();
        S.m_t = mt.data();
        S.m_n = mn.data();
        S.m_inv = mi.data();
        S.m_x = mx.data();
        S.spec_code = scode.data();
        S.spec_cnt = scnt.data();
        S.spec_inv = sinv.data();
        S.spec_ret = sret.data();
        S.spec_keep = skeep.data();
        S.spec_thr = sthr.data();
        S.spec_mask = smask.data();
    }
};

int base_code(char c)
{
    switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    case 'N': return 4;
    default: return 5;
    }
}

void load_read(Buffers &B, HostWave &w, const rco_params *p, const rco_table *t, const char *seq)
{
    int len = (int)strlen(seq);
    B.S.len = len;
    B.S.kcnt = len >= p->k ? len - p->k + 1 : 0;
    for (int i = 0; i < len; ++i) B.S.base[i] = (unsigned char)base_code(seq[i]);
    if (B.S.kcnt > 0) rco_kmer_counts(p, t, seq, B.S.counts);  // stands in for the probe kernel
    rc_build_masks(w, B.S);
    rc_pack_read(w, B.S);
}

}  // namespace

extern "C" {

struct hostsim_stats {
    long probes4, probes1, max_stack, reads;
};

// same contract as rco_correct_batch (oracle/rc_oracle.h)
void hostsim_correct_batch(const rco_params *p, const rco_table *t, rco_batch *b, hostsim_stats *st)
{
    rc_run_params P;
    P.k = p->k;
    P.max_fix_per_k = p->max_fix_per_k;
    P.error_rate = p->error_rate;
    P.bad_qual = (int)(signed char)p->bad_qual;
    std::vector<uint32_t> bsteps(RC_BOUND_STEPS);
    rc_bound_steps_build(P.error_rate, bsteps.data());
    for (int v = 0; v < RC_BS_INLINE; ++v) P.bs[v] = bsteps[v];
    P.bs_ext = getenv("HOSTSIM_NO_BS_EXT") ? nullptr : bsteps.data();
    P.bound_small = nullptr;
    P.flags = getenv("HOSTSIM_NO_ALT") ? RC_PF_NO_ALT : 0;
    Buffers B(RC_MAX_READ_LENGTH + 64);
    HostWave w;
    w.tab = t;
    w.k = p->k;
    const size_t total = b->mode == 1 ? 2 * b->n : b->n;
    std::vector<int> strong(total), info(total);
    auto seq_of = [&](size_t r) -> char * { return r < b->n ? b->seq + b->off[r] : b->seq2 + b->off2[r - b->n]; };
    auto qual_of = [&](size_t r) -> const char * {
        return r < b->n ? b->qual + b->off[r] : b->qual2 + b->off2[r - b->n];
    };
    // threshold kernel
    for (size_t r = 0; r < total; ++r) {
        load_read(B, w, p, t, seq_of(r));
        int inf;
        strong[r] = rc_front_end(w, B.S, P, &inf);
        info[r] = inf;
    }
    // correction kernel
    for (size_t r = 0; r < total; ++r) {
        char *seq = seq_of(r);
        load_read(B, w, p, t, seq);
        w.qualp = qual_of(r);
        for (int i = 0; i < B.S.len; ++i) B.S.qual[i] = (signed char)w.qualp[i];
        int pair_t = -1;
        if (b->mode == 1) {
            size_t mate = r < b->n ? r + b->n : r - b->n;
            pair_t = std::min(strong[r], strong[mate]);
        } else if (b->mode == 2) {
            pair_t = std::min(strong[r], strong[r ^ 1]);
        }
        if (B.S.kcnt > 0 && !(info[r] & 4)) rc_polya_flags(w, B.S, P.k);
        int ret = rc_correct_read(w, B.S, P, pair_t, strong[r], info[r]);
        if (ret > 0) {
            for (int i = 0; i < B.S.len; ++i)
                if (B.S.best[i] != -1) {
                    seq[i] = "ACGT"[B.S.best[i]];
                    B.S.base[i] = (unsigned char)B.S.best[i];
                }
            rc_pack_read(w, B.S);
        }
        int l, m, h;
        rc_kmer_info(w, B.S, P, ret, &l, &m, &h);
        b->ret[r] = ret;
        b->l[r] = l;
        b->m[r] = m;
        b->h[r] = h;
    }
    if (st) {
        st->probes4 = w.gets;
        st->probes1 = 0;
        st->max_stack = w.max_sp;
        if (getenv("HOSTSIM_STATS")) fprintf(stderr, "keep_run calls=%ld sumR=%ld sum_avail=%ld refills=%ld gap_attempts=%ld gap_probes=%ld pushes=%ld pops=%ld reads=%ld alt_runs=%ld alt_full=%ld gets=%ld round_probes=%ld alt_rounds=%ld\n", w.stats[0], w.stats[1], w.stats[2], w.stats[3], w.stats[4], w.stats[5], w.pushes, w.pops, (long)total, w.stats[6], w.stats[7], w.gets, w.stats[8], w.stats[9]);
        st->reads = (long)total;
    }
}
}
