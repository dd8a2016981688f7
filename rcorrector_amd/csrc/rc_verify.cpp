// rc_verify.cpp -- `verify`: accuracy of a corrected read file against the truth a read simulator
// (Mason) left in the header lines.  Host-only companion of the `rcorrector` binary; restates what
// the reference's stand-alone scorer computes and prints (verify.cpp:131-483) so that quality
// regressions of the correction path can be tracked with the same numbers:
//
//   verify reads.cor.fq [-v] [-bv] [-exp] [-noindel]                      (verify.cpp:147-157)
//
// Header fields (verify.cpp:45-56,193-232,290-298): haplotype_infix=<true bases, forward strand>,
// edit_string=<one letter per true base, M = sequenced correctly>, strand=reverse (the true bases
// are reverse-complemented before comparing), exp=high|medium|low (expression class), trim=<n>.
//
// Read level (verify.cpp:234-288): a read "differs" when it is not a prefix-compatible copy of the
// truth; differs & had errors -> FN, differs & had none -> FP, same & had errors -> TP.
// Base level (verify.cpp:300-384): read bases are aligned to true bases (identity when the lengths
// agree, else a longest-common-subsequence alignment, verify.cpp:58-129) and every aligned base is
// scored against its edit letter; true bases skipped inside the aligned span count as FP (M) or
// FN (otherwise).  Lines are limited to 2047 characters and unequal-length alignment to 501
// bases, as in the reference (verify.cpp:8-17).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

namespace {

const size_t LINE_MAX_ = 2048;  // verify.cpp:8-15
const int LCS_MAX = 501;        // verify.cpp:16-17

// complement of the reference's 26-entry tables (verify.cpp:19-24,26-35): letters other than
// ACGT have no code; 3 - (-1) indexes the empty fifth slot, i.e. a NUL that ends the string
char complement(char c)
{
    switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return '\0';
    }
}

void reverse_complement(std::string &s)
{
    std::string r(s.size(), '\0');
    for (size_t i = 0; i < s.size(); ++i) r[i] = complement(s[s.size() - 1 - i]);
    const size_t cut = r.find('\0');
    if (cut != std::string::npos) r.resize(cut);
    s.swap(r);
}

// value of "<tag>...=<value>" inside the header: the text after the first '=' at or after the
// first occurrence of tag (verify.cpp:45-56).  Returns false if the tag does not occur.
bool find_column(const std::string &id, const char *tag, size_t *value_at)
{
    size_t p = id.find(tag);
    if (p == std::string::npos) return false;
    while (p < id.size() && id[p] != '=') ++p;
    *value_at = p < id.size() ? p + 1 : id.size();
    return true;
}

// sscanf("%s"): skip white space, take the run of non-space characters
std::string token_at(const std::string &id, size_t at)
{
    auto is_ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; };
    while (at < id.size() && is_ws(id[at])) ++at;
    size_t e = at;
    while (e < id.size() && !is_ws(id[e])) ++e;
    return id.substr(at, e - at);
}

// verify.cpp:37-43: true when `s` departs from `ref` before `s` ends
bool differs_with_trim(const std::string &ref, const std::string &s)
{
    size_t i = 0;
    while (i < s.size() && i < ref.size() && s[i] == ref[i]) ++i;
    return i < s.size();
}

// align[j] = index of the true base read base j is matched to, or -1 (verify.cpp:58-129).  The
// recurrence prefers the diagonal, then (only above the diagonal, j > i) skipping a read base,
// then (only below, i > j) skipping a true base; ties keep the earlier choice.
void align_read(const std::string &a, const std::string &b, std::vector<int> &align)
{
    const int la = (int)a.size(), lb = (int)b.size();
    align.assign((size_t)(lb > 0 ? lb : 1) + 1, 0);
    if (la == lb) {
        for (int j = 0; j < lb; ++j) align[(size_t)j] = j;
        return;
    }
    if (la > LCS_MAX || lb > LCS_MAX) {
        fprintf(stderr, "verify: reads of different length are aligned up to %d bases only\n", LCS_MAX);
        exit(1);
    }
    std::vector<int> best((size_t)la * (size_t)lb, 0);
    std::vector<signed char> how((size_t)la * (size_t)lb, 1);
    auto at = [&](int i, int j) -> int { return (i < 0 || j < 0) ? 0 : best[(size_t)i * (size_t)lb + (size_t)j]; };
    for (int i = 0; i < la; ++i)
        for (int j = 0; j < lb; ++j) {
            int mx = (a[(size_t)i] == b[(size_t)j] ? 1 : 0) + at(i - 1, j - 1);
            signed char h = 1;
            if (j > i && at(i, j - 1) > mx) {
                mx = at(i, j - 1);
                h = 0;
            }
            if (i > j && at(i - 1, j) > mx) {
                mx = at(i - 1, j);
                h = 2;
            }
            best[(size_t)i * (size_t)lb + (size_t)j] = mx;
            how[(size_t)i * (size_t)lb + (size_t)j] = h;
        }
    int i = la - 1, j = lb - 1;
    while (j >= 0 && i >= 0) {
        const signed char h = how[(size_t)i * (size_t)lb + (size_t)j];
        if (h == 1) {
            align[(size_t)j] = i;
            --i;
            --j;
        } else if (h == 0) {
            align[(size_t)j] = -1;
            --j;
        } else {
            align[(size_t)j] = i - 1;
            --i;
        }
    }
    for (; j >= 0; --j) align[(size_t)j] = -1;
}

struct Tally {
    int tp[4] = {0, 0, 0, 0}, fp[4] = {0, 0, 0, 0}, fn[4] = {0, 0, 0, 0};
};

void report(const char *title, int tp, int fp, int fn)
{
    const double recall = (double)tp / (tp + fn), precision = (double)tp / (tp + fp);
    printf("\n%s\n", title);
    printf("TP: %d\nFP: %d\nFN: %d\n", tp, fp, fn);
    printf("Recall: %lf\nPrecision: %lf\nF-score: %lf\nGain: %lf\n", recall, precision,
           2 * recall * precision / (recall + precision), (double)(tp - fp) / (tp + fn));
}

bool get_line(FILE *fp, std::string &out)
{
    char buf[LINE_MAX_];
    if (!fgets(buf, sizeof buf, fp)) return false;
    out = buf;
    return true;
}

void chomp(std::string &s)
{
    if (!s.empty() && s.back() == '\n') s.pop_back();
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "Usage: verify reads.cor.fq|fa [-v] [-bv] [-exp] [-noindel]\n");
        return 1;
    }
    bool verbose = false, base_verbose = false, use_exp = false, no_indel = false;
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[i], "-v"))
            verbose = true;
        else if (!strcmp(argv[i], "-bv"))
            base_verbose = true;
        else if (!strcmp(argv[i], "-exp"))
            use_exp = true;
        else if (!strcmp(argv[i], "-noindel"))
            no_indel = true;
    }
    FILE *fp = fopen(argv[1], "r");
    if (!fp) {
        fprintf(stderr, "verify: could not open %s\n", argv[1]);
        return 1;
    }
    int c0 = fgetc(fp);
    while (c0 == ' ' || c0 == '\n' || c0 == '\t' || c0 == '\r') c0 = fgetc(fp);
    const bool fastq = c0 != '>';  // verify.cpp:159-167
    rewind(fp);

    int correct_reads = 0, error_reads = 0, untouched_truth = 0, trim_count = 0, trim_sum = 0;
    Tally base, read;
    std::string id, seq, plus, qual, truth, edits;
    std::vector<int> align;
    while (get_line(fp, id)) {
        seq.clear();
        get_line(fp, seq);
        if (fastq) {
            get_line(fp, plus);
            get_line(fp, qual);
        }
        chomp(id);
        chomp(seq);
        size_t at;
        truth = find_column(id, "haplotype_infix", &at) ? token_at(id, at) : std::string();
        if (no_indel && seq.size() != truth.size()) continue;
        if (seq.size() != truth.size()) printf("%s\n%s\n", id.c_str(), seq.c_str());
        edits = find_column(id, "edit_string", &at) ? token_at(id, at) : std::string();
        if (find_column(id, "strand=reverse", &at)) reverse_complement(truth);
        int exp = 3;
        if (find_column(id, "exp", &at)) {
            const std::string v = token_at(id, at);
            exp = v == "high" ? 2 : (v == "medium" ? 1 : (v == "low" ? 0 : 3));
        }
        if (verbose || base_verbose) printf("%s\n", id.c_str());

        const bool had_errors = edits.find_first_not_of('M') != std::string::npos;
        if (differs_with_trim(truth, seq)) {
            ++error_reads;
            if (had_errors) {
                if (verbose) printf("FN\n");
                ++read.fn[exp];
            } else {
                if (verbose) printf("FP\n");
                ++read.fp[exp];
            }
        } else {
            ++correct_reads;
            if (had_errors) {
                if (verbose) printf("TP\n");
                ++read.tp[exp];
            }
        }
        if (!had_errors) ++untouched_truth;
        if (find_column(id, "trim", &at)) {
            ++trim_count;
            trim_sum += atoi(id.c_str() + at);
        }

        // base level, verify.cpp:300-384
        const int la = (int)truth.size(), lb = (int)seq.size();
        auto edit = [&](int i) -> char { return i >= 0 && (size_t)i < edits.size() ? edits[(size_t)i] : '\0'; };
        std::vector<char> visited((size_t)la + 1, 0);
        align_read(truth, seq, align);
        int kind = 0;  // what -bv prints: 1 FP, 2 TP, 3 FN
        for (int i = 0; i < lb; ++i) {
            const int t = align[(size_t)i];
            if (t == -1) {
                ++base.fp[exp];
                continue;
            }
            if (t < la) visited[(size_t)t] = 1;
            const char e = edit(t), tb = t < la ? truth[(size_t)t] : '\0';
            if (e == 'M') {
                if (seq[(size_t)i] != tb) {
                    kind = 1;
                    ++base.fp[exp];
                }
            } else if (e == 'E') {
                if (seq[(size_t)i] == tb) {
                    if (kind == 0) kind = 2;
                    ++base.tp[exp];
                } else {
                    if (kind == 0 || kind == 2) kind = 3;
                    ++base.fn[exp];
                }
            }
        }
        int last = la - 1;
        while (last >= 0 && !visited[(size_t)last]) --last;
        for (int i = 0; i <= last; ++i)
            if (!visited[(size_t)i]) {
                if (edit(i) == 'M')
                    ++base.fp[exp];
                else
                    ++base.fn[exp];
            }
        // unaligned read bases at either end are not held against the corrector
        for (int i = 0; i < lb && align[(size_t)i] == -1; ++i) --base.fp[exp];
        for (int i = lb - 1; i >= 0 && align[(size_t)i] == -1; --i) --base.fp[exp];
        if (base_verbose) {
            if (kind == 1)
                printf("FP\n");
            else if (kind == 2)
                printf("TP\n");
            else if (kind == 3)
                printf("FN\n");
        }
    }
    fclose(fp);

    printf("correct #: %d\nerror #: %d\n", correct_reads, error_reads);
    printf("Original Correct Reads Count: %d\n", untouched_truth);
    printf("Trimmed Reads Count: %d. Average trim length: %lf\n", trim_count, (double)trim_sum / trim_count);
    printf("Overall:\n");
    auto sum = [](const int v[4]) { return v[0] + v[1] + v[2] + v[3]; };
    report("Base level:", sum(base.tp), sum(base.fp), sum(base.fn));
    report("Read level:", sum(read.tp), sum(read.fp), sum(read.fn));
    if (use_exp)
        for (int i = 0; i < 3; ++i) {
            printf("\nExpress level: %d", i);
            report("Base level:", base.tp[i], base.fp[i], base.fn[i]);
            report("Read level:", read.tp[i], read.fp[i], read.fn[i]);
        }
    return 0;
}
