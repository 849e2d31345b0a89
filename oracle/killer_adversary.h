// killer_adversary.h — TEST INFRASTRUCTURE (never linked into or called by the product).
//
// Worst-case inputs for the selection step of the hot path.  torch-CPU topk (aten TopKImpl.h) is libstdc++'s
// std::nth_element + std::sort (or std::partial_sort) on (value, index) pairs, and attn_score.sort (AdaKV,
// pyramidkv_utils.py:702) is std::sort: introselect / introsort, a median-of-three quicksort with a depth budget of
// 2 * lg(n) partitions, after which they fall back to heap select / heap sort.  Random or plateau-heavy scores never spend
// that budget, so the fallbacks of the emulation (kvc_stl_emul.h, kvc_select_exact.hip) would go untested.
//
// These functions build inputs that DO spend it, with M. D. McIlroy's adversary ("A Killer Adversary for Quicksort",
// 1999): the routine under attack — the REAL libstdc++ one, on this machine — runs with a comparator that decides values
// lazily ("gas" items are frozen to the next small value as late as possible), so that every pivot it picks is extreme.
// Replayed with an ordinary comparator the frozen values reproduce the same comparison outcomes, hence the same
// degenerate partitions, and the depth budget runs out.
#pragma once
#include <algorithm>
#include <vector>

namespace killer {

struct Adversary {
    std::vector<int> val;       // frozen value of item i (its rank in the attacked order), or gas
    int gas, nsolid = 0, candidate = 0;
    explicit Adversary(int n) : val(n, n - 1), gas(n - 1) {}
    // "a sorts before b".  The product's comparator is score(a) > score(b): the emitted score of item i is gas - val[i].
    bool less(int a, int b) {
        if (val[a] == gas && val[b] == gas) {
            if (a == candidate) val[a] = nsolid++; else val[b] = nsolid++;
        }
        if (val[a] == gas) candidate = a; else if (val[b] == gas) candidate = b;
        return val[a] < val[b];
    }
    std::vector<int> scores() const {      // integer score codes: larger = higher score, equal codes = equal scores
        std::vector<int> s(val.size());
        for (size_t i = 0; i < val.size(); ++i) s[i] = gas - val[i];
        return s;
    }
};

// torch-CPU topk's nth_element + sort regime (k * 64 > n), attacked as a whole
inline std::vector<int> topk_scores(int n, int k) {
    Adversary A(n);
    std::vector<int> q(n);
    for (int i = 0; i < n; ++i) q[i] = i;
    auto less = [&](int a, int b) { return A.less(a, b); };
    std::nth_element(q.begin(), q.begin() + (k - 1), q.end(), less);
    std::sort(q.begin(), q.begin() + (k - 1), less);
    return A.scores();
}

// std::sort of the whole row (AdaKV / HeadKV's descending sort)
inline std::vector<int> sort_scores(int n) {
    Adversary A(n);
    std::vector<int> q(n);
    for (int i = 0; i < n; ++i) q[i] = i;
    std::sort(q.begin(), q.end(), [&](int a, int b) { return A.less(a, b); });
    return A.scores();
}

}  // namespace killer
