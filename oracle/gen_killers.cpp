// gen_killers.cpp — TEST INFRASTRUCTURE (never linked into or called by the product).
// Prints the adversarial score rows of killer_adversary.h (one line per case: name n k code_0 ... code_{n-1}) for
// oracle/gen_killers.py, which turns them into tests/golden/killer_*.npz; stderr gets the number of comparisons the real
// libstdc++ routine makes on each row next to the count for the same values shuffled.
#include <cstdio>
#include <cstdlib>

#include "killer_adversary.h"

static long comparisons(const std::vector<int>& score, int k) {
    const int n = (int)score.size();
    std::vector<int> q(n);
    for (int i = 0; i < n; ++i) q[i] = i;
    long c = 0;
    auto less = [&](int a, int b) { ++c; return score[a] > score[b]; };
    std::nth_element(q.begin(), q.begin() + (k - 1), q.end(), less);
    std::sort(q.begin(), q.begin() + (k - 1), less);
    return c;
}

static void topk_case(int n, int k) {
    const std::vector<int> s = killer::topk_scores(n, k);
    std::printf("topk_n%d_k%d %d %d", n, k, n, k);
    for (int v : s) std::printf(" %d", v);
    std::printf("\n");
    std::vector<int> sh(s);
    std::srand(12345);
    for (int i = n - 1; i > 0; --i) std::swap(sh[i], sh[std::rand() % (i + 1)]);
    std::fprintf(stderr, "topk n=%-6d k=%-5d comparisons: %8ld   (the same values shuffled: %8ld)\n", n, k, comparisons(s, k), comparisons(sh, k));
}

int main() {
    topk_case(2000, 60);          // the sort of the 59 leaders is one <= 64 range from the start
    topk_case(600, 100);
    topk_case(1000, 999);
    topk_case(1000, 1000);
    topk_case(7992, 234);         // C4 layer 0's shape
    topk_case(7992, 2040);
    topk_case(20000, 400);        // array in the workspace
    return 0;
}
