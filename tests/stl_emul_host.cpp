// Host harness for kvcache_factory_amd/csrc/kvc_stl_emul.h: the move-for-move restatement of libstdc++'s
// partial_sort / nth_element + sort (what torch-CPU topk runs) is compared with the REAL library on tie-heavy
// inputs.  Built with g++ -fsanitize=address,undefined by tests/test_stl_emul.py.  Also mirrors the kernel's
// streamed heap-select (64 candidates screened per step).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <utility>
#include <vector>

#include "../kvcache_factory_amd/csrc/kvc_stl_emul.h"
#include "../oracle/killer_adversary.h"

using kvc::u64;
typedef std::pair<float, int64_t> elem;

static uint32_t key_of(float f) { uint32_t u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

static std::vector<int64_t> real_topk(const std::vector<float>& v, int k) {
    const int n = (int)v.size();
    std::vector<elem> q(n);
    for (int i = 0; i < n; ++i) q[i] = elem(v[i], i);
    auto comp = [](const elem& x, const elem& y) { return x.first > y.first; };
    if ((int64_t)k * 64 <= n) std::partial_sort(q.begin(), q.begin() + k, q.end(), comp);
    else { std::nth_element(q.begin(), q.begin() + (k - 1), q.end(), comp); std::sort(q.begin(), q.begin() + (k - 1), comp); }
    std::vector<int64_t> r(k);
    for (int t = 0; t < k; ++t) r[t] = q[t].second;
    return r;
}

static std::vector<int64_t> emul_topk(const std::vector<float>& v, int k) {
    const int n = (int)v.size();
    std::vector<int> stack(3 * 96);
    std::vector<int64_t> r(k);
    if ((int64_t)k * 64 <= n) {
        std::vector<u64> heap(k);
        for (int i = 0; i < k; ++i) heap[i] = ((u64)key_of(v[i]) << 32) | (uint32_t)i;
        kvc::Arr H{heap.data()};
        kvc::make_heap_(H, 0, k);
        for (int base = k; base < n; base += 64) {            // same screening as select_exact_kernel
            u64 root = H.get(0);
            bool pend[64];
            for (int l = 0; l < 64; ++l) pend[l] = base + l < n && key_of(v[base + l]) > (uint32_t)(root >> 32);
            for (int l = 0; l < 64; ++l) {
                if (!pend[l]) continue;
                const uint32_t kk = key_of(v[base + l]);
                if (kk > (uint32_t)(root >> 32)) { kvc::adjust_heap_(H, 0, 0, k, ((u64)kk << 32) | (uint32_t)(base + l)); root = H.get(0); }
            }
        }
        kvc::sort_heap_(H, 0, k);
        for (int t = 0; t < k; ++t) r[t] = (int64_t)(heap[t] & 0xffffffffull);
    } else {
        std::vector<u64> arr(n);
        for (int i = 0; i < n; ++i) arr[i] = ((u64)key_of(v[i]) << 32) | (uint32_t)i;
        kvc::Arr A{arr.data()};
        kvc::introselect_(A, 0, k - 1, n, kvc::lg_(n) * 2);
        kvc::sort_(A, 0, k - 1, stack.data());
        for (int t = 0; t < k; ++t) r[t] = (int64_t)(arr[t] & 0xffffffffull);
    }
    return r;
}

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 300;
    std::mt19937 rng(12345);
    long checked = 0;
    for (int t = 0; t < trials; ++t) {
        int n, k, distinct;
        switch (t % 8) {
            case 0: n = 7992; k = 120; break;                 // C2: partial_sort regime
            case 1: n = 7992; k = 234; break;                 // C4 layer 0: nth_element regime
            case 2: n = 640; k = 10; break;                   // k*64 == n
            case 3: n = 640; k = 11; break;
            case 4: n = 1 + rng() % 3000; k = 1 + rng() % n; break;
            case 5: n = 24; k = 24; break;                    // k == n
            case 6: n = 64 + rng() % 5000; k = 1; break;
            default: n = 2000 + rng() % 30000; k = 1 + rng() % 4000; if (k > n) k = n; break;
        }
        distinct = (t % 3 == 0) ? 3 : (t % 3 == 1 ? 260 : 100000);      // tie-heavy .. nearly tie-free
        std::vector<float> v(n);
        for (int i = 0; i < n; ++i) v[i] = (float)(rng() % distinct) / (float)distinct;
        if (t % 5 == 0) for (int i = 0; i + 7 < n; i += 7) { float m = v[i]; for (int j = 1; j < 7; ++j) m = std::max(m, v[i + j]); for (int j = 0; j < 7; ++j) v[i + j] = m; }   // plateaus
        if (t % 11 == 0) std::fill(v.begin(), v.end(), 0.125f);           // all equal
        if (t % 13 == 0) std::sort(v.begin(), v.end());                    // sorted input (depth-limit pressure)
        const auto a = real_topk(v, k), b = emul_topk(v, k);
        for (int i = 0; i < k; ++i)
            if (a[i] != b[i]) { printf("MISMATCH trial %d n=%d k=%d at %d: real %ld emul %ld\n", t, n, k, i, (long)a[i], (long)b[i]); return 1; }
        checked += k;
    }
    // AdaKV / HeadKV: a prefix of the FULL std::sort (torch-CPU sort, descending) through sort_prefix_
    for (int t = 0; t < trials; ++t) {
        const int n = (t % 4 == 0) ? 7992 : 17 + (int)(rng() % 9000);
        int want = (t % 5 == 0) ? n : 1 + (int)(rng() % (n < 600 ? n : 600));
        if (t % 7 == 0) want = 1 + (int)(rng() % n);
        const int distinct = (t % 3 == 0) ? 3 : (t % 3 == 1 ? 260 : 100000);
        std::vector<float> v(n);
        for (int i = 0; i < n; ++i) v[i] = (float)(rng() % distinct) / (float)distinct;
        if (t % 5 == 1) for (int i = 0; i + 7 < n; i += 7) { float m = v[i]; for (int j = 1; j < 7; ++j) m = std::max(m, v[i + j]); for (int j = 0; j < 7; ++j) v[i + j] = m; }
        if (t % 11 == 0) std::fill(v.begin(), v.end(), 0.125f);
        if (t % 13 == 0) std::sort(v.begin(), v.end());
        std::vector<elem> q(n);
        for (int i = 0; i < n; ++i) q[i] = elem(v[i], i);
        std::sort(q.begin(), q.end(), [](const elem& x, const elem& y) { return x.first > y.first; });
        std::vector<u64> arr(n);
        for (int i = 0; i < n; ++i) arr[i] = ((u64)key_of(v[i]) << 32) | (uint32_t)i;
        kvc::Arr A{arr.data()};
        std::vector<int> stack(3 * 96);
        std::vector<int64_t> out(want, -1);
        kvc::sort_prefix_(A, 0, n, want, stack.data(), out.data());
        for (int i = 0; i < want; ++i)
            if (out[i] != q[i].second) { printf("SORT-PREFIX MISMATCH trial %d n=%d want=%d at %d: real %ld emul %ld\n", t, n, want, i, (long)q[i].second, (long)out[i]); return 1; }
        checked += want;
    }
    // Adversarial rows (oracle/killer_adversary.h): every partition degenerate, the depth budgets of introselect and of
    // introsort run out and the heap fallbacks of the emulation run
    const int kn[][2] = {{2000, 60}, {600, 100}, {1000, 999}, {1000, 1000}, {7992, 234}, {7992, 2040}, {20000, 400}, {64, 64}, {300, 40}};
    for (const auto& c : kn) {
        const std::vector<int> codes = killer::topk_scores(c[0], c[1]);
        std::vector<float> v(codes.begin(), codes.end());
        const auto a = real_topk(v, c[1]), b = emul_topk(v, c[1]);
        for (int i = 0; i < c[1]; ++i)
            if (a[i] != b[i]) { printf("KILLER MISMATCH n=%d k=%d at %d: real %ld emul %ld\n", c[0], c[1], i, (long)a[i], (long)b[i]); return 1; }
        checked += c[1];
    }
    for (int n : {64, 65, 600, 4000, 7992}) {
        const std::vector<int> codes = killer::sort_scores(n);
        std::vector<elem> q(n);
        for (int i = 0; i < n; ++i) q[i] = elem((float)codes[i], i);
        std::sort(q.begin(), q.end(), [](const elem& x, const elem& y) { return x.first > y.first; });
        for (int want : {n, n / 3 + 1, 17}) {
            std::vector<u64> arr(n);
            for (int i = 0; i < n; ++i) arr[i] = ((u64)key_of((float)codes[i]) << 32) | (uint32_t)i;
            kvc::Arr A{arr.data()};
            std::vector<int> stack(3 * 96);
            std::vector<int64_t> out(want, -1);
            kvc::sort_prefix_(A, 0, n, want, stack.data(), out.data());
            for (int i = 0; i < want; ++i)
                if (out[i] != q[i].second) { printf("KILLER SORT-PREFIX MISMATCH n=%d want=%d at %d\n", n, want, i); return 1; }
            checked += want;
        }
    }
    printf("OK %d trials, %ld indices identical to libstdc++\n", trials, checked);
    return 0;
}
