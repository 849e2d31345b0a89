// kvc_select_exact.hip — A7 with tie_mode KVC_TIES_TORCH_CPU: membership AND order of equal scores exactly as
// torch-CPU `topk` produces them, so indices are bit-identical to the reference's CPU path (pyramidkv_utils.py:334).
//
// torch-CPU topk (aten/native/cpu TopKImpl.h) sorts pair<value,index> with a value-only comparator using
// libstdc++: std::partial_sort when k*64 <= n, otherwise std::nth_element(k-1) + std::sort(first k-1).  What those do
// with equal values is a property of their exact sequence of moves, so this kernel EXECUTES that sequence:
// one wavefront per head walks the same heap / introselect / introsort steps on order-preserving integer keys.
// Every lane of the wave runs the same scalar program (values are made wave-uniform with readfirstlane), the
// array lives in LDS when it fits (n <= 18000 elements of 8 bytes) and in the caller's workspace otherwise.
// The partial_sort scan over the n-k tail is the only data-parallel part: 64 candidates are tested against the
// heap root per step and only the ones that beat it are replayed serially.
//
// This is the slow, exact path (serial by construction); the canonical path is kvc_select.hip.
#include "kvc_common.h"
#include "kvc_launch.h"
#include "kvc_stl_emul.h"

namespace kvc {

// ---------------------------------------------------------------------------------------------------------
// grid = heads, block = 64 (one wave).  LDS: array (or heap) + introsort stack.
// ---------------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(64) void select_exact_kernel(const SelectArgs a, u64* gscratch /*[heads][n] or null*/, int arr_in_lds) {
    typedef typename Dt<DT>::raw raw;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* stack = reinterpret_cast<int*>(smem);                        // 3 * 96 ints
    u64* lds_arr = reinterpret_cast<u64*>(smem + 1152);
    const int lane = threadIdx.x, head = blockIdx.x, item = blockIdx.y;
    const int n = a.n, k = a.k.v[item];
    const raw* s = reinterpret_cast<const raw*>(a.scores.p[item]) + (int64_t)head * n;
    int64_t* out = reinterpret_cast<int64_t*>(const_cast<void*>(a.idx.p[item])) + (int64_t)head * k;
    const bool use_partial_sort = (int64_t)k * 64 <= (int64_t)n;

    if (use_partial_sort) {
        // heap of the first k in LDS; the tail is streamed 64 at a time
        Arr H{lds_arr};
        for (int i = lane; i < k; i += 64) lds_arr[i] = ((u64)Key<DT>::of(s[i]) << 32) | (uint32_t)i;
        __syncthreads();
        make_heap_(H, 0, k);
        for (int base = k; base < n; base += 64) {
            const int i = base + lane;
            const uint32_t key = i < n ? Key<DT>::of(s[i]) : 0u;
            u64 root = H.get(0);
            unsigned long long pending = __ballot(i < n && key > (uint32_t)(root >> 32));
            while (pending) {
                const int src = __builtin_ctzll(pending);
                pending &= pending - 1;
                const uint32_t kk = __builtin_amdgcn_readlane(key, src);
                if (kk > (uint32_t)(root >> 32)) {                       // re-test against the live root
                    adjust_heap_(H, 0, 0, k, ((u64)kk << 32) | (uint32_t)(base + src));
                    root = H.get(0);
                }
            }
        }
        sort_heap_(H, 0, k);
        __syncthreads();
        for (int t = lane; t < k; t += 64) out[t] = (int64_t)(lds_arr[t] & 0xffffffffull);
    } else {
        u64* arr = arr_in_lds ? lds_arr : gscratch + ((int64_t)item * a.heads + head) * n;
        for (int i = lane; i < n; i += 64) arr[i] = ((u64)Key<DT>::of(s[i]) << 32) | (uint32_t)i;
        __syncthreads();
        Arr A{arr};
        // std::nth_element(first, first + k - 1, last)
        if (k - 1 != n) introselect_(A, 0, k - 1, n, lg_(n) * 2);
        // std::sort(first, first + k - 1)
        sort_(A, 0, k - 1, stack);
        __syncthreads();
        for (int t = lane; t < k; t += 64) out[t] = (int64_t)(arr[t] & 0xffffffffull);
    }
}

size_t select_exact_scratch_bytes(int heads, int n, int k) {
    const bool partial = (int64_t)k * 64 <= (int64_t)n;
    if (partial || n <= 18000) return 0;
    return (size_t)heads * n * 8;
}

template <int DT>
static int launch_exact_t(const SelectArgs& a, void* scratch, hipStream_t st) {
    // LDS sized for the worst item: a heap of k_max (partial_sort regime) or the whole array (nth_element regime)
    bool any_nth = false;
    for (int i = 0; i < a.n_items; ++i) any_nth = any_nth || !((int64_t)a.k.v[i] * 64 <= (int64_t)a.n);
    const int in_lds = !any_nth || a.n <= 18000;
    const size_t elems = in_lds ? (any_nth ? (size_t)a.n : (size_t)a.k_max) : (size_t)a.k_max;
    const size_t lds = 1152 + elems * 8;
    if (!in_lds && !scratch) return KVC_ERR_WORKSPACE;
    static LdsCache lds_cache = {};
    if (ensure_lds(reinterpret_cast<const void*>(&select_exact_kernel<DT>), lds, lds_cache) != 0) return KVC_ERR_HIP;
    hipLaunchKernelGGL((select_exact_kernel<DT>), dim3((unsigned)a.heads, (unsigned)a.n_items), dim3(64), lds, st, a,
                       reinterpret_cast<u64*>(scratch), in_lds);
    return 0;
}

int launch_select_exact(const SelectArgs& a, int dtype, void* scratch, hipStream_t st) {
    switch (dtype) {
        case KVC_BF16: return launch_exact_t<KVC_BF16>(a, scratch, st);
        case KVC_FP16: return launch_exact_t<KVC_FP16>(a, scratch, st);
        case KVC_FP32: return launch_exact_t<KVC_FP32>(a, scratch, st);
    }
    return KVC_ERR_INVALID;
}

}  // namespace kvc
