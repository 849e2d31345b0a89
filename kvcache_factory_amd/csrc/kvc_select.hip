// kvc_select.hip — A7: per-head top-k of the pooled scores (pyramidkv_utils.py:334,
// `attn_cache.topk(k, dim=-1).indices`) on gfx950.
//
// select_canon_kernel (tie_mode KVC_TIES_CANONICAL): one 1024-thread workgroup per head.
//   1. MSB-first radix select over order-preserving integer keys (16-bit keys for bf16/fp16 scores:
//      two 8-bit passes; 32-bit keys for fp32: four) -> threshold key T and r = how many of the
//      candidates equal to T are needed;
//   2. one ordered sweep: every key > T plus the r lowest-index keys == T go to an LDS list as
//      (inverted key << 32 | index);
//   3. bitonic sort of the list in LDS -> value descending, index ascending; written as int64.
// The scores of a head (n * 2 bytes) are read three times from L2; nothing else touches HBM.
#include "kvc_common.h"
#include "kvc_launch.h"

namespace kvc {

static constexpr int SEL_THREADS = 1024;
static constexpr int SEL_WAVES = SEL_THREADS / 64;

size_t select_lds_bytes(int k) {
    int p = 1;
    while (p < k) p <<= 1;
    return (size_t)p * 8 + (256 + 256 + 16 + SEL_WAVES) * 4;
}

template <int DT>
__global__ __launch_bounds__(SEL_THREADS) void select_canon_kernel(const SelectArgs a) {
    typedef typename Dt<DT>::raw raw;
    constexpr int KB = Key<DT>::bits;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* list = reinterpret_cast<unsigned long long*>(smem);
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem + (size_t)a.pow2 * 8);   // [256]
    uint32_t* scan = hist + 256;                                               // [256]
    uint32_t* misc = scan + 256;                                               // [16]
    uint32_t* wcnt = misc + 16;                                                // [SEL_WAVES]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n, k = a.k;
    const raw* s = reinterpret_cast<const raw*>(a.scores) + (int64_t)blockIdx.x * n;

    // ---- 1. radix select of the k-th largest key ----
    uint32_t prefix = 0;
    uint32_t k_rem = (uint32_t)k;
    for (int shift = KB - 8; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += SEL_THREADS) {
            const uint32_t key = Key<DT>::of(s[i]);
            if (shift + 8 == KB || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        // inclusive suffix sums: scan[d] = sum_{d' >= d} hist[d']
        if (tid < 256) scan[tid] = hist[tid];
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            uint32_t v = 0;
            if (tid < 256) v = scan[tid] + (tid + off < 256 ? scan[tid + off] : 0u);
            __syncthreads();
            if (tid < 256) scan[tid] = v;
            __syncthreads();
        }
        if (tid < 256) {
            const uint32_t ge = scan[tid], gt = tid < 255 ? scan[tid + 1] : 0u;
            if (ge >= k_rem && gt < k_rem) { misc[0] = (uint32_t)tid; misc[1] = k_rem - gt; }
        }
        __syncthreads();
        prefix = (prefix << 8) | misc[0];
        k_rem = misc[1];
        __syncthreads();
    }
    const uint32_t T = prefix, r = k_rem;

    // ---- 2. ordered sweep: keys > T, and the r first keys == T ----
    if (tid == 0) misc[2] = 0;
    __syncthreads();
    uint32_t eq_base = 0;
    for (int base = 0; base < n; base += SEL_THREADS) {
        const int i = base + tid;
        uint32_t key = 0;
        bool gt = false, eq = false;
        if (i < n) { key = Key<DT>::of(s[i]); gt = key > T; eq = key == T; }
        const unsigned long long bal = __ballot(eq);
        const uint32_t lane_prefix = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t wave_off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < SEL_WAVES; ++w) { const uint32_t c = wcnt[w]; if (w < wave) wave_off += c; total += c; }
        const bool take = gt || (eq && (eq_base + wave_off + lane_prefix) < r);
        if (take) {
            const uint32_t slot = atomicAdd(&misc[2], 1u);
            list[slot] = ((unsigned long long)(0xffffffffu - key) << 32) | (uint32_t)i;
        }
        eq_base += total;
        __syncthreads();
    }
    for (int i = k + tid; i < a.pow2; i += SEL_THREADS) list[i] = ~0ull;
    __syncthreads();

    // ---- 3. bitonic sort (ascending composite == value descending, index ascending) ----
    const int P = a.pow2;
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < P / 2; t += SEL_THREADS) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool asc = (lo & size) == 0;
                const unsigned long long x = list[lo], y = list[hi];
                if ((x > y) == asc) { list[lo] = y; list[hi] = x; }
            }
            __syncthreads();
        }
    }
    int64_t* out = a.idx + (int64_t)blockIdx.x * k;
    for (int t = tid; t < k; t += SEL_THREADS) out[t] = (int64_t)(list[t] & 0xffffffffull);
}

template <int DT>
static int launch_canon(const SelectArgs& a, hipStream_t st) {
    const size_t lds = select_lds_bytes(a.k);
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&select_canon_kernel<DT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return KVC_ERR_HIP;
    }
    hipLaunchKernelGGL((select_canon_kernel<DT>), dim3((unsigned)a.heads), dim3(SEL_THREADS), lds, st, a);
    return 0;
}

int launch_select(const SelectArgs& a, int dtype, int tie_mode, hipStream_t st) {
    if (tie_mode != KVC_TIES_CANONICAL) return KVC_ERR_UNSUPPORTED;
    switch (dtype) {
        case KVC_BF16: return launch_canon<KVC_BF16>(a, st);
        case KVC_FP16: return launch_canon<KVC_FP16>(a, st);
        case KVC_FP32: return launch_canon<KVC_FP32>(a, st);
    }
    return KVC_ERR_INVALID;
}

}  // namespace kvc
