// kvc_select.hip — A7: per-head top-k of the pooled scores (pyramidkv_utils.py:334,
// `attn_cache.topk(k, dim=-1).indices`) on gfx950, optionally followed in the same workgroup by A8 (the
// gather of that head's K/V rows) when k is small.
//
// select_kernel: one 1024-thread workgroup per head, the head's scores held in registers as
// order-preserving integer keys (thread t owns candidates [t*EPT, (t+1)*EPT)).
//   1. threshold: bit-by-bit search of the k-th largest key T (one block-wide count per key bit:
//      16 rounds for bf16/fp16, 32 for fp32) — no atomics, no histogram contention on the few hundred
//      distinct values a head really has;
//   2. membership: every key > T, plus the r = k - count(key > T) candidates equal to T chosen by tie_mode:
//        KVC_TIES_CANONICAL  lowest index first (ordered block scan);
//   3. order: value descending, index ascending (rank sort in LDS for k <= 256, bitonic sort otherwise);
//   4. (fused) gather of the head's k + W rows of K and V straight from the sorted LDS list.
// The scores of a head (n * 2 bytes) are read once from L2; nothing else touches HBM except the rows.
#include "kvc_common.h"
#include "kvc_launch.h"

namespace kvc {

static constexpr int SEL_THREADS = 1024;
static constexpr int SEL_WAVES = SEL_THREADS / 64;
static constexpr int SEL_MAX_EPT = 64;          // n <= 65536 candidates per head

size_t select_lds_bytes(int k) {
    int p = 1;
    while (p < k) p <<= 1;
    return (size_t)p * 8 + (2 * SEL_WAVES + 256) * 4;
}

// Sum of one small per-thread count over the block; result broadcast to every thread.
// `slot` alternates between two LDS arrays so that one barrier per call is enough.
__device__ __forceinline__ uint32_t block_sum(uint32_t wave_total, uint32_t* buf /*[2][SEL_WAVES]*/, int slot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) buf[slot * SEL_WAVES + wave] = wave_total;
    __syncthreads();
    uint32_t s = 0;
#pragma unroll
    for (int w = 0; w < SEL_WAVES; ++w) s += buf[slot * SEL_WAVES + w];
    return s;
}

// Exclusive prefix (in thread order) of a per-thread count, plus the block total.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* buf, int slot, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) buf[slot * SEL_WAVES + wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SEL_WAVES; ++w) { const uint32_t c = buf[slot * SEL_WAVES + w]; if (w < wave) base += c; tot += c; }
    *total = tot;
    return base + inc - v;
}

template <int DT, int EPT>
__global__ __launch_bounds__(SEL_THREADS) void select_kernel(const SelectArgs a) {
    typedef typename Dt<DT>::raw raw;
    constexpr int KB = Key<DT>::bits;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* list = reinterpret_cast<unsigned long long*>(smem);      // [pow2]
    uint32_t* buf = reinterpret_cast<uint32_t*>(smem + (size_t)a.pow2 * 8);      // [2][SEL_WAVES] + spare

    const int tid = threadIdx.x;
    const int n = a.n, k = a.k;
    const int head = blockIdx.x;
    const raw* s = reinterpret_cast<const raw*>(a.scores) + (int64_t)head * n;

    // ---- candidates into registers ----
    uint32_t key[EPT];
    const int i0 = tid * EPT;
#pragma unroll
    for (int e = 0; e < EPT; ++e) key[e] = (i0 + e < n) ? Key<DT>::of(s[i0 + e]) : 0u;

    // ---- 1. threshold: largest T with count(key >= T) >= k ----
    uint32_t T = 0;
    int slot = 0;
    for (int bit = KB - 1; bit >= 0; --bit) {
        const uint32_t cand = T | (1u << bit);
        uint32_t c = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) c += (uint32_t)__popcll(__ballot(key[e] >= cand && (i0 + e) < n));
        if (block_sum(c, buf, slot) >= (uint32_t)k) T = cand;
        slot ^= 1;
    }
    // ---- 2. membership ----
    uint32_t cg = 0, ce = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const bool valid = (i0 + e) < n;
        cg += (valid && key[e] > T) ? 1u : 0u;
        ce += (valid && key[e] == T) ? 1u : 0u;
    }
    uint32_t tot_gt, tot_eq, tot_take;
    (void)block_excl_scan(cg, buf, slot, &tot_gt); slot ^= 1;
    const uint32_t r = (uint32_t)k - tot_gt;                       // ties needed (>= 1)
    const uint32_t eq_before = block_excl_scan(ce, buf, slot, &tot_eq); slot ^= 1;
    const uint32_t eq_take = eq_before >= r ? 0u : (r - eq_before < ce ? r - eq_before : ce);
    uint32_t pos = block_excl_scan(cg + eq_take, buf, slot, &tot_take); slot ^= 1;
    {
        uint32_t eq_seen = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const bool valid = (i0 + e) < n;
            bool take = valid && key[e] > T;
            if (valid && key[e] == T) { take = eq_seen < eq_take; ++eq_seen; }
            if (take) list[pos++] = ((unsigned long long)(0xffffffffu - key[e]) << 32) | (uint32_t)(i0 + e);
        }
    }
    __syncthreads();

    // ---- 3. order ----
    int64_t* out = a.idx + (int64_t)head * k;
    if (k <= 256) {
        // rank sort spread over the whole workgroup: composites are unique, rank = number of smaller composites.
        // thread t: element e = t % E (E = pow2 >= k), comparison slice t / E; slice partials are summed in LDS.
        const int E = a.pow2 < 32 ? 32 : a.pow2;            // 32..256
        const int slices = SEL_THREADS / E;                 // 4..32
        const int e = tid % E, sl = tid / E;
        const int per = (k + slices - 1) / slices;
        uint32_t* rk = buf + 2 * SEL_WAVES;                 // [256] partial-rank accumulators (LDS, after the count buffers)
        if (tid < E) rk[tid] = 0;
        __syncthreads();
        unsigned long long mine = e < k ? list[e] : 0ull;
        if (e < k) {
            uint32_t part = 0;
            const int u0 = sl * per, u1 = u0 + per < k ? u0 + per : k;
            for (int u = u0; u < u1; ++u) part += list[u] < mine ? 1u : 0u;
            if (part) atomicAdd(&rk[e], part);
        }
        __syncthreads();
        if (tid < k) list[rk[tid]] = mine;                  // tid < k <= E: this thread's `mine` is element tid (slice 0)
        __syncthreads();
    } else {
        const int P = a.pow2;
        for (int i = k + tid; i < P; i += SEL_THREADS) list[i] = ~0ull;
        __syncthreads();
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
    }
    for (int t = tid; t < k; t += SEL_THREADS) out[t] = (int64_t)(list[t] & 0xffffffffull);

    // ---- 4. fused gather of this head's rows (small k only): all loads of a batch are issued before its stores ----
    if (a.fuse) {
        const int lanes_per_row = a.gk.row_bytes >> 4;
        const int cap = k + a.gk.window;
        const int per_tensor = cap * lanes_per_row;           // 16-byte pieces per tensor
        const int total = 2 * per_tensor;
        const int b = head / a.gk.n_q_heads, h = head % a.gk.n_q_heads;
        const int64_t esz = a.gk.esize;
        constexpr int BATCH = 4;
        for (int base = 0; base < total; base += BATCH * SEL_THREADS) {
            uint4 v[BATCH];
            char* dst[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int piece = base + u * SEL_THREADS + tid;
                dst[u] = nullptr;
                if (piece < total) {
                    const int which = piece >= per_tensor ? 1 : 0;
                    const int pp = piece - which * per_tensor;
                    const int t = pp / lanes_per_row, cc = pp % lanes_per_row;
                    const GatherArgs& g = which == 0 ? a.gk : a.gv;
                    const int64_t srow = t < k ? (int64_t)(list[t] & 0xffffffffull) : (int64_t)(g.q_len - g.window) + (t - k);
                    const char* src = reinterpret_cast<const char*>(g.src) +
                        ((int64_t)b * g.stride_b + (int64_t)(h / g.group) * g.stride_h + srow * g.stride_l) * esz + cc * 16;
                    v[u] = *reinterpret_cast<const uint4*>(src);
                    dst[u] = reinterpret_cast<char*>(g.out) + ((int64_t)head * cap + t) * g.row_bytes + cc * 16;
                }
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u)
                if (dst[u]) *reinterpret_cast<uint4*>(dst[u]) = v[u];
        }
    }
}

template <int DT, int EPT>
static int launch_t(const SelectArgs& a, hipStream_t st) {
    const size_t lds = select_lds_bytes(a.k);
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&select_kernel<DT, EPT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return KVC_ERR_HIP;
    }
    hipLaunchKernelGGL((select_kernel<DT, EPT>), dim3((unsigned)a.heads), dim3(SEL_THREADS), lds, st, a);
    return 0;
}

template <int DT>
static int launch_dt(const SelectArgs& a, hipStream_t st) {
    const int ept = (a.n + SEL_THREADS - 1) / SEL_THREADS;
    if (ept <= 8) return launch_t<DT, 8>(a, st);
    if (ept <= 16) return launch_t<DT, 16>(a, st);
    if (ept <= 32) return launch_t<DT, 32>(a, st);
    if (ept <= SEL_MAX_EPT) return launch_t<DT, SEL_MAX_EPT>(a, st);
    return KVC_ERR_UNSUPPORTED;
}

int launch_select(const SelectArgs& a, int dtype, int tie_mode, hipStream_t st) {
    if (tie_mode != KVC_TIES_CANONICAL) return KVC_ERR_UNSUPPORTED;
    switch (dtype) {
        case KVC_BF16: return launch_dt<KVC_BF16>(a, st);
        case KVC_FP16: return launch_dt<KVC_FP16>(a, st);
        case KVC_FP32: return launch_dt<KVC_FP32>(a, st);
    }
    return KVC_ERR_INVALID;
}

}  // namespace kvc
