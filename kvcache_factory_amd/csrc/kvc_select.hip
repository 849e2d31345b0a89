// kvc_select.hip — A7: per-head top-k of the pooled scores (pyramidkv_utils.py:334,
// `attn_cache.topk(k, dim=-1).indices`) on gfx950, optionally followed in the same workgroup by A8 (the
// gather of that head's K/V rows) when k is small.
//
// select_kernel: one 256-thread workgroup per head, the head's scores held in registers as
// order-preserving integer keys (thread t owns candidates [t*EPT, (t+1)*EPT)).
//   1. threshold: bit-by-bit search of the k-th largest key T (one block-wide count per key bit:
//      16 rounds for bf16/fp16, 32 for fp32) — no atomics, no histogram contention on the few hundred
//      distinct values a head really has.  Measured issue-bound (stamps, round 1): 4 waves x 32 keys per lane
//      spend a third of the instructions of 16 waves x 8 keys;
//   2. membership: every key > T, plus the r = k - count(key > T) candidates equal to T chosen by tie_mode:
//        KVC_TIES_CANONICAL  lowest index first (ordered block scan);
//   3. order: value descending, index ascending (rank sort in LDS for k <= 256, bitonic sort otherwise);
//   4. (fused) gather of the head's k + W rows of K and V straight from the sorted LDS list.
// The scores of a head (n * 2 bytes) are read once from L2; nothing else touches HBM except the rows.
#include "kvc_common.h"
#include "kvc_launch.h"

namespace kvc {

// Diagnostic build only (-DKVC_STAMPS): phase stamps of wave 0, written over this head's index output.
#if defined(KVC_STAMPS)
#define KVC_SSTAMP(slot)                                                                     \
    do {                                                                                     \
        unsigned long long t_;                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");         \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        if (tid == 0) stamps_[slot] = t_;                                                    \
    } while (0)
#else
#define KVC_SSTAMP(slot) do { } while (0)
#endif

static constexpr int SEL_MAX_WAVES = 16;           // workgroup size THR is a template parameter: 256, 512 or 1024 threads

size_t select_lds_bytes(int k) {
    int p = 1;
    while (p < k) p <<= 1;
    return (size_t)p * 8 + (2 * SEL_MAX_WAVES + 256) * 4;
}

// Wave-wide sum / exclusive prefix of a SMALL per-lane count (< 2^NB) without any LDS round trip: one ballot per
// value bit, popcount (s_bcnt1) for the total and mbcnt for the lanes below.  (The ds_bpermute butterflies these
// replace cost ~120 cycles per step and dominated the kernel: stamps, round 1.)
template <int NB>
__device__ __forceinline__ uint32_t wave_sum_small(uint32_t v) {
    uint32_t s = 0;
#pragma unroll
    for (int bit = 0; bit < NB; ++bit) s += (uint32_t)__popcll(__ballot((v >> bit) & 1u)) << bit;
    return s;
}
template <int NB>
__device__ __forceinline__ uint32_t wave_excl_scan_small(uint32_t v, uint32_t* wave_total) {
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int bit = 0; bit < NB; ++bit) {
        const unsigned long long m = __ballot((v >> bit) & 1u);
        pre += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) << bit;
        tot += (uint32_t)__popcll(m) << bit;
    }
    *wave_total = tot;
    return pre;
}

// Sum of one wave-uniform count over the block; result broadcast to every thread.
// `slot` alternates between two LDS arrays so that one barrier per call is enough.
template <int SEL_WAVES>
__device__ __forceinline__ uint32_t block_sum(uint32_t wave_total, uint32_t* buf /*[2][SEL_MAX_WAVES]*/, int slot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) buf[slot * SEL_MAX_WAVES + wave] = wave_total;
    __syncthreads();
    uint32_t s = 0;
#pragma unroll
    for (int w = 0; w < SEL_WAVES; ++w) s += buf[slot * SEL_MAX_WAVES + w];
    return s;
}

// Exclusive prefix (in thread order) of a small per-thread count, plus the block total.
template <int NB, int SEL_WAVES>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* buf, int slot, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t wtot;
    const uint32_t pre = wave_excl_scan_small<NB>(v, &wtot);
    if (lane == 0) buf[slot * SEL_MAX_WAVES + wave] = wtot;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SEL_WAVES; ++w) { const uint32_t c = buf[slot * SEL_MAX_WAVES + w]; if (w < wave) base += c; tot += c; }
    *total = tot;
    return base + pre;
}

// Copy one head's k selected rows + W tail rows of one tensor in 16-byte pieces; the 8 loads of a batch are all
// issued (unconditionally, clamped) before its stores so the batch stays in registers and the latencies overlap.
template <int SEL_THREADS>
__device__ __forceinline__ void gather_head_rows(const unsigned long long* list, int k, int64_t tail0, int per_tensor,
                                                 int lanes_per_row, int64_t esz, const char* sbase, char* obase,
                                                 int64_t stride_l) {
    const int tid = threadIdx.x;
    for (int base = 0; base < per_tensor; base += 8 * SEL_THREADS) {
        uint4 v0, v1, v2, v3, v4, v5, v6, v7;
#define KVC_GLOAD(V, U)                                                                                     \
        {                                                                                                   \
            const int piece = base + (U) * SEL_THREADS + tid;                                               \
            const int pc = piece < per_tensor ? piece : per_tensor - 1;                                     \
            const int t = pc / lanes_per_row, cc = pc - t * lanes_per_row;                                  \
            const int64_t srow = t < k ? (int64_t)(list[t] & 0xffffffffull) : tail0 + t;                    \
            V = *reinterpret_cast<const uint4*>(sbase + srow * stride_l * esz + cc * 16);                   \
        }
        KVC_GLOAD(v0, 0) KVC_GLOAD(v1, 1) KVC_GLOAD(v2, 2) KVC_GLOAD(v3, 3)
        KVC_GLOAD(v4, 4) KVC_GLOAD(v5, 5) KVC_GLOAD(v6, 6) KVC_GLOAD(v7, 7)
#undef KVC_GLOAD
#define KVC_GSTORE(V, U)                                                                                    \
        {                                                                                                   \
            const int piece = base + (U) * SEL_THREADS + tid;                                               \
            if (piece < per_tensor) *reinterpret_cast<uint4*>(obase + (int64_t)piece * 16) = V;             \
        }
        KVC_GSTORE(v0, 0) KVC_GSTORE(v1, 1) KVC_GSTORE(v2, 2) KVC_GSTORE(v3, 3)
        KVC_GSTORE(v4, 4) KVC_GSTORE(v5, 5) KVC_GSTORE(v6, 6) KVC_GSTORE(v7, 7)
#undef KVC_GSTORE
    }
}

template <int DT, int EPT, int SEL_THREADS>
__global__ __launch_bounds__(SEL_THREADS) void select_kernel(const SelectArgs a) {
    constexpr int SEL_WAVES = SEL_THREADS / 64;
    typedef typename Dt<DT>::raw raw;
    constexpr int KB = Key<DT>::bits;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* list = reinterpret_cast<unsigned long long*>(smem);      // [pow2]
    uint32_t* buf = reinterpret_cast<uint32_t*>(smem + (size_t)a.pow2 * 8);      // [2][SEL_WAVES] + spare

    const int tid = threadIdx.x;
    const int head = blockIdx.x, item = blockIdx.y;
    const int n = a.n, k = a.k.v[item];
    const raw* s = reinterpret_cast<const raw*>(a.scores.p[item]) + (int64_t)head * n;

#if defined(KVC_STAMPS)
    unsigned long long stamps_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    KVC_SSTAMP(0);
    // ---- candidates into registers ----
    constexpr int NB = EPT <= 8 ? 4 : EPT <= 32 ? 6 : EPT <= 64 ? 7 : EPT <= 128 ? 8 : 9;   // bits of a per-thread count
    constexpr int ES = Dt<DT>::esize, PER16 = 16 / ES;
    uint32_t key[EPT];
    const int i0 = tid * EPT;
    if ((reinterpret_cast<uintptr_t>(s) & 15) == 0 && i0 + EPT <= n) {       // whole 16-byte pieces, coalesced
#pragma unroll
        for (int c = 0; c < EPT / PER16; ++c) {
            const uint4 v = reinterpret_cast<const uint4*>(s + i0)[c];
            const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if constexpr (ES == 2) {
                    key[c * 8 + 2 * d] = Key<DT>::of((raw)(wd[d] & 0xffffu));
                    key[c * 8 + 2 * d + 1] = Key<DT>::of((raw)(wd[d] >> 16));
                } else {
                    key[c * 4 + d] = Key<DT>::of(u2f(wd[d]));
                }
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) key[e] = (i0 + e < n) ? Key<DT>::of(s[i0 + e]) : 0u;
    }

    asm volatile("" :: "v"(key[0]), "v"(key[EPT - 1]));
    KVC_SSTAMP(1);
    // ---- 1. threshold: largest T with count(key >= T) >= k ----
    uint32_t T = 0;
    int slot = 0;
    for (int bit = KB - 1; bit >= 0; --bit) {
        const uint32_t cand = T | (1u << bit);
        uint32_t ct = 0, ct2 = 0;                            // padded keys are 0 and cand >= 1: they never count
#pragma unroll
        for (int e = 0; e < EPT; e += 2) { ct += key[e] >= cand ? 1u : 0u; ct2 += key[e + 1] >= cand ? 1u : 0u; }
        ct += ct2;
        if (block_sum<SEL_WAVES>(wave_sum_small<NB>(ct), buf, slot) >= (uint32_t)k) T = cand;
        slot ^= 1;
    }
    KVC_SSTAMP(2);
    // ---- 2. membership ----
    uint32_t cg = 0, ce = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const bool valid = (i0 + e) < n;
        cg += (valid && key[e] > T) ? 1u : 0u;
        ce += (valid && key[e] == T) ? 1u : 0u;
    }
    uint32_t tot_gt, tot_eq, tot_take;
    (void)block_excl_scan<NB, SEL_WAVES>(cg, buf, slot, &tot_gt); slot ^= 1;
    const uint32_t r = (uint32_t)k - tot_gt;                       // ties needed (>= 1)
    const uint32_t eq_before = block_excl_scan<NB, SEL_WAVES>(ce, buf, slot, &tot_eq); slot ^= 1;
    const uint32_t eq_take = eq_before >= r ? 0u : (r - eq_before < ce ? r - eq_before : ce);
    uint32_t pos = block_excl_scan<NB, SEL_WAVES>(cg + eq_take, buf, slot, &tot_take); slot ^= 1;
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
    KVC_SSTAMP(3);

    // ---- 3. order ----
    int64_t* out = reinterpret_cast<int64_t*>(const_cast<void*>(a.idx.p[item])) + (int64_t)head * k;
    if (k <= 256) {
        // rank sort spread over the whole workgroup: composites are unique, rank = number of smaller composites.
        // thread t: element e = t % E (E = pow2 >= k), comparison slice t / E; slice partials are summed in LDS.
        const int E = a.pow2 < 32 ? 32 : a.pow2;            // 32..256
        const int slices = SEL_THREADS / E;                 // 4..32
        const int e = tid % E, sl = tid / E;
        const int per = (k + slices - 1) / slices;
        uint32_t* rk = buf + 2 * SEL_MAX_WAVES;                 // [256] partial-rank accumulators (LDS, after the count buffers)
        if (tid < E) rk[tid] = 0;
        __syncthreads();
        unsigned long long mine = e < k ? list[e] : 0ull;
        if (e < k) {
            uint32_t part = 0;
            const int u0 = sl * per, u1 = u0 + per < k ? u0 + per : k;
#pragma unroll 8
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
    KVC_SSTAMP(4);
    for (int t = tid; t < k; t += SEL_THREADS) out[t] = (int64_t)(list[t] & 0xffffffffull);

    // ---- 4. fused gather of this head's rows (small k only): a batch's loads are all issued before its stores ----
    if (a.fuse) {
        const int lanes_per_row = a.gk.row_bytes >> 4;
        const int cap = k + a.gk.window;
        const int per_tensor = cap * lanes_per_row;           // 16-byte pieces per tensor
        const int b = head / a.gk.n_q_heads, h = head % a.gk.n_q_heads;
        const int64_t esz = a.gk.esize, row_bytes = a.gk.row_bytes;
        const int64_t ohb = a.gk.out_head_bytes.v[item];
        const int64_t head_bytes = ohb ? ohb : (int64_t)cap * row_bytes;
        const int64_t tail0 = (int64_t)(a.gk.q_len - a.gk.window) - k;
        const int hk = h / a.gk.group;
        gather_head_rows<SEL_THREADS>(list, k, tail0, per_tensor, lanes_per_row, esz,
                         reinterpret_cast<const char*>(a.gk.src.p[item]) + ((int64_t)b * a.gk.stride_b + (int64_t)hk * a.gk.stride_h) * esz,
                         reinterpret_cast<char*>(const_cast<void*>(a.gk.out.p[item])) + (int64_t)head * head_bytes, a.gk.stride_l);
        gather_head_rows<SEL_THREADS>(list, k, tail0, per_tensor, lanes_per_row, esz,
                         reinterpret_cast<const char*>(a.gv.src.p[item]) + ((int64_t)b * a.gv.stride_b + (int64_t)hk * a.gv.stride_h) * esz,
                         reinterpret_cast<char*>(const_cast<void*>(a.gv.out.p[item])) + (int64_t)head * head_bytes, a.gv.stride_l);
    }
#if defined(KVC_STAMPS)
    __syncthreads();
    KVC_SSTAMP(5);
    if (tid == 0) for (int i = 0; i < 6; ++i) out[i] = (int64_t)stamps_[i];
#endif
}

template <int DT, int EPT, int THR>
static int launch_t(const SelectArgs& a, hipStream_t st) {
    const size_t lds = select_lds_bytes(a.k_max);
    static LdsCache lds_cache = {};
    if (ensure_lds(reinterpret_cast<const void*>(&select_kernel<DT, EPT, THR>), lds, lds_cache) != 0) return KVC_ERR_HIP;
    hipLaunchKernelGGL((select_kernel<DT, EPT, THR>), dim3((unsigned)a.heads, (unsigned)a.n_items), dim3(THR), lds, st, a);
    return 0;
}

// (threads, keys per thread): two waves per SIMD hide the dependent-issue latency of the per-key compares (one wave per
// SIMD measured ~1.3 k cycles per search round); more waves only repeat the per-round scalar work.
template <int DT>
static int launch_dt(const SelectArgs& a, hipStream_t st) {
    const int n = a.n;
    if (n <= 2048) return launch_t<DT, 8, 256>(a, st);
    if (n <= 8192) return launch_t<DT, 16, 512>(a, st);
    if (n <= 16384) return launch_t<DT, 32, 512>(a, st);
    if (n <= 32768) return launch_t<DT, 64, 512>(a, st);
    if (n <= 65536) return launch_t<DT, 64, 1024>(a, st);
    if (n <= 131072) return launch_t<DT, 128, 1024>(a, st);   // 128k contexts: 128 keys per thread (they spill: correct, not quick)
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
