// kvc_select_exact.hip — A7 with tie_mode KVC_TIES_TORCH_CPU: membership AND order of equal scores exactly as
// torch-CPU `topk` produces them, so indices are bit-identical to the reference's CPU path (pyramidkv_utils.py:334).
//
// torch-CPU topk (aten/native/cpu TopKImpl.h) sorts pair<value,index> with a value-only comparator using
// libstdc++: std::partial_sort when k*64 <= n, otherwise std::nth_element(k-1) + std::sort(first k-1).  What those do
// with equal values is a property of their exact sequence of moves, so this kernel reproduces that sequence:
// one wavefront per head.
//
//  * partial_sort with k <= 128 (the 8k -> 128 configurations): WaveHeap below.  The heap lives one or two nodes per
//    lane; a whole __adjust_heap (sift the hole to the bottom, push the value back up) is ONE lane-parallel step —
//    the sift-down path is known from a 64-bit mask of "which child wins" bits, so every node decides by itself
//    whether it is on the path, the push-up stop is a ballot, and only the path's bits are refreshed afterwards.
//    Same moves, same final array as the serial routine (kvc_stl_emul.h), ~4x fewer dependent steps.
//  * nth_element + sort (k*64 > n): WaveSel below — lane-parallel __unguarded_partition and final insertion
//    placement, array in LDS when it fits (n <= 18000 elements of 8 bytes) and in the caller's workspace otherwise.
//  * partial_sort with k > 128 and the depth-limit fallbacks: the scalar program of kvc_stl_emul.h, every lane
//    running the same code on wave-uniform values.
//
// One wave per head: the run time is the latency of that head's chain of moves, not a throughput; the canonical
// path (kvc_select.hip) is the fast one.
#include "kvc_common.h"
#include "kvc_launch.h"
#include "kvc_stl_emul.h"

namespace kvc {

// ---------------------------------------------------------------------------------------------------------
// WaveHeap: libstdc++'s binary heap (bits/stl_heap.h) of up to 128 (key<<32|index) elements held by one wavefront.
// Node i lives on lane i (i < 64, `lo`) or lane i-64 (`hi`; for len <= 128 those are leaves).  arr[] (LDS) mirrors
// the nodes so a lane can fetch its children; xl/xr cache them.  M bit p = 1  <=>  __adjust_heap at node p moves to
// the LEFT child (comp(a[2p+2], a[2p+1]), or p has only a left child).
// ---------------------------------------------------------------------------------------------------------
// Node word: (key << 32 | index) in 64 bits, or — 16-bit dtypes, n <= 65536 — (key << 16 | index) in 32 bits, which
// halves every move, compare and LDS access of the heap.
// Lane-wise select by a wave mask held in SGPRs: bit l of `mask` set -> b, else a.  One v_cndmask per dword, no
// VGPR boolean in between (the compiler's own lowering of `mask-bit ? b : a` goes through v_lshrrev_b64 / v_cmp).
__device__ __forceinline__ uint32_t lane_sel(uint32_t a, uint32_t b, u64 mask) {
    uint32_t d;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(mask));
    return d;
}
__device__ __forceinline__ u64 lane_sel(u64 a, u64 b, u64 mask) {
    const uint32_t lo = lane_sel((uint32_t)a, (uint32_t)b, mask), hi = lane_sel((uint32_t)(a >> 32), (uint32_t)(b >> 32), mask);
    return ((u64)hi << 32) | lo;
}

template <class NT> struct HeapNode;
template <> struct HeapNode<u64> {
    __device__ __forceinline__ static u64 make(uint32_t key, int idx) { return ((u64)key << 32) | (uint32_t)idx; }
    __device__ __forceinline__ static bool gt(u64 a, u64 b) { return (uint32_t)(a >> 32) > (uint32_t)(b >> 32); }      // key(a) > key(b)
    __device__ __forceinline__ static uint32_t key(u64 v) { return (uint32_t)(v >> 32); }
    __device__ __forceinline__ static int64_t index(u64 v) { return (int64_t)(v & 0xffffffffull); }
    __device__ __forceinline__ static u64 rdlane(u64 v, int l) {
        const uint32_t a = __builtin_amdgcn_readlane((uint32_t)v, l), b = __builtin_amdgcn_readlane((uint32_t)(v >> 32), l);
        return ((u64)b << 32) | a;
    }
};
template <> struct HeapNode<uint32_t> {
    __device__ __forceinline__ static uint32_t make(uint32_t key, int idx) { return (key << 16) | (uint32_t)idx; }
    __device__ __forceinline__ static bool gt(uint32_t a, uint32_t b) { return a > (b | 0xffffu); }                    // key(a) > key(b)
    __device__ __forceinline__ static uint32_t key(uint32_t v) { return v >> 16; }
    __device__ __forceinline__ static int64_t index(uint32_t v) { return (int64_t)(v & 0xffffu); }
    __device__ __forceinline__ static uint32_t rdlane(uint32_t v, int l) { return __builtin_amdgcn_readlane(v, l); }
};

template <class NT>
struct WaveHeap {
    typedef HeapNode<NT> N;
    NT lo, hi;             // node values
    NT xl, xr;             // children of node `lane`
    u64 A_lo, R_lo, A_hi, R_hi;   // ancestors of the node(s) and the M bits they must show for the node to be reached
    u64 M;                 // wave-uniform
    NT* arr;
    int lane;

    __device__ __forceinline__ NT node(int i) const { return i < 64 ? N::rdlane(lo, i) : N::rdlane(hi, i - 64); }

    __device__ __forceinline__ void init(NT* lds, int len) {        // lo/hi set by the caller; arr[0..191] is ours
        lane = threadIdx.x;
        arr = lds;
        A_lo = R_lo = A_hi = R_hi = 0;
        for (int c = lane; c > 0;) { const int p = (c - 1) >> 1; A_lo |= 1ull << p; if (c & 1) R_lo |= 1ull << p; c = p; }
        for (int c = lane + 64; c > 0;) { const int p = (c - 1) >> 1; A_hi |= 1ull << p; if (c & 1) R_hi |= 1ull << p; c = p; }
        arr[lane] = lo; arr[lane + 64] = hi; arr[lane + 128] = 0;
        __syncthreads();
        xl = arr[2 * lane + 1]; xr = arr[2 * lane + 2];
        M = __ballot(N::gt(xr, xl));
        set_len(len);
    }
    __device__ __forceinline__ void set_len(int len) {               // a node with only a left child moves left
        if (len >= 2 && (len & 1) == 0) M |= 1ull << ((len - 2) >> 1);
    }
    // The same step for t == 0 (every replacement of the tail scan, every pop of __sort_heap), written on wave masks:
    // predicates are ballots combined on the scalar unit, values move through mask-driven v_cndmask, every lane writes
    // its node(s) and re-reads its children unconditionally — no branch, no EXEC change, ~40 instructions.
    // in_lo / in_hi: lanes whose node index is below len.
    __device__ __forceinline__ void adjust_top(u64 in_lo, u64 in_hi, int len, NT value) {
        const u64 on_lo = __ballot(((M ^ R_lo) & A_lo) == 0) & in_lo;        // lane 0 has no ancestors: always on
        const u64 on_hi = __ballot(((M ^ R_hi) & A_hi) == 0) & in_hi;
        const u64 c_lo = on_lo & ~__ballot(N::gt(lo, value)) & ~1ull;        // push-up stops: path nodes not sorting before value
        const u64 c_hi = on_hi & ~__ballot(N::gt(hi, value));
        const int stop = c_hi ? 127 - __builtin_clzll(c_hi) : (c_lo ? 63 - __builtin_clzll(c_lo) : 0);
        const u64 below = stop >= 64 ? ~0ull : ((1ull << stop) - 1);
        const u64 eq_lo = stop >= 64 ? 0ull : (1ull << stop), eq_hi = stop >= 64 ? (1ull << (stop - 64)) : 0ull;
        const NT child = lane_sel(xr, xl, M);                               // the child the hole moved to (bit set: left)
        NT vv = value;
        asm volatile("" : "+v"(vv));                                          // the value in a VGPR, once
        lo = lane_sel(lane_sel(lo, child, on_lo & below), vv, eq_lo);
        hi = lane_sel(hi, vv, eq_hi);
        arr[lane] = lo;
        arr[lane + 64] = hi;
        asm volatile("" ::: "memory");                                        // one wave, LDS in order
        xl = arr[2 * lane + 1]; xr = arr[2 * lane + 2];
        M = __ballot(N::gt(xr, xl));                                          // every node's bit anew; unused ones are ignored
        set_len(len);
    }
    // __adjust_heap(first, t, len, value) + its __push_heap, all levels at once.  At = ancestors of t.  TOP: t == 0.
    template <bool TOP>
    __device__ __forceinline__ void adjust(int t, u64 At, int len, NT value) {
        bool on_lo, on_hi;
        if constexpr (TOP) {
            on_lo = lane < len && (((M ^ R_lo) & A_lo) == 0);
            on_hi = lane + 64 < len && (((M ^ R_hi) & A_hi) == 0);
        } else {
            on_lo = lane < len && (lane == t || (((A_lo >> t) & 1) && (((M ^ R_lo) & A_lo & ~At) == 0)));
            on_hi = lane + 64 < len && ((A_hi >> t) & 1) && (((M ^ R_hi) & A_hi & ~At) == 0);
        }
        // __push_heap climbs from the bottom of the path while comp(parent, value): it stops at the deepest path
        // node (other than t) whose ORIGINAL value does not sort before `value`
        const u64 c_lo = __ballot(on_lo && lane != t && !N::gt(lo, value));
        const u64 c_hi = __ballot(on_hi && !N::gt(hi, value));
        const int stop = c_hi ? 127 - __builtin_clzll(c_hi) : (c_lo ? 63 - __builtin_clzll(c_lo) : t);
        const NT child = ((M >> lane) & 1) ? xl : xr;                 // the child the hole moved to
        if (on_lo && lane <= stop) { lo = lane == stop ? value : child; arr[lane] = lo; }
        if (on_hi && lane + 64 == stop) { hi = value; arr[lane + 64] = hi; }
        // one wave, and the LDS serves a wave's instructions in order: the reads below see the writes above without a
        // barrier or a wait — only the compiler must keep them in this order
        asm volatile("" ::: "memory");
        // refresh the children (and the "which child" bit) of the path's nodes and of t's parent
        const bool refresh = TOP ? on_lo : (on_lo || (t > 0 && lane == ((t - 1) >> 1)));
        if (refresh) { xl = arr[2 * lane + 1]; xr = arr[2 * lane + 2]; }
        const u64 P = __ballot(refresh), B = __ballot(refresh && N::gt(xr, xl));
        M = (M & ~P) | B;
        set_len(len);
    }
};

// std::partial_sort(first, first + k, last, greater-by-value) for k <= 128, k >= 1; out[0..k) = indices.
template <int DT, class NT>
__device__ __forceinline__ void partial_sort_wave(const typename Dt<DT>::raw* s, int n, int k, NT* lds, int64_t* out) {
    typedef HeapNode<NT> N;
    const int lane = threadIdx.x;
    WaveHeap<NT> H;
    H.lo = lane < k ? N::make(Key<DT>::of(s[lane]), lane) : 0;
    H.hi = lane + 64 < k ? N::make(Key<DT>::of(s[lane + 64]), lane + 64) : 0;
    H.init(lds, k);
    // __make_heap
    for (int t = (k - 2) / 2; t >= 0 && k >= 2; --t) {
        const uint32_t a0 = __builtin_amdgcn_readlane((uint32_t)H.A_lo, t), a1 = __builtin_amdgcn_readlane((uint32_t)(H.A_lo >> 32), t);
        H.template adjust<false>(t, ((u64)a1 << 32) | a0, k, N::rdlane(H.lo, t));
    }
    // __heap_select: every later element that sorts before the root replaces it (__pop_heap + __adjust_heap)
    uint32_t root = N::key(N::rdlane(H.lo, 0));
    u64 in_lo = k >= 64 ? ~0ull : ((1ull << k) - 1), in_hi = k > 64 ? ((k >= 128 ? 0ull : (1ull << (k - 64))) - 1) : 0ull;
    constexpr int U = 8;
    uint32_t cur[U], nxt[U];
#pragma unroll
    for (int j = 0; j < U; ++j) { const int i = k + j * 64 + lane; cur[j] = i < n ? Key<DT>::of(s[i]) : 0u; }
    for (int base = k; base < n; base += U * 64) {
#pragma unroll
        for (int j = 0; j < U; ++j) { const int i = base + (U + j) * 64 + lane; nxt[j] = i < n ? Key<DT>::of(s[i]) : 0u; }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int i0 = base + j * 64;
            u64 pending = __ballot(i0 + lane < n && cur[j] > root);
            while (pending) {
                const int src = __builtin_ctzll(pending);
                pending &= pending - 1;
                const uint32_t kk = __builtin_amdgcn_readlane(cur[j], src);
                if (kk > root) {                                       // re-test against the live root
                    H.adjust_top(in_lo, in_hi, k, N::make(kk, i0 + src));
                    root = N::key(N::rdlane(H.lo, 0));
                }
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) cur[j] = nxt[j];
    }
    // __sort_heap: the root goes to a[last], a[last] is re-inserted from the top
    NT* res = lds + 192;
    for (int last = k - 1; last >= 1; --last) {
        const NT value = H.node(last), top = N::rdlane(H.lo, 0);
        if (lane == 0) res[last] = top;
        H.set_len(last);
        in_lo = last >= 64 ? ~0ull : ((1ull << last) - 1);
        in_hi = last > 64 ? ((1ull << (last - 64)) - 1) : 0ull;
        H.adjust_top(in_lo, in_hi, last, value);
    }
    if (lane == 0) res[0] = H.lo;
    __syncthreads();
    for (int t = lane; t < k; t += 64) out[t] = N::index(res[t]);
}

// ---------------------------------------------------------------------------------------------------------
// nth_element + sort regime, one wavefront: the same moves as kvc_stl_emul.h, with the two loops that dominate made
// lane-parallel.
//
// __unguarded_partition(first, last, pivot): the serial scans pair the t-th element from the left that does not sort
// before the pivot (key <= pk, a "left stopper") with the t-th element from the right that the pivot does not sort
// before (key >= pk, a "right stopper") and swap them for as long as left < right.  Positions between the two cursors
// are untouched while that goes on, so both stopper sequences can be read off the ORIGINAL array: T = number of t with
// l_t < r_t, swap those pairs, and the returned cut is min(l_{T+1}, r_T) (the left scan stops at the next original
// stopper or, if it gets there first, at r_T, which now holds a left stopper).  Two compaction passes build the
// position lists (caller's scratch, <= m/2 + 2 entries each), one pass swaps.
//
// __final_insertion_sort after the introsort loop: insertion sort is stable, and every unsorted run left by the loop
// is at most 16 long with runs already ordered among themselves, so the final place of element i is
// i - #(j in [i-15, i): key_j < key_i) + #(j in (i, i+15]: key_j > key_i) — 30 reads per element, all lanes at once.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSmallRange = 1024;                        // partitions up to this many elements keep their lists in LDS
constexpr size_t kSmallListBytes = (size_t)(kSmallRange / 2 + 2) * 8;
struct WaveSel {
    u64* arr;
    int* Lbig;                                           // position lists for any range (caller's workspace)
    int* Rbig;
    int* small;                                          // LDS lists for ranges <= kSmallRange, or null
    int lane;
    __device__ __forceinline__ static uint32_t key(u64 v) { return (uint32_t)(v >> 32); }
    __device__ __forceinline__ u64 get(int i) const { return uni(arr[i]); }

    // the two compaction passes of partition(): left stoppers in ascending, right stoppers in descending position order
    template <int UF>
    __device__ __forceinline__ void scan(int first, int last, uint32_t pk, int cap, int* Lp, int* Rp, int& NL, int& NR) {
        for (int base = first; base < last; base += 64 * UF) {
            uint32_t kx[UF];
#pragma unroll
            for (int u = 0; u < UF; ++u) { const int i = base + u * 64 + lane; kx[u] = i < last ? key(arr[i]) : 0u; }
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                const int i = base + u * 64 + lane;
                const bool sl = i < last && !(kx[u] > pk);
                const u64 mask = __ballot(sl);
                const int rank = NL + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                if (sl && rank < cap) Lp[rank] = i;
                NL += __builtin_popcountll(mask);
            }
        }
        for (int top = last; top > first; top -= 64 * UF) {
            uint32_t kx[UF];
#pragma unroll
            for (int u = 0; u < UF; ++u) { const int i = top - 1 - u * 64 - lane; kx[u] = i >= first ? key(arr[i]) : 0u; }
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                const int i = top - 1 - u * 64 - lane;
                const bool sr = i >= first && !(pk > kx[u]);
                const u64 mask = __ballot(sr);
                const int rank = NR + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                if (sr && rank < cap) Rp[rank] = i;
                NR += __builtin_popcountll(mask);
            }
        }
    }
    template <int US>
    __device__ __forceinline__ void swap_pairs(int T, const int* Lp, const int* Rp) {
        for (int t0 = 0; t0 < T; t0 += 64 * US) {
            int l[US], r[US];
            u64 av[US], bv[US];
#pragma unroll
            for (int u = 0; u < US; ++u) { const int t = t0 + u * 64 + lane; l[u] = t < T ? Lp[t] : -1; r[u] = t < T ? Rp[t] : -1; }
#pragma unroll
            for (int u = 0; u < US; ++u) if (l[u] >= 0) { av[u] = arr[l[u]]; bv[u] = arr[r[u]]; }
#pragma unroll
            for (int u = 0; u < US; ++u) if (l[u] >= 0) { arr[l[u]] = bv[u]; arr[r[u]] = av[u]; }
        }
    }
    __device__ int partition(int first, int last, int pivot) {
        const uint32_t pk = key(get(pivot));
        const int cap = (last - first) / 2 + 2;
        // the sort phase makes hundreds of small partitions: their lists live in LDS (a list round trip through the
        // workspace costs a memory latency per partition)
        const bool use_small = small != nullptr && last - first <= kSmallRange;
        int* const Lp = use_small ? small : Lbig;
        int* const Rp = use_small ? small + (kSmallRange / 2 + 2) : Rbig;
        int NL = 0, NR = 0;
        // Loads are grouped UF chunks of 64 elements at a time: a long range in the workspace (n > 18000) is bound by memory
        // round trips, a short one (the hundreds of sort partitions) by the instructions of the padded trips.
        const int m = last - first;
        if (m > 8192) scan<16>(first, last, pk, cap, Lp, Rp, NL, NR);
        else if (m > 256) scan<4>(first, last, pk, cap, Lp, Rp, NL, NR);
        else scan<1>(first, last, pk, cap, Lp, Rp, NL, NR);
        __syncthreads();
        int lim = NL < NR ? NL : NR;
        if (lim > cap) lim = cap;
        // l_t < r_t holds exactly for a prefix of t (l ascending, r descending): T by bisection over the wave — 64 probes
        // per step, log64(lim) dependent list reads instead of lim / 64
        int T;
        {
            int lo_t = 0, hi_t = lim;                                     // answer in [lo_t, hi_t]
            while (hi_t - lo_t > 0) {
                const int span = hi_t - lo_t, step = (span + 63) / 64;    // probe t = lo_t + lane * step
                const int t = lo_t + lane * step;
                const bool ok = t < hi_t && Lp[t] < Rp[t];
                const u64 okm = __ballot(ok);
                const int good = __builtin_popcountll(okm);               // probes 0 .. good-1 hold (prefix property)
                if (good == 0) { hi_t = lo_t; break; }
                const int last_ok = lo_t + (good - 1) * step;
                lo_t = last_ok + 1;
                const int next_bad = lo_t + step - 1;                     // the first failing probe, or the end
                hi_t = next_bad < hi_t ? next_bad : hi_t;
                if (step == 1) break;
            }
            T = lo_t;
        }
        if (T > 2048) swap_pairs<8>(T, Lp, Rp); else if (T > 128) swap_pairs<2>(T, Lp, Rp); else swap_pairs<1>(T, Lp, Rp);
        const int lnext = T < NL ? uni(Lp[T]) : 0x7fffffff;
        const int rlast = T > 0 ? uni(Rp[T - 1]) : 0x7fffffff;
        __syncthreads();
        return lnext < rlast ? lnext : rlast;
    }
    // __unguarded_partition_pivot of a range of at most 64 elements (most calls of the sort phase), one element per lane
    // in registers: median-of-three and its swap by readlane, the two stopper sets as ballots, the pairing by popcounts
    // (left stopper i of rank t is swapped iff at least t right stoppers lie above it), partners through two 65-entry
    // LDS tables and one ds_bpermute.  Same moves as partition(); ~3 LDS round trips instead of ~10.  Kept out of line:
    // inlined, its registers slowed the long scans of partition() by 20 %.
    __device__ __forceinline__ static u64 rd64(u64 v, int l) {
        const uint32_t a = __builtin_amdgcn_readlane((uint32_t)v, l), b = __builtin_amdgcn_readlane((uint32_t)(v >> 32), l);
        return ((u64)b << 32) | a;
    }
    __device__ __noinline__ int partition_pivot_small(int first, int last) {
        const int m = last - first;                                       // 4 .. 64
        u64 x = lane < m ? arr[first + lane] : 0ull;
        // __move_median_to_first(first, first + 1, mid, last - 1)
        const int ia = 1, ib = m / 2, ic = m - 1;
        const u64 va = rd64(x, ia), vb = rd64(x, ib), vc = rd64(x, ic);
        int sw;
        if (comp(va, vb)) sw = comp(vb, vc) ? ib : (comp(va, vc) ? ic : ia);
        else sw = comp(va, vc) ? ia : (comp(vb, vc) ? ic : ib);
        const u64 v0 = rd64(x, 0), vs = rd64(x, sw);
        x = lane == 0 ? vs : (lane == sw ? v0 : x);
        // __unguarded_partition(first + 1, last, pivot = first)
        const uint32_t pk = key(vs), kx = key(x);
        const bool in = lane >= 1 && lane < m;
        const u64 SL = __ballot(in && !(kx > pk)), SR = __ballot(in && !(pk > kx));
        const u64 below_incl = lane >= 63 ? ~0ull : ((2ull << lane) - 1), above = lane >= 63 ? 0ull : (~0ull << (lane + 1));
        const bool isl = (SL >> lane) & 1, isr = (SR >> lane) & 1;
        const int t = __builtin_popcountll(SL & below_incl);              // rank among left stoppers, 1-based, ascending
        const int u = __builtin_popcountll(SR & (above | (1ull << lane)));  // rank among right stoppers, 1-based, descending
        const bool swl = isl && __builtin_popcountll(SR & above) >= t;
        const bool swr = isr && __builtin_popcountll(SL & (below_incl >> 1)) >= u;   // left stoppers strictly below
        const int T = __builtin_popcountll(__ballot(swl));
        int* const tl = small;                                            // positions by rank (LDS, 2 x 65 ints)
        int* const tr = small + 65;
        if (isl) tl[t] = lane;
        if (isr) tr[u] = lane;
        __syncthreads();
        const int partner = swl ? tr[t] : (swr ? tl[u] : lane);
        const int NL = __builtin_popcountll(SL);
        const int lnext = T < NL ? uni(tl[T + 1]) : 0x7fffffff;
        const int rlast = T > 0 ? uni(tr[T]) : 0x7fffffff;
        const uint32_t plo = (uint32_t)__builtin_amdgcn_ds_bpermute(partner << 2, (int)(uint32_t)x);
        const uint32_t phi = (uint32_t)__builtin_amdgcn_ds_bpermute(partner << 2, (int)(uint32_t)(x >> 32));
        x = ((u64)phi << 32) | plo;
        if (lane < m) arr[first + lane] = x;
        __syncthreads();
        return first + (lnext < rlast ? lnext : rlast);
    }
    __device__ int partition_pivot(int first, int last) {                 // __unguarded_partition_pivot
        if (small != nullptr && last - first <= 64) return partition_pivot_small(first, last);
        const int mid = first + (last - first) / 2;
        Arr A{arr};
        if (lane == 0) move_median_to_first_(A, first, first + 1, mid, last - 1);
        __syncthreads();
        return partition(first + 1, last, first);
    }
    __device__ void introselect(int first, int nth, int last, int depth_limit) {
        Arr A{arr};
        while (last - first > 3) {
            if (depth_limit == 0) {
                heap_select_(A, first, nth + 1, last);
                A.swap(first, nth);
                __syncthreads();
                return;
            }
            --depth_limit;
            const int cut = partition_pivot(first, last);
            if (cut <= nth) first = cut; else last = cut;
        }
        insertion_sort_(A, first, last);
        __syncthreads();
    }
    // std::sort(first, last) whose result goes straight to out[first..last) as indices
    __device__ void sort_to(int first, int last, int* stack, int64_t* out) {
        if (first == last) return;
        Arr A{arr};
        int sp = 0;
        int f = first, l = last, d = lg_(last - first) * 2;
        while (true) {
            while (l - f > 16) {
                if (d == 0) {
                    heap_select_(A, f, l, l);
                    sort_heap_(A, f, l);
                    __syncthreads();
                    break;
                }
                --d;
                const int cut = partition_pivot(f, l);
                if (lane == 0) { stack[3 * sp] = cut; stack[3 * sp + 1] = l; stack[3 * sp + 2] = d; }
                ++sp;
                l = cut;
            }
            if (sp == 0) break;
            --sp;
            __syncthreads();
            f = uni(stack[3 * sp]); l = uni(stack[3 * sp + 1]); d = uni(stack[3 * sp + 2]);
        }
        __syncthreads();
        for (int base = first; base < last; base += 64) {                  // __final_insertion_sort, all elements at once
            const int i = base + lane;
            if (i < last) {
                const u64 v = arr[i];
                const uint32_t ki = key(v);
                int pos = i;
#pragma unroll 5
                for (int dlt = 1; dlt < 16; ++dlt) {
                    const int jb = i - dlt, ja = i + dlt;
                    if (jb >= first && key(arr[jb]) < ki) --pos;
                    if (ja < last && key(arr[ja]) > ki) ++pos;
                }
                out[pos] = (int64_t)(v & 0xffffffffull);
            }
        }
    }
    // (caps == nullptr in the kernel below: the whole sort.)
    // The first `want` elements of std::sort(first, last), written to out[0..want) as indices (AdaKV / HeadKV keep a prefix of
    // the head's full descending sort, pyramidkv_utils.py:702-724).  std::sort = introsort loop + final insertion sort: the
    // loop only ever splits ranges, and a range that starts at or beyond `want` can be left alone — everything in it sorts
    // behind every element to its left, it is never mixed with them again, and the final insertion sort moves an element
    // by at most 15 places inside its own leaf range.  So: partition exactly as libstdc++ does, descend only into ranges
    // that start before `want`, then place the elements below want + 16 by the stable-insertion count of sort_to().
    __device__ void sort_prefix_to(int first, int last, int want, int* stack, int64_t* out) {
        if (first == last || want <= 0) return;
        Arr A{arr};
        int sp = 0;
        int f = first, l = last, d = lg_(last - first) * 2;
        while (true) {
            while (l - f > 16) {
                if (d == 0) {
                    heap_select_(A, f, l, l);
                    sort_heap_(A, f, l);
                    __syncthreads();
                    break;
                }
                --d;
                const int cut = partition_pivot(f, l);
                if (cut < first + want) {                                  // the right part still reaches into the prefix
                    if (lane == 0) { stack[3 * sp] = cut; stack[3 * sp + 1] = l; stack[3 * sp + 2] = d; }
                    ++sp;
                }
                l = cut;
            }
            if (sp == 0) break;
            --sp;
            __syncthreads();
            f = uni(stack[3 * sp]); l = uni(stack[3 * sp + 1]); d = uni(stack[3 * sp + 2]);
        }
        __syncthreads();
        const int lim = first + want + 16 < last ? first + want + 16 : last;
        for (int base = first; base < lim; base += 64) {
            const int i = base + lane;
            if (i < lim) {
                const u64 v = arr[i];
                const uint32_t ki = key(v);
                int pos = i;
#pragma unroll 5
                for (int dlt = 1; dlt < 16; ++dlt) {
                    const int jb = i - dlt, ja = i + dlt;
                    if (jb >= first && key(arr[jb]) < ki) --pos;
                    if (ja < last && key(arr[ja]) > ki) ++pos;
                }
                if (pos - first < want) out[pos - first] = (int64_t)(v & 0xffffffffull);
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------------------
// WaveHeap9: the same heap for up to 511 nodes (16-bit dtypes, packed 32-bit nodes): budgets 129..511 in the
// partial_sort regime (e.g. cap 256 at 16k / 32k, PyramidKV's small layers at 32k), which the scalar program runs at
// ~2400 cycles per step.
//   levels 0-5 (nodes 0..62): one node per lane, path by the "which child" mask, exactly as WaveHeap;
//   levels 6-8: node 63+j and its six descendants (127+2j, 128+2j, 255+4j .. 258+4j) are PRIVATE to lane j, in
//   registers: once the path leaves the top part it is that lane's own business — no communication.
// arr[] (LDS) mirrors nodes 0..126 so that the lanes of levels 0-5 can fetch their children (level 6 = the p0 of the
// lanes below them).
// ---------------------------------------------------------------------------------------------------------
struct WaveHeap9 {
    typedef HeapNode<uint32_t> N;
    uint32_t top;          // node `lane` (lanes 0..62)
    uint32_t xl, xr;       // its children
    uint32_t p0, p1, p2, p3, p4, p5, p6;   // private sub-heap of lane j: nodes 63+j; 127+2j, 128+2j; 255+4j..258+4j
    u64 A, R, M;
    uint32_t* arr;
    int lane;
    // what the private walk of this lane found for the current step (valid for every lane, used for one)
    int d1, d2;            // chosen child at level 7 (1 | 2) and at level 8 (slot 3..6)
    uint32_t v1, v2;
    bool has1, has2;

    __device__ __forceinline__ int q1() const { return 127 + 2 * lane; }
    __device__ __forceinline__ int q3() const { return 255 + 4 * lane; }
    __device__ __forceinline__ uint32_t slot(int sl) const {
        return sl == 0 ? p0 : sl == 1 ? p1 : sl == 2 ? p2 : sl == 3 ? p3 : sl == 4 ? p4 : sl == 5 ? p5 : p6;
    }
    __device__ __forceinline__ void set_slot(int sl, uint32_t v) {
        p0 = sl == 0 ? v : p0; p1 = sl == 1 ? v : p1; p2 = sl == 2 ? v : p2; p3 = sl == 3 ? v : p3;
        p4 = sl == 4 ? v : p4; p5 = sl == 5 ? v : p5; p6 = sl == 6 ? v : p6;
    }
    // node index -> value (wave-uniform index)
    __device__ __forceinline__ uint32_t node(int i) const {
        if (i < 63) return N::rdlane(top, i);
        if (i < 127) return N::rdlane(p0, i - 63);
        if (i < 255) { const int l = (i - 127) >> 1; return ((i - 127) & 1) ? N::rdlane(p2, l) : N::rdlane(p1, l); }
        const int l = (i - 255) >> 2, w = (i - 255) & 3;
        return w == 0 ? N::rdlane(p3, l) : w == 1 ? N::rdlane(p4, l) : w == 2 ? N::rdlane(p5, l) : N::rdlane(p6, l);
    }
    __device__ __forceinline__ void force_bit(int len) {             // a top node with only a left child moves left
        if (len >= 2 && (len & 1) == 0 && ((len - 2) >> 1) < 63) M |= 1ull << ((len - 2) >> 1);
    }
    // the private walk from this lane's level-6 node: choices by libstdc++'s rule (right child unless it sorts before
    // the left one; a lone left child is taken)
    __device__ __forceinline__ void walk(int len) {
        const int a1 = q1(), a3 = q3();
        has1 = a1 < len;
        const bool r1 = a1 + 1 < len && !N::gt(p2, p1);            // move right at level 7
        d1 = r1 ? 2 : 1;
        v1 = r1 ? p2 : p1;
        const int al = r1 ? a3 + 2 : a3;                             // left grandchild under the chosen child
        const uint32_t cl = r1 ? p5 : p3, cr = r1 ? p6 : p4;
        has2 = has1 && al < len;
        const bool r2 = al + 1 < len && !N::gt(cr, cl);
        d2 = (r1 ? 5 : 3) + (r2 ? 1 : 0);
        v2 = r2 ? cr : cl;
    }
    // deepest private path node (depth 2, 1, or 0 = the level-6 node itself when `with0`) that does not sort before
    // `value`; -1 if none
    __device__ __forceinline__ int private_stop(uint32_t value, bool with0) const {
        if (has2 && !N::gt(v2, value)) return 2;
        if (has1 && !N::gt(v1, value)) return 1;
        if (with0 && !N::gt(p0, value)) return 0;
        return -1;
    }
    // shift the private path up to depth sp and drop `value` there (this lane only)
    __device__ __forceinline__ void private_apply(int sp, uint32_t value, bool mine) {
        if (!mine) return;
        if (sp == 0) { p0 = value; return; }
        p0 = v1;
        if (sp == 1) { set_slot(d1, value); return; }
        set_slot(d1, v2);
        set_slot(d2, value);
    }
    // __make_heap inside the private sub-heaps: every level-7 node, then every level-6 node — disjoint subtrees, so all
    // lanes at once give the array the sequential loop gives
    __device__ __forceinline__ void private_make_heap(int len) {
        {   // level 7: nodes q1, q2 with their (at most two) level-8 children
            const int a1 = q1(), a3 = q3();
#pragma unroll
            for (int side = 1; side >= 0; --side) {                   // higher index first, like the sequential loop
                const int a = a1 + side, al = a3 + 2 * side;
                const uint32_t cur = side ? p2 : p1, cl = side ? p5 : p3, cr = side ? p6 : p4;
                if (a < len && al < len) {
                    const bool r = al + 1 < len && !N::gt(cr, cl);
                    const uint32_t ch = r ? cr : cl;
                    if (!N::gt(ch, cur)) {                             // __push_heap stops at the child: the two swap
                        set_slot(side ? 2 : 1, ch);
                        set_slot((side ? 5 : 3) + (r ? 1 : 0), cur);
                    }
                }
            }
        }
        if (63 + lane < len) {                                        // level 6: the whole private sub-heap
            walk(len);
            const uint32_t value = p0;
            int sp = private_stop(value, false);
            if (sp < 0) sp = 0;
            private_apply(sp, value, true);
        }
    }
    __device__ __forceinline__ void init(uint32_t* lds, int len) {   // top / p0..p6 set by the caller
        lane = threadIdx.x;
        arr = lds;
        A = R = 0;
        for (int c = lane; c > 0;) { const int p = (c - 1) >> 1; A |= 1ull << p; if (c & 1) R |= 1ull << p; c = p; }
        private_make_heap(len);
        if (lane < 63) arr[lane] = top;
        arr[63 + lane] = p0;
        arr[127 + lane] = 0;                                          // (lane 63 reads arr[127], arr[128] as "children")
        __syncthreads();
        xl = arr[2 * lane + 1]; xr = arr[2 * lane + 2];
        M = __ballot(N::gt(xr, xl));
        force_bit(len);
    }
    // __adjust_heap(first, t, len, value) for a node t of the top part (t <= 62), with its __push_heap.
    template <bool TOP = false>
    __device__ __forceinline__ void adjust(int t, u64 At, int len, uint32_t value) {
        const u64 in_top = len >= 63 ? 0x7fffffffffffffffull : ((1ull << len) - 1);
        u64 on_top;
        if constexpr (TOP) on_top = __ballot(((M ^ R) & A) == 0) & in_top;          // t == 0: lane 0 has no ancestors
        else on_top = __ballot(lane == t || (((A >> t) & 1) && (((M ^ R) & A & ~At) == 0))) & in_top;
        // does the path leave the top part?  the on-path level-5 node (31..62) and its chosen level-6 child
        const u64 m5 = on_top & 0x7fffffff80000000ull;
        int j6 = -1;
        if (m5) {
            const int l5 = __builtin_ctzll(m5);
            const int c6 = 2 * l5 + 2 - (int)((M >> l5) & 1);
            if (c6 < len) j6 = c6 - 63;
        }
        int sp = -1;
        if (j6 >= 0) {
            walk(len);
            sp = __builtin_amdgcn_readlane(private_stop(value, true), j6);
        }
        const u64 g = __ballot(N::gt(top, value));
        const u64 c_top = on_top & ~g & ~(1ull << t);
        const int stop = c_top ? 63 - __builtin_clzll(c_top) : t;
        const u64 below = (1ull << stop) - 1;
        const u64 take = sp >= 0 ? on_top : (on_top & below);
        const u64 eq = sp >= 0 ? 0ull : (1ull << stop);
        const uint32_t child = lane_sel(xr, xl, M);
        uint32_t vv = value;
        asm volatile("" : "+v"(vv));
        top = lane_sel(lane_sel(top, child, take), vv, eq);
        private_apply(sp, value, sp >= 0 && lane == j6);
        if (lane < 63) arr[lane] = top;
        arr[63 + lane] = p0;
        asm volatile("" ::: "memory");                                  // one wave, LDS in order
        xl = arr[2 * lane + 1]; xr = arr[2 * lane + 2];
        M = __ballot(N::gt(xr, xl));
        force_bit(len);
    }
};

// std::partial_sort for 128 < k <= 511 (16-bit dtypes).
template <int DT>
__device__ __forceinline__ void partial_sort_wave9(const typename Dt<DT>::raw* s, int n, int k, uint32_t* lds, int64_t* out) {
    typedef HeapNode<uint32_t> N;
    const int lane = threadIdx.x;
    WaveHeap9 H;
    auto ld = [&](int i) -> uint32_t { return i < k ? N::make(Key<DT>::of(s[i]), i) : 0u; };
    H.lane = lane;
    H.top = lane < 63 ? ld(lane) : 0u;
    H.p0 = ld(63 + lane);
    H.p1 = ld(127 + 2 * lane); H.p2 = ld(128 + 2 * lane);
    H.p3 = ld(255 + 4 * lane); H.p4 = ld(256 + 4 * lane); H.p5 = ld(257 + 4 * lane); H.p6 = ld(258 + 4 * lane);
    H.init(lds, k);
    // __make_heap, the top part (the private levels are done in init)
    const int t_hi = (k - 2) / 2 < 62 ? (k - 2) / 2 : 62;
    for (int t = t_hi; t >= 0; --t) {
        const uint32_t a0 = __builtin_amdgcn_readlane((uint32_t)H.A, t), a1 = __builtin_amdgcn_readlane((uint32_t)(H.A >> 32), t);
        H.adjust(t, ((u64)a1 << 32) | a0, k, N::rdlane(H.top, t));
    }
    // __heap_select
    uint32_t root = N::key(N::rdlane(H.top, 0));
    constexpr int U = 8;
    uint32_t cur[U], nxt[U];
#pragma unroll
    for (int j = 0; j < U; ++j) { const int i = k + j * 64 + lane; cur[j] = i < n ? Key<DT>::of(s[i]) : 0u; }
    for (int base = k; base < n; base += U * 64) {
#pragma unroll
        for (int j = 0; j < U; ++j) { const int i = base + (U + j) * 64 + lane; nxt[j] = i < n ? Key<DT>::of(s[i]) : 0u; }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int i0 = base + j * 64;
            u64 pending = __ballot(i0 + lane < n && cur[j] > root);
            while (pending) {
                const int src = __builtin_ctzll(pending);
                pending &= pending - 1;
                const uint32_t kk = __builtin_amdgcn_readlane(cur[j], src);
                if (kk > root) {
                    H.adjust<true>(0, 0, k, N::make(kk, i0 + src));
                    root = N::key(N::rdlane(H.top, 0));
                }
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) cur[j] = nxt[j];
    }
    // __sort_heap
    uint32_t* res = lds + 192;
    for (int last = k - 1; last >= 1; --last) {
        const uint32_t value = H.node(last), topv = N::rdlane(H.top, 0);
        if (lane == 0) res[last] = topv;
        H.force_bit(last);
        H.adjust<true>(0, 0, last, value);
    }
    if (lane == 0) res[0] = H.top;
    __syncthreads();
    for (int t = lane; t < k; t += 64) out[t] = N::index(res[t]);
}
constexpr int kWaveHeap9MaxK = 511;
constexpr size_t kWaveHeap9Lds = (192 + 512) * 4;

constexpr int kWaveHeapMaxK = 128;
constexpr size_t kWaveHeapLds = (192 + 128) * 8;

// ---------------------------------------------------------------------------------------------------------
// grid = (heads, items), block = 64 (one wave).  LDS: introsort stack + array (or heap).
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

    // (key << 16 | index) nodes need indices below 65536; longer rows (128k contexts, AdaKV's flattened [H * n] array) take
    // the 64-bit nodes
    if (use_partial_sort && k <= kWaveHeapMaxK) {
        if (Key<DT>::bits == 16 && n <= 65536) {
            if constexpr (Key<DT>::bits == 16) partial_sort_wave<DT, uint32_t>(s, n, k, reinterpret_cast<uint32_t*>(lds_arr), out);
        } else {
            partial_sort_wave<DT, u64>(s, n, k, lds_arr, out);
        }
    } else if (use_partial_sort && k <= kWaveHeap9MaxK && Key<DT>::bits == 16 && n <= 65536) {
        if constexpr (Key<DT>::bits == 16) partial_sort_wave9<DT>(s, n, k, reinterpret_cast<uint32_t*>(lds_arr), out);
    } else if (use_partial_sort) {
        // heap of the first k in LDS; the tail is streamed 64 at a time
        Arr H{lds_arr};
        for (int i = lane; i < k; i += 64) lds_arr[i] = ((u64)Key<DT>::of(s[i]) << 32) | (uint32_t)i;
        __syncthreads();
        make_heap_(H, 0, k);
        // the tail is read eight 64-key chunks ahead (a chunk fetched on demand costs a full memory round trip per 64 keys)
        constexpr int U = 8;
        uint32_t cur[U], nxt[U];
#pragma unroll
        for (int j = 0; j < U; ++j) { const int i = k + j * 64 + lane; cur[j] = i < n ? Key<DT>::of(s[i]) : 0u; }
        u64 root = H.get(0);
        for (int base = k; base < n; base += U * 64) {
#pragma unroll
            for (int j = 0; j < U; ++j) { const int i = base + (U + j) * 64 + lane; nxt[j] = i < n ? Key<DT>::of(s[i]) : 0u; }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int i0 = base + j * 64;
                unsigned long long pending = __ballot(i0 + lane < n && cur[j] > (uint32_t)(root >> 32));
                while (pending) {
                    const int src = __builtin_ctzll(pending);
                    pending &= pending - 1;
                    const uint32_t kk = __builtin_amdgcn_readlane(cur[j], src);
                    if (kk > (uint32_t)(root >> 32)) {                   // re-test against the live root
                        adjust_heap_(H, 0, 0, k, ((u64)kk << 32) | (uint32_t)(i0 + src));
                        root = H.get(0);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) cur[j] = nxt[j];
        }
        sort_heap_(H, 0, k);
        __syncthreads();
        for (int t = lane; t < k; t += 64) out[t] = (int64_t)(lds_arr[t] & 0xffffffffull);
    } else {
        // scratch per (item, head): [n u64 array, when it does not fit in LDS][2 x (n/2 + 2) int position lists]
        const bool small_lists = (arr_in_lds & 2) != 0;
        arr_in_lds &= 1;
        const int64_t per_head = (arr_in_lds ? 0 : (int64_t)n) + (n / 2 + 2);
        u64* const hs = gscratch + ((int64_t)item * a.heads + head) * per_head;
        u64* arr = arr_in_lds ? lds_arr : hs;
        int* const lists = reinterpret_cast<int*>(hs + (arr_in_lds ? 0 : n));
        // LDS behind the array region (n elements, or the k_max - 1 being sorted): lists for small partitions
        int* const small = small_lists ? reinterpret_cast<int*>(lds_arr + (arr_in_lds ? n : a.k_max)) : nullptr;
        for (int i = lane; i < n; i += 64) arr[i] = ((u64)Key<DT>::of(s[i]) << 32) | (uint32_t)i;
        __syncthreads();
        WaveSel S{arr, lists, lists + (n / 2 + 2), small, lane};
        // std::nth_element(first, first + k - 1, last)
        if (k - 1 != n) S.introselect(0, k - 1, n, lg_(n) * 2);
        // std::sort(first, first + k - 1), written to out as it is placed; the nth element follows.  When the array lives
        // in the workspace the k - 1 leading elements move to LDS first: the sort makes hundreds of small partitions, each
        // a handful of dependent accesses — a memory round trip apiece on the workspace copy.
        const u64 nth = arr[k - 1];
        if (!arr_in_lds && k - 1 > 0) {
            for (int i = lane; i < k - 1; i += 64) lds_arr[i] = arr[i];
            __syncthreads();
            WaveSel S2{lds_arr, lists, lists + (n / 2 + 2), small, lane};
            S2.sort_to(0, k - 1, stack, out);
        } else {
            S.sort_to(0, k - 1, stack, out);
        }
        if (lane == 0) out[k - 1] = (int64_t)(nth & 0xffffffffull);
    }
}

size_t select_exact_scratch_bytes(int heads, int n, int k) {
    const bool partial = (int64_t)k * 64 <= (int64_t)n;
    if (partial) return 0;
    return (size_t)heads * ((n <= 18000 ? 0 : (size_t)n) + (size_t)(n / 2 + 2)) * 8;
}

template <int DT>
static int launch_exact_t(const SelectArgs& a, void* scratch, hipStream_t st) {
    // LDS sized for the worst item: a heap of k_max (partial_sort regime) or the whole array (nth_element regime)
    bool any_nth = false;
    for (int i = 0; i < a.n_items; ++i) any_nth = any_nth || !((int64_t)a.k.v[i] * 64 <= (int64_t)a.n);
    const int in_lds = !any_nth || a.n <= 18000;
    // LDS: the whole array (nth_element items, n <= 18000), or the heap of k_max (partial_sort) / the k_max - 1 elements
    // being sorted; plus 4 KB of lists for small partitions — unless that costs a resident wave per CU (160 KB of LDS, one
    // wave = one workgroup here: these kernels are latency-bound, a second round of workgroups doubles their time)
    size_t body = (in_lds && any_nth) ? (size_t)a.n * 8 : (size_t)a.k_max * 8;
    if (body < kWaveHeap9Lds) body = kWaveHeap9Lds;
    if (body < kWaveHeapLds) body = kWaveHeapLds;                   // 64-bit nodes (fp32, or n > 65536)
    const bool small_lists = any_nth && (160 * 1024) / (1152 + body + kSmallListBytes) == (160 * 1024) / (1152 + body) &&
                             1152 + body + kSmallListBytes <= 150 * 1024;
    if (small_lists) body += kSmallListBytes;
    const size_t lds = 1152 + body;
    if (any_nth && !scratch) return KVC_ERR_WORKSPACE;
    static LdsCache lds_cache = {};
    if (ensure_lds(reinterpret_cast<const void*>(&select_exact_kernel<DT>), lds, lds_cache) != 0) return KVC_ERR_HIP;
    hipLaunchKernelGGL((select_exact_kernel<DT>), dim3((unsigned)a.heads, (unsigned)a.n_items), dim3(64), lds, st, a,
                       reinterpret_cast<u64*>(scratch), in_lds | (small_lists ? 2 : 0));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// sort_prefix_kernel (AdaKV / HeadKV): out[head][0..caps[head]) = the first caps[head] indices of
// scores[head].sort(descending=True) as torch-CPU produces them (libstdc++ std::sort, value-only comparator).
// grid = (heads), block = 64.  The (key, index) array lives in LDS (n <= 18000) or in `gscratch` ([heads][n] u64).
// ---------------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(64) void sort_prefix_kernel(const RaggedSortArgs a, u64* gscratch, int arr_in_lds) {
    typedef typename Dt<DT>::raw raw;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* stack = reinterpret_cast<int*>(smem);                        // 3 * 96 ints
    u64* lds_arr = reinterpret_cast<u64*>(smem + 1152);
    const int lane = threadIdx.x, head = blockIdx.x, n = a.n;
    int want = a.caps ? a.caps[head] : n;
    want = want < 0 ? 0 : (want > n ? n : want);
    const raw* s = reinterpret_cast<const raw*>(a.scores) + (int64_t)head * n;
    int64_t* out = a.idx_out + (int64_t)head * a.out_stride;
    const int64_t per_head = (arr_in_lds ? 0 : (int64_t)n) + (n / 2 + 2);
    u64* const hs = gscratch + (int64_t)head * per_head;
    u64* arr = arr_in_lds ? lds_arr : hs;
    int* const lists = reinterpret_cast<int*>(hs + (arr_in_lds ? 0 : n));
    int* const small = arr_in_lds ? reinterpret_cast<int*>(lds_arr + n) : reinterpret_cast<int*>(lds_arr);
    for (int i = lane; i < n; i += 64) arr[i] = ((u64)Key<DT>::of(s[i]) << 32) | (uint32_t)i;
    __syncthreads();
    WaveSel S{arr, lists, lists + (n / 2 + 2), small, lane};
    S.sort_prefix_to(0, n, want, stack, out);
}

size_t sort_prefix_scratch_bytes(int heads, int n) {
    return (size_t)heads * ((n <= 18000 ? 0 : (size_t)n) + (size_t)(n / 2 + 2)) * 8;
}

template <int DT>
static int launch_sort_prefix_t(const RaggedSortArgs& a, void* scratch, hipStream_t st) {
    const int in_lds = a.n <= 18000;
    const size_t lds = 1152 + (in_lds ? (size_t)a.n * 8 : 0) + kSmallListBytes;
    if (!scratch) return KVC_ERR_WORKSPACE;
    static LdsCache lds_cache = {};
    if (ensure_lds(reinterpret_cast<const void*>(&sort_prefix_kernel<DT>), lds, lds_cache) != 0) return KVC_ERR_HIP;
    hipLaunchKernelGGL((sort_prefix_kernel<DT>), dim3((unsigned)a.heads), dim3(64), lds, st, a, reinterpret_cast<u64*>(scratch), in_lds);
    return 0;
}

int launch_sort_prefix(const RaggedSortArgs& a, int dtype, void* scratch, hipStream_t st) {
    switch (dtype) {
        case KVC_BF16: return launch_sort_prefix_t<KVC_BF16>(a, scratch, st);
        case KVC_FP16: return launch_sort_prefix_t<KVC_FP16>(a, scratch, st);
        case KVC_FP32: return launch_sort_prefix_t<KVC_FP32>(a, scratch, st);
    }
    return KVC_ERR_INVALID;
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
