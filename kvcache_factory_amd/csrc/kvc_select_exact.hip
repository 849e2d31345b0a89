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
#include <stdlib.h>
#include "kvc_common.h"
#include "kvc_launch.h"
#include "kvc_stl_emul.h"
#include <type_traits>

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
// WaveHeapL (round 3): the same heap of up to 128 packed 32-bit nodes, rebuilt around what tools/hop_probe.hip measured
// for a wave that is alone on its SIMD: every instruction costs >= 4.4 cycles of issue whatever its unit, a scalar
// instruction between two vector instructions that depend on each other through a mask costs 20 more, a taken branch
// 28, an LDS write -> read -> use 72.  WaveHeap's step (mask algebra for the stop position, LDS exchange of every node
// inside the step, 58 instructions) ran at 570-800 cycles.  Here:
//  * along a root-to-leaf path of a heap the keys never decrease, so "__push_heap stops at the deepest path node that
//    does not sort before `value`" is a LOCAL rule: a path node p with key(p) <= key(value) receives its chosen child's
//    value if key(child) <= key(value) too, and `value` itself otherwise.  No stop position, no mask arithmetic.
//  * a lane that caches its two children (xl, xr), its chosen child (xc) and the chosen child of THAT child (xcc) can
//    bring xl / xr / xc up to date by itself: the chosen child, if it is on the path and takes part, becomes
//    (key(xcc) <= key(value) ? xcc : value).  Only xcc has to be fetched again: one LDS exchange per step whose result
//    is not needed before the middle of the NEXT step.  The caches are exact copies at all times, so nodes 64..127
//    (leaves) live in their parents' xl / xr only, and the root's next key is min(key(value), key(xc of the root)) on
//    the scalar unit — the test of the next candidate does not wait for the vector step.
//  * __make_heap level by level: the sift-downs of one level work on disjoint subtrees, so all of them are one step
//    with a per-lane value (6 steps instead of 60).
// Nodes beyond the heap's length hold SENT (largest key): a lone left child wins against it, a leaf's chosen child
// never takes part.  (A real key of 0xffff would be a NaN score: excluded, as everywhere in the exact mode.)
// Same moves, same array as libstdc++ (tools/heap_probe.hip: 1 024 tie-heavy rows against std::partial_sort; the
// exact-tie tests of tests/test_gpu_parity.py against the reference's fixtures).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 key_gt_mask(uint32_t a, uint32_t b) {        // lanes with key(a) > key(b): one SDWA compare
    u64 m;
    asm volatile("v_cmp_gt_u32_sdwa %0, %1, %2 src0_sel:WORD_1 src1_sel:WORD_1" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
// The step's instructions are single `asm volatile` statements: their order is the schedule (the compiler still places the
// hazard nops between them); left to itself it puts a scalar mask operation right behind the compare that feeds it.
__device__ __forceinline__ uint32_t vsel(uint32_t a, uint32_t b, u64 mask) {                  // mask bit ? b : a, in order
    uint32_t d;
    asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(mask));
    return d;
}
struct WaveHeapL {
    typedef HeapNode<uint32_t> N;
    static constexpr uint32_t SENT = 0xffffffffu;
    uint32_t lo;                     // node `lane` (nodes 0..63)
    uint32_t xl, xr, xc, xcc;        // children of node `lane`, the chosen one, and ITS chosen child
    uint32_t A, R;                   // ancestors of node `lane` (all <= 31) / those at which the path to `lane` goes left
    uint32_t c0;                     // the root's chosen child (SGPR); its key is shifted out where it is used, not here:
                                     // a scalar instruction right behind the v_readlane that feeds it waits 20 cycles
    uint32_t* XC;                    // LDS: XC[p] = xc of node p for p < 64, SENT for 64 <= p < 192
    int adL, adR;                    // LDS addresses of XC[2 * lane + 1], XC[2 * lane + 2]
    int lane, lvl;

    __device__ __forceinline__ uint32_t node(int i) const {         // wave-uniform i < 128
        if (i == 0) return N::rdlane(lo, 0);
        const int p = (i - 1) >> 1;
        return (i & 1) ? N::rdlane(xl, p) : N::rdlane(xr, p);
    }
    __device__ __forceinline__ void exchange(uint32_t ad) {          // publish xc, fetch xcc (LDS, compiler-tracked), c0
        XC[lane] = xc;
        asm volatile("" ::: "memory");                                // one wave, LDS in order
        typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
        xcc = *(lds_cu32*)(uintptr_t)ad;                              // ad: an LDS address
        c0 = N::rdlane(xc, 0);
    }
    __device__ __forceinline__ void refresh() {                     // xc from xl / xr; the exchange
        uint32_t ad;
        u64 M;
        asm volatile("v_cmp_gt_u32_sdwa %[M], %[xr], %[xl] src0_sel:WORD_1 src1_sel:WORD_1\n\t"
                     "s_nop 1\n\t"                                            // (hazard: see step())
                     "v_cndmask_b32_e64 %[xc], %[xr], %[xl], %[M]\n\t"
                     "v_cndmask_b32_e64 %[ad], %[adR], %[adL], %[M]"
                     : [M] "=&s"(M), [xc] "=&v"(xc), [ad] "=&v"(ad) : [xl] "v"(xl), [xr] "v"(xr), [adL] "v"(adL), [adR] "v"(adR));
        exchange(ad);
    }
    // The moves of one __adjust_heap + __push_heap of the wave-uniform `value` from the root (if `take`: a candidate that the
    // root has overtaken since the ballot runs the same instructions with nobody taking part — a branch costs 25 cycles
    // whether taken or not, and nearly every candidate is live).  Two asm blocks whose instruction order is the schedule:
    // left to itself the compiler puts a scalar mask operation right behind the compare that feeds it (20 cycles each) and a
    // hazard nop behind every single-instruction asm statement.  Between the blocks it places the wait for the LDS read of
    // the previous step.  gfx950 hazard (the compiler's own rule, GCNHazardRecognizer VALUWriteSGPRVALURead): a VALU
    // instruction that reads an SGPR / VCC written by a VALU instruction needs two other instructions in between — the
    // order below provides them (tools/heap_probe.hip without them: wrong paths).
    // M is made from xl / xr in every step rather than carried: the compiler moves an asm-made 64-bit mask that lives across
    // a loop into VGPRs (and then cannot hand it to a scalar operand).
#define KVC_HEAPL_BLOCK_A                                                                                                   \
        asm volatile("v_cmp_gt_u32_sdwa vcc, %[xr], %[xl] src0_sel:WORD_1 src1_sel:WORD_1\n\t"   /* M: the hole at node p moves left */ \
                     "v_cmp_gt_u32_e64 %[g], %[lo], %[vor]\n\t"             /* g: key(node) > key(value) */                 \
                     "v_cmp_gt_u32_e64 %[c], %[xc], %[vor]\n\t"             /* c: key(chosen child) > key(value) */         \
                     "v_and_b32_e32 %[t], vcc_lo, %[A]\n\t"                                                                 \
                     "s_mov_b64 %[M], vcc\n\t"                                                                              \
                     "v_mov_b32_e32 %[vv], %[value]\n\t"                                                                    \
                     "v_cmp_eq_u32_e64 %[on], %[t], %[R]\n\t"               /* on: the sift path under M (lane 0: no ancestors) */ \
                     "v_cndmask_b32_e64 %[inner], %[xc], %[vv], %[c]"       /* what a participating node receives */        \
                     : [t] "=&v"(t), [M] "=&s"(M), [on] "=&s"(on), [g] "=&s"(g), [c] "=&s"(c), [vv] "=&v"(vv), [inner] "=&v"(inner) \
                     : [xl] "v"(xl), [xr] "v"(xr), [A] "v"(A), [R] "v"(R), [lo] "v"(lo), [xc] "v"(xc), [value] "s"(value), [vor] "s"(vor) : "vcc")
    // the moves: the chosen child is xl where M is set and xr elsewhere, so only its copy xc is selected by `upd`
#define KVC_HEAPL_MOVES                                                                                                     \
                     "s_andn2_b64 %[upd], %[cm], %[c]\n\t"                  /* ... whose chosen child takes part too */     \
                     "v_cndmask_b32_e32 %[nc], %[xcc], %[vv], vcc\n\t"      /* what a participating chosen child receives */ \
                     "v_cndmask_b32_e64 %[nc], %[xc], %[nc], %[upd]\n\t"    /* the chosen child after the step */           \
                     "v_cndmask_b32_e64 %[xl], %[xl], %[nc], %[M]\n\t"                                                      \
                     "v_cndmask_b32_e64 %[xr], %[nc], %[xr], %[M]\n\t"
    template <bool PRED, int PROBE = 0>                               // (tools/heap_probe.hip times the step with parts left out)
    __device__ __forceinline__ void step(uint32_t value, uint32_t& rootor) {       // rootor: the root's (key << 16 | 0xffff), brought up to date
        const uint32_t vor = value | 0xffffu;
        u64 M, on, g, c, cm, upd, Mn;
        uint32_t vv, inner, nc, ad, t, rnew;
        KVC_HEAPL_BLOCK_A;
        if (PRED) {
            asm volatile("v_cmp_gt_u32_e64 vcc, %[xcc], %[vor]\n\t"
                         "s_or_b32 %[rnew], %[c0], 0xffff\n\t"              // (c0 comes from a v_readlane: first scalar use here, late)
                         "s_min_u32 %[rnew], %[rnew], %[vor]\n\t"           // the root's key after the step: the smaller of value's and
                         "s_cmp_gt_u32 %[value], %[rootor]\n\t"             // its chosen child's — if the candidate is still live
                         "s_cselect_b32 %[rnew], %[rnew], %[rootor]\n\t"
                         "s_cselect_b64 %[cm], %[on], 0\n\t"
                         "s_andn2_b64 %[cm], %[cm], %[g]\n\t"               // path nodes that take part
                         KVC_HEAPL_MOVES
                         "v_cmp_gt_u32_sdwa %[Mn], %[xr], %[xl] src0_sel:WORD_1 src1_sel:WORD_1\n\t"
                         "v_cndmask_b32_e64 %[lo], %[lo], %[inner], %[cm]\n\t"
                         "s_nop 0\n\t"
                         "v_cndmask_b32_e64 %[xc], %[xr], %[xl], %[Mn]\n\t"
                         "v_cndmask_b32_e64 %[ad], %[adR], %[adL], %[Mn]"
                         : [nc] "=&v"(nc), [cm] "=&s"(cm), [upd] "=&s"(upd), [lo] "+v"(lo), [xl] "+v"(xl), [xr] "+v"(xr), [Mn] "=&s"(Mn),
                           [xc] "+v"(xc), [ad] "=&v"(ad), [rnew] "=&s"(rnew)
                         : [xcc] "v"(xcc), [vor] "s"(vor), [vv] "v"(vv), [g] "s"(g), [c] "s"(c), [inner] "v"(inner), [adL] "v"(adL), [adR] "v"(adR),
                           [M] "s"(M), [on] "s"(on), [value] "s"(value), [rootor] "s"(rootor), [c0] "s"(c0)
                         : "vcc", "scc");
            rootor = rnew;
        } else {
            asm volatile("v_cmp_gt_u32_e64 vcc, %[xcc], %[vor]\n\t"
                         "s_andn2_b64 %[cm], %[on], %[g]\n\t"               // path nodes that take part
                         KVC_HEAPL_MOVES
                         "v_cmp_gt_u32_sdwa %[Mn], %[xr], %[xl] src0_sel:WORD_1 src1_sel:WORD_1\n\t"
                         "v_cndmask_b32_e64 %[lo], %[lo], %[inner], %[cm]\n\t"
                         "s_nop 0\n\t"
                         "v_cndmask_b32_e64 %[xc], %[xr], %[xl], %[Mn]\n\t"
                         "v_cndmask_b32_e64 %[ad], %[adR], %[adL], %[Mn]"
                         : [nc] "=&v"(nc), [cm] "=&s"(cm), [upd] "=&s"(upd), [lo] "+v"(lo), [xl] "+v"(xl), [xr] "+v"(xr), [Mn] "=&s"(Mn),
                           [xc] "+v"(xc), [ad] "=&v"(ad)
                         : [xcc] "v"(xcc), [vor] "s"(vor), [vv] "v"(vv), [g] "s"(g), [c] "s"(c), [inner] "v"(inner), [adL] "v"(adL), [adR] "v"(adR),
                           [M] "s"(M), [on] "s"(on)
                         : "vcc", "scc");
        }
        if (PROBE == 0) exchange(ad);
        else if (PROBE == 1) { XC[lane] = xc; asm volatile("" ::: "memory"); typedef __attribute__((address_space(3))) const uint32_t lds_cu32; xcc = *(lds_cu32*)(uintptr_t)ad; }
        else if (PROBE == 2) c0 = N::rdlane(xc, 0);
        else xcc = ad;
    }
    // __sort_heap's step: the same moves, with the NEXT pop's bookkeeping (read node last - 1, park the root in the result
    // array, take the node out of its parent's cache) between the moves and the new M — one exchange per pop instead of two.
    // `value` = the node this pop removed (already out of the caches); returns the next pop's value.  res: LDS, k words.
    __device__ __forceinline__ uint32_t pop_step(uint32_t value, int next_last, uint32_t* res) {
        const uint32_t vor = value | 0xffffu;
        u64 M, on, g, c, cm, upd;
        uint32_t vv, inner, nc, t;
        KVC_HEAPL_BLOCK_A;
        asm volatile("v_cmp_gt_u32_e64 vcc, %[xcc], %[vor]\n\t"
                     "s_andn2_b64 %[cm], %[on], %[g]\n\t"
                     KVC_HEAPL_MOVES
                     "v_cndmask_b32_e64 %[lo], %[lo], %[inner], %[cm]"
                     : [nc] "=&v"(nc), [cm] "=&s"(cm), [upd] "=&s"(upd), [lo] "+v"(lo), [xl] "+v"(xl), [xr] "+v"(xr)
                     : [xcc] "v"(xcc), [vor] "s"(vor), [vv] "v"(vv), [g] "s"(g), [c] "s"(c), [inner] "v"(inner), [M] "s"(M), [on] "s"(on), [xc] "v"(xc)
                     : "vcc", "scc");
        // the next pop: its node is read out of the parent's cache, the root goes to res[next_last] (every lane stores the same
        // word: no branch), SENT takes the node's place.  next_last == 0: node 0 is its own parent's ... nobody's: harmless.
        const int p = (next_last - 1) >> 1;
        const uint32_t nl = N::rdlane(xl, p & 63), nr = N::rdlane(xr, p & 63);
        uint32_t top = N::rdlane(lo, 0);
        asm volatile("" : "+v"(top));
        res[next_last] = top;
        const u64 pm = next_last >= 1 ? 1ull << p : 0ull;
        xl = lane_sel(xl, SENT, (next_last & 1) ? pm : 0ull);
        xr = lane_sel(xr, SENT, (next_last & 1) ? 0ull : pm);
        refresh();
        return (next_last & 1) ? nl : nr;
    }
#undef KVC_HEAPL_BLOCK_A
#undef KVC_HEAPL_MOVES
    // lo_init / hi_init: nodes lane, lane + 64 (SENT beyond the length)
    __device__ __forceinline__ void init(uint32_t* lds, uint32_t lo_init, uint32_t hi_init) {
        lane = threadIdx.x;
        lvl = 31 - __clz(lane + 1);
        XC = lds;
        {
            typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
            const int base = (int)(uintptr_t)(lds_u32_t*)lds;
            adL = base + 8 * lane + 4; adR = base + 8 * lane + 8;
        }
        A = R = 0;
        for (int c = lane; c > 0;) { const int p = (c - 1) >> 1; A |= 1u << p; if (c & 1) R |= 1u << p; c = p; }
        lo = lo_init;
        lds[lane] = lo_init; lds[lane + 64] = hi_init; lds[lane + 128] = SENT;
        __syncthreads();
        xl = lds[2 * lane + 1]; xr = lds[2 * lane + 2];
        __syncthreads();
        lds[lane + 64] = SENT;
        refresh();
    }
    // __make_heap(first, first + k): libstdc++ sifts node (k - 2) / 2 .. 0 down one after the other; the nodes of one level
    // own disjoint subtrees, so one level is one step in which every lane works with its own subtree's value.
    __device__ __forceinline__ void make_heap(int k) {
        if (k < 2) return;
        uint32_t* const LO = XC + 64;                                  // XC[64..128) is SENT between steps: restored below
        for (int d = 31 - __clz((k - 2) / 2 + 1); d >= 0; --d) {
            const uint32_t maskd = d >= 6 ? 0u : ~((1u << ((1 << d) - 1)) - 1u);   // nodes of level >= d
            const int anc = lvl >= d ? ((lane + 1) >> (lvl - d)) - 1 : lane;
            // (plain C++ selects: the compiler places the hazard nops between a compare and the select it feeds)
            const bool left = (xr >> 16) > (xl >> 16);                 // the hole at this node moves to the LEFT child
            const uint32_t Mlo = (uint32_t)__ballot(left);
            LO[lane] = lo;
            asm volatile("" ::: "memory");
            const uint32_t v = LO[anc], vor = v | 0xffffu;
            const bool onp = lvl >= d && ((Mlo & A & maskd) == (R & maskd));
            const bool takes = onp && !(lo > vor), child_takes = takes && !(xc > vor);
            const uint32_t inner = xc > vor ? v : xc, nc = xcc > vor ? v : xcc;
            lo = takes ? inner : lo;
            xl = (child_takes && left) ? nc : xl;
            xr = (child_takes && !left) ? nc : xr;
            if (d > 0) {                                               // the level above caches this level's nodes
                asm volatile("" ::: "memory");
                LO[lane] = lo;
                asm volatile("" ::: "memory");
                const uint32_t cl = LO[(2 * lane + 1) & 63], cr = LO[(2 * lane + 2) & 63];     // no divergent branch here: it
                xl = (lvl == d - 1 && 2 * lane + 1 < 64) ? cl : xl;                             // would move M into VGPRs
                xr = (lvl == d - 1 && 2 * lane + 2 < 64) ? cr : xr;
            }
            asm volatile("" ::: "memory");
            LO[lane] = SENT;
            refresh();
        }
    }
    // node `last` leaves the heap (__sort_heap): SENT in its parent's cache (its own lane never takes part again: a parent
    // whose children are both SENT moves right, and a chosen SENT child stops every update)
    __device__ __forceinline__ void remove(int last) {
        const u64 pm = 1ull << ((last - 1) >> 1);
        if (last & 1) xl = lane_sel(xl, SENT, pm); else xr = lane_sel(xr, SENT, pm);
    }
};

// std::partial_sort(first, first + k, last, greater-by-value) for 1 <= k <= 128, 16-bit dtypes, n <= 65536.
#if defined(KVC_HEAP_STAMPS)        // tools/heap_probe.hip: phase stamps of one wave (never defined in the library build)
__device__ unsigned long long g_heap_stamps[8];
#define KVC_HSTAMP(i) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_heap_stamps[i] = t_; } while (0)
#else
#define KVC_HSTAMP(i) do { } while (0)
#endif
// The row reaches the wave through LDS: 16-byte loads of the aligned vectors that cover it (an aligned vector that holds one
// valid byte lies within that byte's page: the few elements before the row and beyond its end are loaded, never looked at),
// 1 024 vectors = 8 192 keys per segment, every load of a segment in flight at once — the first one feeds the heap, the
// others land while __make_heap runs; a later segment is requested before the scan of the one in LDS begins.  (Round 2 read
// the tail with predicated 2-byte loads, which the compiler serialises: one memory round trip per 64 keys, 53 000 cycles of
// a C2 head's 380 000.)  Keys are made while the vectors are written to LDS, two per operation.  The scan looks at 512 keys
// per step first (each lane the largest of its eight, packed maxima): late in the row nearly every such batch holds nothing
// above the root.
constexpr int kSegKeys = 8192;
constexpr size_t kWaveHeapLLds = (192 + 128) * 4 + (size_t)kSegKeys * 2;      // heap exchange area, one segment of keys
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
template <int DT>
__device__ __forceinline__ uint32_t keys16x2(uint32_t x) {                     // Key<bf16 | fp16>::of on both halves
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const s16x2 sg = __builtin_bit_cast(s16x2, x) >> (s16x2)15;                // v_pk_ashrrev_i16: 0xffff where negative
    const uint32_t k = x ^ ((__builtin_bit_cast(uint32_t, sg) & 0x7fff7fffu) | 0x80008000u);
    // NaN halves -> 0xfffe (kvc_common.h): saturating |x| - threshold is non-zero exactly there
    constexpr uint32_t thr = Key<DT>::nan_above * 0x10001u;
    const u16x2 over = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, x & 0x7fff7fffu), __builtin_bit_cast(u16x2, thr));
    const u16x2 one = __builtin_elementwise_min(over, (u16x2)1);
    const uint32_t m = __builtin_bit_cast(uint32_t, (u16x2)(one * (u16x2)0xffff));
    return (k & ~m) | (0xfffefffeu & m);
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
template <int DT>
__device__ __forceinline__ void partial_sort_waveL(const typename Dt<DT>::raw* s, int n_, int k_, uint32_t* lds, int64_t* out) {
    static_assert(Key<DT>::bits == 16, "packed 32-bit nodes");
    typedef HeapNode<uint32_t> N;
    const int lane = threadIdx.x;
    uint16_t* const seg = reinterpret_cast<uint16_t*>(lds + 320);
    // wave-uniform by construction; said so (an item's pointer and budget come out of by-value argument tables, which the
    // compiler may index through scratch: without this, loop bounds and the step's scalar operands would sit in VGPRs)
    const int n = uni(n_), k = uni(k_);
    const uintptr_t addr = (uintptr_t)uni((u64)reinterpret_cast<uintptr_t>(s));
    const int e0 = (int)(addr & 15) >> 1;                                      // row element i = key e0 + i of the vectors
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) const u32x4 glb_vec;               // global_load, not flat_load (which would tie every LDS wait to the loads in flight)
    glb_vec* const vec = (glb_vec*)(addr & ~(uintptr_t)15);
    const int total = e0 + n, nvec = (total + 7) >> 3;
    constexpr int VPL = kSegKeys / 8 / 64;                                      // vectors per lane and segment: 16
    u32x4 r[VPL];
    auto request = [&](int g) {
#pragma unroll
        for (int u = 0; u < VPL; ++u) { int v = g * (kSegKeys / 8) + u * 64 + lane; v = v < nvec ? v : nvec - 1; r[u] = vec[v]; }
    };
    auto land = [&](int u) {
        u32x4 x = r[u];
        x.x = keys16x2<DT>(x.x); x.y = keys16x2<DT>(x.y); x.z = keys16x2<DT>(x.z); x.w = keys16x2<DT>(x.w);
        reinterpret_cast<u32x4*>(seg)[u * 64 + lane] = x;
    };
    request(0);
    KVC_HSTAMP(0);
    land(0);                                                                    // keys 0 .. 511 - e0: the heap's k <= 128
    __syncthreads();
    WaveHeapL H;
    H.init(lds, lane < k ? N::make(seg[e0 + lane], lane) : WaveHeapL::SENT,
           lane + 64 < k ? N::make(seg[e0 + lane + 64], lane + 64) : WaveHeapL::SENT);
    KVC_HSTAMP(1);
    H.make_heap(k);
    KVC_HSTAMP(2);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int u = 1; u < VPL; ++u) land(u);
    // __heap_select: every later element that sorts before the root replaces it (__pop_heap + __adjust_heap)
    uint32_t rootor = N::rdlane(H.lo, 0) | 0xffffu;                            // the root's key, as (key << 16 | 0xffff)
    constexpr int U = 8;
    for (int g = 0; g * kSegKeys < total; ++g) {
        const bool more = (g + 1) * kSegKeys < total;
        __syncthreads();
        if (more) request(g + 1);
        for (int jb = 0; jb < kSegKeys; jb += U * 64) {
            const int ib = g * kSegKeys + jb - e0;                               // row element of the batch's first key
            if (ib + U * 64 <= k || ib >= n) continue;
            {   // can any of the 512 keys replace the root?  (keys outside [k, n) only make this test conservative)
                const u32x4 x = reinterpret_cast<const u32x4*>(seg)[jb / 8 + lane];
                const uint32_t rr = (rootor >> 16) | (rootor & 0xffff0000u);
                const uint32_t mx = pk_max_u16(pk_max_u16(pk_max_u16(x.x, x.y), pk_max_u16(x.z, x.w)), rr);
                if (__ballot(mx != rr) == 0) continue;
            }
            uint32_t cur[U];
#pragma unroll
            for (int j = 0; j < U; ++j) cur[j] = seg[jb + j * 64 + lane];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int i0 = ib + j * 64;
                const uint32_t nodev = (cur[j] << 16) | (uint32_t)(i0 + lane);       // the candidates as heap nodes
                u64 pending = __ballot(nodev > rootor);
                if (i0 < k || i0 + 64 > n) pending &= __ballot(i0 + lane >= k && i0 + lane < n);
                if (pending) {
                    // One branch per candidate.  The candidate after this one is read out of its lane before the step: a scalar
                    // compare right behind the v_readlane that feeds it would wait 20 cycles.
                    int src = __builtin_ctzll(pending);
                    uint32_t v = __builtin_amdgcn_readlane(nodev, src);
                    do {
                        asm volatile("s_bitset0_b64 %0, %1" : "+s"(pending) : "s"(src));
                        const int src2 = __builtin_ctzll(pending | (1ull << 63));      // (lane 63 when nothing is left: never used)
                        const uint32_t v2 = __builtin_amdgcn_readlane(nodev, src2);
                        H.template step<true>(v, rootor);                          // (re-tests v against the live root)
                        src = src2; v = v2;
                    } while (pending);
                }
            }
        }
        if (more) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < VPL; ++u) land(u);
        }
    }
    KVC_HSTAMP(3);
    // __sort_heap: the root goes to a[last], a[last] is re-inserted from the top
    uint32_t* const res = lds + 192;
    if (k >= 2) {
        const int last = k - 1;
        uint32_t value = H.node(last), top = N::rdlane(H.lo, 0);
        asm volatile("" : "+v"(top));
        res[last] = top;
        H.remove(last);
        H.refresh();
        for (int l = last; l >= 1; --l) value = H.pop_step(value, l - 1, res);      // (the last one stores res[0])
    } else {
        res[0] = H.lo;                                                           // k == 1: lane 0's node; every lane writes, lane 63 last
        asm volatile("" ::: "memory");
        if (lane == 0) res[0] = H.lo;
    }
    KVC_HSTAMP(4);
    __syncthreads();
    if (lane < k) out[lane] = N::index(res[lane]);
    if (lane + 64 < k) out[lane + 64] = N::index(res[lane + 64]);
    KVC_HSTAMP(5);
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
#if defined(KVC_STAMPS)
#define KVC_XSTAMP(slot)                                                                     \
    do {                                                                                     \
        unsigned long long t_;                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");         \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        xstamps_[slot] = t_;                                                                 \
    } while (0)
#define KVC_XCOUNT(slot) (++xstamps_[slot])
#define KVC_XTIC(var)                                                                        \
    unsigned long long var;                                                                  \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#define KVC_XACC(slot, a, b) (xstamps_[slot] += (b) - (a))
#else
#define KVC_XTIC(var) do { } while (0)
#define KVC_XACC(slot, a, b) do { } while (0)
#define KVC_XSTAMP(slot) do { } while (0)
#define KVC_XCOUNT(slot) do { } while (0)
#endif
// Position lists of one partition: the first cap = m/2 + 2 left stoppers (ascending) and the last `ring` right stoppers
// (a ring buffer, newest = highest position), as offsets from `first`.  16-bit offsets in LDS whenever the range's lists
// fit the list region (m <= 65536), 32-bit in the caller's workspace otherwise: a list round trip through the workspace
// costs several memory latencies per partition, and nth_element makes a dozen of them one after the other.
constexpr size_t kMinListBytes = 1024;                   // the two 65-int tables of partition_pivot_small; lists for m < 500
__host__ __device__ inline size_t list_bytes_for(int m) {   // LDS bytes of the 16-bit lists of an m-element range
    const int cap = m / 2 + 2, ring = cap > 64 ? cap : 64;
    return (size_t)(cap + ring) * 2;
}
// Pointers carry their address space (LDS = 3, global = 1): a pointer that may be either compiles to flat_load / flat_store,
// whose LDS round trip is several times a ds_read's and which serialises on both memory counters — on this
// latency-bound kernel that alone cost a factor of two.  AP = the array's element type WITH its address space.
typedef __attribute__((address_space(3))) u64 lds_u64;
typedef __attribute__((address_space(1))) u64 glb_u64;
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(1))) int glb_int;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(1))) int64_t glb_i64;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
// Element access: 64-bit (key << 32 | index) nodes, or — lds_u32: 16-bit dtypes, rows up to 65 536, array in LDS — the same
// node packed as (key << 16 | index) in 32 bits: half the LDS, so that twice as many workgroups stay resident per CU.
template <class AP> struct NodeIO {
    __device__ __forceinline__ static u64 ld(const AP* p, int i) { return (u64)p[i]; }
    __device__ __forceinline__ static void st(AP* p, int i, u64 v) { p[i] = v; }
    __device__ __forceinline__ static uint32_t key(const AP* p, int i) { return (uint32_t)((u64)p[i] >> 32); }
};
typedef __attribute__((address_space(1))) uint32_t glb_u32;
template <class P32> struct NodeIO32 {
    __device__ __forceinline__ static u64 ld(const P32* p, int i) { const uint32_t x = p[i]; return ((u64)(x >> 16) << 32) | (x & 0xffffu); }
    __device__ __forceinline__ static void st(P32* p, int i, u64 v) { p[i] = ((uint32_t)(v >> 32) << 16) | ((uint32_t)v & 0xffffu); }
    __device__ __forceinline__ static uint32_t key(const P32* p, int i) { return p[i] >> 16; }
};
template <> struct NodeIO<lds_u32> : NodeIO32<lds_u32> {};
template <> struct NodeIO<glb_u32> : NodeIO32<glb_u32> {};      // the workspace copy of a long row: half the bytes to scan and to swap
template <class AP>
struct ArrT {                // the view kvc_stl_emul.h's routines take, on an address-space-typed array
    AP* p;
    __device__ __forceinline__ u64 get(int i) const { return uni(NodeIO<AP>::ld(p, i)); }
    __device__ __forceinline__ void set(int i, u64 v) const { NodeIO<AP>::st(p, i, v); }
    __device__ __forceinline__ void swap(int i, int j) const { const u64 a = get(i), b = get(j); set(i, b); set(j, a); }
};
// A store for the lanes of `mask` only, WITHOUT a branch: the compiler turns `if (pred) p[i] = v` into s_and_saveexec +
// s_cbranch_execz, and a branch costs a lone wave ~40 cycles whether taken or not (tools/step_probe.hip).  exec is put back
// inside the same statement.  (Loads need none of this: an index clamped into range makes them unconditional — the compiler
// serialises PREDICATED loads, one memory round trip each, which is what made the partition scan slow in round 2.)
__device__ __forceinline__ void store_lanes(lds_u16* p, int v, u64 mask) {
    u64 keep;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b16 %2, %3\n\ts_mov_b64 exec, %0" : "=&s"(keep) : "s"(mask), "v"((uint32_t)(uintptr_t)p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_lanes(glb_int* p, int v, u64 mask) {
    u64 keep;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dword %2, %3, off\n\ts_mov_b64 exec, %0" : "=&s"(keep) : "s"(mask), "v"(p), "v"(v) : "memory");
}
template <class AP>
struct WaveSel {
    typedef ArrT<AP> Arr;
    AP* arr;
    glb_int* Lbig;                                       // 32-bit lists for any range (caller's workspace, n/2 + 2 each)
    glb_int* Rbig;
    lds_int* small;                                      // LDS list region (>= kMinListBytes)
    int lane;
    int small_bytes = (int)kMinListBytes;                // its size
#if defined(KVC_STAMPS)
    unsigned long long* xstamps_ = nullptr;              // the kernel's stamp array (diagnostic build)
#endif
    __device__ __forceinline__ static uint32_t key(u64 v) { return (uint32_t)(v >> 32); }
    typedef NodeIO<AP> IO;
    __device__ __forceinline__ u64 get(int i) const { return uni(IO::ld(arr, i)); }

    // ONE pass over [first, last): left stoppers (!(key > pk)) in ascending position order, capped at `cap` entries; right
    // stoppers (!(pk > key)) in ascending order too, into a ring of `ring` >= max(cap, 64) slots, so that the serial scan's
    // t-th right stopper FROM THE RIGHT is ring[(NR - 1 - t) mod ring] for every t < cap.
    template <class LT, int UF>
    __device__ __forceinline__ void scan(int first, int last, uint32_t pk, int cap, int ring, LT* Lp, LT* Rr, int& NL, int& NR, int& nrm) {
        for (int base = first; base < last; base += 64 * UF) {
            uint32_t kx[UF];
#pragma unroll
            for (int u = 0; u < UF; ++u) { const int i = base + u * 64 + lane; kx[u] = IO::key(arr, i < last ? i : last - 1); }   // unconditional
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                const int i = base + u * 64 + lane;
                const bool sl = i < last && !(kx[u] > pk), sr = i < last && !(pk > kx[u]);
                const u64 ml = __ballot(sl), mr = __ballot(sr);
                const int rank = NL + __builtin_amdgcn_mbcnt_hi((uint32_t)(ml >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ml, 0));
                store_lanes(Lp + rank, i - first, __ballot(sl && rank < cap));
                NL += __builtin_popcountll(ml);
                int slot = nrm + __builtin_amdgcn_mbcnt_hi((uint32_t)(mr >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mr, 0));
                if (slot >= ring) slot -= ring;
                store_lanes(Rr + slot, i - first, mr);
                const int c = __builtin_popcountll(mr);
                NR += c;
                nrm += c;
                if (nrm >= ring) nrm -= ring;
            }
        }
    }
    template <class LT>
    __device__ __forceinline__ static int r_at(const LT* Rr, int nrm, int ring, int t) {      // t-th right stopper from the right
        int e = nrm - 1 - t;
        if (e < 0) e += ring;
        return (int)Rr[e];
    }
    // LEFT_ONLY (nth_element about to continue in [first, cut)): everything at and beyond the cut is never read again — the
    // selection only ever narrows around nth and the caller sorts [0, nth) afterwards — so the right-hand halves of the swaps
    // (a random read and a random write each) are skipped.
    template <class LT, int US, bool LEFT_ONLY>
    __device__ __forceinline__ void swap_pairs(int first, int T, const LT* Lp, const LT* Rr, int nrm, int ring) {
        // lanes beyond the last pair repeat pair T - 1: they load what its lane loads and store what it stores (every load of an
        // iteration precedes its stores) — no predicate, no branch, no serialised loads
        for (int t0 = 0; t0 < T; t0 += 64 * US) {
            int l[US], r[US];
            u64 av[US], bv[US];
#pragma unroll
            for (int u = 0; u < US; ++u) {
                int t = t0 + u * 64 + lane;
                t = t < T ? t : T - 1;
                l[u] = first + (int)Lp[t];
                r[u] = first + r_at(Rr, nrm, ring, t);
            }
#pragma unroll
            for (int u = 0; u < US; ++u) { if (!LEFT_ONLY) av[u] = IO::ld(arr, l[u]); bv[u] = IO::ld(arr, r[u]); }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < US; ++u) { IO::st(arr, l[u], bv[u]); if (!LEFT_ONLY) IO::st(arr, r[u], av[u]); }
        }
    }
    template <class LT, bool LEFT_ONLY>
    __device__ __forceinline__ void swap_all(int first, int T, const LT* Lp, const LT* Rr, int nrm, int ring) {
        if (T > 2048) swap_pairs<LT, 8, LEFT_ONLY>(first, T, Lp, Rr, nrm, ring);
        else if (T > 128) swap_pairs<LT, 2, LEFT_ONLY>(first, T, Lp, Rr, nrm, ring);
        else swap_pairs<LT, 1, LEFT_ONLY>(first, T, Lp, Rr, nrm, ring);
    }
    template <class LT>
    __device__ __forceinline__ int partition_t(int first, int last, uint32_t pk, LT* Lp, LT* Rr, int nth) {
        const int m = last - first, cap = m / 2 + 2, ring = cap > 64 ? cap : 64;
        int NL = 0, NR = 0, nrm = 0;
        KVC_XTIC(x0_);
        // Loads are grouped UF chunks of 64 elements at a time: a long range in the workspace (n > 18000) is bound by memory
        // round trips, a short one (the hundreds of sort partitions) by the instructions of the padded trips.
        if (m > 16384) scan<LT, 32>(first, last, pk, cap, ring, Lp, Rr, NL, NR, nrm);
        else if (m > 2048) scan<LT, 16>(first, last, pk, cap, ring, Lp, Rr, NL, NR, nrm);
        else if (m > 256) scan<LT, 4>(first, last, pk, cap, ring, Lp, Rr, NL, NR, nrm);
        else scan<LT, 1>(first, last, pk, cap, ring, Lp, Rr, NL, NR, nrm);
        __syncthreads();
        KVC_XTIC(x1_);
        KVC_XACC(12, x0_, x1_);
        int lim = NL < NR ? NL : NR;
        if (lim > cap) lim = cap;
        // l_t < r_t holds exactly for a prefix of t (l ascending, r descending): T by bisection over the wave — 64 probes
        // per step, log64(lim) dependent list reads instead of lim / 64
        int T;
        {
            int lo_t = 0, hi_t = lim;                                     // answer in [lo_t, hi_t]
            while (hi_t - lo_t > 0) {
                const int span = hi_t - lo_t, step = (span + 63) / 64;    // probe t = lo_t + lane * step
                const int t = lo_t + lane * step, tc = t < hi_t ? t : hi_t - 1;                   // (clamped: unconditional loads)
                const bool ok = t < hi_t && (int)Lp[tc] < r_at(Rr, nrm, ring, tc);
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
        KVC_XTIC(x2_);
        KVC_XACC(13, x1_, x2_);
        const int lnext = T < NL ? first + uni((int)Lp[T]) : 0x7fffffff;
        const int rlast = T > 0 ? first + uni(r_at(Rr, nrm, ring, T - 1)) : 0x7fffffff;
        const int cut = lnext < rlast ? lnext : rlast;
        if (nth >= 0 && cut > nth) swap_all<LT, true>(first, T, Lp, Rr, nrm, ring);
        else swap_all<LT, false>(first, T, Lp, Rr, nrm, ring);
        __syncthreads();
        KVC_XTIC(x3_);
        KVC_XACC(14, x2_, x3_);
        return cut;
    }
    __device__ __forceinline__ int partition(int first, int last, int pivot, int nth) {
        const uint32_t pk = key(get(pivot));
        const int m = last - first, cap = m / 2 + 2;
        if (m <= 65536 && list_bytes_for(m) <= (size_t)small_bytes) {
            lds_u16* const l16 = (lds_u16*)small;
            return partition_t<lds_u16>(first, last, pk, l16, l16 + cap, nth);
        }
        return partition_t<glb_int>(first, last, pk, Lbig, Rbig, nth);
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
    __device__ __forceinline__ int partition_pivot_small_body(int first, int last) {
        const int m = last - first;                                       // 4 .. 64
        u64 x = lane < m ? IO::ld(arr, first + lane) : 0ull;
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
        lds_int* const tl = small;                                            // positions by rank (LDS, 2 x 65 ints)
        lds_int* const tr = small + 65;
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
        if (lane < m) IO::st(arr, first + lane, x);
        __syncthreads();
        return first + (lnext < rlast ? lnext : rlast);
    }
    __device__ __forceinline__ static u64 bperm64(u64 v, int src_lane) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)v);
        const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)(v >> 32));
        return ((u64)b << 32) | a;
    }
    __device__ __forceinline__ static u64 bits_below(int h) { return h >= 64 ? ~0ull : ((1ull << h) - 1); }
    // __introsort_loop(first, last, d) of a range of at most 64 elements, one element per lane in a register until every
    // run is <= 16 long.  The loop's recursion only ever splits a range in two and the halves never interact again, so
    // all pending sub-ranges ("segments": lane intervals [sf, sl)) take their next __unguarded_partition_pivot in the SAME
    // step — median-of-three through ds_bpermute, stopper sets as ballots masked to the lane's segment, the pairing by
    // popcounts exactly as in partition_pivot_small — and every segment at step s has depth budget d - s, as in the
    // serial recursion.  A segment that runs out of budget is left to the serial heap sort afterwards (libstdc++'s
    // fallback), same moves as the scalar program.  A 64-element range costs 2-4 steps instead of 3-7 partitions.
    __device__ __forceinline__ void sort64_body(int first, int last, int d) {
        const int m = last - first;                                       // 17 .. 64
        u64 x = lane < m ? IO::ld(arr, first + lane) : 0ull;
        int sf = lane < m ? 0 : lane, sl = lane < m ? m : lane;           // lanes beyond the range: empty segments
        bool done = false;
        u64 bailed = 0;                                                   // first lanes of segments left to the heap sort
        lds_int* const tl = small;                                            // positions by (segment start + rank), 2 x 65 ints
        lds_int* const tr = small + 65;
        const u64 self = 1ull << lane, below_incl = lane >= 63 ? ~0ull : ((2ull << lane) - 1);
        const u64 above = lane >= 63 ? 0ull : (~0ull << (lane + 1));
        while (true) {
            bool active = !done && sl - sf > 16;
            if (__ballot(active) == 0) break;
            if (d == 0) {                                                 // every live segment is out of depth budget
                bailed = __ballot(active && lane == sf);
                break;
            }
            --d;
            KVC_XCOUNT(8);
            // __move_median_to_first(sf, sf + 1, mid, sl - 1)
            const int ia = sf + 1, ib = sf + (sl - sf) / 2, ic = sl - 1;
            const uint32_t ka = (uint32_t)__builtin_amdgcn_ds_bpermute((active ? ia : lane) << 2, (int)(uint32_t)(x >> 32));
            const uint32_t kb = (uint32_t)__builtin_amdgcn_ds_bpermute((active ? ib : lane) << 2, (int)(uint32_t)(x >> 32));
            const uint32_t kc = (uint32_t)__builtin_amdgcn_ds_bpermute((active ? ic : lane) << 2, (int)(uint32_t)(x >> 32));
            int sw;
            if (ka > kb) sw = kb > kc ? ib : (ka > kc ? ic : ia);
            else sw = ka > kc ? ia : (kb > kc ? ic : ib);
            const u64 v0 = bperm64(x, active ? sf : lane), vs = bperm64(x, active ? sw : lane);
            if (active) x = lane == sf ? vs : (lane == sw ? v0 : x);
            // __unguarded_partition(sf + 1, sl, pivot = sf)
            const uint32_t pk = key(vs), kx = key(x);
            const bool in = active && lane > sf;
            const u64 segm = bits_below(sl) & ~bits_below(sf);
            const u64 SL = __ballot(in && !(kx > pk)) & segm, SR = __ballot(in && !(pk > kx)) & segm;
            const bool isl = in && (SL & self), isr = in && (SR & self);
            const int t = __builtin_popcountll(SL & below_incl);          // rank among the segment's left stoppers, 1-based
            const int u = __builtin_popcountll(SR & (above | self));      // rank among its right stoppers, 1-based from the right
            const bool swl = isl && __builtin_popcountll(SR & above) >= t;
            const bool swr = isr && __builtin_popcountll(SL & (below_incl >> 1)) >= u;
            const int T = __builtin_popcountll(__ballot(swl) & segm);
            if (isl) tl[sf + t] = lane;
            if (isr) tr[sf + u] = lane;
            __syncthreads();
            const int partner = swl ? tr[sf + t] : (swr ? tl[sf + u] : lane);
            const int NL = __builtin_popcountll(SL);
            const int lnext = active && T < NL ? tl[sf + T + 1] : 0x7fffffff;
            const int rlast = active && T > 0 ? tr[sf + T] : 0x7fffffff;
            x = bperm64(x, partner);
            __syncthreads();
            if (active) {
                const int cut = lnext < rlast ? lnext : rlast;
                if (lane < cut) sl = cut; else sf = cut;
            }
        }
        if (lane < m) IO::st(arr, first + lane, x);
        __syncthreads();
        Arr A{arr};
        while (bailed) {                                                  // __partial_sort(f, l, l) = heap_select + sort_heap
            const int b = __builtin_ctzll(bailed);
            bailed &= bailed - 1;
            const int f = first + __builtin_amdgcn_readlane(sf, b), l = first + __builtin_amdgcn_readlane(sl, b);
            heap_select_(A, f, l, l);
            sort_heap_(A, f, l);
            __syncthreads();
        }
    }
    // The three big pieces stay out of line (inlined at every call site they cost instruction cache and registers in the
    // long scans), and take the view BY VALUE: a handful of pointers in registers.  As member functions they would get
    // `this`, a pointer to the object in scratch memory, and reload every field through flat loads.
    static __device__ __noinline__ int partition_pivot_small(WaveSel S, int first, int last) { return S.partition_pivot_small_body(first, last); }
    static __device__ __noinline__ void sort64(WaveSel S, int first, int last, int d) { S.sort64_body(first, last, d); }
    // nth >= 0: called by nth_element, which continues on the side that holds nth (see swap_pairs); -1: both sides are needed
    static __device__ __noinline__ int partition_pivot(WaveSel S, int first, int last, int nth) { return S.partition_pivot_body(first, last, nth); }
    __device__ __forceinline__ int partition_pivot_body(int first, int last, int nth) {   // __unguarded_partition_pivot
        if (last - first <= 64) { KVC_XCOUNT(8); return partition_pivot_small(*this, first, last); }
        KVC_XCOUNT(9);
        const int mid = first + (last - first) / 2;
        Arr A{arr};
        KVC_XTIC(m0_);
        if (lane == 0) move_median_to_first_(A, first, first + 1, mid, last - 1);
        __syncthreads();
        KVC_XTIC(m1_);
        KVC_XACC(15, m0_, m1_);
        return partition(first + 1, last, first, nth);
    }
    __device__ __forceinline__ void introselect(int first, int nth, int last, int depth_limit) {
        Arr A{arr};
        while (last - first > 3) {
            if (depth_limit == 0) {
                heap_select_(A, first, nth + 1, last);
                A.swap(first, nth);
                __syncthreads();
                return;
            }
            --depth_limit;
            const int cut = partition_pivot(*this, first, last, nth);
            if (cut <= nth) first = cut; else last = cut;
        }
        insertion_sort_(A, first, last);
        __syncthreads();
    }
    // __final_insertion_sort of [first, last) once every run left by the introsort loop is <= 16 long (see the header of
    // this section): element i ends at i - #(j in [i-15, i): key_j < key_i) + #(j in (i, i+15]: key_j > key_i).  Elements
    // below `lim` are placed; out[pos - first] is written when pos - first < want.  The 30 neighbour keys of a lane are
    // requested in one go (clamped addresses, range tests on the counts): one LDS round trip per 64 elements.
    __device__ __forceinline__ void final_place(int first, int last, int lim, int want, glb_i64* out) {
        for (int base = first; base < lim; base += 64) {
            const int i = base + lane, ic = i < last ? i : last - 1;
            uint32_t kb[15], ka[15];
#pragma unroll
            for (int dlt = 1; dlt < 16; ++dlt) {
                const int jb = ic - dlt, ja = ic + dlt;
                kb[dlt - 1] = IO::key(arr, jb >= first ? jb : first);
                ka[dlt - 1] = IO::key(arr, ja < last ? ja : last - 1);
            }
            const u64 v = IO::ld(arr, ic);
            const uint32_t ki = key(v);
            int pos = ic;
#pragma unroll
            for (int dlt = 1; dlt < 16; ++dlt) {
                if (ic - dlt >= first && kb[dlt - 1] < ki) --pos;
                if (ic + dlt < last && ka[dlt - 1] > ki) ++pos;
            }
            if (i < lim && pos - first < want) out[pos - first] = (int64_t)(v & 0xffffffffull);
        }
    }
    // std::sort(first, last) whose result goes straight to out[first..last) as indices
    __device__ __forceinline__ void sort_to(int first, int last, lds_int* stack, glb_i64* out) {
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
                if (l - f <= 64) { sort64(*this, f, l, d); break; }           // the whole subtree, in registers
                --d;
                const int cut = partition_pivot(*this, f, l, -1);
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
        KVC_XSTAMP(3);
        final_place(first, last, last, last - first, out + first);        // __final_insertion_sort, all elements at once
    }
    // (caps == nullptr in the kernel below: the whole sort.)
    // The first `want` elements of std::sort(first, last), written to out[0..want) as indices (AdaKV / HeadKV keep a prefix of
    // the head's full descending sort, pyramidkv_utils.py:702-724).  std::sort = introsort loop + final insertion sort: the
    // loop only ever splits ranges, and a range that starts at or beyond `want` can be left alone — everything in it sorts
    // behind every element to its left, it is never mixed with them again, and the final insertion sort moves an element
    // by at most 15 places inside its own leaf range.  So: partition exactly as libstdc++ does, descend only into ranges
    // that start before `want`, then place the elements below want + 16 by the stable-insertion count of sort_to().
    __device__ __forceinline__ void sort_prefix_to(int first, int last, int want, lds_int* stack, glb_i64* out) {
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
                if (l - f <= 64) { sort64(*this, f, l, d); break; }
                --d;
                const int cut = partition_pivot(*this, f, l, -1);
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
        final_place(first, last, lim, want, out);
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
    u64 A9, R9;            // the same for this lane's level-6 node 63 + lane (adjust_top)
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
        A9 = R9 = 0;
        for (int c = 63 + lane; c > 0;) { const int p = (c - 1) >> 1; A9 |= 1ull << p; if (c & 1) R9 |= 1ull << p; c = p; }
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
    // adjust<true>(0, 0, len, value) — the step of __heap_select and __sort_heap — without a branch: a branch costs a lone wave
    // ~40 cycles taken or not (tools/hop_probe.hip) and the general form above compiles to fifteen of them per step (the
    // private walk, its three depths, the exits of the top part); here every choice is a select, the private walk always runs
    // and the lane it concerns is picked afterwards.  Same moves, same array (test_compress_exact_ties_*, every C5 layer).
    __device__ __forceinline__ void adjust_top(int len, uint32_t value) {
        // Every lane decides for itself, as top node `lane` and as the owner of level-6 node 63 + lane: is my node on the sift path
        // (its ancestors' choices, the bits of M, all point to it)?  The one thing that needs the other lanes is "does the value
        // land deeper than me" — one ballot of the path nodes that do not sort before the value (and one for the private part),
        // read back as vector operands.  No readlane, no scalar chain, no branch — and no `&&` of two per-lane conditions: the
        // compiler turns each into an s_and_b64 between the v_cmp that made them and the v_cndmask that uses them (~20 cycles per
        // hand-off, ~25 of them in the first branch-free form).  Conditions are folded into the VALUES that are compared instead
        // (an absent node is all ones, an excluded lane compares as zero), so every select hangs on ONE fresh compare.
        const uint32_t vor = value | 0xffffu, ones = 0xffffffffu;
        auto absent = [](int idx, int len_) -> uint32_t { return (uint32_t)((len_ - 1 - idx) >> 31); };   // 0xffffffff if idx >= len_, else 0
        // ---- the private walk (walk(len) + private_stop(value, true)), this lane's sub-heap ----
        const int a1 = q1(), a3 = q3();
        const bool r1 = (p2 | absent(a1 + 1, len)) <= (p1 | 0xffffu);  // move right at level 7: the right child exists and does not sort before the left
        d1 = r1 ? 2 : 1;
        v1 = r1 ? p2 : p1;
        const uint32_t cl = r1 ? p5 : p3, cr = r1 ? p6 : p4;
        const int al = r1 ? a3 + 2 : a3;
        const bool r2 = (cr | absent(al + 1, len)) <= (cl | 0xffffu);
        d2 = (r1 ? 5 : 3) + (r2 ? 1 : 0);
        v2 = r2 ? cr : cl;
        const uint32_t no1 = absent(a1, len), no2 = no1 | absent(al, len);
        int ps = (p0 <= vor) ? 0 : -1;                                 // deepest private path node that does not sort before the value
        ps = ((v1 | no1) <= vor) ? 1 : ps;
        ps = ((v2 | no2) <= vor) ? 2 : ps;
        // ---- on the path? ----
        const int lim = len < 63 ? len : 63;
        const uint32_t t_on = (((uint32_t)M ^ (uint32_t)R) & (uint32_t)A) | absent(lane, lim);          // 0: top node `lane` is on the path
        const u64 t9w = (M ^ R9) & A9;
        const uint32_t t_on9 = (uint32_t)t9w | (uint32_t)(t9w >> 32) | absent(63 + lane, len);            // 0: level-6 node 63 + lane is
        const uint32_t lane0 = lane == 0 ? ones : 0u;
        const uint32_t cand = (t_on | lane0) == 0 ? top : ones;        // (the root is the fallback, not a candidate)
        const u64 c = __ballot(cand <= vor);                           // path nodes the value may stop at
        const int code = t_on9 == 0 ? ps : -1;                         // private stop of the ONE lane whose sub-heap the path enters
        const u64 d = __ballot(code >= 0);
        const u64 cs = c >> 1;
        const u64 dxw = (cs >> lane) | d;                              // != 0: the value lands deeper than top node `lane`
        const uint32_t child = lane_sel(xr, xl, M);
        const uint32_t va = (top & ~lane0) <= vor ? value : top;       // it lands here if nothing deeper takes it (the root: always)
        const uint32_t vb = dxw != 0 ? child : va;
        top = t_on == 0 ? vb : top;
        // ---- private_apply(code, value) ----
        const uint32_t np0 = code == 0 ? value : v1, w1 = code == 1 ? value : v2;
        p0 = code >= 0 ? np0 : p0;
        const int z1 = code >= 1 ? d1 : 0, z2 = code >= 2 ? d2 : 0;
        p1 = z1 == 1 ? w1 : p1; p2 = z1 == 2 ? w1 : p2;
        p3 = z2 == 3 ? value : p3; p4 = z2 == 4 ? value : p4; p5 = z2 == 5 ? value : p5; p6 = z2 == 6 ? value : p6;
        arr[lane] = top;                                               // (lane 63 has no top node: its store to arr[63] is overwritten
        arr[63 + lane] = p0;                                           //  by lane 0's p0 right behind it — one wave, LDS in order)
        asm volatile("" ::: "memory");
        xl = arr[2 * lane + 1]; xr = arr[2 * lane + 2];
        M = __ballot(xr > (xl | 0xffffu));
        const bool fb = len >= 2 && (len & 1) == 0 && ((len - 2) >> 1) < 63;
        M |= fb ? (1ull << (((len - 2) >> 1) & 63)) : 0ull;
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
                    H.adjust_top(k, N::make(kk, i0 + src));
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
        H.adjust_top(last, value);
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
__global__ __launch_bounds__(64) void select_exact_kernel(const SelectArgs a, u64* gscratch /*[heads][n] or null*/, int arr_in_lds,
                                                           int list_bytes /*LDS behind the array / sort region*/) {
    typedef typename Dt<DT>::raw raw;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* stack = reinterpret_cast<int*>(smem);                        // 3 * 96 ints
    u64* lds_arr = reinterpret_cast<u64*>(smem + 1152);
    const int lane = threadIdx.x, head = blockIdx.x, item = blockIdx.y;
    // wave-uniform by construction; said so: the per-item tables are by-value arguments indexed at run time, which the
    // compiler may read through scratch — every branch below would then count as divergent
    const int n = uni(a.n), k = uni(a.k.v[item]);
    const raw* s = reinterpret_cast<const raw*>((uintptr_t)uni((u64)reinterpret_cast<uintptr_t>(a.scores.p[item]))) + (int64_t)head * n;
    int64_t* out = reinterpret_cast<int64_t*>((uintptr_t)uni((u64)reinterpret_cast<uintptr_t>(a.idx.p[item]))) + (int64_t)head * k;
    const bool use_partial_sort = (int64_t)k * 64 <= (int64_t)n;

    // (key << 16 | index) nodes need indices below 65536; longer rows (128k contexts, AdaKV's flattened [H * n] array) take
    // the 64-bit nodes
    if (use_partial_sort && k <= kWaveHeapMaxK) {
        if (Key<DT>::bits == 16 && n <= 65536) {
            if constexpr (Key<DT>::bits == 16) partial_sort_waveL<DT>(s, n, k, reinterpret_cast<uint32_t*>(lds_arr), out);
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
        if (!a.members_only) sort_heap_(H, 0, k);                     // (the heap's content IS the selection; only its order needs the pops)
        __syncthreads();
        for (int t = lane; t < k; t += 64) out[t] = (int64_t)(lds_arr[t] & 0xffffffffull);
    } else {
        // scratch per (item, head): [n u64 array, when it does not fit in LDS][2 x (n/2 + 2) int position lists]
        const int64_t per_head = (arr_in_lds ? 0 : (int64_t)n) + (n / 2 + 2);
        u64* const hs = gscratch + ((int64_t)item * a.heads + head) * per_head;
        glb_int* const lists = (glb_int*)(hs + (arr_in_lds ? 0 : n));
        lds_u64* const larr = (lds_u64*)lds_arr;
        lds_int* const lstack = (lds_int*)stack;
        glb_i64* const gout = (glb_i64*)out;
        // LDS list region: behind the array (n elements), or behind the k_max - 1 elements being sorted — and while
        // nth_element runs on the workspace copy, that sort region is free too: its lists start at lds_arr
        lds_int* const small = (lds_int*)(larr + (arr_in_lds ? n : a.k_max));
#if defined(KVC_STAMPS)
        unsigned long long xstamps_[20] = {}, xsa[20] = {}, xsb[20] = {};
#endif
        u64 nth;
        if (arr_in_lds) {
            auto run = [&](auto* arr_t, lds_int* small_t) {
                typedef typename std::remove_pointer<decltype(arr_t)>::type AP;
                for (int i = lane; i < n; i += 64) NodeIO<AP>::st(arr_t, i, ((u64)Key<DT>::of(s[i]) << 32) | (uint32_t)i);
                __syncthreads();
                WaveSel<AP> S{arr_t, lists, lists + (n / 2 + 2), small_t, lane};
                S.small_bytes = list_bytes;
#if defined(KVC_STAMPS)
                S.xstamps_ = xsa;
#endif
                KVC_XSTAMP(0);
                if (k - 1 != n) S.introselect(0, k - 1, n, lg_(n) * 2);      // std::nth_element(first, first + k - 1, last)
#if defined(KVC_STAMPS)
                xstamps_[10] = xsa[8]; xstamps_[11] = xsa[9];
                for (int i = 0; i < 4; ++i) xstamps_[16 + i] = xsa[12 + i];
#endif
                KVC_XSTAMP(1);
                nth = NodeIO<AP>::ld(arr_t, k - 1);
                KVC_XSTAMP(2);
                S.sort_to(0, k - 1, lstack, gout);                           // std::sort(first, first + k - 1), straight to out
#if defined(KVC_STAMPS)
                xstamps_[3] = xsa[3]; xstamps_[8] = xsa[8]; xstamps_[9] = xsa[9];
                for (int i = 12; i < 16; ++i) xstamps_[i] = xsa[i];
#endif
            };
            if constexpr (Key<DT>::bits == 16) {
                if ((arr_in_lds & 2) != 0) {                                 // packed 32-bit nodes (16-bit dtypes, n <= 65536)
                    lds_u32* const parr = (lds_u32*)lds_arr;
                    run(parr, (lds_int*)(parr + ((n + 1) & ~1)));
                } else {
                    run(larr, small);
                }
            } else {
                run(larr, small);
            }
        } else {
            auto run = [&](auto* garr) {
                typedef typename std::remove_pointer<decltype(garr)>::type AP;
                for (int i = lane; i < n; i += 64) NodeIO<AP>::st(garr, i, ((u64)Key<DT>::of(s[i]) << 32) | (uint32_t)i);
                __syncthreads();
                WaveSel<AP> S{garr, lists, lists + (n / 2 + 2), (lds_int*)larr, lane};
                S.small_bytes = list_bytes + a.k_max * 8;
#if defined(KVC_STAMPS)
                S.xstamps_ = xsa;
#endif
                KVC_XSTAMP(0);
                if (k - 1 != n) S.introselect(0, k - 1, n, lg_(n) * 2);
#if defined(KVC_STAMPS)
                xstamps_[10] = xsa[8]; xstamps_[11] = xsa[9];
                for (int i = 0; i < 4; ++i) xstamps_[16 + i] = xsa[12 + i];
#endif
                KVC_XSTAMP(1);
                // The k - 1 leading elements move to LDS for the sort: it makes hundreds of small partitions, each a handful of
                // dependent accesses — a memory round trip apiece on the workspace copy.
                nth = NodeIO<AP>::ld(garr, k - 1);
                for (int i = lane; i < k - 1; i += 64) larr[i] = NodeIO<AP>::ld(garr, i);
                __syncthreads();
            };
            // (Packed 32-bit nodes in the workspace were measured: the swaps gain 12 %, but the scan is latency-bound and each of its
            // loads then carries half the bytes — 242 -> 518 us at 32k.  64-bit nodes stay.)
            run((glb_u64*)hs);
            WaveSel<lds_u64> S2{larr, lists, lists + (n / 2 + 2), small, lane};
            S2.small_bytes = list_bytes;
#if defined(KVC_STAMPS)
            S2.xstamps_ = xsb;
#endif
            KVC_XSTAMP(2);
            S2.sort_to(0, k - 1, lstack, gout);
#if defined(KVC_STAMPS)
            xstamps_[3] = xsb[3]; xstamps_[8] = xsa[8] + xsb[8]; xstamps_[9] = xsa[9] + xsb[9];
            for (int i = 12; i < 16; ++i) xstamps_[i] = xsa[i] + xsb[i];
#endif
        }
        if (lane == 0) out[k - 1] = (int64_t)(nth & 0xffffffffull);
#if defined(KVC_STAMPS)
        KVC_XSTAMP(4);
        __syncthreads();
        if (lane == 0 && a.fuse == 2 && k >= 20) for (int i = 0; i < 20; ++i) out[i] = (int64_t)xstamps_[i];
#endif
    }
}

// LDS list region behind `fixed` bytes of a workgroup: what the lists of an n-element range need, cut back to what keeps
// as many workgroups resident per CU as the minimal region would (160 KB per CU, counted in 1280-byte granules).
static size_t list_region_bytes(size_t fixed, int n) {
    const size_t cu = 160 * 1024, gran = 1280, full = (list_bytes_for(n < 65536 ? n : 65536) + 15) & ~(size_t)15;
    size_t wgs = cu / ((fixed + kMinListBytes + gran - 1) / gran * gran);
    if (wgs < 1) wgs = 1;
    if (wgs > 32) wgs = 32;
    size_t room = cu / wgs / gran * gran;
    room = room > fixed ? (room - fixed) & ~(size_t)15 : 0;
    size_t bytes = full < room ? full : room;
    return bytes < kMinListBytes ? kMinListBytes : bytes;
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
    int lds_max_n = 18000;
    if (const char* e = getenv("KVC_EXACT_LDS_MAXN")) { const long v = atol(e); if (v >= 1024 && v <= 36000) lds_max_n = (int)v; }   // tuning aid
    const int in_lds = !any_nth || a.n <= lds_max_n;
    // LDS: the whole array (nth_element items, n <= 18000), or the heap of k_max (partial_sort) / the k_max - 1 elements
    // being sorted; behind it the list region of the partitions, as large as the resident workgroups per CU allow (160 KB
    // of LDS, one wave = one workgroup here: these kernels are latency-bound, a second round of workgroups doubles
    // their time) up to what the longest range needs
    const bool packed = in_lds && any_nth && Key<DT>::bits == 16 && a.n <= 65536;
    size_t body = (in_lds && any_nth) ? (packed ? (size_t)((a.n + 1) & ~1) * 4 : (size_t)a.n * 8) : (size_t)a.k_max * 8;
    if (body < kWaveHeap9Lds) body = kWaveHeap9Lds;
    if (body < kWaveHeapLds) body = kWaveHeapLds;                   // 64-bit nodes (fp32, or n > 65536)
    bool any_heap_l = false;                                        // WaveHeapL items: heap + one staged segment of keys
    for (int i = 0; i < a.n_items; ++i) any_heap_l = any_heap_l || (Key<DT>::bits == 16 && a.n <= 65536 && a.k.v[i] <= kWaveHeapMaxK && (int64_t)a.k.v[i] * 64 <= (int64_t)a.n);
    if (any_heap_l && body < kWaveHeapLLds) body = kWaveHeapLLds;
    const size_t list_bytes = any_nth ? list_region_bytes(1152 + body, a.n) : 0;
    const size_t lds = 1152 + body + list_bytes;
    if (any_nth && !scratch) return KVC_ERR_WORKSPACE;
    static LdsCache lds_cache = {};
    if (ensure_lds(reinterpret_cast<const void*>(&select_exact_kernel<DT>), lds, lds_cache) != 0) return KVC_ERR_HIP;
    hipLaunchKernelGGL((select_exact_kernel<DT>), dim3((unsigned)a.heads, (unsigned)a.n_items), dim3(64), lds, st, a,
                       reinterpret_cast<u64*>(scratch), in_lds | (packed ? 2 : 0), (int)list_bytes);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// sort_prefix_kernel (AdaKV / HeadKV): out[head][0..caps[head]) = the first caps[head] indices of
// scores[head].sort(descending=True) as torch-CPU produces them (libstdc++ std::sort, value-only comparator).
// grid = (heads), block = 64.  The (key, index) array lives in LDS (n <= 18000) or in `gscratch` ([heads][n] u64).
// ---------------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(64) void sort_prefix_kernel(const RaggedSortArgs a, u64* gscratch, int arr_in_lds, int list_bytes) {
    typedef typename Dt<DT>::raw raw;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* stack = reinterpret_cast<int*>(smem);                        // 3 * 96 ints
    u64* lds_arr = reinterpret_cast<u64*>(smem + 1152);
    const int lane = threadIdx.x, head = blockIdx.x, n = a.n;
    int want = a.caps ? a.caps[head] : (a.want_fixed > 0 ? a.want_fixed : n);
    want = want < 0 ? 0 : (want > n ? n : want);
    const raw* s = reinterpret_cast<const raw*>(a.scores) + (int64_t)head * n;
    int64_t* out = a.idx_out + (int64_t)head * a.out_stride;
    const int64_t per_head = (arr_in_lds ? 0 : (int64_t)n) + (n / 2 + 2);
    u64* const hs = gscratch + (int64_t)head * per_head;
    glb_int* const lists = (glb_int*)(hs + (arr_in_lds ? 0 : n));
    lds_u64* const larr = (lds_u64*)lds_arr;
    // ascending order = the descending program on complemented keys (comp(a, b) = key'(a) > key'(b) <=> key(a) < key(b))
    const uint32_t flip = a.ascending ? 0xffffffffu : 0u;
    if (arr_in_lds) {
        for (int i = lane; i < n; i += 64) larr[i] = ((u64)(Key<DT>::of(s[i]) ^ flip) << 32) | (uint32_t)i;
        __syncthreads();
        WaveSel<lds_u64> S{larr, lists, lists + (n / 2 + 2), (lds_int*)(larr + n), lane};
        S.small_bytes = list_bytes;
#if defined(KVC_STAMPS)
        unsigned long long xs[20] = {};
        S.xstamps_ = xs;
#endif
        S.sort_prefix_to(0, n, want, (lds_int*)stack, (glb_i64*)out);
    } else {
        glb_u64* const garr = (glb_u64*)hs;
        for (int i = lane; i < n; i += 64) garr[i] = ((u64)(Key<DT>::of(s[i]) ^ flip) << 32) | (uint32_t)i;
        __syncthreads();
        WaveSel<glb_u64> S{garr, lists, lists + (n / 2 + 2), (lds_int*)larr, lane};
        S.small_bytes = list_bytes;
#if defined(KVC_STAMPS)
        unsigned long long xs[20] = {};
        S.xstamps_ = xs;
#endif
        S.sort_prefix_to(0, n, want, (lds_int*)stack, (glb_i64*)out);
    }
}

size_t sort_prefix_scratch_bytes(int heads, int n) {
    return (size_t)heads * ((n <= 18000 ? 0 : (size_t)n) + (size_t)(n / 2 + 2)) * 8;
}

template <int DT>
static int launch_sort_prefix_t(const RaggedSortArgs& a, void* scratch, hipStream_t st) {
    const int in_lds = a.n <= 18000;
    // a handful of workgroups (one per head): the lists of the whole range in LDS when the array leaves room for them
    const size_t fixed = 1152 + (in_lds ? (size_t)a.n * 8 : 0);
    size_t list_bytes = (list_bytes_for(a.n < 65536 ? a.n : 65536) + 15) & ~(size_t)15;
    if (fixed + list_bytes > 160 * 1024) list_bytes = (160 * 1024 - fixed) & ~(size_t)15;
    if (list_bytes < kMinListBytes) list_bytes = kMinListBytes;
    const size_t lds = fixed + list_bytes;
    if (!scratch) return KVC_ERR_WORKSPACE;
    static LdsCache lds_cache = {};
    if (ensure_lds(reinterpret_cast<const void*>(&sort_prefix_kernel<DT>), lds, lds_cache) != 0) return KVC_ERR_HIP;
    hipLaunchKernelGGL((sort_prefix_kernel<DT>), dim3((unsigned)a.heads), dim3(64), lds, st, a, reinterpret_cast<u64*>(scratch), in_lds,
                       (int)list_bytes);
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
