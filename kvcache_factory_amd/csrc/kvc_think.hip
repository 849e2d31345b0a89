// kvc_think.hip — SURVEY §8(f) N4: ThinK channel pruning, key_pruner_query_driven (pyramidkv_utils.py:13-26), the step
// SnapKVCluster.update_think (:349-392) takes after its compression: per head, the int(D * ratio) channels with the smallest
// (mean of q^2 over the last 32 query rows) * (mean of k'^2 over the compressed keys) are dropped from all but the last
// `recent` compressed keys.
//
//   think_select_kernel   one workgroup per head, one thread per channel: the two means as torch computes them (x * x rounded to
//                         the dtype, fp32 outer-dimension cascade over the rows, / rows, one rounding), their rounded product; then
//                         torch-CPU topk(largest=False) on the D channel scores: libstdc++ nth_element + sort on (value, index),
//                         i.e. the descending program of kvc_stl_emul.h on complemented keys, run by one wave (D <= 128 elements)
//   think_compact_kernel  kv_pruned[h][r][:] = the kept channels of compressed row r, in channel order
// Arithmetic = oracle/kvc_oracle.cpp kvco_think_prune, which equals the imported reference bit for bit on every fixture.
#include "kvc_common.h"
#include "kvc_launch.h"
#include "kvc_stl_emul.h"

namespace kvc {

// grid = bsz * H, block = 128 (threads >= D idle in the sums)
template <int DT>
__global__ __launch_bounds__(128) void think_select_kernel(const ThinkArgs a) {
    typedef typename Dt<DT>::raw raw;
    __shared__ u64 arr[128];
    __shared__ int stack[3 * 96];
    const int d = threadIdx.x, D = a.head_dim;
    const int64_t hb = blockIdx.x;
    const int b = (int)(hb / a.n_heads), h = (int)(hb % a.n_heads);
    if (d < D) {
        const raw* q = reinterpret_cast<const raw*>(a.q) + (int64_t)b * a.q_stride_b + (int64_t)h * a.q_stride_h + (int64_t)(a.q_len - 32) * a.q_stride_l + d;
        CascadeSum qs;
        qs.init(32);
        for (int r = 0; r < 32; ++r) { const float x = Dt<DT>::ld(q[(int64_t)r * a.q_stride_l]); qs.add(rnd<DT>(x * x)); }
        const raw* kc = reinterpret_cast<const raw*>(a.kc) + hb * a.kc_stride_h + d;
        CascadeSum ks;
        ks.init(a.rows);
        for (int r = 0; r < a.rows; ++r) { const float x = Dt<DT>::ld(kc[(int64_t)r * D]); ks.add(rnd<DT>(x * x)); }
        const float sc = rnd<DT>(rnd<DT>(qs.result() / 32.0f) * rnd<DT>(ks.result() / (float)a.rows));
        if (a.scores) reinterpret_cast<raw*>(a.scores)[hb * D + d] = Dt<DT>::st(sc);
        // smallest first = the descending program on complemented keys
        arr[d] = ((u64)(~Key<DT>::of(Dt<DT>::st(sc))) << 32) | (uint32_t)d;
        a.keep[hb * D + d] = 1;
    }
    __syncthreads();
    if (threadIdx.x < 64 && a.n_prune > 0) {                           // one wave walks libstdc++'s moves (uniform values)
        Arr A{arr};
        const int k = a.n_prune;
        if ((int64_t)k * 64 <= D) {
            heap_select_(A, 0, k, D);
            sort_heap_(A, 0, k);
        } else {
            if (k - 1 != D) introselect_(A, 0, k - 1, D, lg_(D) * 2);
            sort_(A, 0, k - 1, stack);
        }
    }
    __syncthreads();
    if (d < a.n_prune) a.keep[hb * D + (int)(arr[d] & 0xffffffffull)] = 0;
}

// grid = (ceil(rows_out / 4), bsz * H), block = 256: thread (row slot, channel)
template <int DT>
__global__ __launch_bounds__(256) void think_compact_kernel(const ThinkArgs a) {
    typedef typename Dt<DT>::raw raw;
    __shared__ int dst_of[128];
    const int D = a.head_dim, tid = threadIdx.x;
    const int64_t hb = blockIdx.y;
    if (tid < 64) {                                                    // exclusive scan of the keep mask: output column of channel d
        int run = 0;
        for (int d0 = 0; d0 < D; d0 += 64) {
            const int d = d0 + tid;
            const bool kp = d < D && a.keep[hb * D + d] != 0;
            const unsigned long long m = __ballot(kp);
            const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (d < D) dst_of[d] = kp ? run + rank : -1;
            run += __builtin_popcountll(m);
        }
    }
    __syncthreads();
    const int per = 256 / D;                                           // rows per block iteration (2 at D = 128, 4 at D = 64)
    const int slot = tid / D, d = tid % D;
    const int rows_out = a.rows - a.recent, Dk = D - a.n_prune;
    const raw* kc = reinterpret_cast<const raw*>(a.kc) + hb * a.kc_stride_h;
    raw* out = reinterpret_cast<raw*>(a.pruned) + hb * (int64_t)rows_out * Dk;
    for (int r = blockIdx.x * 4 + slot; r < rows_out && r < (int)blockIdx.x * 4 + 4; r += per) {
        const int c = dst_of[d];
        if (c >= 0) out[(int64_t)r * Dk + c] = kc[(int64_t)r * D + d];
    }
}

int launch_think(const ThinkArgs& a, int dtype, hipStream_t st) {
    const int heads = a.bsz * a.n_heads, rows_out = a.rows - a.recent;
#define KVC_CASE(DT_)                                                                                                         \
    case DT_:                                                                                                                 \
        hipLaunchKernelGGL((think_select_kernel<DT_>), dim3((unsigned)heads), dim3(128), 0, st, a);                          \
        if (rows_out > 0 && a.head_dim > a.n_prune)                                                                           \
            hipLaunchKernelGGL((think_compact_kernel<DT_>), dim3((unsigned)((rows_out + 3) / 4), (unsigned)heads), dim3(256), 0, st, a); \
        return 0;
    switch (dtype) {
        KVC_CASE(KVC_BF16) KVC_CASE(KVC_FP16) KVC_CASE(KVC_FP32)
    }
#undef KVC_CASE
    return KVC_ERR_INVALID;
}

}  // namespace kvc
