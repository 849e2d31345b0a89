// kvc_l2norm.hip — SURVEY §8(f) N4: the scoring half of L2NormCluster.update_kv (pyramidkv_utils.py:394-429).
//
// Reference: token_norms = torch.norm(key_states, p=2, dim=-1) (:419) over the repeat_kv-expanded keys [1, H_q, L, D];
// argsort ascending (:420); the first max_capacity_prompt rows of the sorted K and V are the new cache (:423-427).  No
// query, no window, no pooling.
//
// Here: l2norm_kernel reads every KEY row once per KV head (the group's query heads share it: H_kv * L * D * esize bytes,
// not H_q times that) and writes the norm to each of the group's query-head rows of the score buffer; the ascending
// std::sort prefix (sort_prefix_kernel, kvc_select_exact.hip, keys complemented) and the gather kernel do the rest.
//
// Arithmetic = torch-CPU's norm kernel for reduced floating types (oracle/kvc_oracle.cpp, kvco_l2norm): 8 fp32 accumulators,
// element d into accumulator d % 8 as acc = acc + x * x (two roundings; exact products for 16-bit inputs), accumulators
// added 0..7 in order, correctly rounded fp32 sqrt, one rounding to the dtype.  One thread per row: a 16-byte load is one
// "vector step" of the 8 accumulators (16-bit dtypes) or half of one (fp32).  HBM-bound on the K read, like the K scan.
#include "kvc_common.h"
#include "kvc_launch.h"

namespace kvc {

template <int DT>
__global__ __launch_bounds__(256) void l2norm_kernel(const L2NormArgs a) {
    typedef typename Dt<DT>::raw raw;
    constexpr int ES = Dt<DT>::esize;
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= a.q_len) return;
    const int bg = blockIdx.y, b = bg / a.n_kv_heads, g = bg % a.n_kv_heads;
    const char* src = reinterpret_cast<const char*>(a.k) + ((int64_t)b * a.stride_b + (int64_t)g * a.stride_h + (int64_t)row * a.stride_l) * ES;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    const int chunks = a.head_dim * ES / 16;
    for (int c = 0; c < chunks; ++c) {
        const uint4 v = reinterpret_cast<const uint4*>(src)[c];
        if constexpr (DT == KVC_FP32) {
            const float x[4] = {u2f(v.x), u2f(v.y), u2f(v.z), u2f(v.w)};
            const int o = (c & 1) * 4;                                   // elements 4c .. 4c+3 -> accumulators (4c + j) % 8
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float sq = x[j] * x[j];
                if (o == 0) acc[j] = acc[j] + sq; else acc[4 + j] = acc[4 + j] + sq;
            }
        } else {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x0 = Dt<DT>::ld((uint16_t)(w[j] & 0xffffu)), x1 = Dt<DT>::ld((uint16_t)(w[j] >> 16));
                acc[2 * j] = acc[2 * j] + x0 * x0;
                acc[2 * j + 1] = acc[2 * j + 1] + x1 * x1;
            }
        }
    }
    float tot = acc[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) tot = tot + acc[j];
    const raw out = Dt<DT>::st(__builtin_sqrtf(tot));
    const int group = a.n_q_heads / a.n_kv_heads;
    raw* dst = reinterpret_cast<raw*>(a.norms) + ((int64_t)b * a.n_q_heads + (int64_t)g * group) * a.q_len + row;
    for (int r = 0; r < group; ++r) dst[(int64_t)r * a.q_len] = out;
}

int launch_l2norm(const L2NormArgs& a, int dtype, hipStream_t st) {
    const dim3 grid((unsigned)((a.q_len + 255) / 256), (unsigned)(a.bsz * a.n_kv_heads));
    switch (dtype) {
        case KVC_BF16: hipLaunchKernelGGL((l2norm_kernel<KVC_BF16>), grid, dim3(256), 0, st, a); return 0;
        case KVC_FP16: hipLaunchKernelGGL((l2norm_kernel<KVC_FP16>), grid, dim3(256), 0, st, a); return 0;
        case KVC_FP32: hipLaunchKernelGGL((l2norm_kernel<KVC_FP32>), grid, dim3(256), 0, st, a); return 0;
    }
    return KVC_ERR_INVALID;
}

}  // namespace kvc
