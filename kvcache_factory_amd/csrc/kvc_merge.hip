// kvc_merge.hip — SURVEY §8(f) N4: merge_kv(..., merge="pivot") (pyramidkv_utils.py:119-170, LOOK-M), the step SnapKV /
// PyramidKV / H2O take instead of the plain gather when the cluster was built with merge="pivot" (:337-339).
//
// Reference, per head (all H_q heads; K, V repeat_kv-expanded):
//   dropped  = the positions NO head selected — torch.isin against every head's indices (:131-134) — the window included;
//   kept K   = [window rows, selected rows] (:145);   kept V = [selected rows, window rows] (:147)   (the orders differ);
//   similarity = normalise(dropped K) @ normalise(kept K)^T, cosine, in the dtype (:149);  pivot = its first maximal column;
//   merged   = (dropped + kept[pivot]) / 2 (:155, :158);  kept = scatter_reduce(kept, pivot, merged, 'mean', include_self) (:156, :160)
//   — the pivot found on the KEY order is applied to the value rows of the same number.
//
// Here, six launches after the method's scoring + selection (the indices are the select stage's output):
//   merge_mask_kernel     every selected index marks its position                                  (H_q * k threads)
//   merge_drop_kernel     ordered compaction of the unmarked positions -> drop[], drop_len          (one workgroup per batch row)
//   merge_kept_kernel     the normalised kept keys of every head, fp32 in the workspace             (one wave per row)
//   merge_pivot_kernel    the cosines on the f32 MFMA (32 dropped tokens per wave x 128 kept rows per LDS stage; 64
//                         v_mfma_f32_32x32x2_f32 per block = the d-ascending fp32 chain: products of two 16-bit values are exact
//                         in fp32), rounded to the dtype, first maximum
//   merge_reduce_kernel   one wave per (head, kept row) for K and V together: walks the pivots 64 at a time (ballot -> ascending order), adds the
//                         merged rows in fp32, then sum -> dtype, count -> dtype, quotient -> dtype
// Arithmetic = oracle/kvc_oracle.cpp kvco_merge_pivot (norms in torch's 8-accumulator order, fp32 divides, one rounding each),
// which equals the imported reference bit for bit on every fixture (tests/golden/merge_*).
// Cost: the pivot stage is H_q * drop_len * (k + W) * D multiply-adds (4.2 G at 8k -> 128) at the f32 MFMA rate (157 TFLOP/s peak;
// a scalar-FMA version with the kept rows broadcast from LDS ran at 43 TFLOP/s, 194 us); everything else is small.  The K scan of the scoring stage is unchanged.
#include "kvc_common.h"
#include "kvc_launch.h"

namespace kvc {

__global__ __launch_bounds__(256) void merge_mask_kernel(const MergeArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per_b = (int64_t)a.n_q_heads * a.n_keep;
    if (i >= per_b * a.bsz) return;
    const int b = (int)(i / per_b);
    a.mask[(int64_t)b * a.q_len + a.idx[i]] = 1;
}

// grid = bsz, block = 1024: drop[b][0 .. drop_len[b]) = ascending positions with mask == 0
__global__ __launch_bounds__(1024) void merge_drop_kernel(const MergeArgs a) {
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int32_t* mask = a.mask + (int64_t)b * a.q_len;
    int32_t* drop = a.drop + (int64_t)b * a.q_len;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int j0 = 0; j0 < a.q_len; j0 += 1024) {
        const int j = j0 + tid;
        const bool keep = j < a.q_len && mask[j] == 0;
        const unsigned long long m = __ballot(keep);
        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
        if (lane == 0) wave_cnt[wave] = __builtin_popcountll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (keep) drop[off + rank] = j;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wave_cnt[w]; base_s += t; }
        __syncthreads();
    }
    if (tid == 0) a.drop_len[b] = base_s;
}

// torch-CPU's norm of one row held two elements per lane (head_dim 128): accumulator j sums the elements d with d % 8 == j in
// ascending d, then the accumulators are added 0..7.  Lane l holds d = 2l, 2l + 1: accumulator (2l) % 8 and (2l + 1) % 8 get
// their terms from lanes l, l + 4, l + 8, ... in that order.
template <int DT>
__device__ __forceinline__ float row_norm128(float x0, float x1, int lane) {
    const float s0 = x0 * x0, s1 = x1 * x1;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int step = 0; step < 16; ++step) {                                  // elements 8 * step .. 8 * step + 7 = lanes 4 * step .. + 3
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int src = 4 * step + q;
            acc[2 * q] = acc[2 * q] + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s0), src));
            acc[2 * q + 1] = acc[2 * q + 1] + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s1), src));
        }
    }
    float tot = acc[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) tot = tot + acc[j];
    return rnd<DT>(__builtin_sqrtf(tot));
}

// Source position of kept row r of head hb: KEY order [window, selected] / VALUE order [selected, window].
__device__ __forceinline__ int64_t kept_pos(const MergeArgs& a, int64_t hb, int r, bool value_order) {
    if (value_order) return r < a.n_keep ? a.idx[hb * a.n_keep + r] : (int64_t)(a.q_len - a.window) + (r - a.n_keep);
    return r < a.window ? (int64_t)(a.q_len - a.window) + r : a.idx[hb * a.n_keep + (r - a.window)];
}

// grid = (R, bsz * H_q), block = 64: kn[hb][r][:] = dtype(x / norm) as fp32
template <int DT>
__global__ __launch_bounds__(64) void merge_kept_kernel(const MergeArgs a) {
    typedef typename Dt<DT>::raw raw;
    const int r = blockIdx.x, lane = threadIdx.x;
    const int64_t hb = blockIdx.y;
    const int b = (int)(hb / a.n_q_heads), h = (int)(hb % a.n_q_heads);
    const raw* row = reinterpret_cast<const raw*>(a.k) + (int64_t)b * a.k_stride_b + (int64_t)(h / a.group) * a.k_stride_h +
                     kept_pos(a, hb, r, false) * a.k_stride_l;
    const uint32_t w = reinterpret_cast<const uint32_t*>(row)[lane];
    const float x0 = Dt<DT>::ld((raw)(w & 0xffffu)), x1 = Dt<DT>::ld((raw)(w >> 16));
    const float n = row_norm128<DT>(x0, x1, lane);
    float2 o;
    o.x = rnd<DT>(x0 / n);
    o.y = rnd<DT>(x1 / n);
    reinterpret_cast<float2*>(a.kn + (hb * a.rows + r) * 128)[lane] = o;
}

// grid = (ceil(q_len / 128), bsz * H_q), block = 256 = 4 waves: pivot[hb][p] for p < drop_len[b].
// Each wave owns 32 dropped tokens (the B operand of v_mfma_f32_32x32x2_f32: lane (kh, j) keeps the elements d = 2s + kh of
// its token j, normalised, in 64 registers); the kept rows are the A operand, 64 rows at a time through LDS (de-interleaved:
// [row][kh][s], so that four steps of a lane are one 16-byte read).  64 MFMAs per 32 x 32 block of cosines = the d-ascending
// fp32 chain (each MFMA adds its two products in k order; products of two 16-bit values are exact in fp32), at twice the
// rate of the scalar FMA; the first maximum over the kept rows is taken with an explicit (value, smaller row) rule.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MP_STAGE = 64;                               // kept rows per LDS stage: 33 KB, four workgroups per CU
constexpr int MP_PITCH = 132;                              // floats per kept row in LDS (16-byte reads of a quarter wave: all banks)
template <int DT>
__global__ __launch_bounds__(256) void merge_pivot_kernel(const MergeArgs a) {
    typedef typename Dt<DT>::raw raw;
    extern __shared__ __attribute__((aligned(16))) float kept[];                            // [MP_STAGE][MP_PITCH]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kh = lane >> 5, j = lane & 31;
    const int64_t hb = blockIdx.y;
    const int b = (int)(hb / a.n_q_heads), h = (int)(hb % a.n_q_heads);
    const int M = a.drop_len[b];
    if ((int)blockIdx.x * 128 >= M) return;                                                 // uniform per workgroup
    const int p = blockIdx.x * 128 + wave * 32 + j;
    const bool live = p < M;
    float breg[64];
    {
        const int64_t pos = live ? a.drop[(int64_t)b * a.q_len + p] : 0;
        const raw* row = reinterpret_cast<const raw*>(a.k) + (int64_t)b * a.k_stride_b + (int64_t)(h / a.group) * a.k_stride_h + pos * a.k_stride_l;
        float acc4[4] = {0.0f, 0.0f, 0.0f, 0.0f};                                          // accumulators 2i + kh of torch's eight
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const uint4 v = reinterpret_cast<const uint4*>(row)[c];
            const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {                                                   // elements 8c + 2i + kh
                const float e = Dt<DT>::ld((raw)(kh ? (wv[i] >> 16) : (wv[i] & 0xffffu)));
                breg[4 * c + i] = e;
                acc4[i] = acc4[i] + e * e;
            }
        }
        float oth[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) oth[i] = xor_lane<32>(acc4[i]);
        float tot = kh ? oth[0] : acc4[0];                                                  // accumulators 0..7 in order
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float ev = kh ? oth[i] : acc4[i], od = kh ? acc4[i] : oth[i];
            if (i > 0) tot = tot + ev;
            tot = tot + od;
        }
        const float n = rnd<DT>(__builtin_sqrtf(tot));
#pragma unroll
        for (int s_ = 0; s_ < 64; ++s_) breg[s_] = rnd<DT>(breg[s_] / n);
    }
    float bestv = 0.0f;
    int best = 0x7fffffff;
    const float* kn = a.kn + hb * a.rows * 128;
    for (int r0 = 0; r0 < a.rows; r0 += MP_STAGE) {
        const int nr = a.rows - r0 < MP_STAGE ? a.rows - r0 : MP_STAGE;
        __syncthreads();
        for (int i = tid; i < MP_STAGE * 32; i += 256) {                                         // (row, 4 consecutive d) -> evens | odds
            const int row = i >> 5, c = i & 31;
            const float4 z = {0.0f, 0.0f, 0.0f, 0.0f};
            const float4 u = row < nr ? reinterpret_cast<const float4*>(kn + (int64_t)(r0 + row) * 128)[c] : z;
            *reinterpret_cast<float2*>(kept + row * MP_PITCH + 2 * c) = float2{u.x, u.z};
            *reinterpret_cast<float2*>(kept + row * MP_PITCH + 64 + 2 * c) = float2{u.y, u.w};
        }
        __syncthreads();
        const int n_mt = (nr + 31) >> 5;
        for (int mt = 0; mt < n_mt; ++mt) {
            const float* arow = kept + (mt * 32 + j) * MP_PITCH + kh * 64;
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float4 av = *reinterpret_cast<const float4*>(arow + 4 * g);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, breg[4 * g], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, breg[4 * g + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, breg[4 * g + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, breg[4 * g + 3], acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = r0 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                const float v = rnd<DT>(acc[e]);
                if (r < a.rows && (best == 0x7fffffff || v > bestv || (v == bestv && r < best))) { best = r; bestv = v; }
            }
        }
    }
    {   // the two half-waves hold different kept rows of the same token: first maximum of the pair
        const float ov = xor_lane<32>(bestv);
        const int orow = __float_as_int(xor_lane<32>(__int_as_float(best)));
        if (orow != 0x7fffffff && (best == 0x7fffffff || ov > bestv || (ov == bestv && orow < best))) { best = orow; bestv = ov; }
    }
    if (live && kh == 0) a.pivot[hb * a.q_len + p] = best;
}

// grid = (R, bsz * H_q), block = 64: kept row r of the KEY order and kept row r of the VALUE order (the same pivots feed both,
// :155-160); lane l holds elements 2l, 2l + 1 of each
template <int DT>
__global__ __launch_bounds__(64) void merge_reduce_kernel(const MergeArgs a) {
    typedef typename Dt<DT>::raw raw;
    const int r = blockIdx.x, lane = threadIdx.x;
    const int64_t hb = blockIdx.y;
    const int b = (int)(hb / a.n_q_heads), h = (int)(hb % a.n_q_heads);
    const raw* kbase = reinterpret_cast<const raw*>(a.k) + (int64_t)b * a.k_stride_b + (int64_t)(h / a.group) * a.k_stride_h;
    const raw* vbase = reinterpret_cast<const raw*>(a.v) + (int64_t)b * a.v_stride_b + (int64_t)(h / a.group) * a.v_stride_h;
    const uint32_t wk = reinterpret_cast<const uint32_t*>(kbase + kept_pos(a, hb, r, false) * a.k_stride_l)[lane];
    const uint32_t wv = reinterpret_cast<const uint32_t*>(vbase + kept_pos(a, hb, r, true) * a.v_stride_l)[lane];
    const float ks0 = Dt<DT>::ld((raw)(wk & 0xffffu)), ks1 = Dt<DT>::ld((raw)(wk >> 16));
    const float vs0 = Dt<DT>::ld((raw)(wv & 0xffffu)), vs1 = Dt<DT>::ld((raw)(wv >> 16));
    float ksum0 = ks0, ksum1 = ks1, vsum0 = vs0, vsum1 = vs1;
    int cnt = 1;
    const int M = a.drop_len[b];
    const int32_t* pv = a.pivot + hb * a.q_len;
    const int32_t* drop = a.drop + (int64_t)b * a.q_len;
    constexpr int UW = 8;                                                // chunks of 64 pivots requested together (the walk is a
    for (int p0 = 0; p0 < M; p0 += 64 * UW) {                            // chain of L2 round trips otherwise)
        int pvv[UW], posv[UW];
#pragma unroll
        for (int u = 0; u < UW; ++u) {
            const int p = p0 + u * 64 + lane;
            pvv[u] = p < M ? pv[p] : -1;
            posv[u] = p < M ? drop[p] : 0;
        }
#pragma unroll
        for (int u = 0; u < UW; ++u) {
            unsigned long long m = __ballot(pvv[u] == r);
            while (m) {
                const int i = __builtin_ctzll(m);
                m &= m - 1;
                const int64_t src = __builtin_amdgcn_readlane(posv[u], i);
                const uint32_t dk = reinterpret_cast<const uint32_t*>(kbase + src * a.k_stride_l)[lane];
                const uint32_t dv = reinterpret_cast<const uint32_t*>(vbase + src * a.v_stride_l)[lane];
                ksum0 = ksum0 + rnd<DT>(rnd<DT>(Dt<DT>::ld((raw)(dk & 0xffffu)) + ks0) / 2.0f);
                ksum1 = ksum1 + rnd<DT>(rnd<DT>(Dt<DT>::ld((raw)(dk >> 16)) + ks1) / 2.0f);
                vsum0 = vsum0 + rnd<DT>(rnd<DT>(Dt<DT>::ld((raw)(dv & 0xffffu)) + vs0) / 2.0f);
                vsum1 = vsum1 + rnd<DT>(rnd<DT>(Dt<DT>::ld((raw)(dv >> 16)) + vs1) / 2.0f);
                ++cnt;
            }
        }
    }
    const float c = rnd<DT>((float)cnt);
    const int64_t head_elems = a.out_stride_h ? a.out_stride_h : (int64_t)a.rows * 128;
    raw* ko = reinterpret_cast<raw*>(a.k_out) + hb * head_elems + (int64_t)r * 128;
    raw* vo = reinterpret_cast<raw*>(a.v_out) + hb * head_elems + (int64_t)r * 128;
    reinterpret_cast<uint32_t*>(ko)[lane] = (uint32_t)Dt<DT>::st(rnd<DT>(ksum0) / c) | ((uint32_t)Dt<DT>::st(rnd<DT>(ksum1) / c) << 16);
    reinterpret_cast<uint32_t*>(vo)[lane] = (uint32_t)Dt<DT>::st(rnd<DT>(vsum0) / c) | ((uint32_t)Dt<DT>::st(rnd<DT>(vsum1) / c) << 16);
}

template <int DT>
static int launch_merge_t(const MergeArgs& a, hipStream_t st) {
    const int heads = a.bsz * a.n_q_heads;
    if (hipMemsetAsync(a.mask, 0, (size_t)a.bsz * a.q_len * sizeof(int32_t), st) != hipSuccess) return KVC_ERR_HIP;
    const int64_t n_idx = (int64_t)heads * a.n_keep;
    hipLaunchKernelGGL(merge_mask_kernel, dim3((unsigned)((n_idx + 255) / 256)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(merge_drop_kernel, dim3((unsigned)a.bsz), dim3(1024), 0, st, a);
    hipLaunchKernelGGL((merge_kept_kernel<DT>), dim3((unsigned)a.rows, (unsigned)heads), dim3(64), 0, st, a);
    static LdsCache lds_cache = {};
    const size_t lds = (size_t)MP_STAGE * MP_PITCH * sizeof(float);
    if (ensure_lds(reinterpret_cast<const void*>(&merge_pivot_kernel<DT>), lds, lds_cache) != 0) return KVC_ERR_HIP;
    hipLaunchKernelGGL((merge_pivot_kernel<DT>), dim3((unsigned)((a.q_len + 127) / 128), (unsigned)heads), dim3(256), lds, st, a);
    hipLaunchKernelGGL((merge_reduce_kernel<DT>), dim3((unsigned)a.rows, (unsigned)heads), dim3(64), 0, st, a);
    return 0;
}

int launch_merge(const MergeArgs& a, int dtype, hipStream_t st) {
    switch (dtype) {
        case KVC_BF16: return launch_merge_t<KVC_BF16>(a, st);
        case KVC_FP16: return launch_merge_t<KVC_FP16>(a, st);
    }
    return KVC_ERR_UNSUPPORTED;
}

}  // namespace kvc
