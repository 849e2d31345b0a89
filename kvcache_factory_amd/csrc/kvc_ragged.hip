// kvc_ragged.hip — SURVEY §8(f) N3: AdaKV / HeadKV per-head budgets and the ragged ("flattened") compressed cache.
//   AdaKVCluster.update_kv   pyramidkv/pyramidkv_utils.py:674-757      HeadKVCluster.update_kv   :813-878
//   the repo's only native kernel, update_flatten_view, csrc/csrc/cuda_api.cu:12-85 (decode-time append)
//
// Reference flow: pooled window-MEAN scores [H][n] -> per-head full descending sort (torch-CPU sort = libstdc++ std::sort) ->
// AdaKV: scores normalised by ratio_h = sum(top base) / sum(all), global top-(H*base) over the flattened [H*n] values,
// count per head, capacity_h = round(count_h * (1 - floor) + floor_capacity); HeadKV: capacities given -> every head keeps
// the first capacity_h indices of ITS sort + the W window rows, all heads concatenated into one [sum_h len_h, D] tensor.  At
// decode, update_flatten_view allocates a NEW tensor of sum_h len_h + H rows and copies the whole cache with one row
// inserted per head — O(cache) per token and layer.
//
// Here:
//   ragged_head_stats_kernel  one workgroup per head: the base-th largest score by a 16/32-round bit search over keys held in
//                             registers, the two sums, ratio_h (dtype roundings as the reference: sum -> dtype, / -> dtype);
//   ragged_budget_kernel      one workgroup per batch row: bit search of the (H*base)-th largest normalised score over all
//                             H*n values, per-head counts above / at the threshold, capacities and lengths.  Ties AT the global
//                             threshold that span several heads are shared out in flattened (head-major) order and
//                             reported in flags[0] — torch's partial_sort would need a 3840-node heap walk to reproduce; inside
//                             one head they do not change a count;
//   sort_prefix_kernel        (kvc_select_exact.hip) the first capacity_h indices of the head's std::sort, tie order included;
//   ragged_gather_kernel      rows -> the flat cache, each head at seg_off[h] (the host leaves `slack` spare rows behind every
//                             head so decode appends IN PLACE instead of re-copying the cache);
//   ragged_append_kernel + ragged_attention_kernel   one decode step over the ragged cache (the reference: update_flatten_view
//                             + flash_attn_varlen_func, llama_model.py:2363-2390).
#include "kvc_common.h"
#include "kvc_launch.h"

namespace kvc {

constexpr int RG_THREADS = 1024;

// Block-wide sum of an int over RG_THREADS threads (all threads get the result).  scratch: 17 ints of LDS.
__device__ __forceinline__ int block_sum_i(int v, int* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    int t = 0;
    for (int w = 0; w < RG_THREADS / 64; ++w) t += scratch[w];
    return t;
}
__device__ __forceinline__ float block_sum_f(float v, float* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float t = 0.0f;
    for (int w = 0; w < RG_THREADS / 64; ++w) t += scratch[w];
    return t;
}

// grid = (bsz * H), block = 1024.  ratio[hb] = rnd(rnd(sum of the `base` largest scores) / rnd(sum of all scores)).
template <int DT>
__global__ __launch_bounds__(RG_THREADS) void ragged_head_stats_kernel(const RaggedArgs a) {
    typedef typename Dt<DT>::raw raw;
    __shared__ int si[17];
    __shared__ float sf[17];
    const int hb = blockIdx.x, n = a.n, base = a.base;
    const raw* s = reinterpret_cast<const raw*>(a.scores) + (int64_t)hb * n;
    constexpr int KB = Key<DT>::bits;
    uint32_t P = 0;
    for (int b = KB - 1; b >= 0; --b) {
        const uint32_t cand = P | (1u << b);
        int c = 0;
        for (int j = threadIdx.x; j < n; j += RG_THREADS) c += Key<DT>::of(s[j]) >= cand ? 1 : 0;
        if (block_sum_i(c, si) >= base) P = cand;
    }
    int gt = 0;
    float sum_gt = 0.0f, sum_all = 0.0f, tval = 0.0f;
    for (int j = threadIdx.x; j < n; j += RG_THREADS) {
        const float v = Dt<DT>::ld(s[j]);
        const uint32_t key = Key<DT>::of(s[j]);
        sum_all += v;
        if (key > P) { ++gt; sum_gt += v; }
        if (key == P) tval = v;
    }
    const int gt_all = block_sum_i(gt, si);
    const float s_gt = block_sum_f(sum_gt, sf), s_all = block_sum_f(sum_all, sf);
    // the threshold's value: any thread that saw it (all the same bits)
    float tv = tval;
    for (int o = 32; o >= 1; o >>= 1) { const float x = __shfl_xor(tv, o, 64); tv = tv > x ? tv : x; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sf[threadIdx.x >> 6] = tv;
    __syncthreads();
    if (threadIdx.x == 0) {
        float T = 0.0f;
        for (int w = 0; w < RG_THREADS / 64; ++w) T = sf[w] > T ? sf[w] : T;
        const float top = s_gt + (float)(base - gt_all) * T;
        a.ratio[hb] = rnd<DT>(rnd<DT>(top) / rnd<DT>(s_all));
    }
}

// grid = (bsz), block = 1024.  caps[b][h], lens[b][h] (= caps + W), flags[b][0..1].
template <int DT>
__global__ __launch_bounds__(RG_THREADS) void ragged_budget_kernel(const RaggedArgs a) {
    typedef typename Dt<DT>::raw raw;
    __shared__ int si[17];
    __shared__ int gt_h[256], eq_h[256];
    const int b = blockIdx.x, n = a.n, H = a.heads_per_batch;
    const raw* s = reinterpret_cast<const raw*>(a.scores) + (int64_t)b * H * n;
    const float* ratio = a.ratio + (int64_t)b * H;
    const int64_t k_tot = (int64_t)H * a.base;
    constexpr int KB = Key<DT>::bits;
    auto key_of = [&](int h, int j) -> uint32_t {
        const raw r = s[(int64_t)h * n + j];
        if (!a.normalize) return Key<DT>::of(r);
        return Key<DT>::of(Dt<DT>::st(Dt<DT>::ld(r) * ratio[h]));              // adaptive_attn_score * ratio_weight, rounded to dtype
    };
    uint32_t P = 0;
    for (int bit = KB - 1; bit >= 0; --bit) {
        const uint32_t cand = P | (1u << bit);
        int c = 0;
        for (int h = 0; h < H; ++h)
            for (int j = threadIdx.x; j < n; j += RG_THREADS) c += key_of(h, j) >= cand ? 1 : 0;
        if ((int64_t)block_sum_i(c, si) >= k_tot) P = cand;
    }
    for (int h = threadIdx.x; h < H; h += RG_THREADS) { gt_h[h] = 0; eq_h[h] = 0; }
    __syncthreads();
    for (int h = 0; h < H; ++h) {                                             // per-head counts (H <= 256)
        int g = 0, q = 0;
        for (int j = threadIdx.x; j < n; j += RG_THREADS) {
            const uint32_t key = key_of(h, j);
            g += key > P ? 1 : 0;
            q += key == P ? 1 : 0;
        }
        const int G = block_sum_i(g, si), Q = block_sum_i(q, si);
        if (threadIdx.x == 0) { gt_h[h] = G; eq_h[h] = Q; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t above = 0, ties = 0;
        int tie_heads = 0;
        for (int h = 0; h < H; ++h) { above += gt_h[h]; ties += eq_h[h]; tie_heads += eq_h[h] > 0 ? 1 : 0; }
        int64_t remaining = k_tot - above;                                    // slots for values equal to the threshold
        a.flags[2 * b] = (ties > remaining && tie_heads > 1) ? 1 : 0;         // cross-head tie at the global threshold
        a.flags[2 * b + 1] = (int)remaining;
        const int floor_cap = a.floor_capacity;
        for (int h = 0; h < H; ++h) {
            const int64_t take = eq_h[h] < remaining ? eq_h[h] : remaining;   // flattened (head-major) order
            remaining -= take;
            const int cnt = gt_h[h] + (int)take;
            const int cap = (int)__builtin_rintf((float)cnt * a.mix + (float)floor_cap);    // torch.round: half to even
            a.caps[(int64_t)b * H + h] = cap;
        }
    }
}

// ---- tie_mode torch_cpu: the reference's own top-(H*base), tie order included --------------------------------------------
// flat[b][h * n + r] = round(sorted_score[h][r] * ratio[h]): the array the reference hands to torch.topk (:705-711) — every head's
// scores in ITS descending sort order (idx from sort_prefix_kernel, whole sort), normalised.  grid = (ceil(n/256), bsz*H).
template <int DT>
__global__ __launch_bounds__(256) void ragged_flatten_kernel(const RaggedArgs a, const int64_t* idx, void* flat) {
    typedef typename Dt<DT>::raw raw;
    const int hb = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x, n = a.n;
    if (r >= n) return;
    const raw* s = reinterpret_cast<const raw*>(a.scores) + (int64_t)hb * n;
    const float v = Dt<DT>::ld(s[idx[(int64_t)hb * n + r]]);
    reinterpret_cast<raw*>(flat)[(int64_t)hb * n + r] = a.normalize ? Dt<DT>::st(v * a.ratio[hb]) : Dt<DT>::st(v);
}
// caps[b][h] = round(#(top indices that fall into head h) * (1 - floor) + floor_capacity)   (:708-717).  grid = (bsz), block = 1024.
__global__ __launch_bounds__(RG_THREADS) void ragged_recount_kernel(const RaggedArgs a, const int64_t* top) {
    __shared__ int cnt[256];
    const int b = blockIdx.x, H = a.heads_per_batch;
    const int64_t k_tot = (int64_t)H * a.base;
    for (int h = threadIdx.x; h < H; h += RG_THREADS) cnt[h] = 0;
    __syncthreads();
    for (int64_t t = threadIdx.x; t < k_tot; t += RG_THREADS) atomicAdd(&cnt[(int)(top[(int64_t)b * k_tot + t] / a.n)], 1);
    __syncthreads();
    for (int h = threadIdx.x; h < H; h += RG_THREADS)
        a.caps[(int64_t)b * H + h] = (int)__builtin_rintf((float)cnt[h] * a.mix + (float)a.floor_capacity);
    if (threadIdx.x == 0) { a.flags[2 * b] = 0; a.flags[2 * b + 1] = 0; }
}

// grid = (ceil(rows_max / rows_per_block), bsz * H, 2 {K, V}), block = 256
__global__ __launch_bounds__(256) void ragged_gather_kernel(const RaggedGatherArgs a) {
    const int which = blockIdx.z, hb = blockIdx.y;
    const int lanes_per_row = a.row_bytes >> 4, rows_per_block = 256 / lanes_per_row;
    const int r = threadIdx.x / lanes_per_row, c = threadIdx.x % lanes_per_row;
    const int cap = a.caps[hb];
    const int t = blockIdx.x * rows_per_block + r;
    if (r >= rows_per_block || t >= cap + a.window) return;
    const int b = hb / a.n_q_heads, h = hb % a.n_q_heads;
    const int64_t srow = t < cap ? a.idx[(int64_t)hb * a.idx_stride + t] : (int64_t)(a.q_len - a.window) + (t - cap);
    const char* src = reinterpret_cast<const char*>(which ? a.v : a.k) +
        ((int64_t)b * (which ? a.v_stride_b : a.k_stride_b) + (int64_t)(h / a.group) * (which ? a.v_stride_h : a.k_stride_h) +
         srow * (which ? a.v_stride_l : a.k_stride_l)) * a.esize + c * 16;
    char* dst = reinterpret_cast<char*>(which ? a.v_flat : a.k_flat) + (a.seg_off[hb] + t) * (int64_t)a.row_bytes + c * 16;
    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
}

// ---- decode over the ragged cache -------------------------------------------------------------------------------------------
// Head hb's rows live at flat + seg_off[hb] * row; seg_len[hb] of them after the prefill, `appended` more since.
// grid = (new_rows, bsz * H, 2), block = 64
__global__ __launch_bounds__(64) void ragged_append_kernel(const RaggedDecodeArgs a) {
    const int i = blockIdx.x, hb = blockIdx.y, which = blockIdx.z;
    const int b = hb / a.n_q_heads, h = hb % a.n_q_heads;
    const int pieces = a.row_bytes >> 4;
    const char* src = reinterpret_cast<const char*>(which ? a.v_new : a.k_new) +
        ((int64_t)b * a.new_stride_b + (int64_t)(h / a.group) * a.new_stride_h + (int64_t)i * a.new_stride_l) * a.esize;
    char* dst = reinterpret_cast<char*>(which ? a.v_flat : a.k_flat) +
        (a.seg_off[hb] + a.seg_len[hb] + a.appended + i) * (int64_t)a.row_bytes;
    for (int c = threadIdx.x; c < pieces; c += 64)
        reinterpret_cast<uint4*>(dst)[c] = reinterpret_cast<const uint4*>(src)[c];
}

template <int DT> __device__ __forceinline__ void widen16(const uint4& v, float* x) {
    if constexpr (DT == KVC_FP32) {
        x[0] = u2f(v.x); x[1] = u2f(v.y); x[2] = u2f(v.z); x[3] = u2f(v.w);
    } else {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[2 * i] = Dt<DT>::ld((uint16_t)(w[i] & 0xffffu)); x[2 * i + 1] = Dt<DT>::ld((uint16_t)(w[i] >> 16)); }
    }
}

// grid = (bsz * H, new_rows), block = 256: the same online-softmax attention as kvc_decode.hip over ONE contiguous segment
template <int DT, int D>
__global__ __launch_bounds__(256) void ragged_attention_kernel(const RaggedDecodeArgs a) {
    typedef typename Dt<DT>::raw raw;
    constexpr int ES = Dt<DT>::esize, EPL = 16 / ES, LPR = D / EPL, RPW = 64 / LPR, RPI = 4 * RPW, ROWB = D * ES;
    __shared__ float sm_m[4 * RPW], sm_l[4 * RPW];
    __shared__ float sm_acc[4 * RPW][D];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, slot = lane / LPR, c = lane % LPR;
    const int hb = blockIdx.x, b = hb / a.n_q_heads, h = hb % a.n_q_heads, i = blockIdx.y;
    const int R = a.seg_len[hb] + a.appended + i + 1;                       // rows visible to token i of the step
    const float scale = a.scaling * 1.44269504088896340736f;
    float qf[EPL];
    {
        const char* qrow = reinterpret_cast<const char*>(a.q) + ((int64_t)b * a.q_stride_b + (int64_t)h * a.q_stride_h + (int64_t)i * a.q_stride_l) * ES;
        widen16<DT>(reinterpret_cast<const uint4*>(qrow)[c], qf);
#pragma unroll
        for (int e = 0; e < EPL; ++e) qf[e] *= scale;
    }
    const char* kp = reinterpret_cast<const char*>(a.k_flat) + a.seg_off[hb] * (int64_t)ROWB;
    const char* vp = reinterpret_cast<const char*>(a.v_flat) + a.seg_off[hb] * (int64_t)ROWB;
    float m = -1.0e30f, l = 0.0f, acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.0f;
    for (int r0 = 0; r0 < R; r0 += RPI) {
        const int r = r0 + wave * RPW + slot;
        const bool valid = r < R;
        float kf[EPL], vf[EPL];
        if (valid) {
            widen16<DT>(reinterpret_cast<const uint4*>(kp + (int64_t)r * ROWB)[c], kf);
            widen16<DT>(reinterpret_cast<const uint4*>(vp + (int64_t)r * ROWB)[c], vf);
        } else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) { kf[e] = 0.0f; vf[e] = 0.0f; }
        }
        float s = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) s = __builtin_fmaf(qf[e], kf[e], s);
        if constexpr (LPR > 1) s += xor_lane<1>(s);
        if constexpr (LPR > 2) s += xor_lane<2>(s);
        if constexpr (LPR > 4) s += xor_lane<4>(s);
        if constexpr (LPR > 8) s += xor_lane<8>(s);
        if constexpr (LPR > 16) s += xor_lane<16>(s);
        if (valid) {
            const float mn = s > m ? s : m;
            const float f = __builtin_amdgcn_exp2f(m - mn), p = __builtin_amdgcn_exp2f(s - mn);
            l = __builtin_fmaf(l, f, p);
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] = __builtin_fmaf(acc[e], f, p * vf[e]);
            m = mn;
        }
    }
    const int sl = wave * RPW + slot;
    if (c == 0) { sm_m[sl] = m; sm_l[sl] = l; }
#pragma unroll
    for (int e = 0; e < EPL; ++e) sm_acc[sl][c * EPL + e] = acc[e];
    __syncthreads();
    if (tid < D) {
        float M = sm_m[0];
        for (int s_ = 1; s_ < 4 * RPW; ++s_) M = sm_m[s_] > M ? sm_m[s_] : M;
        float Lsum = 0.0f, o = 0.0f;
        for (int s_ = 0; s_ < 4 * RPW; ++s_) {
            const float f = __builtin_amdgcn_exp2f(sm_m[s_] - M);
            Lsum = __builtin_fmaf(sm_l[s_], f, Lsum);
            o = __builtin_fmaf(sm_acc[s_][tid], f, o);
        }
        raw* out = reinterpret_cast<raw*>(a.out) + ((int64_t)b * a.out_stride_b + (int64_t)h * a.out_stride_h + (int64_t)i * a.out_stride_l);
        out[tid] = Dt<DT>::st(o / Lsum);
    }
}

int launch_ragged_plan(const RaggedArgs& a, int dtype, int bsz, hipStream_t st) {
#define KVC_CASE(DT_) if (dtype == DT_) { \
        if (a.normalize) hipLaunchKernelGGL((ragged_head_stats_kernel<DT_>), dim3((unsigned)(bsz * a.heads_per_batch)), dim3(RG_THREADS), 0, st, a); \
        hipLaunchKernelGGL((ragged_budget_kernel<DT_>), dim3((unsigned)bsz), dim3(RG_THREADS), 0, st, a); return 0; }
    KVC_CASE(KVC_BF16) KVC_CASE(KVC_FP16) KVC_CASE(KVC_FP32)
#undef KVC_CASE
    return KVC_ERR_INVALID;
}

int launch_ragged_head_stats(const RaggedArgs& a, int dtype, int bsz, hipStream_t st) {
#define KVC_CASE(DT_) if (dtype == DT_) { hipLaunchKernelGGL((ragged_head_stats_kernel<DT_>), dim3((unsigned)(bsz * a.heads_per_batch)), dim3(RG_THREADS), 0, st, a); return 0; }
    KVC_CASE(KVC_BF16) KVC_CASE(KVC_FP16) KVC_CASE(KVC_FP32)
#undef KVC_CASE
    return KVC_ERR_INVALID;
}
int launch_ragged_flatten(const RaggedArgs& a, int dtype, int bsz, const int64_t* idx, void* flat, hipStream_t st) {
    const dim3 grid((unsigned)((a.n + 255) / 256), (unsigned)(bsz * a.heads_per_batch));
#define KVC_CASE(DT_) if (dtype == DT_) { hipLaunchKernelGGL((ragged_flatten_kernel<DT_>), grid, dim3(256), 0, st, a, idx, flat); return 0; }
    KVC_CASE(KVC_BF16) KVC_CASE(KVC_FP16) KVC_CASE(KVC_FP32)
#undef KVC_CASE
    return KVC_ERR_INVALID;
}
int launch_ragged_recount(const RaggedArgs& a, int bsz, const int64_t* top, hipStream_t st) {
    hipLaunchKernelGGL(ragged_recount_kernel, dim3((unsigned)bsz), dim3(RG_THREADS), 0, st, a, top);
    return 0;
}

int launch_ragged_gather(const RaggedGatherArgs& a, int rows_max, int heads, hipStream_t st) {
    const int lanes_per_row = a.row_bytes / 16, rows_per_block = 256 / lanes_per_row;
    hipLaunchKernelGGL(ragged_gather_kernel, dim3((unsigned)((rows_max + rows_per_block - 1) / rows_per_block), (unsigned)heads, 2), dim3(256), 0, st, a);
    return 0;
}

int launch_ragged_decode(const RaggedDecodeArgs& a, int dtype, int head_dim, int heads, hipStream_t st) {
    hipLaunchKernelGGL(ragged_append_kernel, dim3((unsigned)a.new_rows, (unsigned)heads, 2), dim3(64), 0, st, a);
    const dim3 grid((unsigned)heads, (unsigned)a.new_rows);
#define KVC_CASE(DT_, D_) if (dtype == DT_ && head_dim == D_) { hipLaunchKernelGGL((ragged_attention_kernel<DT_, D_>), grid, dim3(256), 0, st, a); return 0; }
    KVC_CASE(KVC_BF16, 128) KVC_CASE(KVC_BF16, 64) KVC_CASE(KVC_FP16, 128) KVC_CASE(KVC_FP16, 64)
    KVC_CASE(KVC_FP32, 128) KVC_CASE(KVC_FP32, 64)
#undef KVC_CASE
    return KVC_ERR_UNSUPPORTED;
}

}  // namespace kvc
