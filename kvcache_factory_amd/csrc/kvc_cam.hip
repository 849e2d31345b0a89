// kvc_cam.hip — SURVEY §8(f) N4, CAMKVCluster.update_kv (pyramidkv_utils.py:431-513), split at its random draw.
//
// The reference, after SnapKV-style window scoring without pooling (:461-476), walks the tokens s = start_budget .. L - W - 1
// in a Python loop (:487-502): merge_prob[h] = colmean[h][s] / max(max(colmean[h][:start_budget]), max(colmean[h][s : s + W])) with
// colmean = the window rows' MEAN probability of a key (torch.mean over dim -2 of the dtype tensor: fp32 sum, / W, one rounding),
// NaN -> 0, inf -> 1, clamp to [0, 1]; merge_mask = torch.bernoulli(merge_prob) — the ONLY random step; then
// value[s + 1 .. s + W] += value[s] * merge_mask / W, in place, in dtype arithmetic (each += rounds).  colmean does not depend on
// the merged values, so every token's probability can be computed up front:
//   cam_colmean_kernel : colmean over ALL L key columns (the merge loop looks W columns ahead, into the window block) from the
//                        scoring stage's logits / row maxima / row sums in the workspace — the same p = round(exp_u20(x - m) / sum)
//   cam_prob_kernel    : merge_prob [heads][L - W - start_budget], dtype
//   cam_merge_kernel   : the value recurrence given the draws (uint8 mask): one thread per (head, channel), rows in order; a row's
//                        W pending successors live in an LDS ring.  Bit-exact given the mask (tests: the reference's own draws,
//                        captured in the fixtures).
// The draw itself is the caller's (torch.bernoulli on the GPU in kvcache_factory_amd.pyramidkv_utils.CAMKVCluster): its stream
// differs from the reference's CPU generator, which is why parity is stated "given the mask".
#include "kvc_common.h"
#include "kvc_launch.h"

namespace kvc {

// grid = (ceil(L / 256), heads), block = 256
template <int DT>
__global__ __launch_bounds__(256) void cam_colmean_kernel(const CamArgs a) {
    typedef typename Dt<DT>::raw raw;
    __shared__ float m[64], rinv[64];
    const int hb = blockIdx.y, W = a.window, L = a.q_len, tid = threadIdx.x;
    if (tid < W) {
        m[tid] = a.rowmax[(int64_t)hb * W + tid];
        rinv[tid] = 1.0f / a.rowsum[(int64_t)hb * W + tid];
    }
    __syncthreads();
    const int j = blockIdx.x * 256 + tid;
    if (j >= L) return;
    const raw* lg = reinterpret_cast<const raw*>(a.logits) + ((int64_t)hb * L + j) * W;
    CascadeSum cs;
    cs.init(W);
    for (int w = 0; w < W; ++w) cs.add(rnd<DT>(exp_u20(Dt<DT>::ld(lg[w]) - m[w]) * rinv[w]));
    reinterpret_cast<raw*>(a.colmean)[(int64_t)hb * L + j] = Dt<DT>::st(cs.result() / (float)W);
}

// grid = (heads), block = 256
template <int DT>
__global__ __launch_bounds__(256) void cam_prob_kernel(const CamArgs a) {
    typedef typename Dt<DT>::raw raw;
    __shared__ float red[256];
    const int hb = blockIdx.x, W = a.window, L = a.q_len, tid = threadIdx.x, start = a.start;
    const raw* cm = reinterpret_cast<const raw*>(a.colmean) + (int64_t)hb * L;
    float mx = -__builtin_inff();
    for (int j = tid; j < start; j += 256) { const float v = Dt<DT>::ld(cm[j]); mx = v > mx ? v : mx; }
    red[tid] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] = red[tid + s] > red[tid] ? red[tid + s] : red[tid]; __syncthreads(); }
    const float m0 = red[0];
    const int nT = L - W - start;
    raw* out = reinterpret_cast<raw*>(a.prob) + (int64_t)hb * nT;
    for (int t = tid; t < nT; t += 256) {
        const int s = start + t;
        float wm = m0;
        for (int j = s; j < s + W; ++j) { const float v = Dt<DT>::ld(cm[j]); wm = v > wm ? v : wm; }
        float p = rnd<DT>(Dt<DT>::ld(cm[s]) / wm);                    // dtype / dtype: fp32 quotient, one rounding
        if (p != p) p = 0.0f;                                          // :497
        if (p == __builtin_inff() || p == -__builtin_inff()) p = 1.0f; // :498
        p = p < 0.0f ? 0.0f : (p > 1.0f ? 1.0f : p);                   // .clamp(min=0, max=1)
        out[t] = Dt<DT>::st(p);
    }
}

// grid = (b * heads), block = head_dim (<= 256).  LDS: ring[W][head_dim] floats.
template <int DT>
__global__ void cam_merge_kernel(const CamArgs a) {
    typedef typename Dt<DT>::raw raw;
    extern __shared__ float ring[];                                    // slot (row - start) % W, channel
    const int hb = blockIdx.x, b = hb / a.n_q_heads, h = hb % a.n_q_heads, d = threadIdx.x, D = blockDim.x;
    const int W = a.window, L = a.q_len, start = a.start, nT = L - W - start;
    raw* v = reinterpret_cast<raw*>(a.v) + (int64_t)b * a.v_stride_b + (int64_t)h * a.v_stride_h + d;
    const uint8_t* mask = a.mask + (int64_t)hb * nT;
    for (int i = 0; i < W; ++i) ring[i * D + d] = Dt<DT>::ld(v[(int64_t)(start + i) * a.v_stride_l]);
    const float fw = (float)W;
    for (int t = 0; t < nT; ++t) {
        const int s = start + t, slot = t % W;
        const float src = ring[slot * D + d];                          // row s: every merge into it has happened
        v[(int64_t)s * a.v_stride_l] = Dt<DT>::st(src);
        ring[slot * D + d] = Dt<DT>::ld(v[(int64_t)(s + W) * a.v_stride_l]);      // row s + W enters
        const float score1 = rnd<DT>(rnd<DT>(src * (mask[t] ? 1.0f : 0.0f)) / fw);  // value * merge_mask / merge_budget (:500)
        for (int i = 0; i < W; ++i) ring[i * D + d] = rnd<DT>(ring[i * D + d] + score1);   // rows s + 1 .. s + W (:501)
    }
    for (int i = 0; i < W; ++i) {                                      // rows L - W .. L - 1
        const int r = L - W + i;
        v[(int64_t)r * a.v_stride_l] = Dt<DT>::st(ring[((r - start) % W) * D + d]);
    }
}

int launch_cam_prob(const CamArgs& a, int dtype, hipStream_t st) {
    const int heads = a.bsz * a.n_q_heads;
    const dim3 g((unsigned)((a.q_len + 255) / 256), (unsigned)heads);
#define KVC_CASE(DT_) if (dtype == DT_) { hipLaunchKernelGGL((cam_colmean_kernel<DT_>), g, dim3(256), 0, st, a); \
        hipLaunchKernelGGL((cam_prob_kernel<DT_>), dim3((unsigned)heads), dim3(256), 0, st, a); return 0; }
    KVC_CASE(KVC_BF16) KVC_CASE(KVC_FP16) KVC_CASE(KVC_FP32)
#undef KVC_CASE
    return KVC_ERR_UNSUPPORTED;
}

int launch_cam_merge(const CamArgs& a, int dtype, int head_dim, hipStream_t st) {
    const size_t lds = (size_t)a.window * head_dim * sizeof(float);
#define KVC_CASE(DT_) if (dtype == DT_) { hipLaunchKernelGGL((cam_merge_kernel<DT_>), dim3((unsigned)(a.bsz * a.n_q_heads)), dim3((unsigned)head_dim), lds, st, a); return 0; }
    KVC_CASE(KVC_BF16) KVC_CASE(KVC_FP16) KVC_CASE(KVC_FP32)
#undef KVC_CASE
    return KVC_ERR_UNSUPPORTED;
}

}  // namespace kvc
