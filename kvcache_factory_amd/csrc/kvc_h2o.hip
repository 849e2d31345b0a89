// kvc_h2o.hip — A10: H2O heavy-hitter scores (pyramidkv_utils.py:544-561) on gfx950.
//   attn = Q K^T / sqrt(D) for ALL L query rows; only the last W x W block is causally masked (rows before the
//   window see every key — a reference quirk that is reproduced); P = softmax_fp32(attn).to(dtype);
//   score[j] = sum over all L rows of P[:, j] (torch's cascade order), j < L - W; no pooling.
//
// Three kernels (exact-arithmetic first version: the logits are materialised once, [Hq][L][L] in dtype):
//   h2o_logits_kernel : one wave = 32 query rows of one head, looping over every 32-key tile of its KV head
//                       (K tile staged in the wave's own double-buffered LDS region, loads of tile t+1 in flight
//                       during the 64 f32-input MFMAs of tile t).  Stores S row-major and the exact row maximum.
//   h2o_rowsum_kernel : softmax denominators in torch's own order (16 strided chains + xor butterfly 8,4,2,1 —
//                       aten vec::reduce_all, 16 fp32 lanes): 4 rows per wave, 16 lanes per row.
//   h2o_colsum_kernel : one thread per key column walks the L rows in order: p = round(exp_u20(s - m) * rinv),
//                       accumulated in torch's cascade order (SumKernel multi_row_sum), rounded once.
// MFMA-bound by construction (2*Hq*L*L*D flops at the f32-MFMA rate); a bf16-MFMA variant is the planned
// follow-up and cannot be bit-exact with the oracle (tools/mfma_probe.hip).
#include "kvc_common.h"
#include "kvc_launch.h"

namespace kvc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int DT> __device__ __forceinline__ float pick2(const uint4& v, int s, int kh);
template <> __device__ __forceinline__ float pick2<KVC_BF16>(const uint4& v, int s, int kh) {
    const uint32_t w = s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w;
    return u2f(__builtin_amdgcn_perm(w, w, kh ? 0x07060c0cu : 0x05040c0cu));
}
template <> __device__ __forceinline__ float pick2<KVC_FP16>(const uint4& v, int s, int kh) {
    const uint32_t w = s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w;
    return Dt<KVC_FP16>::ld((uint16_t)(kh ? (w >> 16) : (w & 0xffffu)));
}
template <> __device__ __forceinline__ float pick2<KVC_FP32>(const uint4& v, int s, int kh) {
    const uint32_t w = s == 0 ? (kh ? v.y : v.x) : (kh ? v.w : v.z);
    return u2f(w);
}

// grid = (ceil(row_tiles / 4), bsz * n_q_heads), block = 256: wave w of block x owns query rows [32*(4x+w), +32).
template <int DT, int D>
__global__ __launch_bounds__(256) void h2o_logits_kernel(const H2OArgs a) {
    typedef typename Dt<DT>::raw raw;
    constexpr int ES = Dt<DT>::esize;
    constexpr int ROWB = D * ES, CH = ROWB / 16, PAIRS = 8 / ES, SWZ = CH < 16 ? CH - 1 : 15, STG = CH / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
    const int hb = blockIdx.y, b = hb / a.n_q_heads, h = hb % a.n_q_heads, g = h / a.group;
    const int L = a.q_len, W = a.window;
    const int r0 = (blockIdx.x * 4 + wave) * 32;
    if (r0 >= L) return;                                   // whole wave idle (no barriers below)
    char* const buf = smem + wave * (2 * 32 * ROWB);
    const int n_t = (L + 31) / 32;
    const float sqrt_d = a.sqrt_d;
    const char* kbase = reinterpret_cast<const char*>(a.k) + ((int64_t)b * a.k_stride_b + (int64_t)g * a.k_stride_h) * ES;

    auto issue = [&](int tile, uint4 (&st)[STG]) {
#pragma unroll
        for (int it = 0; it < STG; ++it) {
            const int c = it * 64 + lane, r = c / CH, cc = c % CH;
            const int key = tile * 32 + r;
            st[it] = key < L ? *reinterpret_cast<const uint4*>(kbase + (int64_t)key * a.k_stride_l * ES + cc * 16) : make_uint4(0, 0, 0, 0);
        }
    };
    auto commit = [&](char* dst, const uint4 (&st)[STG]) {
#pragma unroll
        for (int it = 0; it < STG; ++it) {
            const int c = it * 64 + lane, r = c / CH, cc = c % CH;
            *reinterpret_cast<uint4*>(dst + r * ROWB + ((cc ^ (r & SWZ)) * 16)) = st[it];
        }
    };

    float areg[D / 2];
    uint4 st[STG];
    {
        const int r = r0 + j;
        const bool valid = r < L;
        const char* qrow = reinterpret_cast<const char*>(a.q) +
            ((int64_t)b * a.q_stride_b + (int64_t)h * a.q_stride_h + (int64_t)(valid ? r : 0) * a.q_stride_l) * ES;
        issue(0, st);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const uint4 v = valid ? *reinterpret_cast<const uint4*>(qrow + c * 16) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < PAIRS; ++s) areg[c * PAIRS + s] = pick2<DT>(v, s, kh);
        }
    }
    commit(buf, st);
    float rmax[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) rmax[e] = -__builtin_inff();
    raw* const Sh = reinterpret_cast<raw*>(a.S) + (int64_t)hb * L * L;
    int cur = 0;
    for (int tile = 0; tile < n_t; ++tile) {
        if (tile + 1 < n_t) issue(tile + 1, st);
        __builtin_amdgcn_wave_barrier();
        const char* krow = buf + cur * (32 * ROWB) + j * ROWB;
        const int key = tile * 32 + j;
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const uint4 kv = *reinterpret_cast<const uint4*>(krow + ((c ^ (j & SWZ)) * 16));
#pragma unroll
            for (int s = 0; s < PAIRS; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[c * PAIRS + s], pick2<DT>(kv, s, kh), acc, 0, 0, 0);
        }
        const bool tailk = tile * 32 + 32 > L - W, tailr = r0 + 32 > L - W;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = r0 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            float v = rnd<DT>(acc[e]);
            v = rnd<DT>(v / sqrt_d);
            if (tailk && tailr) {
                if (r >= L - W && key >= L - W && (key - (L - W)) > (r - (L - W))) v = rnd<DT>(v + Dt<DT>::finfo_min());
            }
            if (r < L && key < L) {
                Sh[(int64_t)r * L + key] = Dt<DT>::st(v);
                rmax[e] = v > rmax[e] ? v : rmax[e];
            }
        }
        if (tile + 1 < n_t) commit(buf + (cur ^ 1) * (32 * ROWB), st);
        cur ^= 1;
    }
    // exact row maxima: reduce each register over the 32 key lanes of its half-wave
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float m = half_xor_max(rmax[e]);
        const int r = r0 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (j == 0 && r < L) a.rowmax[(int64_t)hb * L + r] = m;
    }
}

// grid = (ceil(L / 16), bsz * n_q_heads), block = 256: wave handles 4 rows, 16 lanes (chains) per row.
template <int DT>
__global__ __launch_bounds__(256) void h2o_rowsum_kernel(const H2OArgs a) {
    typedef typename Dt<DT>::raw raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hb = blockIdx.y, L = a.q_len;
    const int r = (blockIdx.x * 4 + wave) * 4 + (lane >> 4);
    const int l = lane & 15;
    const bool valid = r < L;
    const raw* row = reinterpret_cast<const raw*>(a.S) + ((int64_t)hb * L + (valid ? r : 0)) * L;
    const float m = valid ? a.rowmax[(int64_t)hb * L + r] : 0.0f;
    float s;
    if (L < 16) {                                         // vec::reduce_all "slow path": plain left-to-right sum
        s = 0.0f;
        for (int i = 0; i < L; ++i) { const float e = exp_u20(Dt<DT>::ld(row[i]) - m); s = i == 0 ? e : s + e; }
    } else {
        float acc = exp_u20(Dt<DT>::ld(row[l]) - m);
        const int full = L - (L % 16);
        int d = 16;
        for (; d < full; d += 16) acc = acc + exp_u20(Dt<DT>::ld(row[d + l]) - m);
        if (l < L - d) acc = acc + exp_u20(Dt<DT>::ld(row[d + l]) - m);
        // xor butterfly 8,4,2,1 inside each 16-lane group (fp32 add is commutative: both partners get the same bits)
        acc = acc + xor_lane<8>(acc);
        acc = acc + xor_lane<4>(acc);
        acc = acc + xor_lane<2>(acc);
        acc = acc + xor_lane<1>(acc);
        s = acc;
    }
    if (valid && l == 0) a.rinv[(int64_t)hb * L + r] = 1.0f / s;
}

// grid = (ceil(n / 256), bsz * n_q_heads), block = 256: one thread per key column.
template <int DT>
__global__ __launch_bounds__(256) void h2o_colsum_kernel(const H2OArgs a) {
    typedef typename Dt<DT>::raw raw;
    const int hb = blockIdx.y, L = a.q_len, n = L - a.window;
    const int jcol = blockIdx.x * 256 + threadIdx.x;
    if (jcol >= n) return;
    const raw* S = reinterpret_cast<const raw*>(a.S) + (int64_t)hb * L * L + jcol;
    const float* m = a.rowmax + (int64_t)hb * L;
    const float* ri = a.rinv + (int64_t)hb * L;
    CascadeSum cs;
    cs.init(L);
    for (int r = 0; r < L; ++r) {
        const float e = exp_u20(Dt<DT>::ld(S[(int64_t)r * L]) - m[r]);
        cs.add(rnd<DT>(e * ri[r]));
    }
    reinterpret_cast<raw*>(a.scores)[(int64_t)hb * n + jcol] = Dt<DT>::st(rnd<DT>(cs.result()));
}

template <int DT, int D>
static int launch_h2o_t(const H2OArgs& a, hipStream_t st) {
    constexpr int ES = Dt<DT>::esize;
    const size_t lds = (size_t)4 * 2 * 32 * D * ES;
    static LdsCache lds_cache = {};
    if (ensure_lds(reinterpret_cast<const void*>(&h2o_logits_kernel<DT, D>), lds, lds_cache) != 0) return KVC_ERR_HIP;
    const int L = a.q_len, n = L - a.window, heads = a.bsz * a.n_q_heads;
    const int row_tiles = (L + 31) / 32;
    hipLaunchKernelGGL((h2o_logits_kernel<DT, D>), dim3((unsigned)((row_tiles + 3) / 4), (unsigned)heads), dim3(256), lds, st, a);
    hipLaunchKernelGGL((h2o_rowsum_kernel<DT>), dim3((unsigned)((L + 15) / 16), (unsigned)heads), dim3(256), 0, st, a);
    hipLaunchKernelGGL((h2o_colsum_kernel<DT>), dim3((unsigned)((n + 255) / 256), (unsigned)heads), dim3(256), 0, st, a);
    return 0;
}

int launch_h2o_scores(const H2OArgs& a, int dtype, int head_dim, hipStream_t st) {
#define KVC_CASE(DT_, D_) if (dtype == DT_ && head_dim == D_) return launch_h2o_t<DT_, D_>(a, st)
    KVC_CASE(KVC_BF16, 128); KVC_CASE(KVC_BF16, 64);
    KVC_CASE(KVC_FP16, 128); KVC_CASE(KVC_FP16, 64);
    KVC_CASE(KVC_FP32, 128); KVC_CASE(KVC_FP32, 64);
#undef KVC_CASE
    return KVC_ERR_UNSUPPORTED;
}

}  // namespace kvc
