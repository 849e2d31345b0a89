// kvc_oracle.cpp — CPU restatement of the KVCache-Factory post-prefill scoring+eviction path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product path (kvcache_factory_amd/) may import,
// link or call this file; it is the checker used by tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py.  It is written in C++ (not C) on purpose: torch-CPU `topk`
// is libstdc++ std::partial_sort / std::nth_element + std::sort on pair<value,index>, and the
// only way to be bit-identical in tie order is to call the same library algorithms.
//
// What it restates (reference = /root/reference @ 2025-02-18, file:line):
//   SnapKVCluster.update_kv          pyramidkv/pyramidkv_utils.py:306-347
//   PyramidKVCluster.update_kv       pyramidkv/pyramidkv_utils.py:197-283 (budget schedule :205-215)
//   H2OKVCluster.update_kv           pyramidkv/pyramidkv_utils.py:533-575
//   StreamingLLMKVCluster.update_kv  pyramidkv/pyramidkv_utils.py:595-620
// The arithmetic below the Python (torch 2.10 CPU kernels: GEMM, div, softmax, sum, pool1d,
// topk, gather) is third-party and not in /root/reference; its rounding points are restated
// here stage by stage and pinned by tests/golden/* (generated from the imported reference by
// oracle/gen_golden.py) — see tests/test_oracle_golden.py.
//
// Arithmetic "modes" (the oracle has to describe two machines):
//   dot_mode  KVCO_DOT_CHAIN   : q·k as a sequential fp32 fmaf chain over d ascending — this is
//                                bit-for-bit what the gfx950 f32-input MFMA computes, i.e. the
//                                GPU product's defined arithmetic.
//             KVCO_DOT_F64     : q·k accumulated in fp64 then rounded to fp32 (an order-free
//                                "ideal" used to quantify 1-ulp flips vs torch's opaque GEMM).
//   sum_mode  KVCO_SUM_TORCH16 : softmax row sum exactly as torch's AVX512 vec::reduce_all
//                                (16 strided chains + xor butterfly 8,4,2,1) — also the GPU
//                                product's order since round 2 (round 1's own 256-key-chunk order
//                                put 126 instead of 20 of 1 023 744 pooled scores off the reference
//                                at L = 32000 and is gone).
//   tie_mode  KVCO_TIES_TORCH  : libstdc++ partial_sort / nth_element+sort (== torch-CPU topk).
//             KVCO_TIES_CANON  : value descending, index ascending.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>

#if defined(_OPENMP)
#include <omp.h>
#endif

#define KVCO_API extern "C" __attribute__((visibility("default")))

enum { KVCO_BF16 = 0, KVCO_FP16 = 1, KVCO_FP32 = 2 };
enum { KVCO_POOL_NONE = 0, KVCO_POOL_AVG = 1, KVCO_POOL_MAX = 2 };
enum { KVCO_DOT_CHAIN = 0, KVCO_DOT_F64 = 1 };
enum { KVCO_SUM_TORCH16 = 0 };
enum { KVCO_TIES_TORCH = 0, KVCO_TIES_CANON = 1 };

// ----------------------------------------------------------------------------------------
// dtype helpers (raw storage <-> float, round-to-nearest-even, NaN kept NaN)
// ----------------------------------------------------------------------------------------
static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

static inline float bf16_to_f32(uint16_t h) { return u2f((uint32_t)h << 16); }
static inline uint16_t f32_to_bf16(float f) {
    uint32_t u = f2u(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t em = h & 0x7fffu;
    if (em >= 0x7c00u) return u2f(sign | 0x7f800000u | ((em & 0x3ffu) << 13));   // inf / nan
    if (em >= 0x0400u) return u2f(sign | ((em << 13) + ((127u - 15u) << 23)));   // normal
    // zero / subnormal: value = em * 2^-24 (exact in fp32)
    const float mag = (float)em * u2f((127u - 24u) << 23);
    return u2f(sign | f2u(mag));
}
static inline uint16_t f32_to_f16(float f) {
    uint32_t u = f2u(f);
    const uint32_t sign = u & 0x80000000u;
    u ^= sign;
    uint16_t o;
    if (u >= ((127u + 16u) << 23)) {                 // |f| >= 65536, inf or nan
        o = (u > 0x7f800000u) ? 0x7e00u : 0x7c00u;
    } else if (u < (113u << 23)) {                   // result is subnormal half or zero
        const uint32_t magic = ((127u - 15u) + (23u - 10u) + 1u) << 23;
        const float t = u2f(u) + u2f(magic);         // the fp32 adder does the RNE for us
        o = (uint16_t)(f2u(t) - magic);
    } else {                                          // normal; may carry into inf (>= 65520)
        const uint32_t odd = (u >> 13) & 1u;
        u += ((uint32_t)(15 - 127) << 23) + 0xfffu;
        u += odd;
        o = (uint16_t)(u >> 13);
    }
    return (uint16_t)(o | (sign >> 16));
}

struct DtBf16 {
    typedef uint16_t raw; static const int code = KVCO_BF16;
    static inline float ld(raw r) { return bf16_to_f32(r); }
    static inline raw st(float f) { return f32_to_bf16(f); }
    static inline float finfo_min() { return u2f(0xff7f0000u); }       // -3.3895314e38
};
struct DtFp16 {
    typedef uint16_t raw; static const int code = KVCO_FP16;
    static inline float ld(raw r) { return f16_to_f32(r); }
    static inline raw st(float f) { return f32_to_f16(f); }
    static inline float finfo_min() { return -65504.0f; }
};
struct DtFp32 {
    typedef float raw; static const int code = KVCO_FP32;
    static inline float ld(raw r) { return r; }
    static inline raw st(float f) { return f; }
    static inline float finfo_min() { return -std::numeric_limits<float>::max(); }
};
template <class DT> static inline float rnd(float f) { return DT::ld(DT::st(f)); }

// ----------------------------------------------------------------------------------------
// exp: scalar restatement of torch 2.10 Vectorized<float>::exp_u20 (AVX512/AVX2 share it):
// clamp, n = floor(x*log2e + 0.5), r = x - n*ln2, degree-5 Horner with FMAs, scale by 2^(n-1)
// then by 2.  Every step is a single IEEE fp32 operation, so the gfx950 kernel can (and does)
// execute the identical sequence.
// ----------------------------------------------------------------------------------------
static inline float exp_u20(float x) {
    const float ln_flt_min = u2f(0xc2aeac50u), ln_flt_max = u2f(0x42b17218u);
    const float log2ef = u2f(0x3fb8aa3bu), ln2f = u2f(0x3f317218u);
    const float c1 = 0.999999701f, c2 = 0.499991506f, c3 = 0.166676521f, c4 = 0.0418978221f,
                c5 = 0.00828929059f;
    const bool below = x < ln_flt_min;
    float s = (x < ln_flt_max) ? x : ln_flt_max;      // _mm512_min_ps(values, max)
    s = (s > ln_flt_min) ? s : ln_flt_min;            // _mm512_max_ps(src, min)
    float fx = std::fmaf(s, log2ef, 0.5f);
    fx = std::floor(fx);
    const float r = std::fmaf(-fx, ln2f, s);
    float p = std::fmaf(r, c5, c4);
    p = std::fmaf(r, p, c3);
    p = std::fmaf(r, p, c2);
    p = std::fmaf(r, p, c1);
    p = std::fmaf(r, p, 1.0f);
    const int n1 = (int)(fx - 1.0f);
    const float two_n = below ? 0.0f : u2f((uint32_t)(n1 + 127) << 23);
    p = p * two_n;
    p = p * 2.0f;
    return p;
}

// ----------------------------------------------------------------------------------------
// Row sum of fp32 values in torch's fixed order.
// ----------------------------------------------------------------------------------------
static float sum_torch16(const float* x, int64_t n) {
    // vec::reduce_all<float>(+) with Vec::size()==16  (functional_base.h: reduce_all,
    // VecReduceAllSIMD<float> AVX512)
    if (n < 16) {  // "slow path": acc = x[0] + x[1] + ... sequentially in lane 0
        float a = n > 0 ? x[0] : 0.0f;
        for (int64_t i = 1; i < n; ++i) a = a + x[i];
        return a;
    }
    float acc[16];
    for (int l = 0; l < 16; ++l) acc[l] = x[l];
    int64_t d = 16;
    const int64_t full = n - (n % 16);
    for (; d < full; d += 16)
        for (int l = 0; l < 16; ++l) acc[l] = acc[l] + x[d + l];
    for (int l = 0; l < (int)(n - d); ++l) acc[l] = acc[l] + x[d + l];
    for (int step = 8; step >= 1; step >>= 1) {
        float nxt[16];
        for (int l = 0; l < 16; ++l) nxt[l] = acc[l] + acc[l ^ step];
        std::memcpy(acc, nxt, sizeof(acc));
    }
    return acc[0];
}
// ----------------------------------------------------------------------------------------
// torch's outer-dimension sum (SumKernel.cpp cascade "multi_row_sum"): used by
// attn_weights[..., -W:, :-W].sum(dim=-2) (pyramidkv_utils.py:327) and by H2O's sum over all
// L rows (:554).  One instance per output column; rows are fed in order.
// ----------------------------------------------------------------------------------------
struct CascadeSum {
    float acc[4];
    int64_t i, level_step, level_mask; int level_power; int64_t in_step;
    void init(int64_t size) {
        int ceil_log2 = 0; while (((int64_t)1 << ceil_log2) < size) ++ceil_log2;
        level_power = std::max(4, ceil_log2 / 4);
        level_step = (int64_t)1 << level_power; level_mask = level_step - 1;
        acc[0] = acc[1] = acc[2] = acc[3] = 0.0f; i = 0; in_step = 0; total = size;
    }
    int64_t total;
    inline void add(float v) {
        // rows beyond the last full level_step group are the scalar tail loop: same acc[0] += v
        acc[0] = acc[0] + v; ++i; ++in_step;
        if (in_step == level_step && i <= total - (total % level_step)) {
            in_step = 0;
            for (int j = 1; j < 4; ++j) {
                acc[j] = acc[j] + acc[j - 1]; acc[j - 1] = 0.0f;
                const int64_t mask = level_mask << (j * level_power);
                if ((i & mask) != 0) break;
            }
        }
    }
    inline float result() const { return ((acc[0] + acc[1]) + acc[2]) + acc[3]; }
};

// ----------------------------------------------------------------------------------------
// A1+A2: one logit, with the reference's three rounding points
//   matmul -> dtype ; / sqrt(D) in fp32 -> dtype ; (+ finfo.min mask in fp32 -> dtype)
// ----------------------------------------------------------------------------------------
template <class DT>
static inline float dot_round(const typename DT::raw* q, const typename DT::raw* k, int D, int dot_mode) {
    if (dot_mode == KVCO_DOT_F64) {
        double a = 0.0;
        for (int d = 0; d < D; ++d) a += (double)DT::ld(q[d]) * (double)DT::ld(k[d]);
        return rnd<DT>((float)a);
    }
    float a = 0.0f;
    for (int d = 0; d < D; ++d) a = std::fmaf(DT::ld(q[d]), DT::ld(k[d]), a);
    return rnd<DT>(a);
}
template <class DT>
static inline float scale_mask(float logit, float sqrt_d, bool masked) {
    float x = rnd<DT>(logit / sqrt_d);
    if (masked) x = rnd<DT>(x + DT::finfo_min());
    return x;
}

// softmax of one fp32 row (already widened from dtype) -> probabilities rounded to dtype.
// torch: softmax(x, -1, dtype=float32).to(dtype)  (pyramidkv_utils.py:326):
//   m = max; e = exp_u20(x - m); s = sum(e); r = 1/s; p = e*r; round to dtype.
template <class DT>
static void softmax_row(const float* x, int64_t L, int sum_mode, float* e_tmp, float* p_out) {
    float m = x[0];
    for (int64_t j = 1; j < L; ++j) m = (x[j] > m) ? x[j] : m;
    for (int64_t j = 0; j < L; ++j) e_tmp[j] = exp_u20(x[j] - m);
    (void)sum_mode;
    const float s = sum_torch16(e_tmp, L);
    const float r = 1.0f / s;
    for (int64_t j = 0; j < L; ++j) p_out[j] = rnd<DT>(e_tmp[j] * r);
}

// A5 pooling on a dtype row (values already exactly representable in dtype).
template <class DT>
static void pool_row(const float* s, int64_t n, int kernel, int pooling, float* out) {
    if (pooling == KVCO_POOL_NONE) { std::memcpy(out, s, sizeof(float) * n); return; }
    const int pad = kernel / 2;
    for (int64_t j = 0; j < n; ++j) {
        const int64_t lo = std::max<int64_t>(0, j - pad), hi = std::min<int64_t>(n, j - pad + kernel);
        if (pooling == KVCO_POOL_MAX) {
            float m = -std::numeric_limits<float>::infinity();
            for (int64_t i = lo; i < hi; ++i) m = (s[i] > m) ? s[i] : m;
            out[j] = m;
        } else {  // avg, count_include_pad=True: fp32 sum of the in-range taps / kernel
            float a = 0.0f;
            for (int64_t i = lo; i < hi; ++i) a = a + s[i];
            out[j] = rnd<DT>(a / (float)kernel);
        }
    }
}

// A7 top-k of one row.
static void topk_row(const float* c, int64_t n, int64_t k, int tie_mode, int64_t* idx, float* val) {
    typedef std::pair<float, int64_t> elem;
    std::vector<elem> q((size_t)n);
    for (int64_t j = 0; j < n; ++j) q[(size_t)j] = elem(c[j], j);
    if (tie_mode == KVCO_TIES_CANON) {
        std::stable_sort(q.begin(), q.end(), [](const elem& a, const elem& b) { return a.first > b.first; });
    } else {
        // aten/src/ATen/native/cpu/SortingKernel / TopKImpl.h topk_impl_loop, largest=True, sorted=True
        auto comp = [](const elem& x, const elem& y) -> bool {
            return ((std::isnan(x.first) && !std::isnan(y.first)) || (x.first > y.first));
        };
        if (k * 64 <= n) {
            std::partial_sort(q.begin(), q.begin() + k, q.end(), comp);
        } else {
            std::nth_element(q.begin(), q.begin() + (k - 1), q.end(), comp);
            std::sort(q.begin(), q.begin() + (k - 1), comp);
        }
    }
    for (int64_t t = 0; t < k; ++t) { idx[t] = q[(size_t)t].second; val[t] = q[(size_t)t].first; }
}

// ----------------------------------------------------------------------------------------
// Parameter block shared by the entry points (host pointers, element strides).
// ----------------------------------------------------------------------------------------
struct kvco_params {
    int32_t dtype;        // KVCO_BF16 / FP16 / FP32
    int32_t n_q_heads;    // H_q : heads of q and of the outputs
    int32_t n_kv_heads;   // heads actually present in k/v (H_q as the reference passes them, or H_kv)
    int32_t q_len;        // L
    int32_t head_dim;     // D
    int32_t window;       // W
    int32_t k;            // number of prefix tokens to keep (already resolved per layer)
    int32_t kernel_size;  // pooling kernel (odd)
    int32_t pooling;      // KVCO_POOL_*
    int32_t full_rows;    // 0: SnapKV/PyramidKV (last W query rows, window SUM) ; 1: H2O (all L rows) ;
                          // 2: AdaKV/HeadKV (last W rows, window MEAN: calcul_attn_sore, pyramidkv_utils.py:647-672)
    int32_t dot_mode, sum_mode, tie_mode;
    int32_t n_threads;    // OpenMP threads (<=0: runtime default)
    int64_t q_stride_h, q_stride_l;   // q[h][l][d] element strides (d contiguous)
    int64_t k_stride_h, k_stride_l;
    int64_t v_stride_h, v_stride_l;
};

template <class DT>
static int scores_impl(const kvco_params& P, const void* q_, const void* k_,
                       void* logits_out /*[H][R][L] dtype or null*/, void* probs_out /*same or null*/,
                       void* wsum_out /*[H][n] dtype or null*/, void* scores_out /*[H][n] dtype*/) {
    typedef typename DT::raw raw;
    const raw* q = (const raw*)q_; const raw* kk = (const raw*)k_;
    const int H = P.n_q_heads, G = H / P.n_kv_heads, W = P.window, D = P.head_dim;
    const int64_t L = P.q_len, n = L - W;
    const int64_t R = P.full_rows == 1 ? L : W;       // query rows scored
    const float sqrt_d = (float)std::sqrt((double)D);
#if defined(_OPENMP)
    if (P.n_threads > 0) omp_set_num_threads(P.n_threads);
#endif
    int err = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int h = 0; h < H; ++h) {
        const raw* kh = kk + (int64_t)(h / G) * P.k_stride_h;
        const raw* qh = q + (int64_t)h * P.q_stride_h;
        std::vector<float> x((size_t)L), e((size_t)L), p((size_t)L), s((size_t)n), c((size_t)n);
        std::vector<CascadeSum> col((size_t)n);
        for (int64_t j = 0; j < n; ++j) col[(size_t)j].init(R);
        for (int64_t r = 0; r < R; ++r) {
            const int64_t qrow = L - R + r;                    // absolute query position
            const raw* qr = qh + qrow * P.q_stride_l;
            const int64_t wrow = qrow - (L - W);               // row inside the W x W mask block (or <0)
            for (int64_t j = 0; j < L; ++j) {
                const float lg = dot_round<DT>(qr, kh + j * P.k_stride_l, D, P.dot_mode);
                const bool masked = (wrow >= 0) && (j >= L - W) && ((j - (L - W)) > wrow);
                x[(size_t)j] = scale_mask<DT>(lg, sqrt_d, masked);
            }
            if (logits_out) { raw* o = (raw*)logits_out + ((int64_t)h * R + r) * L;
                              for (int64_t j = 0; j < L; ++j) o[j] = DT::st(x[(size_t)j]); }
            softmax_row<DT>(x.data(), L, P.sum_mode, e.data(), p.data());
            if (probs_out) { raw* o = (raw*)probs_out + ((int64_t)h * R + r) * L;
                             for (int64_t j = 0; j < L; ++j) o[j] = DT::st(p[(size_t)j]); }
            for (int64_t j = 0; j < n; ++j) col[(size_t)j].add(p[(size_t)j]);
        }
        // .sum(dim=-2): fp32 cascade, rounded once.  .mean(dim=-2) (AdaKV / HeadKV): torch computes reduced-precision means as
        // cast_fp32 -> sum -> div_(W) -> cast (ReduceOps.cpp mean_out: "for accuracy reasons ..."), one rounding as well
        for (int64_t j = 0; j < n; ++j)
            s[(size_t)j] = P.full_rows == 2 ? rnd<DT>(col[(size_t)j].result() / (float)W) : rnd<DT>(col[(size_t)j].result());
        if (wsum_out) { raw* o = (raw*)wsum_out + (int64_t)h * n;
                        for (int64_t j = 0; j < n; ++j) o[j] = DT::st(s[(size_t)j]); }
        pool_row<DT>(s.data(), n, P.kernel_size, P.full_rows == 1 ? KVCO_POOL_NONE : P.pooling, c.data());
        raw* so = (raw*)scores_out + (int64_t)h * n;
        for (int64_t j = 0; j < n; ++j) so[j] = DT::st(c[(size_t)j]);
    }
    return err;
}

template <class DT>
static int topk_impl(const kvco_params& P, const void* scores_, int64_t* idx_out, void* val_out) {
    typedef typename DT::raw raw;
    const int H = P.n_q_heads; const int64_t n = P.q_len - P.window, k = P.k;
#if defined(_OPENMP)
    if (P.n_threads > 0) omp_set_num_threads(P.n_threads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int h = 0; h < H; ++h) {
        const raw* s = (const raw*)scores_ + (int64_t)h * n;
        std::vector<float> c((size_t)n), v((size_t)k);
        for (int64_t j = 0; j < n; ++j) c[(size_t)j] = DT::ld(s[j]);
        topk_row(c.data(), n, k, P.tie_mode, idx_out + (int64_t)h * k, v.data());
        if (val_out) { raw* o = (raw*)val_out + (int64_t)h * k; for (int64_t t = 0; t < k; ++t) o[t] = DT::st(v[(size_t)t]); }
    }
    return 0;
}

// A8: K' = cat(K[:, :, :-W].gather(2, idx), K[:, :, -W:])  (pyramidkv_utils.py:341-346) — pure copy.
static int gather_impl(const kvco_params& P, size_t esize, const void* src_, int64_t stride_h, int64_t stride_l,
                       const int64_t* idx, void* out_) {
    const int H = P.n_q_heads, G = H / P.n_kv_heads, W = P.window, D = P.head_dim;
    const int64_t L = P.q_len, k = P.k, cap = k + W;
    const char* src = (const char*)src_; char* out = (char*)out_;
    for (int h = 0; h < H; ++h) {
        const char* sh = src + (int64_t)(h / G) * stride_h * esize;
        char* oh = out + (int64_t)h * cap * D * esize;
        for (int64_t t = 0; t < k; ++t) {
            const int64_t j = idx[(int64_t)h * k + t];
            if (j < 0 || j >= L - W) return -3;
            std::memcpy(oh + t * D * esize, sh + j * stride_l * esize, (size_t)D * esize);
        }
        for (int w = 0; w < W; ++w)
            std::memcpy(oh + (k + w) * D * esize, sh + (L - W + w) * stride_l * esize, (size_t)D * esize);
    }
    return 0;
}

static int check(const kvco_params& P) {
    if (P.dtype < 0 || P.dtype > 2) return -1;
    if (P.n_q_heads <= 0 || P.n_kv_heads <= 0 || P.n_q_heads % P.n_kv_heads) return -1;
    if (P.window <= 0 || P.q_len <= P.window || P.head_dim <= 0) return -1;
    if (P.k < 0 || P.k > P.q_len - P.window) return -1;
    if (P.pooling != KVCO_POOL_NONE && (P.kernel_size <= 0 || (P.kernel_size % 2) == 0)) return -2;
    return 0;
}

KVCO_API int kvco_version(void) { return 1; }

// A1..A5 (+ H2O's A10 when full_rows=1).  Optional outputs may be null.
KVCO_API int kvco_scores(const kvco_params* P, const void* q, const void* k, void* logits_out,
                         void* probs_out, void* wsum_out, void* scores_out) {
    if (int e = check(*P)) return e;
    switch (P->dtype) {
        case KVCO_BF16: return scores_impl<DtBf16>(*P, q, k, logits_out, probs_out, wsum_out, scores_out);
        case KVCO_FP16: return scores_impl<DtFp16>(*P, q, k, logits_out, probs_out, wsum_out, scores_out);
        default:        return scores_impl<DtFp32>(*P, q, k, logits_out, probs_out, wsum_out, scores_out);
    }
}
// A7 on given pooled scores [H][n] (dtype) -> idx [H][k] int64 (+ values [H][k] dtype).
KVCO_API int kvco_topk(const kvco_params* P, const void* scores, int64_t* idx_out, void* val_out) {
    if (int e = check(*P)) return e;
    switch (P->dtype) {
        case KVCO_BF16: return topk_impl<DtBf16>(*P, scores, idx_out, val_out);
        case KVCO_FP16: return topk_impl<DtFp16>(*P, scores, idx_out, val_out);
        default:        return topk_impl<DtFp32>(*P, scores, idx_out, val_out);
    }
}
// A8 for one of K / V.
KVCO_API int kvco_gather(const kvco_params* P, const void* src, int64_t stride_h, int64_t stride_l,
                         const int64_t* idx, void* out) {
    if (int e = check(*P)) return e;
    return gather_impl(*P, P->dtype == KVCO_FP32 ? 4 : 2, src, stride_h, stride_l, idx, out);
}
// A0/A9/A10: scores -> top-k -> compact K and V.  scores/idx outputs are mandatory scratch.
KVCO_API int kvco_compress(const kvco_params* P, const void* q, const void* k, const void* v,
                           void* k_out, void* v_out, int64_t* idx_out, void* scores_out) {
    int e = kvco_scores(P, q, k, nullptr, nullptr, nullptr, scores_out);
    if (e) return e;
    e = kvco_topk(P, scores_out, idx_out, nullptr);
    if (e) return e;
    e = kvco_gather(P, k, P->k_stride_h, P->k_stride_l, idx_out, k_out);
    if (e) return e;
    return kvco_gather(P, v, P->v_stride_h, P->v_stride_l, idx_out, v_out);
}
// A11 StreamingLLM: idx = arange(k) (pyramidkv_utils.py:607-608) then A8.
KVCO_API int kvco_streaming(const kvco_params* P, const void* k, const void* v, void* k_out, void* v_out,
                            int64_t* idx_out) {
    if (int e = check(*P)) return e;
    for (int h = 0; h < P->n_q_heads; ++h)
        for (int64_t t = 0; t < P->k; ++t) idx_out[(int64_t)h * P->k + t] = t;
    int e = kvco_gather(P, k, P->k_stride_h, P->k_stride_l, idx_out, k_out);
    if (e) return e;
    return kvco_gather(P, v, P->v_stride_h, P->v_stride_l, idx_out, v_out);
}
// A9 PyramidKV per-layer budget (pyramidkv_utils.py:205-215).  Returns k for this layer, or -1 for
// the pass-through branch (q_len < cap, :218); the middle branch (:220) returns cap - W.
KVCO_API int64_t kvco_pyramid_k(int64_t cap, int64_t W, int64_t q_len, int64_t layer_idx, int64_t n_layers,
                                int64_t beta) {
    int64_t min_num = (cap - W) / beta;
    int64_t max_num = (cap - W) * 2 - min_num;
    if (max_num >= q_len - W) { max_num = q_len - W; min_num = (cap - W) * 2 - max_num; }
    // Python floor division (operands may be negative after the clamp)
    const int64_t num = max_num - min_num, den = n_layers - 1;
    int64_t steps = den != 0 ? num / den : 0;
    if (den != 0 && (num % den != 0) && ((num < 0) != (den < 0))) --steps;
    if (q_len < cap) return -1;
    if (q_len < (cap - W) * 2) return cap - W;
    return max_num - layer_idx * steps;
}

// ----------------------------------------------------------------------------------------
// AdaKV / HeadKV (pyramidkv_utils.py:622-757, :760-878) — per-head budgets over ONE scoring front-end
// (window MEAN + pooling), a ragged ("flattened") compressed cache [sum_h (cap_h + W), D].
// ----------------------------------------------------------------------------------------
// attn_score.sort(dim=-1, descending=True) (:702, :851): torch-CPU sort is libstdc++ std::sort on (value, index) with a
// value-only comparator (checked against torch in this container: identical indices, ties included).
template <class DT>
static void sort_desc_impl(const kvco_params& P, const void* scores_, int64_t* idx_out, void* val_out) {
    typedef typename DT::raw raw;
    const int H = P.n_q_heads; const int64_t n = P.q_len - P.window;
#pragma omp parallel for schedule(dynamic, 1)
    for (int h = 0; h < H; ++h) {
        const raw* s = (const raw*)scores_ + (int64_t)h * n;
        std::vector<std::pair<float, int64_t>> a((size_t)n);
        for (int64_t j = 0; j < n; ++j) a[(size_t)j] = std::make_pair(DT::ld(s[j]), j);
        std::sort(a.begin(), a.end(), [](const std::pair<float, int64_t>& x, const std::pair<float, int64_t>& y) {
            return ((std::isnan(x.first) && !std::isnan(y.first)) || (x.first > y.first)); });
        for (int64_t j = 0; j < n; ++j) {
            idx_out[(int64_t)h * n + j] = a[(size_t)j].second;
            if (val_out) ((raw*)val_out)[(int64_t)h * n + j] = DT::st(a[(size_t)j].first);
        }
    }
}
KVCO_API int kvco_sort_desc(const kvco_params* P, const void* scores, int64_t* idx_out, void* val_out) {
    if (int e = check(*P)) return e;
    switch (P->dtype) {
        case KVCO_BF16: sort_desc_impl<DtBf16>(*P, scores, idx_out, val_out); break;
        case KVCO_FP16: sort_desc_impl<DtFp16>(*P, scores, idx_out, val_out); break;
        default:        sort_desc_impl<DtFp32>(*P, scores, idx_out, val_out); break;
    }
    return 0;
}
// AdaKV's per-head capacities (:701-717) from the SORTED scores [H][n] (dtype):
//   ratio = sorted[:, :base].sum(-1) / sorted.sum(-1)  (normalize)  ->  adaptive = sorted * ratio  (each op rounds to dtype)
//   topk(H * base) over the flattened [H * n] values -> count per head -> round(count * (1 - floor) + floor_capacity)
// Sums: fp32 accumulation in sorted order, rounded once to dtype — torch's vectorised inner sum adds in another order;
// the difference is ~1e-7 relative before a rounding to 8 / 11 mantissa bits (pinned by the fixtures).
template <class DT>
static void adakv_caps_impl(const kvco_params& P, const void* sorted_, int64_t base, double floor_ratio, int normalize, int32_t* caps) {
    typedef typename DT::raw raw;
    const int H = P.n_q_heads; const int64_t n = P.q_len - P.window;
    const raw* sv = (const raw*)sorted_;
    std::vector<float> flat((size_t)H * n);
    for (int h = 0; h < H; ++h) {
        float ratio = 1.0f;
        if (normalize) {
            float a = 0.0f, b = 0.0f;
            for (int64_t j = 0; j < n; ++j) { const float v = DT::ld(sv[(int64_t)h * n + j]); b += v; if (j < base) a += v; }
            ratio = rnd<DT>(rnd<DT>(a) / rnd<DT>(b));
        }
        for (int64_t j = 0; j < n; ++j) {
            const float v = DT::ld(sv[(int64_t)h * n + j]);
            flat[(size_t)h * n + j] = normalize ? rnd<DT>(v * ratio) : v;
        }
    }
    const int64_t k = (int64_t)H * base;
    std::vector<int64_t> idx((size_t)k);
    std::vector<float> val((size_t)k);
    topk_row(flat.data(), (int64_t)H * n, k, KVCO_TIES_TORCH, idx.data(), val.data());
    std::vector<int64_t> cnt((size_t)H, 0);
    for (int64_t t = 0; t < k; ++t) cnt[(size_t)(idx[(size_t)t] / n)]++;
    const int64_t floor_capacity = (int64_t)((double)base * floor_ratio);                 // int(base_capacity * floor_ratio)  (:632)
    const float mix = (float)(1.0 - floor_ratio);                                          // python double -> the fp32 op's scalar
    for (int h = 0; h < H; ++h)                                                            // torch.round: half to even
        caps[h] = (int32_t)std::nearbyintf((float)cnt[(size_t)h] * mix + (float)floor_capacity);
}
KVCO_API int kvco_adakv_caps(const kvco_params* P, const void* sorted_vals, int64_t base, double floor_ratio, int normalize, int32_t* caps) {
    if (int e = check(*P)) return e;
    switch (P->dtype) {
        case KVCO_BF16: adakv_caps_impl<DtBf16>(*P, sorted_vals, base, floor_ratio, normalize, caps); break;
        case KVCO_FP16: adakv_caps_impl<DtFp16>(*P, sorted_vals, base, floor_ratio, normalize, caps); break;
        default:        adakv_caps_impl<DtFp32>(*P, sorted_vals, base, floor_ratio, normalize, caps); break;
    }
    return 0;
}
// The flattened cache of one tensor (:721-757): for every head, its first caps[h] sorted indices' rows, then the window rows.
KVCO_API int kvco_ragged_gather(const kvco_params* P, const void* src_, int64_t stride_h, int64_t stride_l, const int64_t* sorted_idx,
                                const int32_t* caps, void* out_) {
    if (int e = check(*P)) return e;
    const size_t esize = P->dtype == KVCO_FP32 ? 4 : 2;
    const int H = P->n_q_heads, G = H / P->n_kv_heads, W = P->window, D = P->head_dim;
    const int64_t L = P->q_len, n = L - W;
    const char* src = (const char*)src_; char* out = (char*)out_;
    int64_t row = 0;
    for (int h = 0; h < H; ++h) {
        const char* sh = src + (int64_t)(h / G) * stride_h * esize;
        for (int64_t t = 0; t < caps[h]; ++t, ++row)
            std::memcpy(out + row * D * esize, sh + sorted_idx[(int64_t)h * n + t] * stride_l * esize, (size_t)D * esize);
        for (int w = 0; w < W; ++w, ++row)
            std::memcpy(out + row * D * esize, sh + (L - W + w) * stride_l * esize, (size_t)D * esize);
    }
    return 0;
}

// ----------------------------------------------------------------------------------------
// L2NormCluster.update_kv (pyramidkv_utils.py:394-429) — keep the max_capacity_prompt tokens of every head
// whose KEY has the smallest L2 norm, in ascending norm order; no window, no query.
//   token_norms = torch.norm(key_states, p=2, dim=-1)         (:419)
//   sorted_indices = token_norms.argsort(dim=-1)              (:420)   ascending, not stable
//   K', V' = gather(sorted_indices)[:, :, :max_capacity_prompt]  (:423-427)
// torch-CPU norm over the contiguous last dimension (aten/native/cpu/ReduceOpsKernel.cpp, norm_kernel_tensor_iterator_impl,
// p = 2, reduced floating types): 8 fp32 accumulators, element d goes to accumulator d % 8 (acc += x * x), the accumulators
// are added one after the other from 0 to 7, sqrt in fp32, ONE rounding to the dtype.  Checked against torch.norm in this
// container on 2 * 10^6 random bf16 rows: identical (a plain d-ascending sum differs on ~1 row in 10^5).  fp32 inputs:
// torch's fp32 path fuses the multiply-add and this restatement does not — about 1 % of fp32 norms differ in the last
// place ("parity unpinned" for fp32 norms; the fp32 fixtures are compared on indices of heads without such a flip).
// argsort = torch-CPU sort, ascending: libstdc++ std::sort on (value, index) with the value-only comparator
// (!isnan(a) && isnan(b)) || a < b for the 16-bit dtypes (same routine as AdaKV's descending sort above).
// ----------------------------------------------------------------------------------------
template <class DT>
static void l2norm_impl(const kvco_params& P, const void* k_, const void* v_, void* norms_out, int64_t* idx_out, void* k_out, void* v_out) {
    typedef typename DT::raw raw;
    const int H = P.n_q_heads, G = H / P.n_kv_heads, D = P.head_dim;
    const int64_t L = P.q_len, cap = P.k;
    const size_t es = sizeof(raw);
#pragma omp parallel for schedule(dynamic, 1)
    for (int h = 0; h < H; ++h) {
        const raw* kh = (const raw*)k_ + (int64_t)(h / G) * P.k_stride_h;
        std::vector<std::pair<float, int64_t>> a((size_t)L);
        for (int64_t l = 0; l < L; ++l) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const raw* row = kh + l * P.k_stride_l;
            for (int d = 0; d < D; ++d) { const float x = DT::ld(row[d]); acc[d & 7] = acc[d & 7] + x * x; }
            float tot = acc[0];
            for (int j = 1; j < 8; ++j) tot = tot + acc[j];
            const float nrm = rnd<DT>(std::sqrt(tot));
            a[(size_t)l] = std::make_pair(nrm, l);
            if (norms_out) ((raw*)norms_out)[(int64_t)h * L + l] = DT::st(nrm);
        }
        std::sort(a.begin(), a.end(), [](const std::pair<float, int64_t>& x, const std::pair<float, int64_t>& y) {
            return ((!std::isnan(x.first) && std::isnan(y.first)) || (x.first < y.first)); });
        const raw* vh = (const raw*)v_ + (int64_t)(h / G) * P.v_stride_h;
        for (int64_t t = 0; t < cap; ++t) {
            const int64_t j = a[(size_t)t].second;
            if (idx_out) idx_out[(int64_t)h * cap + t] = j;
            if (k_out) std::memcpy((char*)k_out + ((int64_t)h * cap + t) * D * es, kh + j * P.k_stride_l, (size_t)D * es);
            if (v_out) std::memcpy((char*)v_out + ((int64_t)h * cap + t) * D * es, vh + j * P.v_stride_l, (size_t)D * es);
        }
    }
}
// P->k = rows kept per head (max_capacity_prompt, <= q_len); P->window and the scoring fields are ignored.
KVCO_API int kvco_l2norm(const kvco_params* P, const void* k, const void* v, void* norms_out, int64_t* idx_out, void* k_out, void* v_out) {
    if (P->dtype < 0 || P->dtype > 2 || P->n_q_heads <= 0 || P->n_kv_heads <= 0 || P->n_q_heads % P->n_kv_heads) return -1;
    if (P->q_len <= 0 || P->head_dim <= 0 || P->k < 0 || P->k > P->q_len) return -1;
    switch (P->dtype) {
        case KVCO_BF16: l2norm_impl<DtBf16>(*P, k, v, norms_out, idx_out, k_out, v_out); break;
        case KVCO_FP16: l2norm_impl<DtFp16>(*P, k, v, norms_out, idx_out, k_out, v_out); break;
        default:        l2norm_impl<DtFp32>(*P, k, v, norms_out, idx_out, k_out, v_out); break;
    }
    return 0;
}

// ----------------------------------------------------------------------------------------
// merge_kv(..., merge="pivot") (pyramidkv_utils.py:119-170, LOOK-M) — called by SnapKV / PyramidKV / H2O update_kv
// right after their .topk when the cluster was built with merge="pivot" (:337-339): instead of dropping the unselected
// tokens, every dropped key is averaged into the kept key it is most similar to (cosine), and its value into the value row of
// the same number.  Restated with the reference's quirks, which parity must reproduce:
//   * "dropped" = positions selected by NO head (torch.isin against the indices of ALL heads flattened, :131-134), the last
//     window_size positions included; the same drop list for every head;
//   * the kept keys are ordered [window rows, selected rows] (:145) but the kept values [selected rows, window rows] (:147),
//     and the pivot index found on the keys is used for both (:155-160);
//   * head_dim is hard-coded as 128 (:149, :154).
// Arithmetic (16-bit dtypes), each step checked against torch in this container:
//   norms: torch.norm's 8-accumulator order (see kvco_l2norm);  x / norm: fp32 divide, one rounding;
//   similarity = normalised_dropped @ normalised_kept^T: torch's reduced-precision GEMM accumulates in fp32 in an order that is
//     not documented; restated as the d-ascending fp32 sum of the (exact) products, rounded once — on 230 400 entries 10
//     differed from torch (0.004 %) and no argmax moved; at the 8k configuration a handful of dropped tokens may pick a
//     different pivot than the reference (measured by the fixtures);
//   max(dim=-1): the FIRST maximal column;  merged = ((dropped + kept[pivot]) -> dtype) / 2;
//   scatter_reduce(mean, include_self=True): fp32 sum of the row itself and its merged rows in ascending dropped position,
//     rounded to the dtype; the count (1 + contributions) rounded to the dtype; their quotient rounded to the dtype.
// ----------------------------------------------------------------------------------------
template <class DT>
static float norm8(const typename DT::raw* row, int D) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int d = 0; d < D; ++d) { const float x = DT::ld(row[d]); acc[d & 7] = acc[d & 7] + x * x; }
    float tot = acc[0];
    for (int j = 1; j < 8; ++j) tot = tot + acc[j];
    return rnd<DT>(std::sqrt(tot));
}
template <class DT>
static int merge_pivot_impl(const kvco_params& P, const void* k_, const void* v_, const int64_t* idx, void* k_out_, void* v_out_,
                            int32_t* pivot_out, int64_t* drop_len_out) {
    typedef typename DT::raw raw;
    const int H = P.n_q_heads, G = H / P.n_kv_heads, D = P.head_dim, W = P.window;
    const int64_t L = P.q_len, kk = P.k, R = kk + W;
    std::vector<char> selected((size_t)L, 0);
    for (int64_t i = 0; i < (int64_t)H * kk; ++i) selected[(size_t)idx[i]] = 1;
    std::vector<int64_t> drop;
    for (int64_t j = 0; j < L; ++j) if (!selected[(size_t)j]) drop.push_back(j);
    const int64_t M = (int64_t)drop.size();
    if (drop_len_out) *drop_len_out = M;
    int err = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int h = 0; h < H; ++h) {
        const raw* kh = (const raw*)k_ + (int64_t)(h / G) * P.k_stride_h;
        const raw* vh = (const raw*)v_ + (int64_t)(h / G) * P.v_stride_h;
        // kept rows: source position of row r in the KEY order [window, selected] and in the VALUE order [selected, window]
        std::vector<int64_t> kpos((size_t)R), vpos((size_t)R);
        for (int64_t r = 0; r < R; ++r) {
            kpos[(size_t)r] = r < W ? L - W + r : idx[(int64_t)h * kk + (r - W)];
            vpos[(size_t)r] = r < kk ? idx[(int64_t)h * kk + r] : L - W + (r - kk);
        }
        std::vector<float> kn((size_t)R * D);                                       // normalised kept keys (dtype values)
        for (int64_t r = 0; r < R; ++r) {
            const raw* row = kh + kpos[(size_t)r] * P.k_stride_l;
            const float n = norm8<DT>(row, D);
            for (int d = 0; d < D; ++d) kn[(size_t)r * D + d] = rnd<DT>(DT::ld(row[d]) / n);
        }
        std::vector<float> ksum((size_t)R * D), vsum((size_t)R * D);
        std::vector<int64_t> cnt((size_t)R, 1);
        for (int64_t r = 0; r < R; ++r)
            for (int d = 0; d < D; ++d) {
                ksum[(size_t)r * D + d] = DT::ld(kh[kpos[(size_t)r] * P.k_stride_l + d]);
                vsum[(size_t)r * D + d] = DT::ld(vh[vpos[(size_t)r] * P.v_stride_l + d]);
            }
        std::vector<float> a((size_t)D);
        for (int64_t p = 0; p < M; ++p) {
            const raw* krow = kh + drop[(size_t)p] * P.k_stride_l;
            const raw* vrow = vh + drop[(size_t)p] * P.v_stride_l;
            const float n = norm8<DT>(krow, D);
            for (int d = 0; d < D; ++d) a[(size_t)d] = rnd<DT>(DT::ld(krow[d]) / n);
            int64_t best = 0; float bestv = 0.0f;
            for (int64_t r = 0; r < R; ++r) {
                float s = 0.0f;
                for (int d = 0; d < D; ++d) s = s + a[(size_t)d] * kn[(size_t)r * D + d];
                s = rnd<DT>(s);
                if (r == 0 || s > bestv) { best = r; bestv = s; }
            }
            if (pivot_out) pivot_out[(int64_t)h * M + p] = (int32_t)best;
            const raw* ksel = kh + kpos[(size_t)best] * P.k_stride_l;
            const raw* vsel = vh + vpos[(size_t)best] * P.v_stride_l;
            for (int d = 0; d < D; ++d) {
                const float km = rnd<DT>(rnd<DT>(DT::ld(krow[d]) + DT::ld(ksel[d])) / 2.0f);
                const float vm = rnd<DT>(rnd<DT>(DT::ld(vrow[d]) + DT::ld(vsel[d])) / 2.0f);
                ksum[(size_t)best * D + d] = ksum[(size_t)best * D + d] + km;
                vsum[(size_t)best * D + d] = vsum[(size_t)best * D + d] + vm;
            }
            cnt[(size_t)best]++;
        }
        raw* ko = (raw*)k_out_ + (int64_t)h * R * D;
        raw* vo = (raw*)v_out_ + (int64_t)h * R * D;
        for (int64_t r = 0; r < R; ++r) {
            const float c = rnd<DT>((float)cnt[(size_t)r]);
            for (int d = 0; d < D; ++d) {
                ko[r * D + d] = DT::st(rnd<DT>(ksum[(size_t)r * D + d]) / c);
                vo[r * D + d] = DT::st(rnd<DT>(vsum[(size_t)r * D + d]) / c);
            }
        }
    }
    return err;
}
// idx: [H][P->k] int64, the indices the method's .topk returned.  k_out / v_out: [H][k + window][D].  pivot_out (optional):
// [H][drop_len] int32, drop_len returned through drop_len_out (size the buffer for q_len).
KVCO_API int kvco_merge_pivot(const kvco_params* P, const void* k, const void* v, const int64_t* idx, void* k_out, void* v_out,
                              int32_t* pivot_out, int64_t* drop_len_out) {
    if (int e = check(*P)) return e;
    if (P->head_dim != 128) return -4;                                            // the reference hard-codes 128 (:149)
    switch (P->dtype) {
        case KVCO_BF16: return merge_pivot_impl<DtBf16>(*P, k, v, idx, k_out, v_out, pivot_out, drop_len_out);
        case KVCO_FP16: return merge_pivot_impl<DtFp16>(*P, k, v, idx, k_out, v_out, pivot_out, drop_len_out);
        default: return -4;                                                      // fp32: torch's fp32 GEMM / norm orders not restated
    }
}

// ----------------------------------------------------------------------------------------
// ThinK channel pruning, key_pruner_query_driven (pyramidkv_utils.py:13-26), called by SnapKVCluster.update_think
// (:391) on the COMPRESSED keys [1, H, S, D] and the full queries:
//   k = int(head_dim * ratio);  queries_norm = pow(q[..., -32:, :], 2).mean(dim=2);  keys_norm = pow(kv, 2).mean(dim=2)      (:15-18)
//   key = queries_norm * keys_norm;  _, indices = topk(key, k, largest=False);  mask = scatter(indices) -> pruned channels      (:19-23)
//   returns kv[:, :, :S - recent][~mask]  ->  [1, H, S - recent, D - k]   (channels in order), kv[:, :, S - recent:], ~mask       (:26)
// Arithmetic: pow(x, 2) = x * x rounded to the dtype; mean(dim=2) = cast to fp32, torch's outer-dimension cascade sum over the
// rows (CascadeSum above), / rows, ONE rounding (ReduceOps mean_out); the product rounded; topk largest=False = libstdc++
// nth_element + sort (k * 64 > 128 always) with the comparator (!isnan(a) && isnan(b)) || a < b on (value, index).
// ----------------------------------------------------------------------------------------
template <class DT>
static int think_impl(const kvco_params& P, const void* q_, const void* kc_, int64_t S, int64_t recent, int64_t n_prune,
                      void* pruned_out, uint8_t* keep_out, void* score_out) {
    typedef typename DT::raw raw;
    const int H = P.n_q_heads, D = P.head_dim;
    const int64_t L = P.q_len, QR = 32;
    const raw* q = (const raw*)q_; const raw* kc = (const raw*)kc_;
    for (int h = 0; h < H; ++h) {
        std::vector<float> sc((size_t)D);
        for (int d = 0; d < D; ++d) {
            CascadeSum a; a.init(QR);
            for (int64_t r = 0; r < QR; ++r) { const float x = DT::ld(q[(int64_t)h * P.q_stride_h + (L - QR + r) * P.q_stride_l + d]); a.add(rnd<DT>(x * x)); }
            CascadeSum b; b.init(S);
            for (int64_t r = 0; r < S; ++r) { const float x = DT::ld(kc[((int64_t)h * S + r) * D + d]); b.add(rnd<DT>(x * x)); }
            sc[(size_t)d] = rnd<DT>(rnd<DT>(a.result() / (float)QR) * rnd<DT>(b.result() / (float)S));
            if (score_out) ((raw*)score_out)[(int64_t)h * D + d] = DT::st(sc[(size_t)d]);
        }
        typedef std::pair<float, int64_t> elem;
        std::vector<elem> e((size_t)D);
        for (int d = 0; d < D; ++d) e[(size_t)d] = elem(sc[(size_t)d], d);
        auto comp = [](const elem& x, const elem& y) -> bool { return ((!std::isnan(x.first) && std::isnan(y.first)) || (x.first < y.first)); };
        if (n_prune > 0) {
            if (n_prune * 64 <= D) std::partial_sort(e.begin(), e.begin() + n_prune, e.end(), comp);
            else { std::nth_element(e.begin(), e.begin() + (n_prune - 1), e.end(), comp); std::sort(e.begin(), e.begin() + (n_prune - 1), comp); }
        }
        std::vector<uint8_t> keep((size_t)D, 1);
        for (int64_t t = 0; t < n_prune; ++t) keep[(size_t)e[(size_t)t].second] = 0;
        for (int d = 0; d < D; ++d) keep_out[(int64_t)h * D + d] = keep[(size_t)d];
        raw* o = (raw*)pruned_out + (int64_t)h * (S - recent) * (D - n_prune);
        for (int64_t r = 0; r < S - recent; ++r)
            for (int d = 0; d < D; ++d)
                if (keep[(size_t)d]) *o++ = kc[((int64_t)h * S + r) * D + d];
    }
    return 0;
}
// q: [H][L][D] by P's q strides (the last 32 rows are read); kc: the compressed keys, dense [H][S][D].
// pruned_out [H][S - recent][D - n_prune]; keep_out [H][D] (1 = channel kept = the reference's returned ~mask); score_out [H][D] or null.
KVCO_API int kvco_think_prune(const kvco_params* P, const void* q, const void* kc, int64_t S, int64_t recent, int64_t n_prune,
                              void* pruned_out, uint8_t* keep_out, void* score_out) {
    if (P->dtype < 0 || P->dtype > 2 || P->n_q_heads <= 0 || P->head_dim <= 0 || P->q_len < 32) return -1;
    if (S <= 0 || recent < 0 || recent > S || n_prune < 0 || n_prune > P->head_dim) return -1;
    switch (P->dtype) {
        case KVCO_BF16: return think_impl<DtBf16>(*P, q, kc, S, recent, n_prune, pruned_out, keep_out, score_out);
        case KVCO_FP16: return think_impl<DtFp16>(*P, q, kc, S, recent, n_prune, pruned_out, keep_out, score_out);
        default:        return think_impl<DtFp32>(*P, q, kc, S, recent, n_prune, pruned_out, keep_out, score_out);
    }
}

// Scalar probes used by tests to pin the helper arithmetic.
KVCO_API float kvco_exp_u20(float x) { return exp_u20(x); }
KVCO_API float kvco_sum(const float* x, int64_t n, int sum_mode) {
    (void)sum_mode;
    return sum_torch16(x, n);
}
KVCO_API uint16_t kvco_f32_to_f16(float f) { return f32_to_f16(f); }
KVCO_API float kvco_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
KVCO_API uint16_t kvco_f32_to_bf16(float f) { return f32_to_bf16(f); }
