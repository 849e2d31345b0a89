// kvc_common.h — device-side arithmetic shared by the gfx950 kernels.
//
// Every function here is the device twin of a host function in oracle/kvc_oracle.cpp and is built
// from single IEEE-754 fp32 operations (add, mul, fma, floor, correctly rounded divide) so the two
// agree bit for bit.  Compile with -ffp-contract=off: a contraction the source does not spell out
// would change a rounding point.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kvc.h"

namespace kvc {

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// ---- storage types -------------------------------------------------------------------------
template <int DT> struct Dt;
template <> struct Dt<KVC_BF16> {
    typedef uint16_t raw;
    static constexpr int esize = 2;
    __device__ static __forceinline__ float ld(raw r) { return u2f((uint32_t)r << 16); }
    __device__ static __forceinline__ raw st(float f) {   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
        asm volatile("" : "+v"(f));                         // the fp32 value is rounded as such (no folding into the producer)
        __bf16 h = (__bf16)f;
        raw r;
        __builtin_memcpy(&r, &h, 2);
        return r;
    }
    __device__ static __forceinline__ float finfo_min() { return u2f(0xff7f0000u); }
};
template <> struct Dt<KVC_FP16> {
    typedef uint16_t raw;
    static constexpr int esize = 2;
    __device__ static __forceinline__ float ld(raw r) {
        _Float16 h;
        __builtin_memcpy(&h, &r, 2);
        return (float)h;
    }
    __device__ static __forceinline__ raw st(float f) {   // v_cvt_f16_f32: RNE, overflow -> inf
        // The fp32 value must exist as such before the conversion: without this barrier the compiler folds
        // `(half)(a * b)` into v_fma_mixlo_f16, which rounds the exact product ONCE to fp16 — not the reference's
        // round-to-fp32-then-to-fp16 (seen as 1-ulp differences on fp16-subnormal probabilities at L = 8000).
        asm volatile("" : "+v"(f));
        _Float16 h = (_Float16)f;
        raw r;
        __builtin_memcpy(&r, &h, 2);
        return r;
    }
    __device__ static __forceinline__ float finfo_min() { return -65504.0f; }
};
template <> struct Dt<KVC_FP32> {
    typedef float raw;
    static constexpr int esize = 4;
    __device__ static __forceinline__ float ld(raw r) { return r; }
    __device__ static __forceinline__ raw st(float f) { return f; }
    __device__ static __forceinline__ float finfo_min() { return u2f(0xff7fffffu); }
};
template <int DT> __device__ __forceinline__ float rnd(float f) { return Dt<DT>::ld(Dt<DT>::st(f)); }

// ---- exp: the reference's softmax exponent (torch Vectorized<float>::exp_u20) ----------------
__device__ __forceinline__ float exp_u20(float x) {
    const float ln_flt_min = u2f(0xc2aeac50u), ln_flt_max = u2f(0x42b17218u);
    const float log2ef = u2f(0x3fb8aa3bu), ln2f = u2f(0x3f317218u);
    const float c1 = 0.999999701f, c2 = 0.499991506f, c3 = 0.166676521f, c4 = 0.0418978221f,
                c5 = 0.00828929059f;
    const bool below = x < ln_flt_min;
    float s = (x < ln_flt_max) ? x : ln_flt_max;
    s = (s > ln_flt_min) ? s : ln_flt_min;
    float fx = __builtin_fmaf(s, log2ef, 0.5f);
    fx = __builtin_floorf(fx);
    const float r = __builtin_fmaf(-fx, ln2f, s);
    float p = __builtin_fmaf(r, c5, c4);
    p = __builtin_fmaf(r, p, c3);
    p = __builtin_fmaf(r, p, c2);
    p = __builtin_fmaf(r, p, c1);
    p = __builtin_fmaf(r, p, 1.0f);
    const int n1 = (int)(fx - 1.0f);
    const float two_n = below ? 0.0f : u2f((uint32_t)(n1 + 127) << 23);
    p = p * two_n;
    p = p * 2.0f;
    return p;
}

// Two exponentials at once: the same operations on the packed-fp32 ALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 are
// per-element IEEE operations, so each half is bit-identical to exp_u20 of that half).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 exp_u20x2(f32x2 x) {
    const float ln_flt_min = u2f(0xc2aeac50u), ln_flt_max = u2f(0x42b17218u);
    const f32x2 log2ef = u2f(0x3fb8aa3bu), nln2f = -u2f(0x3f317218u);
    const f32x2 c1 = 0.999999701f, c2 = 0.499991506f, c3 = 0.166676521f, c4 = 0.0418978221f, c5 = 0.00828929059f;
    const bool below0 = x.x < ln_flt_min, below1 = x.y < ln_flt_min;
    f32x2 s;
    s.x = (x.x < ln_flt_max) ? x.x : ln_flt_max; s.x = (s.x > ln_flt_min) ? s.x : ln_flt_min;
    s.y = (x.y < ln_flt_max) ? x.y : ln_flt_max; s.y = (s.y > ln_flt_min) ? s.y : ln_flt_min;
    f32x2 fx = __builtin_elementwise_fma(s, log2ef, (f32x2)0.5f);
    fx.x = __builtin_floorf(fx.x); fx.y = __builtin_floorf(fx.y);
    const f32x2 r = __builtin_elementwise_fma(fx, nln2f, s);          // fma(-fx, ln2, s): negating an operand is exact
    f32x2 p = __builtin_elementwise_fma(r, c5, c4);
    p = __builtin_elementwise_fma(r, p, c3);
    p = __builtin_elementwise_fma(r, p, c2);
    p = __builtin_elementwise_fma(r, p, c1);
    p = __builtin_elementwise_fma(r, p, (f32x2)1.0f);
    const f32x2 fm1 = fx - (f32x2)1.0f;
    f32x2 two_n;
    two_n.x = below0 ? 0.0f : u2f((uint32_t)((int)fm1.x + 127) << 23);
    two_n.y = below1 ? 0.0f : u2f((uint32_t)((int)fm1.y + 127) << 23);
    p = p * two_n;
    p = p * (f32x2)2.0f;
    return p;
}

// exp_u20x2 for arguments <= 0 (a logit minus its row maximum), bit-identical to it there with 10 instructions fewer (17 for two
// values): one v_med3 clamps to [ln(FLT_MIN), 0]; the clamped lower bound yields fx = -126 and so 2^n = bits 0 = +0.0 — what the
// `below` select of exp_u20x2 returns — and (int)(fx - 1) + 127 = (int)fx + 126 because fx is integral.  An argument > 0 can only
// come from a lane whose result is discarded (a key or row beyond L, padded with zeros): it is treated as 0.
__device__ __forceinline__ f32x2 exp_u20x2_nonpos(f32x2 x) {
    const float ln_flt_min = u2f(0xc2aeac50u);
    const f32x2 log2ef = u2f(0x3fb8aa3bu), nln2f = -u2f(0x3f317218u);
    const f32x2 c1 = 0.999999701f, c2 = 0.499991506f, c3 = 0.166676521f, c4 = 0.0418978221f, c5 = 0.00828929059f;
    f32x2 s_;
    s_.x = __builtin_amdgcn_fmed3f(x.x, ln_flt_min, 0.0f);
    s_.y = __builtin_amdgcn_fmed3f(x.y, ln_flt_min, 0.0f);
    f32x2 fx = __builtin_elementwise_fma(s_, log2ef, (f32x2)0.5f);
    fx.x = __builtin_floorf(fx.x); fx.y = __builtin_floorf(fx.y);
    const f32x2 r = __builtin_elementwise_fma(fx, nln2f, s_);
    f32x2 p = __builtin_elementwise_fma(r, c5, c4);
    p = __builtin_elementwise_fma(r, p, c3);
    p = __builtin_elementwise_fma(r, p, c2);
    p = __builtin_elementwise_fma(r, p, c1);
    p = __builtin_elementwise_fma(r, p, (f32x2)1.0f);
    f32x2 two_n;
    two_n.x = u2f((uint32_t)((int)fx.x + 126) << 23);
    two_n.y = u2f((uint32_t)((int)fx.y + 126) << 23);
    p = p * two_n;
    p = p * (f32x2)2.0f;
    return p;
}

// ---- torch's cascade sum over the scored query rows (SumKernel.cpp multi_row_sum) -------------
struct CascadeSum {
    float a0, a1, a2, a3;
    int i, in_step, level_step, level_power, full;
    __device__ __forceinline__ void init(int size) {
        int cl = 0;
        while ((1 << cl) < size) ++cl;
        level_power = cl / 4 > 4 ? cl / 4 : 4;
        level_step = 1 << level_power;
        full = size - (size % level_step);
        a0 = a1 = a2 = a3 = 0.0f;
        i = 0;
        in_step = 0;
    }
    __device__ __forceinline__ void add(float v) {
        a0 = a0 + v;
        ++i;
        ++in_step;
        if (in_step == level_step && i <= full) {
            in_step = 0;
            const int mask = level_step - 1;
            a1 = a1 + a0; a0 = 0.0f;
            if ((i & (mask << level_power)) == 0) {
                a2 = a2 + a1; a1 = 0.0f;
                if ((i & (mask << (2 * level_power))) == 0) { a3 = a3 + a2; a2 = 0.0f; }
            }
        }
    }
    __device__ __forceinline__ float result() const { return ((a0 + a1) + a2) + a3; }
};

// ---- order-preserving integer keys for top-k (larger float -> larger key) ---------------------
// NaN: torch's top-k / sort comparators put every NaN above every number and leave NaNs equal among themselves
// (TopKImpl.h: (isnan(x) && !isnan(y)) || x > y).  All NaN patterns therefore share ONE key above +inf's — 0xfffe / 0xfffffffe,
// one below the all-ones word that the lane-parallel heaps use as "no node" — and the key comparison is torch's comparator.
template <int DT> struct Key;
template <> struct Key<KVC_BF16> {
    static constexpr int bits = 16;
    static constexpr uint32_t nan_above = 0x7f80u;        // |bits| beyond +inf's
    __device__ static __forceinline__ uint32_t of(uint16_t r) {
        const uint32_t k = (r & 0x8000u) ? (uint32_t)(~r & 0xffffu) : (uint32_t)(r | 0x8000u);
        return (r & 0x7fffu) > nan_above ? 0xfffeu : k;
    }
};
template <> struct Key<KVC_FP16> {
    static constexpr int bits = 16;
    static constexpr uint32_t nan_above = 0x7c00u;
    __device__ static __forceinline__ uint32_t of(uint16_t r) {
        const uint32_t k = (r & 0x8000u) ? (uint32_t)(~r & 0xffffu) : (uint32_t)(r | 0x8000u);
        return (r & 0x7fffu) > nan_above ? 0xfffeu : k;
    }
};
template <> struct Key<KVC_FP32> {
    static constexpr int bits = 32;
    __device__ static __forceinline__ uint32_t of(float f) {
        const uint32_t u = f2u(f), k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        return (u & 0x7fffffffu) > 0x7f800000u ? 0xfffffffeu : k;
    }
};

// Value held by lane (l ^ MASK), MASK in {1,2,4,8,16,32}: DPP quad_perm / row shifts and the gfx950 permlane swaps —
// plain VALU moves, no LDS round trip (a ds_bpermute costs ~120 cycles of latency per step; checked on the GPU by
// tools/dpp_probe.hip).
template <int MASK> __device__ __forceinline__ float xor_lane(float v) {
    const int x = __float_as_int(v);
    int r;
    if constexpr (MASK == 1) r = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);          // quad_perm [1,0,3,2]
    else if constexpr (MASK == 2) r = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);     // quad_perm [2,3,0,1]
    else if constexpr (MASK == 4) {                                                                  // row_shl:4 | row_shr:4
        r = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);
        r = __builtin_amdgcn_update_dpp(r, x, 0x114, 0xF, 0xA, false);
    } else if constexpr (MASK == 8) {                                                                // row_shl:8 | row_shr:8
        r = __builtin_amdgcn_update_dpp(x, x, 0x108, 0xF, 0x3, false);
        r = __builtin_amdgcn_update_dpp(r, x, 0x118, 0xF, 0xC, false);
    } else if constexpr (MASK == 16) {
        auto p = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        r = (threadIdx.x & 16) ? p[0] : p[1];
    } else {
        auto p = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        r = (threadIdx.x & 32) ? p[0] : p[1];
    }
    return __int_as_float(r);
}

__device__ __forceinline__ float wave_xor_sum(float v) {   // butterfly 1,2,4,8,16,32 (oracle: sum_kvc)
    v = v + xor_lane<1>(v);
    v = v + xor_lane<2>(v);
    v = v + xor_lane<4>(v);
    v = v + xor_lane<8>(v);
    v = v + xor_lane<16>(v);
    v = v + xor_lane<32>(v);
    return v;
}
__device__ __forceinline__ float half_xor_max(float v) {   // max over the 32 lanes sharing lane>>5
    float o;
    o = xor_lane<1>(v); v = o > v ? o : v;
    o = xor_lane<2>(v); v = o > v ? o : v;
    o = xor_lane<4>(v); v = o > v ? o : v;
    o = xor_lane<8>(v); v = o > v ? o : v;
    o = xor_lane<16>(v); v = o > v ? o : v;
    return v;
}

}  // namespace kvc
