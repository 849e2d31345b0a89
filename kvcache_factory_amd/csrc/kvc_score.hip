// kvc_score.hip — A1..A5 of the reference hot path on gfx950 (MI355X):
//   window QK^T (+ scale, + local causal mask)   pyramidkv_utils.py:317-324
//   fp32 softmax over all L keys, cast to dtype  :326
//   sum over the W window rows                    :327
//   1-D max / avg pooling                         :328-333
//
// One HBM pass over K, then the softmax / window-sum / pooling stage in one of two forms:
//   logits_kernel        : 4 autonomous waves per workgroup, each walking 32-key tiles of one KV head: K tile ->
//                          registers -> padded LDS rows, contracted against the G*W query rows of the head's query
//                          group with v_mfma_f32_32x32x2_f32, whose result is bit for bit the d-ascending fmaf
//                          chain the oracle computes (KVCO_DOT_CHAIN); dot_mode mfma16 uses the packed 16-bit MFMA
//                          instead (tolerance mode).  Epilogue: the reference's three roundings, logits
//                          [h][L][W] and per-workgroup row maxima.  Design notes at the kernel.
//   softmax_pool_kernel  : one 1024-thread workgroup per head (batched launches): exponentials, fixed-order row
//                          sums (oracle: sum_kvc), p = round(e / sum), window sum in torch's cascade order, round,
//                          pool, write the scores.
//   rowsum_kernel + pool_kernel : the same arithmetic split over many workgroups per head (per-layer calls).
#include <type_traits>

#include "kvc_common.h"
#include "kvc_launch.h"
#include "kvc_ldsasm.h"

namespace kvc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ ScoreView view_of(const ScoreArgs& a, int item) {
    char* w = a.ws + (int64_t)item * a.ws_item_stride;
    ScoreView v;
    v.q = a.q.p[item]; v.k = a.k.p[item]; v.scores = const_cast<void*>(a.scores.p[item]);
    v.logits = w + a.off_logits;
    v.pmax = reinterpret_cast<float*>(w + a.off_pmax);
    v.rowmax = reinterpret_cast<float*>(w + a.off_rowmax);
    v.rowsum = reinterpret_cast<float*>(w + a.off_rowsum);
    v.part16 = a.off_part16 >= 0 ? reinterpret_cast<float*>(w + a.off_part16) : nullptr;
    return v;
}

// Diagnostic build only (-DKVC_STAMPS): per-wave s_memtime stamps at phase boundaries of logits_kernel, written to
// a buffer of their own (ScoreArgs::dbg) that nothing else reads.  Never enabled in the shipped library.
#if defined(KVC_STAMPS)
#define KVC_STAMP(slot)                                                                                       \
    do {                                                                                                      \
        unsigned long long t_;                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (a.dbg && lane == 0) a.dbg[((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8 + (slot)] = t_; \
    } while (0)
#define KVC_SPSTAMP(slot)                                                                                     \
    do {                                                                                                      \
        unsigned long long t_;                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (a.dbg && threadIdx.x == 0 && blockIdx.y == 0) a.dbg[blockIdx.x * 8 + (slot)] = t_;                \
    } while (0)
#else
#define KVC_STAMP(slot) do { } while (0)
#define KVC_SPSTAMP(slot) do { } while (0)
#endif

// Pull the element with parity kh (= lane >> 5) of bf16/fp16 pair s out of a 16-byte chunk (8 elements), or of fp32
// pair s out of a 16-byte chunk (4 elements), widened to fp32.  `sel` is the per-lane constant lane_sel<DT>(kh):
// one v_perm_b32 per bf16 element (selector moves the chosen half to the top and zero-fills), shift+cvt for fp16.
template <int DT> __device__ __forceinline__ uint32_t lane_sel(int kh);
template <> __device__ __forceinline__ uint32_t lane_sel<KVC_BF16>(int kh) { return kh ? 0x07060c0cu : 0x05040c0cu; }
template <> __device__ __forceinline__ uint32_t lane_sel<KVC_FP16>(int kh) { return kh ? 16u : 0u; }
template <> __device__ __forceinline__ uint32_t lane_sel<KVC_FP32>(int kh) { return (uint32_t)kh; }
template <int DT> __device__ __forceinline__ float pick(const uint4& v, int s, uint32_t sel);
template <> __device__ __forceinline__ float pick<KVC_BF16>(const uint4& v, int s, uint32_t sel) {
    const uint32_t w = s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w;
    return u2f(__builtin_amdgcn_perm(w, w, sel));
}
template <> __device__ __forceinline__ float pick<KVC_FP16>(const uint4& v, int s, uint32_t sel) {
    const uint32_t w = s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w;
    return Dt<KVC_FP16>::ld((uint16_t)(w >> sel));
}
template <> __device__ __forceinline__ float pick<KVC_FP32>(const uint4& v, int s, uint32_t sel) {
    // chunk holds d = 4c..4c+3; pair s in {0,1} -> elements 2s, 2s+1
    const uint32_t w = s == 0 ? (sel ? v.y : v.x) : (sel ? v.w : v.z);
    return u2f(w);
}

// ---------------------------------------------------------------------------------------------
// x / sqrt(D) as the reference computes it
// ---------------------------------------------------------------------------------------------
template <int D> struct ScaleDiv;            // x / sqrt(D) in fp32, bit-identical to the IEEE division
template <> struct ScaleDiv<64> {            // sqrt(64) = 8: multiplying by 2^-3 IS the division (exact scaling)
    __device__ static __forceinline__ float apply(float x, float) { return x * 0.125f; }
    __device__ static __forceinline__ float apply_in_guard(float x, float) { return x * 0.125f; }
};
template <> struct ScaleDiv<128> {
    // q0 = x*rc, r = fma(-q0, c, x), q = fma(r, rc, q0) equals RN(x / c) for c = sqrt(128) and EVERY finite fp32 x with
    // |x| >= 2^-100 (checked exhaustively over all 2^32 inputs: tests/test_fastdiv.py); the rest takes the true division.
    __device__ static __forceinline__ float apply(float x, float c) {
        const float rc = u2f(0x3db504f3u);                         // RN(1 / sqrt(128))
        const float ax = __builtin_fabsf(x);
        if (__builtin_expect(!(ax >= u2f(0x0d800000u) && ax < __builtin_inff()), 0)) return x / c;
        const float q0 = x * rc;
        const float r = __builtin_fmaf(-q0, c, x);
        return __builtin_fmaf(r, rc, q0);
    }
    // branch-free form for values already known to lie inside the guard (whole tile checked once: tile_in_guard)
    __device__ static __forceinline__ float apply_in_guard(float x, float c) {
        const float rc = u2f(0x3db504f3u);
        const float q0 = x * rc;
        const float r = __builtin_fmaf(-q0, c, x);
        return __builtin_fmaf(r, rc, q0);
    }
};

constexpr int LOGITS_WAVES = 4;              // waves per workgroup sharing one Q image: 48 KB LDS -> 3 groups per CU.
                                             // (8 waves / 128 VGPRs / 4 waves per SIMD measured 12 % slower in batch mode, round 1.)
constexpr int LOGITS_THREADS = LOGITS_WAVES * 64;

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
// 16-deep MFMA step on packed bf16 / fp16 operands (FAST scan).  Its internal accumulation order is NOT the fmaf chain
// (tools/mfma_probe.hip): logits may differ from the exact path by one unit in the last place on ~1e-4 of the entries.
template <int DT> __device__ __forceinline__ f32x16 mfma16(const uint4& av, const uint4& bv, f32x16 acc) {
    if constexpr (DT == KVC_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, av), __builtin_bit_cast(s16x8, bv), acc, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, av), __builtin_bit_cast(h16x8, bv), acc, 0, 0, 0);
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// Two values rounded to the storage dtype: the packed bits (element 0 in the low half) and the rounded values in fp32.
template <int DT> __device__ __forceinline__ uint32_t pack2(f32x2 v, f32x2& back) {
    if constexpr (DT == KVC_BF16) {
        asm volatile("" : "+v"(v));                                   // round the fp32 values as such
        const uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32
        back.x = u2f(w << 16); back.y = u2f(w & 0xffff0000u);
        return w;
    } else {
        const uint32_t lo = Dt<DT>::st(v.x), hi = Dt<DT>::st(v.y);
        back.x = Dt<DT>::ld((uint16_t)lo); back.y = Dt<DT>::ld((uint16_t)hi);
        return lo | (hi << 16);
    }
}
template <int D> __device__ __forceinline__ f32x2 scale2_in_guard(f32x2 x, float c);   // ScaleDiv<D>::apply_in_guard x 2
template <> __device__ __forceinline__ f32x2 scale2_in_guard<64>(f32x2 x, float) { return x * (f32x2)0.125f; }
template <> __device__ __forceinline__ f32x2 scale2_in_guard<128>(f32x2 x, float c) {
    // three packed-fp32 instructions with the two constants broadcast from SGPR pairs (the compiler scalarises the
    // f32x2 form into six when the constants live in SGPRs)
    const uint32_t rcb = 0x3db504f3u, cb = f2u(c);
    const unsigned long long rc2 = ((unsigned long long)rcb << 32) | rcb, c2 = ((unsigned long long)cb << 32) | cb;
    f32x2 q0, r, q;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(q0) : "v"(x), "s"(rc2));
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(q0), "s"(c2), "v"(x));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(q) : "v"(r), "s"(rc2), "v"(q0));
    return q;
}

// ---------------------------------------------------------------------------------------------
// logits_kernel (v5).  What shapes it: on gfx950 the f32-input MFMA and ordinary VALU instructions do NOT overlap on a
// SIMD (tools/mfma_valu_probe.hip: chain + VALU costs the SUM of the two, even across waves), so every VALU
// instruction of this kernel is paid in full on top of the 64-cycle MFMAs of the exact chain.  Hence:
//   * bf16 K elements reach the MFMA B operand through ds_read_u16_d16_hi, which writes the 16 bits into the high
//     half of the VGPR and clears the low half (tools/d16_probe.hip) — an exact bf16 -> fp32 widening with no VALU op;
//   * LDS / global addresses are one per-lane register + immediate offsets (padded rows instead of an XOR swizzle,
//     chunk-major A image, uniform bases in SGPRs);
//   * the epilogue works on pairs with the packed-fp32 ALU, keeps one running maximum per accumulator register (the
//     cross-lane fold happens once per wave, and the maximum is taken before the last rounding — rounding is monotonic).
// ---------------------------------------------------------------------------------------------
template <int DT, int D, int WV, bool FAST>
__global__ __launch_bounds__(LOGITS_THREADS, 3) void logits_kernel(const ScoreArgs a) {
    const ScoreView vw = view_of(a, blockIdx.z);
    typedef typename Dt<DT>::raw raw;
    constexpr int ES = Dt<DT>::esize;
    constexpr int ROWB = D * ES;             // bytes per key row
    constexpr bool ASM_B = (DT == KVC_BF16) && !FAST;
    // Row pitch in LDS.  The exact bf16 scan reads ONE dword per key row per instruction (ds_read_u16_d16_hi): an odd
    // dword pitch puts the 32 keys in 32 different banks (a 16-byte pad left 4-way conflicts: SQ_LDS_BANK_CONFLICT was
    // 24 % of the kernel's cycles); the rows are then only 4-byte aligned and are staged with dword-pair writes.  The
    // other variants read 16-byte chunks and keep 16-byte aligned rows.
    constexpr int ROWP = ASM_B ? ROWB + 4 : ROWB + 16;
    constexpr int CH = ROWB / 16;            // 16-byte chunks per row
    constexpr int PAIRS = 8 / ES;            // mfma k-pairs per chunk: 4 (16-bit) or 2 (fp32)
    constexpr int STG = CH / 2;              // staging registers (uint4) per lane per tile: 32*CH chunks / 64 lanes
    constexpr int RPI = 64 / CH;             // key rows covered by one staging step
    constexpr int ICH = D / 8;               // 16-byte chunks of one lane's fp32 A fragment (D/2 values)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: [Q image: ICH chunks x 64 lanes x 16 B] [LOGITS_WAVES x 32 keys x ROWP] [LOGITS_WAVES x 32 floats]
    char* const img = smem;
    constexpr int IMG_BYTES = FAST ? 0 : 64 * ICH * 16;  // the FAST scan keeps its packed A fragments in registers
    constexpr int NSTEP = FAST ? D / 16 : ICH;            // MFMA steps per tile: D/16 (packed 16-deep) or one per A chunk
    char* const tiles = smem + IMG_BYTES;
    float* wmax = reinterpret_cast<float*>(tiles + LOGITS_WAVES * 32 * ROWP);   // [waves][32]

    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: tile indices and bases stay in SGPRs
    const uint32_t psel = lane_sel<DT>(kh);
    const int b = blockIdx.y / a.n_kv_heads, g = blockIdx.y % a.n_kv_heads;
    const int L = a.q_len, W = WV > 0 ? WV : a.window, G = a.group;
    const int rows = G * W;                   // query rows sharing this KV head
    const int n_mt = (rows + 31) / 32;
    const int n_t = (L + 31) / 32;            // 32-key tiles of this head
    const int wave_g = blockIdx.x * LOGITS_WAVES + wave, n_waves = gridDim.x * LOGITS_WAVES;
    char* const buf = tiles + wave * (32 * ROWP);
    const float sqrt_d = a.sqrt_d;
    KVC_STAMP(0);

    const char* const kbase = reinterpret_cast<const char*>(vw.k) +
                              ((int64_t)b * a.k_stride_b + (int64_t)g * a.k_stride_h) * ES;
    const int64_t tile_bytes = (int64_t)32 * a.k_stride_l * ES;    // one tile further along the key axis
    uint32_t koff[STG];                                            // this lane's chunks inside a tile (global side)
#pragma unroll
    for (int it = 0; it < STG; ++it) {
        const int c = it * 64 + lane;
        koff[it] = (uint32_t)((c / CH) * (a.k_stride_l * ES) + (c % CH) * 16);
    }
    auto issue = [&](int tile, uint4 (&st)[STG]) {
        const char* const tb = kbase + tile * tile_bytes;          // uniform
        if (tile * 32 + 32 <= L) {
#pragma unroll
            for (int it = 0; it < STG; ++it) st[it] = *reinterpret_cast<const uint4*>(tb + koff[it]);
        } else {
#pragma unroll
            for (int it = 0; it < STG; ++it) {
                const int key = tile * 32 + it * RPI + lane / CH;
                st[it] = key < L ? *reinterpret_cast<const uint4*>(tb + koff[it]) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    char* const cdst = buf + (lane / CH) * ROWP + (lane % CH) * 16;   // LDS side of the same chunks
    auto commit = [&](const uint4 (&st)[STG]) {
#pragma unroll
        for (int it = 0; it < STG; ++it) {
            if constexpr (ASM_B) {
                uint32_t* d = reinterpret_cast<uint32_t*>(cdst + it * (RPI * ROWP));
                d[0] = st[it].x; d[1] = st[it].y; d[2] = st[it].z; d[3] = st[it].w;
            } else {
                *reinterpret_cast<uint4*>(cdst + it * (RPI * ROWP)) = st[it];
            }
        }
    };

    for (int mt = 0; mt < n_mt; ++mt) {
        // ---- A operand: the workgroup converts the 32 query rows of this M-tile ONCE into an fp32 image in LDS, laid
        // out as the MFMA A-fragment of every lane (lane = row + 32*parity, value s = Q[row][2s + parity]), chunk-major
        // ([chunk][lane][4 values]) so that the waves read chunk c at one per-lane address + c*1024.
        uint4 st[STG];
        int tile = wave_g;
        if (tile < n_t) issue(tile, st);
        uint4 aq[FAST ? NSTEP : 1];                        // FAST: this lane's packed A fragments, chunk 2*s + kh of its row
        if constexpr (FAST) {
            const int i = mt * 32 + j;
            const bool valid = i < rows;
            const int hq = g * G + (valid ? i / W : 0), w = valid ? i % W : 0;
            const char* qrow = reinterpret_cast<const char*>(vw.q) +
                ((int64_t)b * a.q_stride_b + (int64_t)hq * a.q_stride_h + (int64_t)(L - W + w) * a.q_stride_l) * ES;
#pragma unroll
            for (int sI = 0; sI < NSTEP; ++sI)
                aq[sI] = valid ? *reinterpret_cast<const uint4*>(qrow + (2 * sI + kh) * 16) : make_uint4(0, 0, 0, 0);
        }
        if (!FAST && mt > 0) __syncthreads();              // previous image no longer read
        if constexpr (!FAST) {
            constexpr int PER_ROW = ROWB / 16;             // 16-byte pieces of one query row
            for (int pc = tid; pc < 32 * PER_ROW; pc += LOGITS_THREADS) {
                const int r = pc / PER_ROW, cc = pc % PER_ROW;
                const int i = mt * 32 + r;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (i < rows) {
                    const int hq = g * G + i / W, w = i % W;
                    const char* qrow = reinterpret_cast<const char*>(vw.q) +
                        ((int64_t)b * a.q_stride_b + (int64_t)hq * a.q_stride_h + (int64_t)(L - W + w) * a.q_stride_l) * ES;
                    v = *reinterpret_cast<const uint4*>(qrow + cc * 16);
                }
                // this piece holds pairs s = cc*PAIRS .. cc*PAIRS+PAIRS-1; element s of lane-row l sits in chunk s/4
#pragma unroll
                for (int sp = 0; sp < PAIRS; ++sp) {
                    const int sidx = cc * PAIRS + sp;
                    const int chunk = sidx >> 2, e = sidx & 3;
                    *reinterpret_cast<float*>(img + chunk * 1024 + r * 16 + e * 4) = pick<DT>(v, sp, lane_sel<DT>(0));
                    *reinterpret_cast<float*>(img + chunk * 1024 + (32 + r) * 16 + e * 4) = pick<DT>(v, sp, lane_sel<DT>(1));
                }
            }
        }
        if (!FAST) __syncthreads();
        if (tile < n_t) commit(st);
        KVC_STAMP(1);
        float rm[16];                          // running maximum per accumulator register (row 8*(e/4) + 4*kh + e%4)
#pragma unroll
        for (int e = 0; e < 16; ++e) rm[e] = -__builtin_inff();
        const char* const krow = buf + j * ROWP;
        const char* const arow = img + lane * 16;
        const uint32_t krow_a = lds_addr(krow) + 2 * kh, arow_a = lds_addr(arow);
        // logits of this lane: element (rg, e) of key `key` goes to lg + soff[rg] + key*W*ES (+ e*ES)
        char* const lg = reinterpret_cast<char*>(vw.logits);
        uint32_t soff[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int i0 = mt * 32 + 8 * rg + 4 * kh;
            const int hq = g * G + (i0 < rows ? i0 / W : 0), w0 = i0 % W;
            soff[rg] = (uint32_t)(((((int64_t)b * a.n_q_heads + hq) * L) * W + w0) * ES);
        }

        // One epilogue element: the reference's three roundings (+ the local causal mask on the last W keys).
        auto finish = [&](float accv, int i, int key, bool tail) -> float {
            float v = rnd<DT>(accv);
            v = rnd<DT>(ScaleDiv<D>::apply(v, sqrt_d));
            if (tail) {
                const int w = i % W;
                if (key >= L - W && (key - (L - W)) > w) v = rnd<DT>(v + Dt<DT>::finfo_min());
            }
            return v;
        };
        // Store 4 consecutive rows (one accumulator register group) of one key: 8 bytes (16-bit) or 16 bytes (fp32).
        auto store4 = [&](const float (&x)[4], int i0, int key) {
            const int hq = g * G + i0 / W, w0 = i0 % W;
            raw* dst = reinterpret_cast<raw*>(vw.logits) + (((int64_t)b * a.n_q_heads + hq) * L + key) * W + w0;
            if constexpr (ES == 2) {
                uint2 pk;
                pk.x = (uint32_t)Dt<DT>::st(x[0]) | ((uint32_t)Dt<DT>::st(x[1]) << 16);
                pk.y = (uint32_t)Dt<DT>::st(x[2]) | ((uint32_t)Dt<DT>::st(x[3]) << 16);
                *reinterpret_cast<uint2*>(dst) = pk;
            } else {
                *reinterpret_cast<float4*>(dst) = make_float4(x[0], x[1], x[2], x[3]);
            }
        };
        // per-row maximum over the 32 key lanes: reduce-scatter over the 5 key bits (16 cross-lane moves): after the
        // step on lane bit t each lane keeps only the registers whose index bit matches its own.
        auto fold_max = [&](const float (&xs)[16]) -> float {
            const bool b4 = (j & 16) != 0, b3 = (j & 8) != 0, b2 = (j & 4) != 0, b1 = (j & 2) != 0;
            float y[8], z[4], u[2];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float o = xor_lane<16>(b4 ? xs[r] : xs[r + 8]);
                const float keep = b4 ? xs[r + 8] : xs[r];
                y[r] = o > keep ? o : keep;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float o = xor_lane<8>(b3 ? y[r] : y[r + 4]);
                const float keep = b3 ? y[r + 4] : y[r];
                z[r] = o > keep ? o : keep;
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float o = xor_lane<4>(b2 ? z[r] : z[r + 2]);
                const float keep = b2 ? z[r + 2] : z[r];
                u[r] = o > keep ? o : keep;
            }
            float m = xor_lane<2>(b1 ? u[0] : u[1]);
            { const float keep = b1 ? u[1] : u[0]; m = m > keep ? m : keep; }
            { const float o = xor_lane<1>(m); m = o > m ? o : m; }
            return m;
        };
        // General epilogue (ragged tile, masked tail, padded rows, W % 4 != 0): straight after its own MFMAs.
        auto epilogue_general = [&](const f32x16& acc, int t) {
            const int key = t * 32 + j;
            const bool tail = t * 32 + 32 > L - W;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float x[4];
                const int i0 = mt * 32 + 8 * rg + 4 * kh;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[e] = finish(acc[rg * 4 + e], i0 + e, key, tail);
                    const float xv = (key < L) ? x[e] : -__builtin_inff();
                    rm[rg * 4 + e] = xv > rm[rg * 4 + e] ? xv : rm[rg * 4 + e];
                }
                if (i0 < rows && key < L) {
                    if ((W % 4) == 0) {
                        store4(x, i0, key);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int i = i0 + e;
                            if (i < rows) {
                                const int hq = g * G + i / W, w = i % W;
                                reinterpret_cast<raw*>(vw.logits)[(((int64_t)b * a.n_q_heads + hq) * L + key) * W + w] = Dt<DT>::st(x[e]);
                            }
                        }
                    }
                }
            }
        };
        // A tile is "plain" when none of those cases applies: its epilogue is then branch-free, a pair of accumulator
        // elements at a time on the packed-fp32 ALU.  (Riding it between the MFMA steps of the next tile, as earlier
        // versions did, buys nothing: the f32 MFMA and the VALU do not overlap on gfx950.)
        const bool rows_plain = (W % 4) == 0 && (mt + 1) * 32 <= rows;
        // Plain epilogue of accumulator elements e0, e0+1 (rows i0+e0%4, +1 of register group e0/4).
        uint32_t pkw[2];                                       // packed words of the current register group
        auto plain_pair = [&](const f32x16& pend, int e0, uint32_t pkeyoff) {     // pend: the tile's accumulators
            if constexpr (DT == KVC_FP32) {
                const float v0 = ScaleDiv<D>::apply_in_guard(pend[e0], sqrt_d), v1 = ScaleDiv<D>::apply_in_guard(pend[e0 + 1], sqrt_d);
                rm[e0] = v0 > rm[e0] ? v0 : rm[e0];
                rm[e0 + 1] = v1 > rm[e0 + 1] ? v1 : rm[e0 + 1];
                *reinterpret_cast<float2*>(lg + (soff[e0 >> 2] + pkeyoff + (e0 & 3) * 4)) = make_float2(v0, v1);
            } else {
                f32x2 x = {pend[e0], pend[e0 + 1]}, back;
                (void)pack2<DT>(x, back);                                         // first rounding
                const f32x2 q = scale2_in_guard<D>(back, sqrt_d);                 // second: after the scaling
                // (maximum before rounding: monotonic.)  One v_max_f32 each, spelled out: fmaxf() makes the compiler
                // re-canonicalise all 16 running maxima at the top of every tile
                asm("v_max_f32 %0, %1, %2" : "=v"(rm[e0]) : "v"(rm[e0]), "v"(q.x));
                asm("v_max_f32 %0, %1, %2" : "=v"(rm[e0 + 1]) : "v"(rm[e0 + 1]), "v"(q.y));
                f32x2 unused;
                pkw[(e0 >> 1) & 1] = pack2<DT>(q, unused);
                if ((e0 & 3) == 2) *reinterpret_cast<uint2*>(lg + (soff[e0 >> 2] + pkeyoff)) = make_uint2(pkw[0], pkw[1]);
            }
        };
        for (; tile < n_t; tile += n_waves) {
            const int next = tile + n_waves;
            if (next < n_t) issue(next, st);                  // in flight during the MFMAs below
            __builtin_amdgcn_wave_barrier();
            KVC_STAMP(2);
            // ---- 32 rows x 32 keys, chain over d = 0..D-1 in order ----
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            auto chain = [&]() {
                if constexpr (ASM_B) {
                    // operands by hand-issued LDS reads, one step ahead of the MFMAs that consume them:
                    //   A: 4 fragment values (ds_read_b128, chunk-major image);  B: 4 x bf16 -> fp32 in the load itself.
                    asm volatile("" ::: "memory");            // this tile's ds_writes (commit) stay above the reads
                    f32x4 A0, A1;
                    uint32_t B0[4], B1[4];
                    ld_step<0>(A0, B0, arow_a, krow_a);
                    static_for<0, NSTEP>([&](auto ic_) {
                        constexpr int ic = decltype(ic_)::value;
                        f32x4& Ac = (ic & 1) ? A1 : A0;
                        uint32_t (&Bc)[4] = (ic & 1) ? B1 : B0;
                        if constexpr (ic + 1 < NSTEP) {
                            ld_step<ic + 1>((ic & 1) ? A0 : A1, (ic & 1) ? B0 : B1, arow_a, krow_a);
                            wait_step<5>(Ac, Bc);             // all but the 5 reads just issued have landed
                        } else {
                            wait_step<0>(Ac, Bc);
                        }
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[s4], u2f(Bc[s4]), acc, 0, 0, 0);
                    });
                    asm volatile("" ::: "memory");            // ... and the next tile's ds_writes stay below them
                } else {
                    static_for<0, NSTEP>([&](auto ic_) {
                        constexpr int sti = decltype(ic_)::value;
                        if constexpr (FAST) {
                            const uint4 kv = *reinterpret_cast<const uint4*>(krow + (2 * sti + kh) * 16);
                            acc = mfma16<DT>(aq[sti], kv, acc);
                        } else {                               // one A chunk = 4 fragment values = 4 exact f32 MFMAs
                            const float4 av = *reinterpret_cast<const float4*>(arow + sti * 1024);
                            const float af[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                            for (int kc = 0; kc < 4 / PAIRS; ++kc) {   // K chunks feeding these 4 values: 1 (16-bit) or 2 (fp32)
                                const int c = sti * (4 / PAIRS) + kc;
                                const uint4 kv = *reinterpret_cast<const uint4*>(krow + c * 16);
#pragma unroll
                                for (int sp = 0; sp < PAIRS; ++sp)
                                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kc * PAIRS + sp], pick<DT>(kv, sp, psel), acc, 0, 0, 0);
                            }
                        }
                    });
                }
            };
            chain();
            asm volatile("" :: "v"(acc[0]), "v"(acc[15]));
            // the tile's LDS reads are all issued (the LDS serves a wave in order): the next tile may overwrite the single
            // buffer now, so its ds_writes overlap the VALU work that follows
            if (next < n_t) commit(st);
            KVC_STAMP(3);
            bool plain = rows_plain && tile * 32 + 32 <= L - W;
            if (plain) {
                // the branch-free scaling needs every |value| of the tile inside [2^-98, 2^126] (rounding to dtype moves a
                // value by < 1 %, so this keeps the rounded value inside the proven [2^-100, inf) guard); a zero logit, an
                // overflow or a NaN sends the whole tile down the general path instead
                // (v_min3_f32 / v_max3_f32 take |x| as a source modifier: 16 instructions where integer compares on the masked bits
                // took 48.  They skip a NaN — which the packed path turns into a NaN like the general one, and NaN scores all
                // order alike; an infinity is caught by the maximum.)
                float amin = __builtin_fminf(__builtin_fabsf(acc[0]), __builtin_fabsf(acc[1])), amax = __builtin_fmaxf(__builtin_fabsf(acc[0]), __builtin_fabsf(acc[1]));
#pragma unroll
                for (int e = 2; e < 16; e += 2) {
                    amin = __builtin_fminf(__builtin_fminf(__builtin_fabsf(acc[e]), __builtin_fabsf(acc[e + 1])), amin);
                    amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(acc[e]), __builtin_fabsf(acc[e + 1])), amax);
                }
                plain = !__any(!(amin >= u2f(0x0e800000u) && amax <= u2f(0x7e800000u)));
            }
            if (plain) {                                        // branch-free epilogue, two accumulator elements at a time
                const uint32_t keyoff = (uint32_t)(tile * 32 + j) * (uint32_t)(W * ES);
#pragma unroll
                for (int pr = 0; pr < 8; ++pr) plain_pair(acc, 2 * pr, keyoff);
            } else {
                epilogue_general(acc, tile);
            }
            KVC_STAMP(5);
        }
        // ---- block-level maximum per row -> pmax[hq][blockIdx.x][w] ----
        {
            float rmr[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) rmr[e] = rnd<DT>(rm[e]);  // plain tiles kept the unrounded value (monotonic)
            const float runmax = fold_max(rmr);
            const int R = ((j >> 1) & 1) + ((j >> 2) & 1) * 2 + ((j >> 3) & 1) * 4 + ((j >> 4) & 1) * 8;   // accumulator register
            if ((j & 1) == 0) wmax[wave * 32 + (R & 3) + 8 * (R >> 2) + 4 * kh] = runmax;
        }
        __syncthreads();
        if (tid < 32) {
            const int i = mt * 32 + tid;
            if (i < rows) {
                float m = wmax[tid];
#pragma unroll
                for (int wv = 1; wv < LOGITS_WAVES; ++wv) { const float o = wmax[wv * 32 + tid]; m = o > m ? o : m; }
                const int hq = g * G + i / W, w = i % W;
                vw.pmax[(((int64_t)b * a.n_q_heads + hq) * a.n_tiles + blockIdx.x) * W + w] = m;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// logits_mt4_kernel: the same scan for G * W == 128 query rows per KV head (W = 32 with four query heads per KV head: the
// library's default window, the needle runner's; also W = 64 with two and W = 16 with eight).  logits_kernel walks the keys once per 32-row M-tile — at four M-tiles the
// PMC counters showed every K byte fetched four times (profiles/r02_pmc_batch_c2_w32_FETCH_SIZE.csv: 2.08 GB per launch for
// 526 MB of keys).  Here the four waves of a workgroup ARE the four M-tiles: they share ONE staged K tile (loaded by all
// 256 threads, requested two tiles ahead), each wave keeps its own Q image / packed A fragments, accumulators, running maxima
// and epilogue.  Two barriers per tile; K is read from HBM once.  Same arithmetic, same bits.
// ---------------------------------------------------------------------------------------------
template <int DT, int D, int WV, bool FAST>
__global__ __launch_bounds__(LOGITS_THREADS, 2) void logits_mt4_kernel(const ScoreArgs a) {
    const ScoreView vw = view_of(a, blockIdx.z);
    typedef typename Dt<DT>::raw raw;
    constexpr int ES = Dt<DT>::esize;
    constexpr int ROWB = D * ES;             // bytes per key row
    constexpr bool ASM_B = (DT == KVC_BF16) && !FAST;
    // Row pitch in LDS.  The exact bf16 scan reads ONE dword per key row per instruction (ds_read_u16_d16_hi): an odd
    // dword pitch puts the 32 keys in 32 different banks (a 16-byte pad left 4-way conflicts: SQ_LDS_BANK_CONFLICT was
    // 24 % of the kernel's cycles); the rows are then only 4-byte aligned and are staged with dword-pair writes.  The
    // other variants read 16-byte chunks and keep 16-byte aligned rows.
    constexpr int ROWP = ASM_B ? ROWB + 4 : ROWB + 16;
    constexpr int CH = ROWB / 16;            // 16-byte chunks per row
    constexpr int PAIRS = 8 / ES;            // mfma k-pairs per chunk: 4 (16-bit) or 2 (fp32)
    constexpr int STG = (32 * CH) / LOGITS_THREADS > 0 ? (32 * CH) / LOGITS_THREADS : 1;   // staging registers (uint4) per THREAD per tile
    constexpr int RPI = LOGITS_THREADS / CH; // key rows covered by one staging step of the workgroup
    constexpr int ICH = D / 8;               // 16-byte chunks of one lane's fp32 A fragment (D/2 values)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: [4 Q images (one per wave = per M-tile): ICH chunks x 64 lanes x 16 B] [ONE tile of 32 keys x ROWP, shared]
    constexpr int IMG_BYTES = FAST ? 0 : 64 * ICH * 16;  // the FAST scan keeps its packed A fragments in registers
    constexpr int NSTEP = FAST ? D / 16 : ICH;            // MFMA steps per tile: D/16 (packed 16-deep) or one per A chunk
    char* const buf = smem + LOGITS_WAVES * IMG_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: tile indices and bases stay in SGPRs
    const uint32_t psel = lane_sel<DT>(kh);
    const int b = blockIdx.y / a.n_kv_heads, g = blockIdx.y % a.n_kv_heads;
    const int L = a.q_len, W = WV > 0 ? WV : a.window, G = a.group;
    const int rows = G * W;                   // query rows sharing this KV head
    const int n_t = (L + 31) / 32;            // 32-key tiles of this head
    const int mt = wave;                      // this wave's M-tile: query rows 32 * wave .. + 31 (rows == 128)
    char* const img = smem + wave * IMG_BYTES;
    const float sqrt_d = a.sqrt_d;
    KVC_STAMP(0);

    const char* const kbase = reinterpret_cast<const char*>(vw.k) +
                              ((int64_t)b * a.k_stride_b + (int64_t)g * a.k_stride_h) * ES;
    const int64_t tile_bytes = (int64_t)32 * a.k_stride_l * ES;    // one tile further along the key axis
    uint32_t koff[STG];                                            // this lane's chunks inside a tile (global side)
#pragma unroll
    for (int it = 0; it < STG; ++it) {
        const int c = it * LOGITS_THREADS + tid;
        koff[it] = (uint32_t)((c / CH) * (a.k_stride_l * ES) + (c % CH) * 16);
    }
    auto issue = [&](int tile, uint4 (&st)[STG]) {
        const char* const tb = kbase + tile * tile_bytes;          // uniform
        if (tile * 32 + 32 <= L) {
#pragma unroll
            for (int it = 0; it < STG; ++it) st[it] = *reinterpret_cast<const uint4*>(tb + koff[it]);
        } else {
#pragma unroll
            for (int it = 0; it < STG; ++it) {
                const int key = tile * 32 + it * RPI + tid / CH;
                st[it] = key < L ? *reinterpret_cast<const uint4*>(tb + koff[it]) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    char* const cdst = buf + (tid / CH) * ROWP + (tid % CH) * 16;     // LDS side of the same chunks
    auto commit = [&](const uint4 (&st)[STG]) {
#pragma unroll
        for (int it = 0; it < STG; ++it) {
            if constexpr (ASM_B) {
                uint32_t* d = reinterpret_cast<uint32_t*>(cdst + it * (RPI * ROWP));
                d[0] = st[it].x; d[1] = st[it].y; d[2] = st[it].z; d[3] = st[it].w;
            } else {
                *reinterpret_cast<uint4*>(cdst + it * (RPI * ROWP)) = st[it];
            }
        }
    };

    {
        // ---- A operand: every wave converts the 32 query rows of ITS M-tile ONCE into an fp32 image in LDS, laid
        // out as the MFMA A-fragment of every lane (lane = row + 32*parity, value s = Q[row][2s + parity]), chunk-major
        // ([chunk][lane][4 values]) so that the waves read chunk c at one per-lane address + c*1024.
        uint4 st[STG], st2[STG];                           // the next tile and the one after (two tiles ahead: a workgroup has
        int tile = blockIdx.x;                             // one tile in LDS, and a tile's MFMAs last about one memory latency)
        const int stride_t = gridDim.x;
        if (tile < n_t) issue(tile, st);
        if (tile + stride_t < n_t) issue(tile + stride_t, st2);
        uint4 aq[FAST ? NSTEP : 1];                        // FAST: this lane's packed A fragments, chunk 2*s + kh of its row
        if constexpr (FAST) {
            const int i = mt * 32 + j;
            const bool valid = i < rows;
            const int hq = g * G + (valid ? i / W : 0), w = valid ? i % W : 0;
            const char* qrow = reinterpret_cast<const char*>(vw.q) +
                ((int64_t)b * a.q_stride_b + (int64_t)hq * a.q_stride_h + (int64_t)(L - W + w) * a.q_stride_l) * ES;
#pragma unroll
            for (int sI = 0; sI < NSTEP; ++sI)
                aq[sI] = valid ? *reinterpret_cast<const uint4*>(qrow + (2 * sI + kh) * 16) : make_uint4(0, 0, 0, 0);
        }
        if constexpr (!FAST) {
            constexpr int PER_ROW = ROWB / 16;             // 16-byte pieces of one query row
            for (int pc = lane; pc < 32 * PER_ROW; pc += 64) {
                const int r = pc / PER_ROW, cc = pc % PER_ROW;
                const int i = mt * 32 + r;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (i < rows) {
                    const int hq = g * G + i / W, w = i % W;
                    const char* qrow = reinterpret_cast<const char*>(vw.q) +
                        ((int64_t)b * a.q_stride_b + (int64_t)hq * a.q_stride_h + (int64_t)(L - W + w) * a.q_stride_l) * ES;
                    v = *reinterpret_cast<const uint4*>(qrow + cc * 16);
                }
                // this piece holds pairs s = cc*PAIRS .. cc*PAIRS+PAIRS-1; element s of lane-row l sits in chunk s/4
#pragma unroll
                for (int sp = 0; sp < PAIRS; ++sp) {
                    const int sidx = cc * PAIRS + sp;
                    const int chunk = sidx >> 2, e = sidx & 3;
                    *reinterpret_cast<float*>(img + chunk * 1024 + r * 16 + e * 4) = pick<DT>(v, sp, lane_sel<DT>(0));
                    *reinterpret_cast<float*>(img + chunk * 1024 + (32 + r) * 16 + e * 4) = pick<DT>(v, sp, lane_sel<DT>(1));
                }
            }
        }
        if (tile < n_t) commit(st);
        __syncthreads();                                   // images and the first tile are in place
        KVC_STAMP(1);
        float rm[16];                          // running maximum per accumulator register (row 8*(e/4) + 4*kh + e%4)
#pragma unroll
        for (int e = 0; e < 16; ++e) rm[e] = -__builtin_inff();
        const char* const krow = buf + j * ROWP;
        const char* const arow = img + lane * 16;
        const uint32_t krow_a = lds_addr(krow) + 2 * kh, arow_a = lds_addr(arow);
        // logits of this lane: element (rg, e) of key `key` goes to lg + soff[rg] + key*W*ES (+ e*ES)
        char* const lg = reinterpret_cast<char*>(vw.logits);
        uint32_t soff[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int i0 = mt * 32 + 8 * rg + 4 * kh;
            const int hq = g * G + (i0 < rows ? i0 / W : 0), w0 = i0 % W;
            soff[rg] = (uint32_t)(((((int64_t)b * a.n_q_heads + hq) * L) * W + w0) * ES);
        }

        // One epilogue element: the reference's three roundings (+ the local causal mask on the last W keys).
        auto finish = [&](float accv, int i, int key, bool tail) -> float {
            float v = rnd<DT>(accv);
            v = rnd<DT>(ScaleDiv<D>::apply(v, sqrt_d));
            if (tail) {
                const int w = i % W;
                if (key >= L - W && (key - (L - W)) > w) v = rnd<DT>(v + Dt<DT>::finfo_min());
            }
            return v;
        };
        // Store 4 consecutive rows (one accumulator register group) of one key: 8 bytes (16-bit) or 16 bytes (fp32).
        auto store4 = [&](const float (&x)[4], int i0, int key) {
            const int hq = g * G + i0 / W, w0 = i0 % W;
            raw* dst = reinterpret_cast<raw*>(vw.logits) + (((int64_t)b * a.n_q_heads + hq) * L + key) * W + w0;
            if constexpr (ES == 2) {
                uint2 pk;
                pk.x = (uint32_t)Dt<DT>::st(x[0]) | ((uint32_t)Dt<DT>::st(x[1]) << 16);
                pk.y = (uint32_t)Dt<DT>::st(x[2]) | ((uint32_t)Dt<DT>::st(x[3]) << 16);
                *reinterpret_cast<uint2*>(dst) = pk;
            } else {
                *reinterpret_cast<float4*>(dst) = make_float4(x[0], x[1], x[2], x[3]);
            }
        };
        // per-row maximum over the 32 key lanes: reduce-scatter over the 5 key bits (16 cross-lane moves): after the
        // step on lane bit t each lane keeps only the registers whose index bit matches its own.
        auto fold_max = [&](const float (&xs)[16]) -> float {
            const bool b4 = (j & 16) != 0, b3 = (j & 8) != 0, b2 = (j & 4) != 0, b1 = (j & 2) != 0;
            float y[8], z[4], u[2];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float o = xor_lane<16>(b4 ? xs[r] : xs[r + 8]);
                const float keep = b4 ? xs[r + 8] : xs[r];
                y[r] = o > keep ? o : keep;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float o = xor_lane<8>(b3 ? y[r] : y[r + 4]);
                const float keep = b3 ? y[r + 4] : y[r];
                z[r] = o > keep ? o : keep;
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float o = xor_lane<4>(b2 ? z[r] : z[r + 2]);
                const float keep = b2 ? z[r + 2] : z[r];
                u[r] = o > keep ? o : keep;
            }
            float m = xor_lane<2>(b1 ? u[0] : u[1]);
            { const float keep = b1 ? u[1] : u[0]; m = m > keep ? m : keep; }
            { const float o = xor_lane<1>(m); m = o > m ? o : m; }
            return m;
        };
        // General epilogue (ragged tile, masked tail, padded rows, W % 4 != 0): straight after its own MFMAs.
        auto epilogue_general = [&](const f32x16& acc, int t) {
            const int key = t * 32 + j;
            const bool tail = t * 32 + 32 > L - W;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float x[4];
                const int i0 = mt * 32 + 8 * rg + 4 * kh;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[e] = finish(acc[rg * 4 + e], i0 + e, key, tail);
                    const float xv = (key < L) ? x[e] : -__builtin_inff();
                    rm[rg * 4 + e] = xv > rm[rg * 4 + e] ? xv : rm[rg * 4 + e];
                }
                if (i0 < rows && key < L) {
                    if ((W % 4) == 0) {
                        store4(x, i0, key);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int i = i0 + e;
                            if (i < rows) {
                                const int hq = g * G + i / W, w = i % W;
                                reinterpret_cast<raw*>(vw.logits)[(((int64_t)b * a.n_q_heads + hq) * L + key) * W + w] = Dt<DT>::st(x[e]);
                            }
                        }
                    }
                }
            }
        };
        // A tile is "plain" when none of those cases applies: its epilogue is then branch-free, a pair of accumulator
        // elements at a time on the packed-fp32 ALU.  (Riding it between the MFMA steps of the next tile, as earlier
        // versions did, buys nothing: the f32 MFMA and the VALU do not overlap on gfx950.)
        const bool rows_plain = (W % 4) == 0 && (mt + 1) * 32 <= rows;
        // Plain epilogue of accumulator elements e0, e0+1 (rows i0+e0%4, +1 of register group e0/4).
        uint32_t pkw[2];                                       // packed words of the current register group
        auto plain_pair = [&](const f32x16& pend, int e0, uint32_t pkeyoff) {     // pend: the tile's accumulators
            if constexpr (DT == KVC_FP32) {
                const float v0 = ScaleDiv<D>::apply_in_guard(pend[e0], sqrt_d), v1 = ScaleDiv<D>::apply_in_guard(pend[e0 + 1], sqrt_d);
                rm[e0] = v0 > rm[e0] ? v0 : rm[e0];
                rm[e0 + 1] = v1 > rm[e0 + 1] ? v1 : rm[e0 + 1];
                *reinterpret_cast<float2*>(lg + (soff[e0 >> 2] + pkeyoff + (e0 & 3) * 4)) = make_float2(v0, v1);
            } else {
                f32x2 x = {pend[e0], pend[e0 + 1]}, back;
                (void)pack2<DT>(x, back);                                         // first rounding
                const f32x2 q = scale2_in_guard<D>(back, sqrt_d);                 // second: after the scaling
                // (maximum before rounding: monotonic.)  One v_max_f32 each, spelled out: fmaxf() makes the compiler
                // re-canonicalise all 16 running maxima at the top of every tile
                asm("v_max_f32 %0, %1, %2" : "=v"(rm[e0]) : "v"(rm[e0]), "v"(q.x));
                asm("v_max_f32 %0, %1, %2" : "=v"(rm[e0 + 1]) : "v"(rm[e0 + 1]), "v"(q.y));
                f32x2 unused;
                pkw[(e0 >> 1) & 1] = pack2<DT>(q, unused);
                if ((e0 & 3) == 2) *reinterpret_cast<uint2*>(lg + (soff[e0 >> 2] + pkeyoff)) = make_uint2(pkw[0], pkw[1]);
            }
        };
        for (; tile < n_t; tile += stride_t) {
            const int next = tile + stride_t, next2 = tile + 2 * stride_t;
#pragma unroll
            for (int it = 0; it < STG; ++it) st[it] = st2[it];  // the tile requested one iteration ago
            if (next2 < n_t) issue(next2, st2);                // in flight during the MFMAs of this tile and the next
            __builtin_amdgcn_wave_barrier();
            KVC_STAMP(2);
            // ---- 32 rows x 32 keys, chain over d = 0..D-1 in order ----
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            auto chain = [&]() {
                if constexpr (ASM_B) {
                    // operands by hand-issued LDS reads, one step ahead of the MFMAs that consume them:
                    //   A: 4 fragment values (ds_read_b128, chunk-major image);  B: 4 x bf16 -> fp32 in the load itself.
                    asm volatile("" ::: "memory");            // this tile's ds_writes (commit) stay above the reads
                    f32x4 A0, A1;
                    uint32_t B0[4], B1[4];
                    ld_step<0>(A0, B0, arow_a, krow_a);
                    static_for<0, NSTEP>([&](auto ic_) {
                        constexpr int ic = decltype(ic_)::value;
                        f32x4& Ac = (ic & 1) ? A1 : A0;
                        uint32_t (&Bc)[4] = (ic & 1) ? B1 : B0;
                        if constexpr (ic + 1 < NSTEP) {
                            ld_step<ic + 1>((ic & 1) ? A0 : A1, (ic & 1) ? B0 : B1, arow_a, krow_a);
                            wait_step<5>(Ac, Bc);             // all but the 5 reads just issued have landed
                        } else {
                            wait_step<0>(Ac, Bc);
                        }
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[s4], u2f(Bc[s4]), acc, 0, 0, 0);
                    });
                    asm volatile("" ::: "memory");            // ... and the next tile's ds_writes stay below them
                } else {
                    static_for<0, NSTEP>([&](auto ic_) {
                        constexpr int sti = decltype(ic_)::value;
                        if constexpr (FAST) {
                            const uint4 kv = *reinterpret_cast<const uint4*>(krow + (2 * sti + kh) * 16);
                            acc = mfma16<DT>(aq[sti], kv, acc);
                        } else {                               // one A chunk = 4 fragment values = 4 exact f32 MFMAs
                            const float4 av = *reinterpret_cast<const float4*>(arow + sti * 1024);
                            const float af[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                            for (int kc = 0; kc < 4 / PAIRS; ++kc) {   // K chunks feeding these 4 values: 1 (16-bit) or 2 (fp32)
                                const int c = sti * (4 / PAIRS) + kc;
                                const uint4 kv = *reinterpret_cast<const uint4*>(krow + c * 16);
#pragma unroll
                                for (int sp = 0; sp < PAIRS; ++sp)
                                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kc * PAIRS + sp], pick<DT>(kv, sp, psel), acc, 0, 0, 0);
                            }
                        }
                    });
                }
            };
            chain();
            asm volatile("" :: "v"(acc[0]), "v"(acc[15]));
            // every wave has its operands of this tile in registers: the next tile may overwrite the shared buffer, its
            // ds_writes overlap the VALU work that follows
            __syncthreads();
            if (next < n_t) commit(st);
            KVC_STAMP(3);
            bool plain = rows_plain && tile * 32 + 32 <= L - W;
            if (plain) {
                // the branch-free scaling needs every |value| of the tile inside [2^-98, 2^126] (rounding to dtype moves a
                // value by < 1 %, so this keeps the rounded value inside the proven [2^-100, inf) guard); a zero logit, an
                // overflow or a NaN sends the whole tile down the general path instead
                // (v_min3_f32 / v_max3_f32 take |x| as a source modifier: 16 instructions where integer compares on the masked bits
                // took 48.  They skip a NaN — which the packed path turns into a NaN like the general one, and NaN scores all
                // order alike; an infinity is caught by the maximum.)
                float amin = __builtin_fminf(__builtin_fabsf(acc[0]), __builtin_fabsf(acc[1])), amax = __builtin_fmaxf(__builtin_fabsf(acc[0]), __builtin_fabsf(acc[1]));
#pragma unroll
                for (int e = 2; e < 16; e += 2) {
                    amin = __builtin_fminf(__builtin_fminf(__builtin_fabsf(acc[e]), __builtin_fabsf(acc[e + 1])), amin);
                    amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(acc[e]), __builtin_fabsf(acc[e + 1])), amax);
                }
                plain = !__any(!(amin >= u2f(0x0e800000u) && amax <= u2f(0x7e800000u)));
            }
            if (plain) {                                        // branch-free epilogue, two accumulator elements at a time
                const uint32_t keyoff = (uint32_t)(tile * 32 + j) * (uint32_t)(W * ES);
#pragma unroll
                for (int pr = 0; pr < 8; ++pr) plain_pair(acc, 2 * pr, keyoff);
            } else {
                epilogue_general(acc, tile);
            }
            __syncthreads();                                  // the next tile is complete
            KVC_STAMP(5);
        }
        // ---- this wave's maximum per row of its M-tile -> pmax[hq][blockIdx.x][w] ----
        {
            float rmr[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) rmr[e] = rnd<DT>(rm[e]);  // plain tiles kept the unrounded value (monotonic)
            const float runmax = fold_max(rmr);
            const int R = ((j >> 1) & 1) + ((j >> 2) & 1) * 2 + ((j >> 3) & 1) * 4 + ((j >> 4) & 1) * 8;   // accumulator register
            if ((j & 1) == 0) {
                const int i = mt * 32 + (R & 3) + 8 * (R >> 2) + 4 * kh;
                if (i < rows) {
                    const int hq = g * G + i / W, w = i % W;
                    vw.pmax[(((int64_t)b * a.n_q_heads + hq) * a.n_tiles + blockIdx.x) * W + w] = runmax;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Row maxima of one head from the tile maxima: result in LDS m[0..W).  >= 256 threads (the first 256 work).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_row_max(const float* pmax /*[n_tiles][W]*/, int n_tiles, int W,
                                               float* m /*LDS [W]*/, float* scratch /*LDS [256]*/) {
    const int tid = threadIdx.x;
    int Wp = 1;
    while (Wp < W) Wp <<= 1;               // W <= 64
    const int parts = 256 / Wp, w = tid % Wp, part = tid / Wp;
    float v = -__builtin_inff();
    if (w < W && tid < 256)
        for (int t = part; t < n_tiles; t += parts) { const float o = pmax[(int64_t)t * W + w]; v = o > v ? o : v; }
    if (tid < 256) scratch[tid] = v;
    __syncthreads();
    if (tid < W) {
        float r = scratch[tid];
        for (int p = 1; p < parts; ++p) { const float o = scratch[p * Wp + tid]; r = o > r ? o : r; }
        m[tid] = r;
    }
    __syncthreads();
}

// Load the W logits of one key (contiguous W*ES bytes) widened to fp32.  WV > 0: W is the
// compile-time constant WV (multiple of 8) and x[] stays in registers; WV == 0: runtime W <= 64.
template <int DT, int WV>
__device__ __forceinline__ void load_logits(const typename Dt<DT>::raw* src, int W, float* x /*[WV or 64]*/) {
    constexpr int ES = Dt<DT>::esize;
    constexpr int PER16 = 16 / ES;
    if (WV > 0 || (W % PER16) == 0) {
#pragma unroll
        for (int c = 0; c < (WV > 0 ? WV : W) / PER16; ++c) {
            const uint4 v = reinterpret_cast<const uint4*>(src)[c];
            const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (ES == 2) {
                    x[c * 8 + 2 * e] = Dt<DT>::ld((uint16_t)(wd[e] & 0xffffu));
                    x[c * 8 + 2 * e + 1] = Dt<DT>::ld((uint16_t)(wd[e] >> 16));
                } else {
                    x[c * 4 + e] = u2f(wd[e]);
                }
            }
        }
    } else {
        for (int w = 0; w < W; ++w) x[w] = Dt<DT>::ld(src[w]);
    }
}

// The W logits of one key from 16-byte vectors already in registers (compile-time W, a multiple of 16 bytes).
template <int DT, int WV>
__device__ __forceinline__ void widen_logits(const uint4 (&v)[(WV * Dt<DT>::esize) / 16 > 0 ? (WV * Dt<DT>::esize) / 16 : 1], float (&x)[WV]) {
    constexpr int ES = Dt<DT>::esize, PER16 = 16 / ES;
#pragma unroll
    for (int c = 0; c < WV / PER16; ++c) {
        const uint32_t wd[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (ES == 2) {
                x[c * 8 + 2 * e] = Dt<DT>::ld((uint16_t)(wd[e] & 0xffffu));
                x[c * 8 + 2 * e + 1] = Dt<DT>::ld((uint16_t)(wd[e] >> 16));
            } else {
                x[c * 4 + e] = u2f(wd[e]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Softmax denominators in torch's own order (aten vec::reduce_all on 16 fp32 lanes, the order of softmax's row sum on
// the reference's AVX-512 host — oracle: sum_torch16): chain l, l = 0..15, adds the exponentials of keys l, l+16, l+32, ...
// one after the other; the 16 chains are folded by an xor butterfly 8, 4, 2, 1.  (L < 16: one plain left-to-right sum.)
// A chain is strictly serial over the whole row, so one workgroup owns a head: its 1024 threads compute the
// exponentials of 1024 consecutive keys, stage them in LDS, and thread c < 16 * W — row c / 16, chain c % 16 — adds its
// 64 terms of the stage in key order.  Two stage buffers: one barrier per 1024 keys.
// With it the C5 fixtures (Mistral 32k -> 2048) select the reference's indices in 32 of 32 heads (round 1's own
// 256-key-chunk order: 31 of 32; 126 -> 20 of 1 023 744 pooled scores off the reference).
// ---------------------------------------------------------------------------------------------
constexpr int SP_THREADS = 1024;
constexpr int EPITCH = SP_THREADS + 16;                      // stage row pitch (floats): rows 16 banks apart
constexpr int ESTAGE_ROWS = 8;                               // rows staged at a time
constexpr int ESTAGE = ESTAGE_ROWS * EPITCH;                 // floats per stage buffer

// acc += the terms of chain `cl` inside the stage that holds keys [base, base + 1024) of one row: 64 dependent adds, their
// operands fetched eight at a time one batch ahead (the adds are the serial part of the whole softmax).
__device__ __forceinline__ float chain16_add(float acc, const float* row /* stage row + cl */, int base, int cl, int L) {
    const int cnt = (L - base - cl + 15) >> 4;               // keys base + cl + 16 i < L
    if (cnt >= 64) {
        float va[8], vb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) va[i] = row[16 * i];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            float (&cur)[8] = (b & 1) ? vb : va;
            float (&nxt)[8] = (b & 1) ? va : vb;
            if (b + 1 < 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) nxt[i] = row[16 * (8 * (b + 1) + i)];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc = acc + cur[i];
        }
    } else {
        for (int i = 0; i < cnt; ++i) acc = acc + row[16 * i];
    }
    return acc;
}
// After the last stage: fold the 16 chains of each row; lanes with cl == 0 hold the row sum.
__device__ __forceinline__ float chain16_fold(float acc) {
    acc = acc + xor_lane<8>(acc);
    acc = acc + xor_lane<4>(acc);
    acc = acc + xor_lane<2>(acc);
    acc = acc + xor_lane<1>(acc);
    return acc;
}

// Row sums of one head.  E(it, w) -> exponential of key it * 1024 + tid, row w (0 for keys >= L).  Returns, in the threads
// c = 16 w (w < W), the sum of row w.  stage: 2 * ESTAGE floats of LDS.  All 1024 threads must call it.
template <class EF>
__device__ __forceinline__ float torch16_rowsums(EF&& E, int L, int W, float* stage) {
    const int tid = threadIdx.x;
    const int cw = tid >> 4, cl = tid & 15;
    const int iters = (L + SP_THREADS - 1) / SP_THREADS;
    const int groups = (W + ESTAGE_ROWS - 1) / ESTAGE_ROWS;
    float acc = 0.0f;                                        // 0 + x == x: the chain starts with its first term
    int fill = 0;
    if (L < 16) {                                            // vec::reduce_all's scalar path: one sum, left to right
        for (int g = 0; g < groups; ++g) {
            float* buf = stage + (fill & 1) * ESTAGE;
            for (int r = 0; r < ESTAGE_ROWS; ++r) if (g * ESTAGE_ROWS + r < W) buf[r * EPITCH + tid] = E(0, g * ESTAGE_ROWS + r);
            __syncthreads();
            if (cl == 0 && (cw / ESTAGE_ROWS) == g && cw < W)
                for (int k = 0; k < L; ++k) acc = acc + buf[(cw % ESTAGE_ROWS) * EPITCH + k];
            ++fill;
        }
        return acc;
    }
    for (int it = 0; it < iters; ++it) {
        for (int g = 0; g < groups; ++g) {
            float* buf = stage + (fill & 1) * ESTAGE;
#pragma unroll
            for (int r = 0; r < ESTAGE_ROWS; ++r)
                if (g * ESTAGE_ROWS + r < W) buf[r * EPITCH + tid] = E(it, g * ESTAGE_ROWS + r);
            __syncthreads();                                 // (the buffer filled two stages ago has been read: its
            if ((cw / ESTAGE_ROWS) == g && cw < W)           //  readers passed the previous barrier after reading)
                acc = chain16_add(acc, buf + (cw % ESTAGE_ROWS) * EPITCH + cl, it * SP_THREADS, cl, L);
            ++fill;
        }
    }
    return chain16_fold(acc);
}

// ---------------------------------------------------------------------------------------------
// rowsum16_kernel (per-layer calls: few heads in flight): grid = (bsz*n_q_heads, items), block = 1024.
// Row maxima from the tile maxima, then the denominators above; writes rowmax[hb][w], rowsum[hb][w].
// ---------------------------------------------------------------------------------------------
template <int DT, int WV>
__global__ __launch_bounds__(SP_THREADS) void rowsum16_kernel(const ScoreArgs a) {
    const ScoreView vw = view_of(a, blockIdx.y);
    typedef typename Dt<DT>::raw raw;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const m = reinterpret_cast<float*>(smem);              // [64]
    float* const scratch = m + 64;                                // [256]
    float* const stage = scratch + 256;                           // [2][ESTAGE]
    const int tid = threadIdx.x, hb = blockIdx.x;
    const int L = a.q_len, W = WV > 0 ? WV : a.window;
    const raw* const lg = reinterpret_cast<const raw*>(vw.logits) + (int64_t)hb * L * W;
    block_row_max(vw.pmax + (int64_t)hb * a.n_tiles * W, a.n_tiles, W, m, scratch);
    float x[WV > 0 ? WV : 64];
    int have = -1;                                                // iteration whose logits x[] holds
    auto E = [&](int it, int w) -> float {
        const int key = it * SP_THREADS + tid;
        if (key >= L) return 0.0f;
        if (have != it) { load_logits<DT, WV>(lg + (int64_t)key * W, W, x); have = it; }
        return exp_u20(x[w] - m[w]);
    };
    const float sum = torch16_rowsums(E, L, W, stage);
    if ((tid & 15) == 0 && (tid >> 4) < W) {
        vw.rowmax[(int64_t)hb * W + (tid >> 4)] = m[tid >> 4];
        vw.rowsum[(int64_t)hb * W + (tid >> 4)] = sum;
    }
}

// ---------------------------------------------------------------------------------------------
// pool_kernel: grid = (ceil(n/256), bsz*n_q_heads), block = 256: p = round(e * (1/sum)), window sum (cascade), round,
// pooling — for 256 keys (+ the pooling halo) of one head, from the row maxima / sums rowsum16_kernel left.
// ---------------------------------------------------------------------------------------------
template <int DT, int WV>
__global__ __launch_bounds__(256) void pool_kernel(const ScoreArgs a) {
    const ScoreView vw = view_of(a, blockIdx.z);
    typedef typename Dt<DT>::raw raw;
    __shared__ float m[64];
    __shared__ float rinv[64];
    __shared__ float s_tile[256 + 64];
    const int tid = threadIdx.x;
    const int hb = blockIdx.y;
    const int L = a.q_len, W = WV > 0 ? WV : a.window, n = L - W;
    const int pad = a.pooling == KVC_POOL_NONE ? 0 : a.kernel_size / 2;
    const int j0 = blockIdx.x * 256;
    if (tid < W) {
        m[tid] = vw.rowmax[(int64_t)hb * W + tid];
        rinv[tid] = 1.0f / vw.rowsum[(int64_t)hb * W + tid];
    }
    __syncthreads();

    for (int t = tid; t < 256 + 2 * pad; t += 256) {
        const int key = j0 - pad + t;
        float sv = 0.0f;
        if (key >= 0 && key < n) {
            float x[WV > 0 ? WV : 64];
            load_logits<DT, WV>(reinterpret_cast<const raw*>(vw.logits) + ((int64_t)hb * L + key) * W, W, x);
            CascadeSum cs;
            cs.init(W);
#pragma unroll
            for (int w = 0; w < (WV > 0 ? WV : W); ++w) {
                if constexpr (WV > 0) {
                    if ((w & 1) == 0) {
                        const f32x2 pr = exp_u20x2_nonpos(f32x2{x[w], x[w + 1]} - f32x2{m[w], m[w + 1]}) * f32x2{rinv[w], rinv[w + 1]};
                        x[w] = pr.x; x[w + 1] = pr.y;
                    }
                    cs.add(rnd<DT>(x[w]));
                } else {
                    cs.add(rnd<DT>(exp_u20(x[w] - m[w]) * rinv[w]));
                }
            }
            // .sum(dim=-2) rounds the fp32 cascade once; .mean(dim=-2) (AdaKV / HeadKV) is cast_fp32 -> sum -> div_(W) -> cast
            sv = rnd<DT>(a.window_mean ? cs.result() / (float)W : cs.result());
        }
        s_tile[t] = sv;
    }
    __syncthreads();

    const int jo = j0 + tid;
    if (jo < n) {
        float c;
        if (a.pooling == KVC_POOL_NONE) {
            c = s_tile[tid];
        } else {
            const int lo = jo - pad < 0 ? 0 : jo - pad;
            const int hi = jo - pad + a.kernel_size > n ? n : jo - pad + a.kernel_size;
            if (a.pooling == KVC_POOL_MAX) {
                c = -__builtin_inff();
                for (int i = lo; i < hi; ++i) { const float v = s_tile[i - j0 + pad]; c = v > c ? v : c; }
            } else {
                float acc = 0.0f;
                for (int i = lo; i < hi; ++i) acc = acc + s_tile[i - j0 + pad];
                c = rnd<DT>(acc / (float)a.kernel_size);
            }
        }
        reinterpret_cast<raw*>(vw.scores)[(int64_t)hb * n + jo] = Dt<DT>::st(c);
    }
}

// ---------------------------------------------------------------------------------------------
// softmax_pool_kernel: rowsum16_kernel + pool_kernel for one (head, item) in ONE workgroup of 1024 threads — the same
// arithmetic in the same order, so the bits are those of the two-kernel path.  Thread t owns keys t, t+1024, ...  With
// KEEP > 0 the exponentials of the first pass stay in registers (KEEP iterations x WV rows) and are not recomputed for
// the normalisation.  The per-key window sums go through a ring of four 1024-key segments in LDS so that pooling reads
// its halo from the neighbouring segments (one barrier per iteration, nothing recomputed); the ring shares its LDS
// with the exponential stages of the first pass.
// grid = (bsz*n_q_heads, items), block = 1024.  LDS: 1.5 KB + 2 stage buffers (65 KB).
// ---------------------------------------------------------------------------------------------
template <int DT, int WV, int KEEP>
__global__ __launch_bounds__(SP_THREADS) void softmax_pool_kernel(const ScoreArgs a) {
    const ScoreView vw = view_of(a, blockIdx.y);
    typedef typename Dt<DT>::raw raw;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int W = WV;
    const int tid = threadIdx.x;
    const int hb = blockIdx.x;
    const int L = a.q_len, n = L - W;
    const int iters = (L + SP_THREADS - 1) / SP_THREADS;
    float* const m = reinterpret_cast<float*>(smem);              // [64]
    float* const rinv = m + 64;                                   // [64]
    float* const scratch = rinv + 64;                             // [256]
    float* const stage = scratch + 256;                           // [2][ESTAGE] (pass 1)
    float* const seg = stage;                                     // [4][1024]   (pass 2; 4096 <= 2 * ESTAGE)
    const raw* const lg = reinterpret_cast<const raw*>(vw.logits) + (int64_t)hb * L * W;

    // the head's logits are requested BEFORE the row maxima are fetched and reduced: two dependent trips to memory (tile
    // maxima, then logits) were a third of a workgroup's lifetime
    KVC_SPSTAMP(0);
    constexpr int VPK = (W * (int)sizeof(raw)) / 16;               // 16-byte vectors per key
    uint4 rawx[KEEP > 0 ? KEEP : 1][VPK > 0 ? VPK : 1];
    if constexpr (KEEP > 0) {
#pragma unroll
        for (int it = 0; it < KEEP; ++it) {
            const int key = it * SP_THREADS + tid;
#pragma unroll
            for (int c = 0; c < VPK; ++c)
                rawx[it][c] = (it < iters && key < L) ? reinterpret_cast<const uint4*>(lg + (int64_t)key * W)[c] : make_uint4(0, 0, 0, 0);
        }
    }
    block_row_max(vw.pmax + (int64_t)hb * a.n_tiles * W, a.n_tiles, W, m, scratch);
    KVC_SPSTAMP(1);
    float mr[W];
#pragma unroll
    for (int w = 0; w < W; ++w) mr[w] = m[w];

    // pass 1: exponentials (kept in registers when they fit) and the row sums in torch's order
    float e[KEEP > 0 ? KEEP : 1][W];
    float sum;
    if constexpr (KEEP > 0) {
        float x[KEEP][W];
#pragma unroll
        for (int it = 0; it < KEEP; ++it) widen_logits<DT, W>(rawx[it], x[it]);
#pragma unroll
        for (int it = 0; it < KEEP; ++it) {
            const int key = it * SP_THREADS + tid;
#pragma unroll
            for (int w = 0; w < W; w += 2) {
                const f32x2 ex = exp_u20x2_nonpos(f32x2{x[it][w], x[it][w + 1]} - f32x2{mr[w], mr[w + 1]});
                e[it][w] = (it < iters && key < L) ? ex.x : 0.0f; e[it][w + 1] = (it < iters && key < L) ? ex.y : 0.0f;
            }
        }
        KVC_SPSTAMP(2);
        // (a compile-time it / w inside the staging loop: the kept values are addressed as registers)
        const int cw = tid >> 4, cl = tid & 15;
        constexpr int groups = (W + ESTAGE_ROWS - 1) / ESTAGE_ROWS;
        float acc = 0.0f;
        if (L < 16) {
            sum = torch16_rowsums([&](int, int w) -> float { float r = 0.0f;
#pragma unroll
                                                             for (int ww = 0; ww < W; ++ww) r = ww == w ? e[0][ww] : r;
                                                             return r; }, L, W, stage);
        } else {
            int fill = 0;
#pragma unroll
            for (int it = 0; it < KEEP; ++it) {
                if (it < iters) {
#pragma unroll
                    for (int g = 0; g < groups; ++g) {
                        float* buf = stage + (fill & 1) * ESTAGE;
#pragma unroll
                        for (int r = 0; r < ESTAGE_ROWS; ++r)
                            if (g * ESTAGE_ROWS + r < W) buf[r * EPITCH + tid] = e[it][(g * ESTAGE_ROWS + r) < W ? (g * ESTAGE_ROWS + r) : 0];
                        __syncthreads();
                        if (it == 1) KVC_SPSTAMP(6);
                        if ((cw / ESTAGE_ROWS) == g && cw < W)
                            acc = chain16_add(acc, buf + (cw % ESTAGE_ROWS) * EPITCH + cl, it * SP_THREADS, cl, L);
                        if (it == 1) KVC_SPSTAMP(7);
                        ++fill;
                    }
                }
            }
            sum = chain16_fold(acc);
        }
    } else {
        float x[W];
        int have = -1;
        auto E = [&](int it, int w) -> float {
            const int key = it * SP_THREADS + tid;
            if (key >= L) return 0.0f;
            if (have != it) {
                load_logits<DT, W>(lg + (int64_t)key * W, W, x);
#pragma unroll
                for (int ww = 0; ww < W; ww += 2) {
                    const f32x2 ex = exp_u20x2_nonpos(f32x2{x[ww], x[ww + 1]} - f32x2{mr[ww], mr[ww + 1]});
                    x[ww] = ex.x; x[ww + 1] = ex.y;
                }
                have = it;
            }
            float r = 0.0f;
#pragma unroll
            for (int ww = 0; ww < W; ++ww) r = ww == w ? x[ww] : r;
            return r;
        };
        sum = torch16_rowsums(E, L, W, stage);
    }
    if ((tid & 15) == 0 && (tid >> 4) < W) {
        rinv[tid >> 4] = 1.0f / sum;
        vw.rowmax[(int64_t)hb * W + (tid >> 4)] = m[tid >> 4];
        vw.rowsum[(int64_t)hb * W + (tid >> 4)] = sum;
    }
    __syncthreads();                                              // rinv visible; the stages are free for the ring
    KVC_SPSTAMP(3);
    float ri[W];
#pragma unroll
    for (int w = 0; w < W; ++w) ri[w] = rinv[w];

    // pass 2: p = round(e / sum), window sum (cascade), round; pooling one iteration behind through the segment ring
    const int pad = a.pooling == KVC_POOL_NONE ? 0 : a.kernel_size / 2;
    raw* const out = reinterpret_cast<raw*>(vw.scores) + (int64_t)hb * n;
    // The per-key window sums go to LDS: all of them at once when the row fits the stage area (up to 16 640 keys: ONE barrier
    // between the sums and the pooling), else through a ring of four 1024-key segments, pooling one iteration behind.
    const bool flat = iters * SP_THREADS <= 2 * ESTAGE;
    const int ring = flat ? ~0 : (4 * SP_THREADS - 1);
    auto pool_iteration = [&](int itp) {
        const int jo = itp * SP_THREADS + tid;
        if (jo >= n) return;
        float c;
        if (a.pooling == KVC_POOL_NONE) {
            c = seg[jo & ring];
        } else {
            const int lo = jo - pad < 0 ? 0 : jo - pad;
            const int hi = jo - pad + a.kernel_size > n ? n : jo - pad + a.kernel_size;
            if (a.pooling == KVC_POOL_MAX) {
                c = -__builtin_inff();
                for (int i = lo; i < hi; ++i) { const float v = seg[i & ring]; c = v > c ? v : c; }
            } else {
                float acc = 0.0f;
                for (int i = lo; i < hi; ++i) acc = acc + seg[i & ring];
                c = rnd<DT>(acc / (float)a.kernel_size);
            }
        }
        out[jo] = Dt<DT>::st(c);
    };
    auto window_sum = [&](const float (&ev)[W]) {
        CascadeSum cs;
        cs.init(W);
#pragma unroll
        for (int w = 0; w < W; w += 2) {
            const f32x2 pr = f32x2{ev[w], ev[w + 1]} * f32x2{ri[w], ri[w + 1]};
            cs.add(rnd<DT>(pr.x));
            cs.add(rnd<DT>(pr.y));
        }
        return rnd<DT>(a.window_mean ? cs.result() / (float)W : cs.result());
    };
    if constexpr (KEEP > 0) {
        if (flat) {
#pragma unroll
            for (int it = 0; it < KEEP; ++it)
                if (it < iters) {
                    const int key = it * SP_THREADS + tid;
                    seg[key] = key < n ? window_sum(e[it]) : 0.0f;
                }
            __syncthreads();
            KVC_SPSTAMP(4);
            for (int it = 0; it < iters; ++it) pool_iteration(it);
            KVC_SPSTAMP(5);
        } else {
#pragma unroll
            for (int it = 0; it <= KEEP; ++it) {
                if (it <= iters) {
                    if (it < iters) {
                        const int key = it * SP_THREADS + tid;
                        seg[(it & 3) * SP_THREADS + tid] = key < n ? window_sum(e[it < KEEP ? it : 0]) : 0.0f;
                    }
                    __syncthreads();
                    if (it >= 1) pool_iteration(it - 1);
                }
            }
        }
    } else {
        for (int it = 0; it <= iters; ++it) {
            if (it < iters) {
                const int key = it * SP_THREADS + tid;
                float sv = 0.0f;
                if (key < n) {
                    float x[W], ev[W];
                    load_logits<DT, W>(lg + (int64_t)key * W, W, x);
#pragma unroll
                    for (int w = 0; w < W; w += 2) {
                        const f32x2 ex = exp_u20x2_nonpos(f32x2{x[w], x[w + 1]} - f32x2{mr[w], mr[w + 1]});
                        ev[w] = ex.x; ev[w + 1] = ex.y;
                    }
                    sv = window_sum(ev);
                }
                seg[flat ? key : (it & 3) * SP_THREADS + tid] = sv;
            }
            if (!flat) {
                __syncthreads();
                if (it >= 1) pool_iteration(it - 1);
            }
        }
        if (flat) {
            __syncthreads();
            for (int it = 0; it < iters; ++it) pool_iteration(it);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// softmax_pool_ws_kernel (round 3, W = 8, rows up to 8 064 keys): the same arithmetic in the same order as
// softmax_pool_kernel, with the workgroup's two jobs on DIFFERENT waves.  In softmax_pool_kernel the 128 threads that add
// the torch-order chains (16 chains x 8 rows) also own keys: their two waves alternate between a stage's 64 dependent
// adds and the next stage's exponentials while the other fourteen waves wait at the barrier (tools/softmax_stamps.py:
// 12.3 of a workgroup's 23.6 us are the 8 stages).  Here waves 0-1 only add chains and waves 2-15 (896 = 56 x 16 threads)
// own the keys: a stage's chain adds run beside the next stage's exponentials.  Stage boundaries stay multiples of 16
// keys, so chain c still sums the keys = c (mod 16) in ascending order — bit-identical row sums
// (test_softmax_pool_forms_identical).  Measured: 97 -> 92 us per 32-layer C2 launch; what bounds the stage is the
// exponentials (14 instructions each, a dependent v_pk_fma chain at four waves per SIMD: 6 cycles per instruction) —
// computing all of a thread's exponentials before the first stage, for more independent chains, was measured and is
// slower (99 us: the chain waves then idle through that phase).
// grid = (bsz*n_q_heads, items), block = 1024.  LDS as softmax_pool_kernel.
// ---------------------------------------------------------------------------------------------
constexpr int WS_CHAIN = 128, WS_WORK = SP_THREADS - WS_CHAIN, WS_ITERS = 9;     // 9 x 896 = 8 064 keys
template <int DT>
__global__ __launch_bounds__(SP_THREADS) void softmax_pool_ws_kernel(const ScoreArgs a) {
    const ScoreView vw = view_of(a, blockIdx.y);
    typedef typename Dt<DT>::raw raw;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int W = 8;
    const int tid = threadIdx.x, hb = blockIdx.x;
    const bool worker = tid >= WS_CHAIN;
    const int wt = tid - WS_CHAIN;
    const int L = a.q_len, n = L - W;
    const int iters = (L + WS_WORK - 1) / WS_WORK;
    float* const m = reinterpret_cast<float*>(smem);              // [64]
    float* const rinv = m + 64;                                   // [64]
    float* const scratch = rinv + 64;                             // [256]
    float* const stage = scratch + 256;                           // [2][ESTAGE] (pass 1)
    float* const seg = stage;                                     // [<= 8064]   (pass 2)
    const raw* const lg = reinterpret_cast<const raw*>(vw.logits) + (int64_t)hb * L * W;
    constexpr int VPK = (W * (int)sizeof(raw)) / 16;
    uint4 rawx[WS_ITERS][VPK];
#pragma unroll
    for (int it = 0; it < WS_ITERS; ++it) {
        const int key = it * WS_WORK + wt;
#pragma unroll
        for (int c = 0; c < VPK; ++c)
            rawx[it][c] = (worker && it < iters && key < L) ? reinterpret_cast<const uint4*>(lg + (int64_t)key * W)[c] : make_uint4(0, 0, 0, 0);
    }
    block_row_max(vw.pmax + (int64_t)hb * a.n_tiles * W, a.n_tiles, W, m, scratch);
    float mr[W];
#pragma unroll
    for (int w = 0; w < W; ++w) mr[w] = m[w];
    // pass 1: the workers' exponentials (kept in registers) go to LDS stage by stage; the chain waves add them in torch's order
    float e[WS_ITERS][W];
    const int cw = tid >> 4, cl = tid & 15;                       // chain threads: row cw, chain cl
    float acc = 0.0f;
#pragma unroll
    for (int it = 0; it < WS_ITERS; ++it) {
        if (it < iters) {                                          // (uniform)
            const int key = it * WS_WORK + wt;
            float x[W];
            widen_logits<DT, W>(rawx[it], x);
#pragma unroll
            for (int w = 0; w < W; w += 2) {
                const f32x2 ex = exp_u20x2_nonpos(f32x2{x[w], x[w + 1]} - f32x2{mr[w], mr[w + 1]});
                e[it][w] = (worker && key < L) ? ex.x : 0.0f; e[it][w + 1] = (worker && key < L) ? ex.y : 0.0f;
            }
            float* buf = stage + (it & 1) * ESTAGE;
            if (worker) {
#pragma unroll
                for (int r = 0; r < W; ++r) buf[r * EPITCH + wt] = e[it][r];
            }
            __syncthreads();       // stage `it` is complete; the chain waves finished stage it - 1 before they arrived here, so
                                   // the workers may overwrite that buffer's twin (stage it + 1) while stage `it` is being added
            if (!worker) {
                const float* row = buf + cw * EPITCH + cl;
                const int cnt = (L - it * WS_WORK - cl + 15) >> 4;     // keys it * 896 + cl + 16 i < L
                if (cnt >= WS_WORK / 16) {
                    float va[8], vb[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) va[i] = row[16 * i];
#pragma unroll
                    for (int b = 0; b < WS_WORK / 16 / 8; ++b) {       // 7 blocks of 8 terms, the next block's reads in flight
                        float (&cur)[8] = (b & 1) ? vb : va;
                        float (&nxt)[8] = (b & 1) ? va : vb;
                        if (b + 1 < WS_WORK / 16 / 8) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) nxt[i] = row[16 * (8 * (b + 1) + i)];
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc = acc + cur[i];
                    }
                } else {
                    for (int i = 0; i < cnt; ++i) acc = acc + row[16 * i];
                }
            }
        }
    }
    const float sum = chain16_fold(acc);
    if (!worker && cl == 0) {
        rinv[cw] = 1.0f / sum;
        vw.rowmax[(int64_t)hb * W + cw] = m[cw];
        vw.rowsum[(int64_t)hb * W + cw] = sum;
    }
    __syncthreads();                                              // rinv visible; the stages are free for the per-key sums
    float ri[W];
#pragma unroll
    for (int w = 0; w < W; ++w) ri[w] = rinv[w];
    // pass 2: p = round(e / sum), window sum (cascade), round -> LDS; then pooling by all 1024 threads
#pragma unroll
    for (int it = 0; it < WS_ITERS; ++it) {
        if (it < iters && worker) {
            const int key = it * WS_WORK + wt;
            if (key < L) {
                CascadeSum cs;
                cs.init(W);
#pragma unroll
                for (int w = 0; w < W; w += 2) {
                    const f32x2 pr = f32x2{e[it][w], e[it][w + 1]} * f32x2{ri[w], ri[w + 1]};
                    cs.add(rnd<DT>(pr.x));
                    cs.add(rnd<DT>(pr.y));
                }
                seg[key] = key < n ? rnd<DT>(a.window_mean ? cs.result() / (float)W : cs.result()) : 0.0f;
            }
        }
    }
    __syncthreads();
    const int pad = a.pooling == KVC_POOL_NONE ? 0 : a.kernel_size / 2;
    raw* const out = reinterpret_cast<raw*>(vw.scores) + (int64_t)hb * n;
    for (int jo = tid; jo < n; jo += SP_THREADS) {
        float c;
        if (a.pooling == KVC_POOL_NONE) {
            c = seg[jo];
        } else {
            const int lo = jo - pad < 0 ? 0 : jo - pad;
            const int hi = jo - pad + a.kernel_size > n ? n : jo - pad + a.kernel_size;
            if (a.pooling == KVC_POOL_MAX) {
                c = -__builtin_inff();
                for (int i = lo; i < hi; ++i) { const float v = seg[i]; c = v > c ? v : c; }
            } else {
                float s2 = 0.0f;
                for (int i = lo; i < hi; ++i) s2 = s2 + seg[i];
                c = rnd<DT>(s2 / (float)a.kernel_size);
            }
        }
        out[jo] = Dt<DT>::st(c);
    }
}

// ---------------------------------------------------------------------------------------------
// softmax_rows16_kernel + softmax_comb16_kernel (round 3; W = 16, 32 or 64, 16-bit dtypes, 16 <= L <= 8 192): the logits are
// read ONCE.  softmax_pool_kernel cannot keep a head's exponentials at these windows (32 rows x 8 keys per thread) and reads
// the logits a second time for its normalising pass: 1.02 GB fetched for 524 MB of logits at C2 / W = 32 (round 2's PMC pass).
// The rows of the window are independent until torch's window sum, which is a cascade with 16-row leaves (SumKernel
// multi_row_sum, level step 16): sum over W rows = (((0 + P0) + P1) + ...) with Pk the in-order sum of rows 16 k .. 16 k + 15.
// So one workgroup takes 16 ROWS of a head: their logits stay in registers as they arrived (8 keys x 32 bytes per thread),
// the exponentials are recomputed for the second pass from those registers, and it writes Pk (fp32) to the workspace;
// softmax_comb16_kernel adds the W / 16 partials in order, rounds, pools and writes the scores.  Same arithmetic in the
// same order as softmax_pool_kernel (test_softmax_pool_forms_identical); debug_stage_mask bit 12 selects that kernel.
// The workgroups of one head sit 8 ids apart — on the same XCD, back to back — so the 64-byte (W = 32) key rows they share
// cache lines of are fetched from HBM once.
//   rows16: grid = (bsz*n_q_heads * W/16, items), block = 1024, LDS = kSoftmaxLds16.   comb16: grid = (ceil(n / 1024), bsz*n_q_heads, items).
// ---------------------------------------------------------------------------------------------
constexpr int R16_ITERS = 8;                                                       // 8 x 1024 keys
constexpr size_t kSoftmaxLds16 = (size_t)(64 + 64 + 256 + 2 * 16 * EPITCH) * sizeof(float);
template <int DT>
__global__ __launch_bounds__(SP_THREADS) void softmax_rows16_kernel(const ScoreArgs a) {
    const ScoreView vw = view_of(a, blockIdx.y);
    typedef typename Dt<DT>::raw raw;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, W = a.window, parts = W / 16, heads = a.bsz * a.n_q_heads;
    int hb, part;
    if ((heads & 7) == 0) { const int id = blockIdx.x, r = id >> 3; part = r % parts; hb = (r / parts) * 8 + (id & 7); }
    else { part = blockIdx.x % parts; hb = blockIdx.x / parts; }
    const int L = a.q_len, n = L - W, n_pad = (n + 1) & ~1;
    const int iters = (L + SP_THREADS - 1) / SP_THREADS;
    float* const m = reinterpret_cast<float*>(smem);              // [64]
    float* const rinv = m + 64;                                   // [64]
    float* const scratch = rinv + 64;                             // [256]
    float* const stage = scratch + 256;                           // [2][16][EPITCH]
    const raw* const lg = reinterpret_cast<const raw*>(vw.logits) + (int64_t)hb * L * W + 16 * part;
    // the row maxima first: loads return in order, so the maxima would otherwise wait behind the workgroup's 256 KB of logits and
    // no exponential could start before the last of them had landed; this way iteration 0 computes while 1 .. 7 are in flight
    block_row_max(vw.pmax + (int64_t)hb * a.n_tiles * W, a.n_tiles, W, m, scratch);
    uint4 rawx[R16_ITERS][2];
#pragma unroll
    for (int it = 0; it < R16_ITERS; ++it) {
        const int key = it * SP_THREADS + tid;
        const uint4* src = reinterpret_cast<const uint4*>(lg + (int64_t)(key < L ? key : L - 1) * W);
        rawx[it][0] = src[0]; rawx[it][1] = src[1];
    }
    f32x2 nm[8];                                                  // minus the maxima of this workgroup's rows, in pairs
#pragma unroll
    for (int w = 0; w < 8; ++w) nm[w] = f32x2{-m[16 * part + 2 * w], -m[16 * part + 2 * w + 1]};
    auto exps = [&](const uint4 (&rx)[2], f32x2 (&e)[8]) {
        const uint32_t wv[8] = {rx[0].x, rx[0].y, rx[0].z, rx[0].w, rx[1].x, rx[1].y, rx[1].z, rx[1].w};
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            f32x2 x;
            if constexpr (DT == KVC_BF16) x = f32x2{u2f(wv[w] << 16), u2f(wv[w] & 0xffff0000u)};
            else x = f32x2{Dt<DT>::ld((uint16_t)(wv[w] & 0xffffu)), Dt<DT>::ld((uint16_t)(wv[w] >> 16))};
            e[w] = exp_u20x2_nonpos(x + nm[w]);
        }
    };
    // pass 1: the denominators in torch's order (16 chains per row, stage by stage: softmax_pool_kernel)
    const int cw = tid >> 4, cl = tid & 15;                       // tid < 256: chain cl of row cw
    float acc = 0.0f;
#pragma unroll
    for (int it = 0; it < R16_ITERS; ++it) {
        if (it < iters) {                                          // (uniform)
            const int key = it * SP_THREADS + tid;
            f32x2 e[8];
            exps(rawx[it], e);
            float* buf = stage + (it & 1) * (16 * EPITCH);
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                buf[(2 * w) * EPITCH + tid] = key < L ? e[w].x : 0.0f;
                buf[(2 * w + 1) * EPITCH + tid] = key < L ? e[w].y : 0.0f;
            }
            __syncthreads();                                       // (the buffer filled two stages ago has been read: its readers passed
            if (tid < 256)                                         //  the previous barrier after reading)
                acc = chain16_add(acc, buf + cw * EPITCH + cl, it * SP_THREADS, cl, L);
        }
    }
    const float sum = chain16_fold(acc);
    if (tid < 256 && cl == 0) {
        rinv[cw] = 1.0f / sum;
        vw.rowmax[(int64_t)hb * W + 16 * part + cw] = m[16 * part + cw];
        vw.rowsum[(int64_t)hb * W + 16 * part + cw] = sum;
    }
    __syncthreads();
    f32x2 ri[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) ri[w] = f32x2{rinv[2 * w], rinv[2 * w + 1]};
    // pass 2: p = round(e / sum) of the 16 rows added in row order: the cascade's leaf
    float* const pk = vw.part16 + ((int64_t)hb * parts + part) * n_pad;
#pragma unroll
    for (int it = 0; it < R16_ITERS; ++it) {
        const int key = it * SP_THREADS + tid;
        if (it < iters && key < n) {
            f32x2 e[8];
            exps(rawx[it], e);
            float a0 = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const f32x2 pr = e[w] * ri[w];
                a0 = a0 + rnd<DT>(pr.x);
                a0 = a0 + rnd<DT>(pr.y);
            }
            pk[key] = a0;
        }
    }
}

template <int DT>
__global__ __launch_bounds__(SP_THREADS) void softmax_comb16_kernel(const ScoreArgs a) {
    const ScoreView vw = view_of(a, blockIdx.z);
    typedef typename Dt<DT>::raw raw;
    __shared__ float seg[SP_THREADS + 64];
    const int tid = threadIdx.x, hb = blockIdx.y, W = a.window, parts = W / 16;
    const int L = a.q_len, n = L - W, n_pad = (n + 1) & ~1;
    const int pad = a.pooling == KVC_POOL_NONE ? 0 : a.kernel_size / 2;       // kernel_size <= 63
    const int j0 = blockIdx.x * SP_THREADS - pad;                               // seg[i] = key j0 + i
    const float* const pk = vw.part16 + (int64_t)hb * parts * n_pad;
    for (int i = tid; i < SP_THREADS + 2 * pad; i += SP_THREADS) {
        const int key = j0 + i;
        float v = 0.0f;
        if (key >= 0 && key < n) {
            float a1 = 0.0f;                                                    // the cascade's a1: its leaves in order
            for (int p_ = 0; p_ < parts; ++p_) a1 = a1 + pk[(int64_t)p_ * n_pad + key];
            v = rnd<DT>(a.window_mean ? a1 / (float)W : a1);
        }
        seg[i] = v;
    }
    __syncthreads();
    const int jo = blockIdx.x * SP_THREADS + tid;
    if (jo >= n) return;
    float c;
    if (a.pooling == KVC_POOL_NONE) {
        c = seg[tid];
    } else {
        const int lo = jo - pad < 0 ? 0 : jo - pad;
        const int hi = jo - pad + a.kernel_size > n ? n : jo - pad + a.kernel_size;
        if (a.pooling == KVC_POOL_MAX) {
            c = -__builtin_inff();
            for (int i = lo; i < hi; ++i) { const float v = seg[i - j0]; c = v > c ? v : c; }
        } else {
            float s2 = 0.0f;
            for (int i = lo; i < hi; ++i) s2 = s2 + seg[i - j0];
            c = rnd<DT>(s2 / (float)a.kernel_size);
        }
    }
    reinterpret_cast<raw*>(vw.scores)[(int64_t)hb * n + jo] = Dt<DT>::st(c);
}

// ---------------------------------------------------------------------------------------------
// host launch
// ---------------------------------------------------------------------------------------------
// One workgroup per head (softmax_pool_kernel) when there are enough heads x items to fill the chip (a prompt's layers
// batched: -25 % on the two kernels' time); the two-kernel split (many workgroups per head) otherwise — with 32 heads
// the fused form would occupy 32 of 256 CUs.  Same bits either way; debug_stage_mask bit3 / bit4 force split / fused
// (parity tests).
// One workgroup per head does everything (softmax_pool_kernel) when there are enough heads x items to fill the chip (a
// prompt's layers batched); with few heads (per-layer calls) only the serial row sums stay one workgroup per head and
// the normalisation / pooling pass is spread over many workgroups (pool_kernel).  Same bits either way;
// debug_stage_mask bit3 / bit4 force split / fused (parity tests).
constexpr size_t kSoftmaxLds = (size_t)(64 + 64 + 256 + 2 * ESTAGE) * sizeof(float);
template <int DT, int WV>
static void launch_softmax_pool_t(const ScoreArgs& a, hipStream_t st) {
    const int m = (a.stage_mask & 7) ? (a.stage_mask & 7) : 7;
    if constexpr (WV > 0) {
        const int iters = (a.q_len + SP_THREADS - 1) / SP_THREADS;
        const int heads = a.bsz * a.n_q_heads * a.n_items, ov = (a.stage_mask & 8) ? 1 : ((a.stage_mask & 16) ? 2 : 0);
        const bool fused = ov != 1 && (ov == 2 || heads >= 128);
        if (fused) {
            if (!(m & 6)) return;
            dim3 g((unsigned)(a.bsz * a.n_q_heads), (unsigned)a.n_items);
            if constexpr (DT != KVC_FP32 && (WV == 16 || WV == 32 || WV == 64)) {   // 16-row workgroups: the logits read once (bit 12: round 2's form)
                if (a.off_part16 >= 0 && a.q_len >= 16 && a.q_len <= R16_ITERS * SP_THREADS && !(a.stage_mask & 4096) && (m & 6) == 6) {
                    static LdsCache c_r16 = {};
                    (void)ensure_lds(reinterpret_cast<const void*>(&softmax_rows16_kernel<DT>), kSoftmaxLds16, c_r16);
                    hipLaunchKernelGGL((softmax_rows16_kernel<DT>), dim3((unsigned)(a.bsz * a.n_q_heads * (WV / 16)), (unsigned)a.n_items), dim3(SP_THREADS), kSoftmaxLds16, st, a);
                    const int n_ = a.q_len - a.window;
                    hipLaunchKernelGGL((softmax_comb16_kernel<DT>), dim3((unsigned)((n_ + SP_THREADS - 1) / SP_THREADS), (unsigned)(a.bsz * a.n_q_heads), (unsigned)a.n_items), dim3(SP_THREADS), 0, st, a);
                    return;
                }
            }
            if constexpr (WV == 8) {                          // chain waves beside worker waves (debug_stage_mask bit 10: round 2's form)
                if (a.q_len >= 1024 && a.q_len <= WS_ITERS * WS_WORK && !(a.stage_mask & (256 | 1024))) {
                    static LdsCache c_ws = {};
                    (void)ensure_lds(reinterpret_cast<const void*>(&softmax_pool_ws_kernel<DT>), kSoftmaxLds, c_ws);
                    hipLaunchKernelGGL((softmax_pool_ws_kernel<DT>), g, dim3(SP_THREADS), kSoftmaxLds, st, a);
                    return;
                }
            }
            constexpr int KEEP = 64 / WV;                     // iterations whose exponentials stay in registers
            static LdsCache c_keep = {}, c_loop = {};
            if (KEEP >= 2 && iters <= KEEP && !(a.stage_mask & 256)) {
                (void)ensure_lds(reinterpret_cast<const void*>(&softmax_pool_kernel<DT, WV, (KEEP >= 2 ? KEEP : 0)>), kSoftmaxLds, c_keep);
                hipLaunchKernelGGL((softmax_pool_kernel<DT, WV, (KEEP >= 2 ? KEEP : 0)>), g, dim3(SP_THREADS), kSoftmaxLds, st, a);
            } else {
                (void)ensure_lds(reinterpret_cast<const void*>(&softmax_pool_kernel<DT, WV, 0>), kSoftmaxLds, c_loop);
                hipLaunchKernelGGL((softmax_pool_kernel<DT, WV, 0>), g, dim3(SP_THREADS), kSoftmaxLds, st, a);
            }
            return;
        }
    }
    static LdsCache c_rs = {};
    (void)ensure_lds(reinterpret_cast<const void*>(&rowsum16_kernel<DT, WV>), kSoftmaxLds, c_rs);
    dim3 g2((unsigned)(a.bsz * a.n_q_heads), (unsigned)a.n_items);
    if (m & 2) hipLaunchKernelGGL((rowsum16_kernel<DT, WV>), g2, dim3(SP_THREADS), kSoftmaxLds, st, a);
    const int n = a.q_len - a.window;
    dim3 g3((unsigned)((n + 255) / 256), (unsigned)(a.bsz * a.n_q_heads), (unsigned)a.n_items);
    if (m & 4) hipLaunchKernelGGL((pool_kernel<DT, WV>), g3, dim3(256), 0, st, a);
}

template <int DT, int D, int WV, bool FAST>
static void launch_logits_t(const ScoreArgs& a, hipStream_t st) {
    constexpr int ES = Dt<DT>::esize;
    if constexpr (WV == 16 || WV == 32 || WV == 64) {
        if (a.group * WV == 128 && !(a.stage_mask & 512)) {       // four M-tiles (W = 32 x 4 heads per KV head, 64 x 2, 16 x 8):
                                                                  // one K tile shared by the workgroup's waves
            constexpr bool ASMB = (DT == KVC_BF16) && !FAST;
            const size_t lds4 = (FAST ? 0 : (size_t)LOGITS_WAVES * 64 * (D / 2) * 4) + (size_t)32 * (D * ES + (ASMB ? 4 : 16));
            static LdsCache lds_cache4 = {};
            (void)ensure_lds(reinterpret_cast<const void*>(&logits_mt4_kernel<DT, D, WV, FAST>), lds4, lds_cache4);
            dim3 g4((unsigned)a.n_tiles, (unsigned)(a.bsz * a.n_kv_heads), (unsigned)a.n_items);
            hipLaunchKernelGGL((logits_mt4_kernel<DT, D, WV, FAST>), g4, dim3(LOGITS_THREADS), lds4, st, a);
            return;
        }
    }
    const size_t lds = (FAST ? 0 : (size_t)64 * (D / 2) * 4) + (size_t)LOGITS_WAVES * 32 * (D * ES + 16) + LOGITS_WAVES * 32 * sizeof(float);
    static LdsCache lds_cache = {};
    (void)ensure_lds(reinterpret_cast<const void*>(&logits_kernel<DT, D, WV, FAST>), lds, lds_cache);   // a failure surfaces as a launch error
    dim3 g1((unsigned)a.n_tiles, (unsigned)(a.bsz * a.n_kv_heads), (unsigned)a.n_items);
    hipLaunchKernelGGL((logits_kernel<DT, D, WV, FAST>), g1, dim3(LOGITS_THREADS), lds, st, a);
}

template <int DT, int D, int WV>
static void launch_all_t(const ScoreArgs& a, hipStream_t st) {
    if (((a.stage_mask & 7) ? (a.stage_mask & 7) : 7) & 1) {
        if constexpr (DT != KVC_FP32) {
            if (a.fast_dot) launch_logits_t<DT, D, WV, true>(a, st);
            else launch_logits_t<DT, D, WV, false>(a, st);
        } else {
            launch_logits_t<DT, D, WV, false>(a, st);
        }
    }
    launch_softmax_pool_t<DT, WV>(a, st);
}

template <int DT, int D>
static int launch_scores_t(const ScoreArgs& a, hipStream_t st) {
    switch (a.window) {
        case 8:  launch_all_t<DT, D, 8>(a, st); break;
        case 16: launch_all_t<DT, D, 16>(a, st); break;
        case 32: launch_all_t<DT, D, 32>(a, st); break;
        case 64: launch_all_t<DT, D, 64>(a, st); break;
        default: launch_all_t<DT, D, 0>(a, st); break;
    }
    return 0;
}

int launch_scores(const ScoreArgs& a, int dtype, int head_dim, hipStream_t st) {
#define KVC_CASE(DT_, D_) if (dtype == DT_ && head_dim == D_) return launch_scores_t<DT_, D_>(a, st)
    KVC_CASE(KVC_BF16, 128); KVC_CASE(KVC_BF16, 64);
    KVC_CASE(KVC_FP16, 128); KVC_CASE(KVC_FP16, 64);
    KVC_CASE(KVC_FP32, 128); KVC_CASE(KVC_FP32, 64);
#undef KVC_CASE
    return KVC_ERR_UNSUPPORTED;
}

}  // namespace kvc
