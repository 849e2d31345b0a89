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
//                       accumulated in torch's cascade order (SumKernel multi_row_sum), rounded once.  Wide forms for
//                       16-bit dtypes: two columns per thread, and (L >= 512) 256-row blocks in parallel combined in the
//                       cascade's own order (h2o_colpart_kernel / h2o_colcomb_kernel).
// MFMA-bound by construction (2*Hq*L*L*D flops at the f32-MFMA rate); a bf16-MFMA variant is the planned
// follow-up and cannot be bit-exact with the oracle (tools/mfma_probe.hip).
#include <stdlib.h>
#include "kvc_common.h"
#include "kvc_launch.h"
#include "kvc_ldsasm.h"

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

// x / sqrt(D), bit-identical to the IEEE division (same construction and guard as ScaleDiv in kvc_score.hip:
// tests/test_fastdiv.py checks all 2^32 inputs of the D = 128 form).
template <int D> __device__ __forceinline__ float h2o_scale(float x, float c) {
    if constexpr (D == 64) {
        return x * 0.125f;
    } else {
        const float rc = u2f(0x3db504f3u);                         // RN(1 / sqrt(128))
        const float ax = __builtin_fabsf(x);
        if (__builtin_expect(!(ax >= u2f(0x0d800000u) && ax < __builtin_inff()), 0)) return x / c;
        const float q0 = x * rc;
        const float r = __builtin_fmaf(-q0, c, x);
        return __builtin_fmaf(r, rc, q0);
    }
}

// grid = (ceil(row_tiles / 4), bsz * n_q_heads), block = 256: wave w of block x owns query rows [32*(4x+w), +32).
template <int DT, int D>
__global__ __launch_bounds__(256) void h2o_logits_kernel(const H2OArgs a) {
    typedef typename Dt<DT>::raw raw;
    constexpr int ES = Dt<DT>::esize;
    constexpr int ROWB = D * ES, CH = ROWB / 16, PAIRS = 8 / ES, SWZ = CH < 16 ? CH - 1 : 15, STG = CH / 2;
    // bf16: the B operand is widened by the LDS load itself (ds_read_u16_d16_hi, kvc_ldsasm.h) as in kvc_score.hip — one
    // VALU instruction (v_perm) less per MFMA on a pipe that does not overlap the f32 MFMA.  Round 1 rejected this here
    // for "rare, run-to-run different" logit errors; the cause was a destination allocated onto the address register
    // of a multi-load asm statement (kvc_ldsasm.h (1)), fixed by early-clobber outputs and audited per build.
    constexpr bool ASM_B = (DT == KVC_BF16);
    constexpr int ROWP = ASM_B ? ROWB + 4 : ROWB;          // odd dword pitch: the 32 keys of a read sit in 32 banks
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
    const int hb = blockIdx.y, b = hb / a.n_q_heads, h = hb % a.n_q_heads, g = h / a.group;
    const int L = a.q_len, W = a.window;
    const int r0 = a.row0 + (blockIdx.x * 4 + wave) * 32;
    if (r0 >= L || r0 >= a.row0 + a.rows) return;          // whole wave idle (no barriers below)
    char* const buf = smem + wave * (2 * 32 * ROWP);
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
            if constexpr (ASM_B) {
                uint32_t* d = reinterpret_cast<uint32_t*>(dst + r * ROWP + cc * 16);
                d[0] = st[it].x; d[1] = st[it].y; d[2] = st[it].z; d[3] = st[it].w;
            } else {
                *reinterpret_cast<uint4*>(dst + r * ROWB + ((cc ^ (r & SWZ)) * 16)) = st[it];
            }
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
    raw* const Sh = reinterpret_cast<raw*>(a.S) + ((int64_t)hb * a.s_rows - a.row0) * L;   // row r of the chunk at Sh + r * L
    int cur = 0;
    for (int tile = 0; tile < n_t; ++tile) {
        if (tile + 1 < n_t) issue(tile + 1, st);
        __builtin_amdgcn_wave_barrier();
        const char* krow = buf + cur * (32 * ROWP) + j * ROWP;
        const int key = tile * 32 + j;
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (ASM_B) {
            // operands one step ahead of the MFMAs that consume them (4 LDS reads per step, kvc_ldsasm.h)
            const uint32_t krow_a = lds_addr(krow) + 2 * kh;
            uint32_t B0[4], B1[4];
            ld_b_step<0>(B0, krow_a);
            static_for<0, D / 8>([&](auto ic_) {
                constexpr int ic = decltype(ic_)::value;
                uint32_t (&Bc)[4] = (ic & 1) ? B1 : B0;
                if constexpr (ic + 1 < D / 8) { ld_b_step<ic + 1>((ic & 1) ? B0 : B1, krow_a); wait_b_step<4>(Bc); }
                else wait_b_step<0>(Bc);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[ic * 4 + s4], u2f(Bc[s4]), acc, 0, 0, 0);
            });
        } else {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const uint4 kv = *reinterpret_cast<const uint4*>(krow + ((c ^ (j & SWZ)) * 16));
#pragma unroll
                for (int s = 0; s < PAIRS; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[c * PAIRS + s], pick2<DT>(kv, s, kh), acc, 0, 0, 0);
            }
        }
        const bool tailk = tile * 32 + 32 > L - W, tailr = r0 + 32 > L - W;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = r0 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            float v = rnd<DT>(acc[e]);
            v = rnd<DT>(h2o_scale<D>(v, sqrt_d));
            if (tailk && tailr) {
                if (r >= L - W && key >= L - W && (key - (L - W)) > (r - (L - W))) v = rnd<DT>(v + Dt<DT>::finfo_min());
            }
            if (r < L && key < L) {
                Sh[(int64_t)r * L + key] = Dt<DT>::st(v);
                rmax[e] = v > rmax[e] ? v : rmax[e];
            }
        }
        if (tile + 1 < n_t) commit(buf + (cur ^ 1) * (32 * ROWP), st);
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
    const int r = a.row0 + (blockIdx.x * 4 + wave) * 4 + (lane >> 4);
    const int l = lane & 15;
    const bool valid = r < L && r < a.row0 + a.rows;
    const raw* row = reinterpret_cast<const raw*>(a.S) + ((int64_t)hb * a.s_rows + (valid ? r - a.row0 : 0)) * L;
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
// ---- wide forms for 16-bit dtypes and L % 8 == 0 (the configuration sizes): same sums in the same order ----
// Two elements of a packed 16-bit pair widened to fp32.
template <int DT> __device__ __forceinline__ f32x2 widen2(uint32_t w) {
    if constexpr (DT == KVC_BF16) return f32x2{u2f(w << 16), u2f(w & 0xffff0000u)};
    else return f32x2{Dt<DT>::ld((uint16_t)(w & 0xffffu)), Dt<DT>::ld((uint16_t)(w >> 16))};
}
// h2o_rowsum_wide: two lanes per row, 32 rows per wave.  Lane `half` of a row owns the strided chains 8*half .. 8*half+7
// (one 16-byte load per 16 elements), packed-fp32 exponentials; the xor-8 butterfly step is the one cross-lane add,
// steps 4, 2, 1 fold the lane's own eight chains — the same tree as 16 lanes with one chain each.
template <int DT>
__global__ __launch_bounds__(256) void h2o_rowsum_wide_kernel(const H2OArgs a) {
    typedef typename Dt<DT>::raw raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hb = blockIdx.y, L = a.q_len;
    const int r = a.row0 + (blockIdx.x * 4 + wave) * 32 + (lane >> 1);
    const int half = lane & 1;
    const bool valid = r < L && r < a.row0 + a.rows;
    const raw* row = reinterpret_cast<const raw*>(a.S) + ((int64_t)hb * a.s_rows + (valid ? r - a.row0 : 0)) * L + 8 * half;
    const float m = valid ? a.rowmax[(int64_t)hb * L + r] : 0.0f;
    const f32x2 m2 = {m, m};
    f32x2 acc[4];
    {
        const uint4 v = *reinterpret_cast<const uint4*>(row);
        acc[0] = exp_u20x2(widen2<DT>(v.x) - m2); acc[1] = exp_u20x2(widen2<DT>(v.y) - m2);
        acc[2] = exp_u20x2(widen2<DT>(v.z) - m2); acc[3] = exp_u20x2(widen2<DT>(v.w) - m2);
    }
    const int full = L - (L % 16);
    int d = 16;
    for (; d < full; d += 16) {
        const uint4 v = *reinterpret_cast<const uint4*>(row + d);
        acc[0] = acc[0] + exp_u20x2(widen2<DT>(v.x) - m2); acc[1] = acc[1] + exp_u20x2(widen2<DT>(v.y) - m2);
        acc[2] = acc[2] + exp_u20x2(widen2<DT>(v.z) - m2); acc[3] = acc[3] + exp_u20x2(widen2<DT>(v.w) - m2);
    }
    if (8 * half < L - d) {                                // L % 16 == 8: the tail holds chains 0..7 only
        const uint4 v = *reinterpret_cast<const uint4*>(row + d);
        acc[0] = acc[0] + exp_u20x2(widen2<DT>(v.x) - m2); acc[1] = acc[1] + exp_u20x2(widen2<DT>(v.y) - m2);
        acc[2] = acc[2] + exp_u20x2(widen2<DT>(v.z) - m2); acc[3] = acc[3] + exp_u20x2(widen2<DT>(v.w) - m2);
    }
    float t[8] = {acc[0].x, acc[0].y, acc[1].x, acc[1].y, acc[2].x, acc[2].y, acc[3].x, acc[3].y};
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = t[e] + xor_lane<1>(t[e]);                  // chain c + chain c^8
    const float u0 = t[0] + t[4], u1 = t[1] + t[5], u2 = t[2] + t[6], u3 = t[3] + t[7];   // ^4
    const float v0 = u0 + u2, v1 = u1 + u3;                                                 // ^2
    const float s = v0 + v1;                                                                // ^1
    if (valid && half == 0) a.rinv[(int64_t)hb * L + r] = 1.0f / s;
}
// h2o_colsum_wide: one thread per PAIR of key columns (4-byte loads, packed-fp32 exp / scale / cascade adds).
struct CascadeSum2 {
    f32x2 a0, a1, a2, a3;
    int i, in_step, level_step, level_power, full;
    __device__ __forceinline__ void init(int size) {
        int cl = 0;
        while ((1 << cl) < size) ++cl;
        level_power = cl / 4 > 4 ? cl / 4 : 4;
        level_step = 1 << level_power;
        full = size - (size % level_step);
        a0 = a1 = a2 = a3 = f32x2{0.0f, 0.0f};
        i = 0;
        in_step = 0;
    }
    __device__ __forceinline__ void add(f32x2 v) {
        a0 = a0 + v;
        ++i;
        ++in_step;
        if (in_step == level_step && i <= full) {
            in_step = 0;
            const int mask = level_step - 1;
            a1 = a1 + a0; a0 = f32x2{0.0f, 0.0f};
            if ((i & (mask << level_power)) == 0) {
                a2 = a2 + a1; a1 = f32x2{0.0f, 0.0f};
                if ((i & (mask << (2 * level_power))) == 0) { a3 = a3 + a2; a2 = f32x2{0.0f, 0.0f}; }
            }
        }
    }
    __device__ __forceinline__ f32x2 result() const { return ((a0 + a1) + a2) + a3; }
};
template <int DT>
__global__ __launch_bounds__(256) void h2o_colsum_wide_kernel(const H2OArgs a) {
    const int hb = blockIdx.y, L = a.q_len, n = L - a.window;       // L and n even
    const int jcol = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (jcol >= n) return;
    const uint32_t* S = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.S) + ((int64_t)hb * L * L + jcol) * 2);
    const float* m = a.rowmax + (int64_t)hb * L;
    const float* ri = a.rinv + (int64_t)hb * L;
    CascadeSum2 cs;
    cs.init(L);
    const int64_t pitch = L / 2;                                     // row pitch in 4-byte words
    auto row = [&](uint32_t w, int r) {
        const float mr = m[r], rr = ri[r];
        const f32x2 e = exp_u20x2(widen2<DT>(w) - f32x2{mr, mr});
        const f32x2 pr = e * f32x2{rr, rr};
        cs.add(f32x2{rnd<DT>(pr.x), rnd<DT>(pr.y)});
    };
    // the rows are independent loads 2*L bytes apart: keep eight in flight while the previous eight are summed
    constexpr int U = 8;
    uint32_t cur[U], nxt[U];
    int r = 0;
    if (L >= U) {
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = S[(int64_t)u * pitch];
        for (; r + 2 * U <= L; r += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = S[(int64_t)(r + U + u) * pitch];
#pragma unroll
            for (int u = 0; u < U; ++u) row(cur[u], r + u);
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) row(cur[u], r + u);
        r += U;
    }
    for (; r < L; ++r) row(S[(int64_t)r * pitch], r);
    const f32x2 res = cs.result();
    reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(a.scores) + ((int64_t)hb * n + jcol) * 2)[0] =
        (uint32_t)Dt<DT>::st(rnd<DT>(res.x)) | ((uint32_t)Dt<DT>::st(rnd<DT>(res.y)) << 16);
}

// Column sums split over the rows.  torch's cascade (SumKernel multi_row_sum, level step 16 for any L <= 2^19) is a tree:
// 16 rows into a0, 16 such chunks into a1 (256 rows), 16 of those into a2, the rest into a3 — and every accumulator
// restarts from zero after it is dumped.  So the sum of one 256-row block is independent of the others:
//   h2o_colpart_kernel : (column pair, 256-row block) -> the block's a1 (and a0 of the rows beyond the last full chunk)
//   h2o_colcomb_kernel : per column pair, the blocks combined in the cascade's own order: a2 over 16 blocks, a3 over
//                        those, result ((a0 + a1) + a2) + a3.
// 32x the parallelism of one thread walking all L rows (which ran at two waves per SIMD).
template <int DT>
__global__ __launch_bounds__(256) void h2o_colpart_kernel(const H2OArgs a) {
    const int hb = blockIdx.y, g = blockIdx.z + a.row0 / 256, L = a.q_len, n = L - a.window;   // g: 256-row block of the prompt
    const int jcol = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (jcol >= n) return;
    const int n_pad = (n + 1) & ~1, n_blk = (L + 255) / 256;
    const int r0 = g * 256, r1 = r0 + 256 < L ? r0 + 256 : L, full = L - (L % 16);
    const uint32_t* S = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.S) + (((int64_t)hb * a.s_rows - a.row0) * L + jcol) * 2);
    const float* m = a.rowmax + (int64_t)hb * L;
    const float* ri = a.rinv + (int64_t)hb * L;
    const int64_t pitch = L / 2;
    f32x2 a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f};
    auto row = [&](uint32_t w, int r) {
        const float mr = m[r], rr = ri[r];
        const f32x2 pr = exp_u20x2(widen2<DT>(w) - f32x2{mr, mr}) * f32x2{rr, rr};
        a0 = a0 + f32x2{rnd<DT>(pr.x), rnd<DT>(pr.y)};
        if (((r + 1) & 15) == 0 && r + 1 <= full) { a1 = a1 + a0; a0 = f32x2{0.0f, 0.0f}; }
    };
    constexpr int U = 8;
    int r = r0;
    for (; r + U <= r1; r += U) {
        uint32_t w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = S[(int64_t)(r + u) * pitch];
#pragma unroll
        for (int u = 0; u < U; ++u) row(w[u], r + u);
    }
    for (; r < r1; ++r) row(S[(int64_t)r * pitch], r);
    float* part = a.part + ((int64_t)hb * (n_blk + 1) + g) * n_pad + jcol;
    part[0] = a1.x; part[1] = a1.y;
    if (g == n_blk - 1) {                                   // rows beyond the last full chunk stay in a0
        float* left = a.part + ((int64_t)hb * (n_blk + 1) + n_blk) * n_pad + jcol;
        left[0] = a0.x; left[1] = a0.y;
    }
}
template <int DT>
__global__ __launch_bounds__(256) void h2o_colcomb_kernel(const H2OArgs a) {
    const int hb = blockIdx.y, L = a.q_len, n = L - a.window;
    const int jcol = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (jcol >= n) return;
    const int n_pad = (n + 1) & ~1, n_blk = (L + 255) / 256, n_complete = L / 256;
    const float* base = a.part + (int64_t)hb * (n_blk + 1) * n_pad + jcol;
    f32x2 a2 = {0.0f, 0.0f}, a3 = {0.0f, 0.0f};
    for (int g = 0; g < n_complete; ++g) {                  // a complete block's a1 is dumped into a2 at row 256 (g + 1)
        const float2 p = *reinterpret_cast<const float2*>(base + (int64_t)g * n_pad);
        a2 = a2 + f32x2{p.x, p.y};
        if (((g + 1) & 15) == 0) { a3 = a3 + a2; a2 = f32x2{0.0f, 0.0f}; }
    }
    f32x2 a1 = {0.0f, 0.0f};
    if (n_complete < n_blk) { const float2 p = *reinterpret_cast<const float2*>(base + (int64_t)n_complete * n_pad); a1 = f32x2{p.x, p.y}; }
    const float2 l0 = *reinterpret_cast<const float2*>(base + (int64_t)n_blk * n_pad);
    const f32x2 a0 = {l0.x, l0.y};
    const f32x2 res = ((a0 + a1) + a2) + a3;
    reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(a.scores) + ((int64_t)hb * n + jcol) * 2)[0] =
        (uint32_t)Dt<DT>::st(rnd<DT>(res.x)) | ((uint32_t)Dt<DT>::st(rnd<DT>(res.y)) << 16);
}

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

// One column per thread forms of the two kernels above (any dtype, any L / W parity): used when the prompt is processed in
// row chunks and the pair form does not apply.
template <int DT>
__global__ __launch_bounds__(256) void h2o_colpart1_kernel(const H2OArgs a) {
    typedef typename Dt<DT>::raw raw;
    const int hb = blockIdx.y, g = blockIdx.z + a.row0 / 256, L = a.q_len, n = L - a.window;
    const int jcol = blockIdx.x * 256 + threadIdx.x;
    if (jcol >= n) return;
    const int n_pad = (n + 1) & ~1, n_blk = (L + 255) / 256;
    const int r0 = g * 256, r1 = r0 + 256 < L ? r0 + 256 : L, full = L - (L % 16);
    const raw* S = reinterpret_cast<const raw*>(a.S) + ((int64_t)hb * a.s_rows - a.row0) * L + jcol;
    const float* m = a.rowmax + (int64_t)hb * L;
    const float* ri = a.rinv + (int64_t)hb * L;
    float a0 = 0.0f, a1 = 0.0f;
    for (int r = r0; r < r1; ++r) {
        a0 = a0 + rnd<DT>(exp_u20(Dt<DT>::ld(S[(int64_t)r * L]) - m[r]) * ri[r]);
        if (((r + 1) & 15) == 0 && r + 1 <= full) { a1 = a1 + a0; a0 = 0.0f; }
    }
    a.part[((int64_t)hb * (n_blk + 1) + g) * n_pad + jcol] = a1;
    if (g == n_blk - 1) a.part[((int64_t)hb * (n_blk + 1) + n_blk) * n_pad + jcol] = a0;
}
template <int DT>
__global__ __launch_bounds__(256) void h2o_colcomb1_kernel(const H2OArgs a) {
    typedef typename Dt<DT>::raw raw;
    const int hb = blockIdx.y, L = a.q_len, n = L - a.window;
    const int jcol = blockIdx.x * 256 + threadIdx.x;
    if (jcol >= n) return;
    const int n_pad = (n + 1) & ~1, n_blk = (L + 255) / 256, n_complete = L / 256;
    const float* base = a.part + (int64_t)hb * (n_blk + 1) * n_pad + jcol;
    float a2 = 0.0f, a3 = 0.0f;
    for (int g = 0; g < n_complete; ++g) {
        a2 = a2 + base[(int64_t)g * n_pad];
        if (((g + 1) & 15) == 0) { a3 = a3 + a2; a2 = 0.0f; }
    }
    const float a1 = n_complete < n_blk ? base[(int64_t)n_complete * n_pad] : 0.0f;
    const float a0 = base[(int64_t)n_blk * n_pad];
    reinterpret_cast<raw*>(a.scores)[(int64_t)hb * n + jcol] = Dt<DT>::st(rnd<DT>(((a0 + a1) + a2) + a3));
}

// ---------------------------------------------------------------------------------------------------------
// Fast mode (dot_mode = KVC_DOT_MFMA16, 16-bit dtypes): no logit matrix at all.  Two passes, each recomputing Q K^T on
// the packed 16-deep MFMA (16x the f32-input rate), written so that every reduction is lane-local:
//   pass 1  h2o_fast_stats_kernel  : S^T tiles (MFMA rows = keys, columns = query rows): a lane owns ONE query row and
//           sixteen keys of each tile, keeps a running maximum and a rescaled running sum of exponentials for it (online
//           softmax; the two lanes of a row are combined once at the end).  Output per row: -max * log2(e) and 1 / sum.
//   pass 2  h2o_fast_colsum_kernel : S tiles (rows = query rows, columns = keys): a lane owns ONE key column and sixteen
//           rows of each tile; p = round(exp2(x * log2(e) - max * log2(e)) * rinv) summed over all row tiles in registers.
// Workspace: two floats per query row.  The arithmetic is the reference's up to (a) the MFMA's internal accumulation order
// (tools/mfma_probe.hip), (b) exp by v_exp_f32 instead of torch's polynomial, (c) fp32 sums in tile order instead of
// torch's 16-lane / cascade orders: a TOLERANCE mode (tests/test_gpu_parity.py::test_h2o_fast_mode_within_tolerance).
// Workgroup -> (head, tile block) mapping: consecutive workgroup ids go to consecutive XCDs, so with the KV head
// (pass 1) / query head (pass 2) taken from id % 8 every XCD streams ONE head's 2 MB of K (Q) through its own L2.
// ---------------------------------------------------------------------------------------------------------
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
template <int DT> __device__ __forceinline__ f32x16 mfma_pk16(const uint4& av, const uint4& bv, f32x16 acc) {
    if constexpr (DT == KVC_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, av), __builtin_bit_cast(s16x8, bv), acc, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, av), __builtin_bit_cast(h16x8, bv), acc, 0, 0, 0);
}
// two values rounded to the storage dtype and widened again
template <int DT> __device__ __forceinline__ f32x2 round2(f32x2 v) {
    if constexpr (DT == KVC_BF16) {
        asm volatile("" : "+v"(v));
        const uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));     // v_cvt_pk_bf16_f32
        return f32x2{u2f(w << 16), u2f(w & 0xffff0000u)};
    } else {
        return f32x2{rnd<DT>(v.x), rnd<DT>(v.y)};
    }
}
// the reference's  round(round(dot) / sqrt(D))  on a pair (x / sqrt(128) as the 2-FMA division of kvc_score.hip, unguarded:
// outside its proven range the quotient may be one fp32 ulp off — inside this mode's tolerance)
template <int DT, int D> __device__ __forceinline__ f32x2 logit2(f32x2 acc, float c) {
    const f32x2 r1 = round2<DT>(acc);
    if constexpr (D == 64) {
        return round2<DT>(r1 * (f32x2)0.125f);
    } else {
        const f32x2 rc = u2f(0x3db504f3u), cc = c;
        const f32x2 q0 = r1 * rc;
        const f32x2 r = __builtin_elementwise_fma(-q0, cc, r1);
        return round2<DT>(__builtin_elementwise_fma(r, rc, q0));
    }
}
constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kNeg = -1.0e30f;                            // "minus infinity" that stays finite under subtraction

struct FastMap { int b, h, g, hb, blk; };
// Workgroup id -> (batch, query head, its kv head, tile block).  Ids x, x + 8, x + 16, ... share an XCD (round-robin
// placement, observed, speed only): the stream of ids of one XCD walks all tile blocks of ONE head (pass 1: one kv head
// and its `group` query heads in turn; pass 2: one query head) before it moves to the next, so the head's K (or Q),
// 2 MB at 8k, is read from HBM once and then served by that XCD's L2.
__device__ __forceinline__ FastMap fast_map(const H2OArgs& a, int id, int n_blk, bool by_kv_head) {
    FastMap m;
    const int n_units = by_kv_head ? a.n_kv_heads : a.n_q_heads;      // heads that own a stream of tile blocks
    const int per_unit = by_kv_head ? a.group : 1;                    // query heads visited per unit
    int unit, rest;
    if (n_units % 8 == 0) {
        const int x = id & 7;
        rest = id >> 3;
        m.blk = rest % n_blk; rest /= n_blk;
        const int qi = rest % per_unit; rest /= per_unit;
        unit = x + 8 * (rest % (n_units / 8));
        m.b = rest / (n_units / 8);
        m.g = by_kv_head ? unit : unit / a.group;
        m.h = by_kv_head ? unit * a.group + qi : unit;
    } else {
        m.blk = id % n_blk; rest = id / n_blk;
        m.h = rest % a.n_q_heads;
        m.b = rest / a.n_q_heads;
        m.g = m.h / a.group;
    }
    m.hb = m.b * a.n_q_heads + m.h;
    return m;
}

constexpr int FAST_THREADS = 256;                            // 4 waves; each owns 2 tiles (64 rows / 64 keys) of the block's 256
template <int DT, int D>
__global__ __launch_bounds__(FAST_THREADS, 2) void h2o_fast_stats_kernel(const H2OArgs a) {
    constexpr int ROWB = D * 2, ROWP = ROWB + 16, NS = D / 16, CPR = ROWB / 16, CHT = 32 * CPR / FAST_THREADS;
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 x 32 x ROWP
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = a.q_len, W = a.window, n_t = (L + 31) / 32;
    const FastMap fm = fast_map(a, blockIdx.x, (L + 255) / 256, true);
    const int r0 = fm.blk * 256 + wave * 64;
    const char* const kbase = reinterpret_cast<const char*>(a.k) + ((int64_t)fm.b * a.k_stride_b + (int64_t)fm.g * a.k_stride_h) * 2;
    const int64_t krow_bytes = a.k_stride_l * 2;
    uint4 bq[2][NS];                                                      // B operand: this lane's query row, chunk 2 s + kh
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int row = r0 + rt * 32 + j;
        const char* qrow = reinterpret_cast<const char*>(a.q) +
            ((int64_t)fm.b * a.q_stride_b + (int64_t)fm.h * a.q_stride_h + (int64_t)(row < L ? row : 0) * a.q_stride_l) * 2;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_)
            bq[rt][s_] = row < L ? *reinterpret_cast<const uint4*>(qrow + (2 * s_ + kh) * 16) : make_uint4(0, 0, 0, 0);
    }
    uint4 st[CHT];
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < CHT; ++i) {
            const int c = tid + FAST_THREADS * i, r = c / CPR, cc = c % CPR, key = t * 32 + r;
            st[i] = key < L ? *reinterpret_cast<const uint4*>(kbase + (int64_t)key * krow_bytes + cc * 16) : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < CHT; ++i) {
            const int c = tid + FAST_THREADS * i, r = c / CPR, cc = c % CPR;
            *reinterpret_cast<uint4*>(buf + r * ROWP + cc * 16) = st[i];
        }
    };
    float mrun[2] = {kNeg, kNeg}, srun[2] = {0.0f, 0.0f};
    const float sqrt_d = a.sqrt_d;
    gload(0);
    lstore(smem);
    __syncthreads();
    for (int t = 0; t < n_t; ++t) {
        if (t + 1 < n_t) gload(t + 1);
        const char* const buf = smem + (t & 1) * (32 * ROWP);
        f32x16 acc[2];
        {
            const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // C = 0 folds into the first MFMA
            const uint4 ak = *reinterpret_cast<const uint4*>(buf + j * ROWP + kh * 16);
            acc[0] = mfma_pk16<DT>(ak, bq[0][0], zero);
            acc[1] = mfma_pk16<DT>(ak, bq[1][0], zero);
        }
#pragma unroll
        for (int s_ = 1; s_ < NS; ++s_) {
            const uint4 ak = *reinterpret_cast<const uint4*>(buf + j * ROWP + (2 * s_ + kh) * 16);   // A operand: key row j of the tile
            acc[0] = mfma_pk16<DT>(ak, bq[0][s_], acc[0]);
            acc[1] = mfma_pk16<DT>(ak, bq[1][s_], acc[1]);
        }
        const bool ragged = t * 32 + 32 > L;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int row = r0 + rt * 32 + j;
            const bool window_tile = (r0 + rt * 32 + 32 > L - W) && (t * 32 + 32 > L - W);   // wave-uniform
            float x[16];
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 v = logit2<DT, D>(f32x2{acc[rt][e], acc[rt][e + 1]}, sqrt_d);
                x[e] = v.x; x[e + 1] = v.y;
            }
            if (ragged || window_tile) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = t * 32 + 8 * (e >> 2) + 4 * kh + (e & 3);
                    const bool masked = row >= L - W && key >= L - W && (key - (L - W)) > (row - (L - W));   // :545-551
                    x[e] = (key >= L || masked) ? kNeg : x[e];
                }
            }
            float mt = __builtin_fmaxf(x[0], x[1]);                        // (no NaN can reach here from finite inputs; v_max_f32)
#pragma unroll
            for (int e = 2; e < 16; ++e) mt = __builtin_fmaxf(mt, x[e]);
            const float mn = __builtin_fmaxf(mt, mrun[rt]);
            const float f = __builtin_amdgcn_exp2f((mrun[rt] - mn) * kLog2e);
            const float c = -mn * kLog2e;
            f32x2 part = {0.0f, 0.0f};
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 arg = __builtin_elementwise_fma(f32x2{x[e], x[e + 1]}, (f32x2)kLog2e, (f32x2)c);
                part = part + f32x2{__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
            }
            srun[rt] = __builtin_fmaf(srun[rt], f, part.x + part.y);
            mrun[rt] = mn;
        }
        if (t + 1 < n_t) lstore(smem + ((t + 1) & 1) * (32 * ROWP));
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {                                      // the two lanes of a row hold disjoint keys
        const float mo = xor_lane<32>(mrun[rt]), so = xor_lane<32>(srun[rt]);
        const float M = mo > mrun[rt] ? mo : mrun[rt];
        const float S = srun[rt] * __builtin_amdgcn_exp2f((mrun[rt] - M) * kLog2e) + so * __builtin_amdgcn_exp2f((mo - M) * kLog2e);
        const int row = r0 + rt * 32 + j;
        if (kh == 0 && row < L) {
            a.rowmax[(int64_t)fm.hb * L + row] = -M * kLog2e;
            a.rinv[(int64_t)fm.hb * L + row] = 1.0f / S;
        }
    }
}

template <int DT, int D>
__global__ __launch_bounds__(FAST_THREADS, 2) void h2o_fast_colsum_kernel(const H2OArgs a) {
    typedef typename Dt<DT>::raw raw;
    constexpr int ROWB = D * 2, ROWP = ROWB + 16, NS = D / 16, CPR = ROWB / 16, CHT = 32 * CPR / FAST_THREADS;
    constexpr int BUF = 32 * ROWP + 256;                                  // Q tile + 32 x (mneg, rinv)
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 x BUF
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = a.q_len, W = a.window, n = L - W, n_t = (L + 31) / 32;
    const FastMap fm = fast_map(a, blockIdx.x, (n + 255) / 256, false);
    const int k0 = fm.blk * 256 + wave * 64;
    const char* const qbase = reinterpret_cast<const char*>(a.q) + ((int64_t)fm.b * a.q_stride_b + (int64_t)fm.h * a.q_stride_h) * 2;
    const int64_t qrow_bytes = a.q_stride_l * 2;
    const float* const mneg = a.rowmax + (int64_t)fm.hb * L;
    const float* const rinv = a.rinv + (int64_t)fm.hb * L;
    uint4 bk[2][NS];                                                      // B operand: this lane's key row, chunk 2 s + kh
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = k0 + kt * 32 + j;
        const char* krow = reinterpret_cast<const char*>(a.k) +
            ((int64_t)fm.b * a.k_stride_b + (int64_t)fm.g * a.k_stride_h + (int64_t)(key < L ? key : 0) * a.k_stride_l) * 2;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_)
            bk[kt][s_] = key < L ? *reinterpret_cast<const uint4*>(krow + (2 * s_ + kh) * 16) : make_uint4(0, 0, 0, 0);
    }
    const bool live[2] = {k0 < n, k0 + 32 < n};                          // wave-uniform: a tile of columns that are all >= n is skipped
    uint4 st[CHT];
    float2 sst = make_float2(0.0f, 0.0f);
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < CHT; ++i) {
            const int c = tid + FAST_THREADS * i, r = c / CPR, cc = c % CPR, row = t * 32 + r;
            st[i] = row < L ? *reinterpret_cast<const uint4*>(qbase + (int64_t)row * qrow_bytes + cc * 16) : make_uint4(0, 0, 0, 0);
        }
        if (tid < 32) { const int row = t * 32 + tid; sst = row < L ? make_float2(mneg[row], rinv[row]) : make_float2(0.0f, 0.0f); }
    };
    auto lstore = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < CHT; ++i) {
            const int c = tid + FAST_THREADS * i, r = c / CPR, cc = c % CPR;
            *reinterpret_cast<uint4*>(buf + r * ROWP + cc * 16) = st[i];
        }
        if (tid < 32) { reinterpret_cast<float*>(buf + 32 * ROWP)[tid] = sst.x; reinterpret_cast<float*>(buf + 32 * ROWP + 128)[tid] = sst.y; }
    };
    f32x2 col[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};
    const float sqrt_d = a.sqrt_d;
    gload(0);
    lstore(smem);
    __syncthreads();
    for (int t = 0; t < n_t; ++t) {
        if (t + 1 < n_t) gload(t + 1);
        const char* const buf = smem + (t & 1) * BUF;
        f32x16 acc[2];
        {
            const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const uint4 aq = *reinterpret_cast<const uint4*>(buf + j * ROWP + kh * 16);
            acc[0] = mfma_pk16<DT>(aq, bk[0][0], zero);                     // (a dead column tile is computed and ignored: no branch
            acc[1] = mfma_pk16<DT>(aq, bk[1][0], zero);                     //  inside the MFMA chain)
        }
#pragma unroll
        for (int s_ = 1; s_ < NS; ++s_) {
            const uint4 aq = *reinterpret_cast<const uint4*>(buf + j * ROWP + (2 * s_ + kh) * 16);   // A operand: query row j of the tile
            acc[0] = mfma_pk16<DT>(aq, bk[0][s_], acc[0]);
            acc[1] = mfma_pk16<DT>(aq, bk[1][s_], acc[1]);
        }
        // this lane's sixteen rows of the tile: 8 g + 4 kh + e
        float mr[16], rr[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 m4 = *reinterpret_cast<const float4*>(buf + 32 * ROWP + (8 * g4 + 4 * kh) * 4);
            const float4 r4 = *reinterpret_cast<const float4*>(buf + 32 * ROWP + 128 + (8 * g4 + 4 * kh) * 4);
            mr[4 * g4] = m4.x; mr[4 * g4 + 1] = m4.y; mr[4 * g4 + 2] = m4.z; mr[4 * g4 + 3] = m4.w;
            rr[4 * g4] = r4.x; rr[4 * g4 + 1] = r4.y; rr[4 * g4 + 2] = r4.z; rr[4 * g4 + 3] = r4.w;
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (!live[kt]) continue;
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 x = logit2<DT, D>(f32x2{acc[kt][e], acc[kt][e + 1]}, sqrt_d);
                const f32x2 arg = __builtin_elementwise_fma(x, (f32x2)kLog2e, f32x2{mr[e], mr[e + 1]});
                const f32x2 pr = f32x2{__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)} * f32x2{rr[e], rr[e + 1]};
                col[kt] = col[kt] + round2<DT>(pr);
            }
        }
        if (t + 1 < n_t) lstore(smem + ((t + 1) & 1) * BUF);
        __syncthreads();
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        float c = col[kt].x + col[kt].y;
        c = c + xor_lane<32>(c);                                          // the two lanes of a key hold disjoint rows
        const int key = k0 + kt * 32 + j;
        if (kh == 0 && key < n) reinterpret_cast<raw*>(a.scores)[(int64_t)fm.hb * n + key] = Dt<DT>::st(c);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Fused exact mode (round 3; 16-bit dtypes, 16 <= L <= 8192): the logits never leave the compute unit.
//   One workgroup (16 waves) = one 256-row block of one head, walked in sub-blocks of 16 query rows — the unit of torch's
//   column cascade (16 rows into a0, a0 into a1).  The 16 x L logits of a sub-block live in REGISTERS, packed two per
//   dword (256 KB of the CU's 512 KB at L = 8192), so every later phase re-reads registers instead of HBM:
//     A  S = round(round(q.k) / sqrt(D)) (+ window mask) on v_mfma_f32_16x16x4_f32 — a k-ordered fmaf chain like the
//        32x32x2 form (tools/mfma16x4_probe.hip: 0 of 51 200 results differ from the chain).  Key tile T = 16 i + wave;
//        Q is the A operand (fp32 in LDS, re-read per step), K the B operand straight from a pre-permuted copy
//        (h2o_kperm_kernel: lane (key, g) needs dims g, g + 4, ...; there they are 64 contiguous bytes), row maxima kept;
//     B  row maxima across the waves (LDS);
//     C  exponentials by all waves into an LDS ring, two tiles per barrier; waves 0-3 add them in torch's order: chain c
//        of row r = keys = c (mod 16) ascending = lane c of the MFMA layout, tile after tile; then the xor butterfly;
//     D  p = round(exp * rinv); the 16 rows of a key are summed IN ROW ORDER by four chained MFMAs with A = 1
//        (fma(1, p, acc) = acc + p: the Q rows are dealt to the MFMA's row slots so that register v of lane group g is
//        row 4 v + g, which makes MFMA v add rows 4 v .. 4 v + 3 in order); a1 += a0 per sub-block.
//   Output: the block's a1 (and the leftover a0) in `part`, combined by h2o_colcomb*_kernel as before.
//   HBM traffic: Q and K once (K re-streamed from L2: the workgroup ids of one XCD walk one KV head, fast_map).
// ---------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x32 __attribute__((ext_vector_type(32)));
constexpr int FZ_WAVES = 16, FZ_THREADS = 64 * FZ_WAVES, FZ_TPW = 32, FZ_MAX_L = 16 * FZ_WAVES * FZ_TPW;   // 8192 keys
constexpr int FZ_EBUF = 32 * 4 * 64;                         // floats per ring half: 32 tile slots x 4 registers x 64 lanes
constexpr int FZ_RING = 4;                                   // 1 KB chunks of K per wave in LDS (phase A)
constexpr size_t fz_lds_bytes(int D) {
    const size_t x = (size_t)2 * FZ_EBUF * 4, r = (size_t)FZ_WAVES * FZ_RING * 1024;
    return (size_t)(16 * D + 256 + 16 + FZ_MAX_L) * 4 + (x > r ? x : r);
}

// The B operand of lane (key n, k-slot gq) over a tile's D / 4 steps is K[key][gq], K[key][4 + gq], ...: 2-byte elements 8 bytes
// apart.  h2o_kperm_kernel lays K out as the fused kernel's LDS wants it, in 1 KB chunks that global_load_lds_dwordx4 copies
// verbatim: Kt[b][g][tile][chunk c] = u16 [8 steps e][64 lanes], entry (e, lane = 16 gq + n) = K[16 tile + n][4 (8 c + e) + gq].
// Element-major, so that one ds_read_u16_d16_hi of a wave reads 128 contiguous bytes — all 32 banks, two lanes per dword.
// (A lane-major chunk, lane l's eight steps in 16 contiguous bytes, put the 64 lanes of a read on 8 banks: with the Q
// operand's conflicts the LDS, not the matrix core, bounded the logits phase.)  Keys beyond L are zeros.
// grid = (ceil(tiles * D/32 * 64 / 256), bsz * n_kv_heads), block = 256: one 16-byte piece (8 lanes of one step) per thread.
template <int DT, int D>
__global__ __launch_bounds__(256) void h2o_kperm_kernel(const H2OArgs a) {
    typedef typename Dt<DT>::raw raw;
    constexpr int KV4 = D / 32;
    const int L = a.q_len, n_tiles = (L + 15) / 16, bg = blockIdx.y, b = bg / a.n_kv_heads, g = bg % a.n_kv_heads;
    const int o = blockIdx.x * 256 + threadIdx.x;                             // output uint4 index within the head
    if (o >= n_tiles * KV4 * 64) return;
    const int p16 = o & 63, c = (o >> 6) % KV4, tile = (o >> 6) / KV4;
    const int e = p16 >> 3, gq = (p16 & 7) >> 1, n0 = 8 * (p16 & 1), dim = 4 * (8 * c + e) + gq;
    const raw* kh = reinterpret_cast<const raw*>(a.k) + (int64_t)b * a.k_stride_b + (int64_t)g * a.k_stride_h + dim;
    uint32_t w[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int k0 = 16 * tile + n0 + 2 * jj, k1 = k0 + 1;
        const uint32_t lo = k0 < L ? (uint32_t)kh[(int64_t)k0 * a.k_stride_l] : 0u, hi = k1 < L ? (uint32_t)kh[(int64_t)k1 * a.k_stride_l] : 0u;
        w[jj] = lo | (hi << 16);
    }
    reinterpret_cast<uint4*>(a.kt)[(int64_t)bg * n_tiles * KV4 * 64 + o] = make_uint4(w[0], w[1], w[2], w[3]);
}

template <int DT> __device__ __forceinline__ float wide_lo(uint32_t w) {
    if constexpr (DT == KVC_BF16) return u2f(w << 16); else return Dt<DT>::ld((uint16_t)(w & 0xffffu));
}
template <int DT> __device__ __forceinline__ float wide_hi(uint32_t w) {
    if constexpr (DT == KVC_BF16) return u2f(w & 0xffff0000u); else return Dt<DT>::ld((uint16_t)(w >> 16));
}
// two values rounded to the storage dtype, packed (low half = first)
template <int DT> __device__ __forceinline__ uint32_t pack2(f32x2 v) {
    if constexpr (DT == KVC_BF16) {
        asm volatile("" : "+v"(v));
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));                   // v_cvt_pk_bf16_f32
    } else {
        return (uint32_t)Dt<DT>::st(v.x) | ((uint32_t)Dt<DT>::st(v.y) << 16);
    }
}
// v_max_f32 as is (fmaxf would canonicalise both operands first: three instructions; no NaN reaches these maxima)
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// One group of four MFMA steps of the fused kernel: A = the group's 16 bytes of the lane's Q fragment (image: [group][lane][4] fp32),
// B = the lane's bf16 element of four consecutive steps of the K chunk ([step][lane] u16), widened by the load (kvc_ldsasm.h).
// Every one of the 5 reads touches each LDS bank once.  Retired by wait_step.
template <int QOFF, int KOFF> __device__ __forceinline__ void fz_ld(f32x4& A, uint32_t (&B)[4], uint32_t q_a, uint32_t k_a) {
    asm volatile("ds_read_b128 %0, %5 offset:%7\n\t"
                 "ds_read_u16_d16_hi %1, %6 offset:%8\n\t"
                 "ds_read_u16_d16_hi %2, %6 offset:%9\n\t"
                 "ds_read_u16_d16_hi %3, %6 offset:%10\n\t"
                 "ds_read_u16_d16_hi %4, %6 offset:%11"
                 : KVC_LD_OUT(A), KVC_LD_OUT(B[0]), KVC_LD_OUT(B[1]), KVC_LD_OUT(B[2]), KVC_LD_OUT(B[3])
                 : "v"(q_a), "v"(k_a), "n"(QOFF), "n"(KOFF), "n"(KOFF + 128), "n"(KOFF + 256), "n"(KOFF + 384)
                 : "memory");
}

// grid = bsz * n_q_heads * ceil(L / 256), block = 1024.
// LDS: Q operand image [D / 16][64][4] f32, maxima [16][16], rinv [16], a1 [8192], then the K rings (phase A) / the exponentials' ring (phase C).
template <int DT, int D>
__global__ __launch_bounds__(FZ_THREADS) void h2o_fused_kernel(const H2OArgs a) {
    typedef typename Dt<DT>::raw raw;
    constexpr int NS = D / 4, KV4 = NS * 2 / 16, NG = NS / 4;                  // NG groups of four MFMA steps per tile
    static_assert(FZ_RING % KV4 == 0 && FZ_RING - 1 <= 2 * KV4, "ring slots follow the tile's chunks");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const qs = reinterpret_cast<float*>(smem);
    float* const mx = qs + 16 * D;                                          // (Q image: [NG][64 lanes][4] fp32)
    float* const rinvs = mx + 256;
    float* const a1s = rinvs + 16;                                           // [FZ_MAX_L] the block's a1 per key
    float* const ebuf = a1s + FZ_MAX_L;                                      // phase C: the exponentials' ring; phase A: the K rings
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, gq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = a.q_len, W = a.window, ncol = L - W;
    const int n_blk = (L + 255) / 256, n_pad = (ncol + 1) & ~1;
    const int n_tiles = (L + 15) / 16, tpw = (n_tiles + FZ_WAVES - 1) / FZ_WAVES;   // key tiles per wave (<= FZ_TPW)
    const FastMap fm = fast_map(a, blockIdx.x, n_blk, true);
    uint64_t ktb;                                                            // (scalar) the head's permuted K
    {
        const uint64_t pk = (uint64_t)(uintptr_t)a.kt + (uint64_t)(fm.b * a.n_kv_heads + fm.g) * (uint64_t)n_tiles * (KV4 * 1024);
        ktb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pk >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pk);
    }
    uint32_t lane16 = lane * 16;
    const raw* const qb = reinterpret_cast<const raw*>(a.q) + (int64_t)fm.b * a.q_stride_b + (int64_t)fm.h * a.q_stride_h;
    float* const part = a.part + ((int64_t)fm.hb * (n_blk + 1) + fm.blk) * n_pad;
    float* const left = a.part + ((int64_t)fm.hb * (n_blk + 1) + n_blk) * n_pad;
    const float sqrt_d = a.sqrt_d;
    const char* const ring = reinterpret_cast<const char*>(ebuf) + wave * (FZ_RING * 1024);
    const uint32_t ring_base = __builtin_amdgcn_readfirstlane(lds_addr(ring));
    u32x32 Slo, Shi;                                                         // the sub-block's logits: tile i at dwords 2 i, 2 i + 1
    for (int key = tid; key < FZ_MAX_L; key += FZ_THREADS) a1s[key] = 0.0f;   // (each key is only ever touched by the lane that owns it)
#ifdef KVC_STAMPS
    uint64_t fz_t[6] = {0, 0, 0, 0, 0, 0}, fz_last = __builtin_amdgcn_s_memtime();
#define KVC_FZ_STAMP(i_) { const uint64_t t_ = __builtin_amdgcn_s_memtime(); fz_t[i_] += t_ - fz_last; fz_last = t_; }
#else
#define KVC_FZ_STAMP(i_)
#endif
    auto s_put = [&](int i, uint32_t w01, uint32_t w23) {
        if (i < 16) { Slo[2 * i] = w01; Slo[2 * i + 1] = w23; } else { Shi[2 * i - 32] = w01; Shi[2 * i - 31] = w23; }
    };
    auto s_get = [&](int i, uint32_t& w01, uint32_t& w23) {
        if (i < 16) { w01 = Slo[2 * i]; w23 = Slo[2 * i + 1]; } else { w01 = Shi[2 * i - 32]; w23 = Shi[2 * i - 31]; }
    };
    // K operand: chunk c (16 bytes per lane = 8 steps) of tile i is chunk q = i * KV4 + c of the wave's stream and lives in ring
    // slot q % FZ_RING.  The chunks travel global -> LDS without passing through registers (global_load_lds_dwordx4: lane l's
    // 16 bytes land at M0 + 16 l), FZ_RING - 2 chunks ahead of the MFMAs that use them.  vmcnt: these loads are the wave's only
    // vector-memory operations in phase A and complete in order.
    auto tile_src = [&](int i) {                                             // (scalar) tile 16 i + wave of the stream, clamped
#ifdef KVC_FZ_EXP1
        const int t = wave + 0 * i;                                          // (timing experiment: K always from the same 16 tiles)
#else
        const int t = 16 * i + wave;
#endif
        return ktb + (uint64_t)(t < n_tiles ? t : n_tiles - 1) * (KV4 * 1024);
    };
    for (int sb = 0; sb < 16; ++sb) {
        const int rb = fm.blk * 256 + sb * 16;
        if (rb >= L) break;
        const bool complete = rb + 16 <= L;
        {   // Q rows -> fp32 MFMA operand: row offset rho sits in MFMA row slot 4 (rho & 3) + (rho >> 2)
            constexpr int EPT = D / 64;
            const int rho = tid >> 6, slot = 4 * (rho & 3) + (rho >> 2);
            const int row = rb + rho < L ? rb + rho : L - 1;
            const raw* qr = qb + (int64_t)row * a.q_stride_l + (tid & 63) * EPT;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int d = (tid & 63) * EPT + e, st_ = d >> 2;             // step st_, k-slot d & 3: group st_ >> 2, lane 16 (d & 3) + slot
                qs[((st_ >> 2) * 64 + (d & 3) * 16 + slot) * 4 + (st_ & 3)] = Dt<DT>::ld(qr[e]);
            }
        }
        __syncthreads();
        KVC_FZ_STAMP(0)
        // ---- A: logits -------------------------------------------------------------------------------------------
        float rmax[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        uint64_t src[3] = {tile_src(0), tile_src(1), tile_src(2)};           // the tiles whose chunks are being fetched (SGPR pairs)
        auto kdma = [ring_base, lane16](uint64_t sp, auto coff_, uint32_t slot) {
            constexpr int coff = decltype(coff_)::value;
            const uint32_t dst = ring_base + slot * 1024u;
            // (an instruction offset would move the LDS destination as well as the source: the chunk offset goes into the scalar base)
            // hipcc pads nothing inside the string: the scalar base may come straight from an s_add (SALU write -> VMEM read of
            // the SGPR: 5 wait states) and M0 is written right before the load that reads it (1 wait state).
            asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(lane16), "s"(sp + (uint64_t)(coff * 1024)) : "memory", "m0");
        };
        static_for<0, FZ_RING - 1>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            kdma(src[q / KV4], std::integral_constant<int, q % KV4>{}, (uint32_t)q);
        });
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(FZ_RING - 2) : "memory");  // chunk 0 has landed
        const uint32_t q_a = lds_addr(qs) + lane * 16, k_a0 = lds_addr(ring) + lane * 2;
        for (int i = 0; i < tpw; ++i) {
            const uint32_t ring_i = (uint32_t)((i * KV4) % FZ_RING);          // slot of this tile's chunk 0 (uniform)
            const uint32_t k_a = k_a0 + ring_i * 1024u;
            f32x4 acc = {0, 0, 0, 0};
            if constexpr (DT == KVC_BF16) {
                f32x4 A0, A1;
                uint32_t B0[4], B1[4];
                fz_ld<0, 0>(A0, B0, q_a, k_a);
                static_for<0, NG>([&](auto g_) {
                    constexpr int g = decltype(g_)::value, c = g / 2;
                    f32x4& Ac = (g & 1) ? A1 : A0;
                    uint32_t (&Bc)[4] = (g & 1) ? B1 : B0;
                    if constexpr ((g & 1) == 0) {                           // first group of chunk c: fetch chunk q + RING - 1, chunk q + 1 must be there
                        constexpr int ahead = c + FZ_RING - 1;
                        kdma(src[ahead / KV4], std::integral_constant<int, ahead % KV4>{}, (ring_i + (uint32_t)ahead) % FZ_RING);
                        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(FZ_RING - 2) : "memory");
                    }
                    if constexpr (g + 1 < NG) {
                        fz_ld<1024 * (g + 1), ((g + 1) / 2) * 1024 + ((g + 1) & 1) * 512>((g & 1) ? A0 : A1, (g & 1) ? B0 : B1, q_a, k_a);
                        wait_step<5>(Ac, Bc);
                    } else {
                        wait_step<0>(Ac, Bc);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Ac[e], u2f(Bc[e]), acc, 0, 0, 0);
                });
            } else {
                static_for<0, KV4>([&](auto c_) {
                    constexpr int c = decltype(c_)::value, ahead = c + FZ_RING - 1;
                    kdma(src[ahead / KV4], std::integral_constant<int, ahead % KV4>{}, (ring_i + (uint32_t)ahead) % FZ_RING);
                    // chunk q has landed.  The lane's offset is an operand of the wait: hipcc moved plain LDS loads ABOVE an asm wait
                    // that only clobbers "memory" (seen in the listing; run-to-run different fp16 scores) — a data dependence it keeps.
                    uint32_t koff = lane * 2;
                    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(koff) : "n"(FZ_RING - 1) : "memory");
                    const raw* const kc = reinterpret_cast<const raw*>(ring + ((ring_i + c) % FZ_RING) * 1024 + koff);
                    uint32_t kr[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) kr[e] = kc[e * 64];
                    // ... and the chunk's reads are RETIRED here, as operands of the wait: the next step's DMA overwrites this slot,
                    // and hipcc had placed that DMA right behind the issue of these reads (write-after-read on a busy LDS).
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kr[0]), "+v"(kr[1]), "+v"(kr[2]), "+v"(kr[3]), "+v"(kr[4]), "+v"(kr[5]), "+v"(kr[6]), "+v"(kr[7]) :: "memory");
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4 qv = *reinterpret_cast<const f32x4*>(qs + ((2 * c + h) * 64 + lane) * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[e], Dt<DT>::ld((raw)kr[4 * h + e]), acc, 0, 0, 0);
                    }
                });
            }
            src[0] = src[1]; src[1] = src[2]; src[2] = tile_src(i + 3);
            const int t0 = 16 * (16 * i + wave), key = t0 + n;
            const bool tail = t0 + 16 > L - W && rb + 16 > L - W;               // (uniform) the tile touches the masked window block
            uint32_t w01, w23;
            if (!tail && t0 + 16 <= L && rb + 16 <= L) {
                // every row and key of the tile is valid and unmasked: packed arithmetic, one guard for the four values
                const f32x2 y01 = round2<DT>(f32x2{acc[0], acc[1]}), y23 = round2<DT>(f32x2{acc[2], acc[3]});
                f32x2 z01, z23;
                if constexpr (D == 64) {
                    z01 = y01 * (f32x2)0.125f; z23 = y23 * (f32x2)0.125f;
                } else {
                    const f32x2 rc = u2f(0x3db504f3u), cc = sqrt_d;             // the 2-FMA division of h2o_scale, on pairs
                    const f32x2 q01 = y01 * rc, q23 = y23 * rc;
                    z01 = __builtin_elementwise_fma(__builtin_elementwise_fma(-q01, cc, y01), rc, q01);
                    z23 = __builtin_elementwise_fma(__builtin_elementwise_fma(-q23, cc, y23), rc, q23);
                    const float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(y01.x), __builtin_fabsf(y01.y)), __builtin_fminf(__builtin_fabsf(y23.x), __builtin_fabsf(y23.y)));
                    const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(y01.x), __builtin_fabsf(y01.y)), __builtin_fmaxf(__builtin_fabsf(y23.x), __builtin_fabsf(y23.y)));
                    if (__builtin_expect(!(lo >= u2f(0x0d800000u) && hi < __builtin_inff()), 0)) {   // outside the proven range: IEEE division
                        z01 = f32x2{h2o_scale<D>(y01.x, sqrt_d), h2o_scale<D>(y01.y, sqrt_d)};
                        z23 = f32x2{h2o_scale<D>(y23.x, sqrt_d), h2o_scale<D>(y23.y, sqrt_d)};
                    }
                }
                // (rounding is monotone: the maximum of the rounded logits is the rounded maximum — rmax is rounded once, below)
                rmax[0] = vmax(rmax[0], z01.x); rmax[1] = vmax(rmax[1], z01.y);
                rmax[2] = vmax(rmax[2], z23.x); rmax[3] = vmax(rmax[3], z23.y);
                w01 = pack2<DT>(z01); w23 = pack2<DT>(z23);
            } else {
                float x[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int r = rb + 4 * v + gq;
                    float y = rnd<DT>(acc[v]);
                    y = rnd<DT>(h2o_scale<D>(y, sqrt_d));
                    if (tail) {
                        if (r >= L - W && key >= L - W && (key - (L - W)) > (r - (L - W))) y = rnd<DT>(y + Dt<DT>::finfo_min());
                    }
                    x[v] = y;
                    rmax[v] = (r < L && key < L && y > rmax[v]) ? y : rmax[v];
                }
                w01 = (uint32_t)Dt<DT>::st(x[0]) | ((uint32_t)Dt<DT>::st(x[1]) << 16);
                w23 = (uint32_t)Dt<DT>::st(x[2]) | ((uint32_t)Dt<DT>::st(x[3]) << 16);
            }
            s_put(i, w01, w23);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the ring (aliased by the exponentials' ring) is quiet
        KVC_FZ_STAMP(1)
        // ---- B: row maxima across the 16 key lanes, then across the waves --------------------------------------------
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float m = rnd<DT>(rmax[v]), o;
            o = xor_lane<1>(m); m = o > m ? o : m;
            o = xor_lane<2>(m); m = o > m ? o : m;
            o = xor_lane<4>(m); m = o > m ? o : m;
            o = xor_lane<8>(m); m = o > m ? o : m;
            if (n == 0) mx[(4 * v + gq) * 16 + wave] = m;
        }
        __syncthreads();
        f32x2 nm01, nm23;                                                   // minus the row maxima of rows gq, 4 + gq | 8 + gq, 12 + gq
        {
            float m[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4* mp = reinterpret_cast<const f32x4*>(mx + (4 * v + gq) * 16);
                float t = -__builtin_inff();
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 u = mp[q4];
                    t = vmax(vmax(t, u[0]), vmax(u[1], vmax(u[2], u[3])));
                }
                m[v] = t;
            }
            nm01 = f32x2{-m[0], -m[1]}; nm23 = f32x2{-m[2], -m[3]};
        }
        KVC_FZ_STAMP(2)
        // ---- C: denominators in torch's order ---------------------------------------------------------------------------
        float cacc = 0.0f;                                                   // waves 0-3: chain n of row 4 wave + gq
        for (int rd = 0; 2 * rd < tpw; ++rd) {
            float* const eb = ebuf + (rd & 1) * FZ_EBUF;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = 2 * rd + u;
                if (i < tpw) {
                    uint32_t w01, w23;
                    s_get(i, w01, w23);
                    f32x2 e01 = exp_u20x2_nonpos(f32x2{wide_lo<DT>(w01), wide_hi<DT>(w01)} + nm01);
                    f32x2 e23 = exp_u20x2_nonpos(f32x2{wide_lo<DT>(w23), wide_hi<DT>(w23)} + nm23);
                    const int t0 = 16 * (16 * i + wave);
                    float* const dst = eb + (u * 16 + wave) * 256 + lane;
                    if (__builtin_expect(t0 + 16 > L, 0)) {                  // (uniform, a real branch) keys beyond L add nothing
                        const bool kvalid = t0 + n < L;
                        asm volatile("" ::: "memory");
                        dst[0] = kvalid ? e01.x : 0.0f; dst[64] = kvalid ? e01.y : 0.0f;
                        dst[128] = kvalid ? e23.x : 0.0f; dst[192] = kvalid ? e23.y : 0.0f;
                    } else {
                        dst[0] = e01.x; dst[64] = e01.y; dst[128] = e23.x; dst[192] = e23.y;
                    }
                }
            }
            __syncthreads();      // this round's exponentials are in the ring; the chain waves finished the previous round's half
                                  // before they arrived here, so the workers may refill that half while this one is being added
            if (wave < 4) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (2 * rd + u < tpw) {
                        const float* srcp = eb + u * 16 * 256 + wave * 64 + lane;
                        float t[16];
#pragma unroll
                        for (int w2 = 0; w2 < 16; ++w2) t[w2] = srcp[w2 * 256];
#pragma unroll
                        for (int w2 = 0; w2 < 16; ++w2) cacc = cacc + t[w2];
                    }
                }
            }
        }
        if (wave < 4) {
            float s2 = cacc;
            s2 = s2 + xor_lane<8>(s2);
            s2 = s2 + xor_lane<4>(s2);
            s2 = s2 + xor_lane<2>(s2);
            s2 = s2 + xor_lane<1>(s2);
            if (n == 0) rinvs[4 * wave + gq] = 1.0f / s2;
        }
        __syncthreads();
        const f32x2 ri01 = {rinvs[gq], rinvs[4 + gq]}, ri23 = {rinvs[8 + gq], rinvs[12 + gq]};
        KVC_FZ_STAMP(3)
        // ---- D: probabilities, summed over the 16 rows in row order by the matrix core ------------------------------------------
        for (int i = 0; i < tpw; ++i) {
            uint32_t w01, w23;
            s_get(i, w01, w23);
            const int t0 = 16 * (16 * i + wave), key = t0 + n;
            const float a1v = a1s[key];                                      // (key < FZ_MAX_L always; read ahead of its use)
            f32x2 p01 = round2<DT>(exp_u20x2_nonpos(f32x2{wide_lo<DT>(w01), wide_hi<DT>(w01)} + nm01) * ri01);
            f32x2 p23 = round2<DT>(exp_u20x2_nonpos(f32x2{wide_lo<DT>(w23), wide_hi<DT>(w23)} + nm23) * ri23);
            if (__builtin_expect(t0 + 16 > L || !complete, 0)) {             // (uniform, a real branch) rows / keys beyond L add nothing
                asm volatile("" ::: "memory");
                const bool kvalid = key < L;
                p01.x = (kvalid && rb + gq < L) ? p01.x : 0.0f; p01.y = (kvalid && rb + 4 + gq < L) ? p01.y : 0.0f;
                p23.x = (kvalid && rb + 8 + gq < L) ? p23.x : 0.0f; p23.y = (kvalid && rb + 12 + gq < L) ? p23.y : 0.0f;
            }
            f32x4 cs = {0, 0, 0, 0};
            cs = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, p01.x, cs, 0, 0, 0);
            cs = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, p01.y, cs, 0, 0, 0);
            cs = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, p23.x, cs, 0, 0, 0);
            cs = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, p23.y, cs, 0, 0, 0);
            const float a0 = cs[0];                                          // the 16-row sum of column `key` (every lane group holds a copy)
            if (complete) {
                if (gq == 0) a1s[key] = a1v + a0;
            } else if (gq == 0 && key < ncol) {
                left[key] = a0;                                              // rows beyond the last full 16-row chunk stay in a0
            }
        }
        KVC_FZ_STAMP(4)
    }
#ifdef KVC_STAMPS
    if (lane == 0 && (wave == 0 || wave == 5)) {
        uint64_t* o = reinterpret_cast<uint64_t*>(a.scores) + ((int64_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 8;
        for (int e = 0; e < 5; ++e) o[e] = fz_t[e];
    }
#endif
    __syncthreads();
    for (int key = tid; key < ncol; key += FZ_THREADS) part[key] = a1s[key];
    if (fm.blk == n_blk - 1 && (L & 15) == 0) {
        for (int key = tid; key < ncol; key += FZ_THREADS) left[key] = 0.0f;
    }
}

// Query rows per chunk of the exact mode's logit matrix S (multiple of 256: the column sums combine 256-row blocks): as
// many as fit in kH2OSBudget bytes, at least 512 — [Hq][rows][L] instead of [Hq][L][L] (4.1 GB at 32 heads x 8k, 65 GB at 32k).
constexpr size_t kH2OSBudget = (size_t)1 << 30;
int h2o_chunk_rows(int heads, int L, int esize) {
    const size_t per256 = (size_t)heads * 256 * (size_t)L * esize;
    size_t rows = 256 * (kH2OSBudget / (per256 ? per256 : 1));
    if (rows < 512) rows = 512;
    if (const char* e = getenv("KVC_H2O_CHUNK_ROWS")) { const long r = atol(e); if (r >= 256 && r % 256 == 0) rows = (size_t)r; }   // tuning aid
    return rows >= (size_t)L ? L : (int)rows;
}

// The exact mode runs as the fused kernel (no logit matrix in the workspace, only the permuted copy of K) for 16-bit dtypes up
// to FZ_MAX_L keys unless debug_stage_mask bit 11 asks for round 2's kernels.  The workspace layout (kvc_api.hip) asks the same question.
bool h2o_fused_eligible(int dtype, int L, int legacy) { return dtype != KVC_FP32 && !legacy && L >= 16 && L <= FZ_MAX_L; }

template <int DT, int D>
static int launch_h2o_t(const H2OArgs& a0, hipStream_t st) {
    constexpr int ES = Dt<DT>::esize;
    H2OArgs a = a0;
    const int L = a.q_len, n = L - a.window, heads = a.bsz * a.n_q_heads;
    if constexpr (DT != KVC_FP32) {
        if (a.fast) {
            const size_t lds1 = (size_t)2 * 32 * (D * 2 + 16), lds2 = (size_t)2 * (32 * (D * 2 + 16) + 256);
            const unsigned g1 = (unsigned)(((L + 255) / 256) * heads), g2 = (unsigned)(((n + 255) / 256) * heads);
            hipLaunchKernelGGL((h2o_fast_stats_kernel<DT, D>), dim3(g1), dim3(FAST_THREADS), lds1, st, a);
            hipLaunchKernelGGL((h2o_fast_colsum_kernel<DT, D>), dim3(g2), dim3(FAST_THREADS), lds2, st, a);
            return 0;
        }
    }
    bool wide = false;
    if constexpr (DT != KVC_FP32) wide = (L % 8) == 0 && (a.window % 2) == 0 && L >= 16;
    if constexpr (DT != KVC_FP32) {
        if (h2o_fused_eligible(DT, L, a.legacy) && a.part && a.kt) {
            const size_t lds_f = fz_lds_bytes(D);
            static LdsCache c_f = {};
            if (ensure_lds(reinterpret_cast<const void*>(&h2o_fused_kernel<DT, D>), lds_f, c_f) != 0) return KVC_ERR_HIP;
            hipLaunchKernelGGL((h2o_kperm_kernel<DT, D>), dim3((unsigned)((((L + 15) / 16) * (D / 32) * 64 + 255) / 256), (unsigned)(a.bsz * a.n_kv_heads)), dim3(256), 0, st, a);
            hipLaunchKernelGGL((h2o_fused_kernel<DT, D>), dim3((unsigned)(heads * ((L + 255) / 256))), dim3(FZ_THREADS), lds_f, st, a);
#ifndef KVC_STAMPS
            if (wide) hipLaunchKernelGGL((h2o_colcomb_kernel<DT>), dim3((unsigned)((n / 2 + 255) / 256), (unsigned)heads), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((h2o_colcomb1_kernel<DT>), dim3((unsigned)((n + 255) / 256), (unsigned)heads), dim3(256), 0, st, a);
#endif
            return 0;
        }
    }
    const size_t lds = (size_t)4 * 2 * 32 * (D * ES + (DT == KVC_BF16 ? 4 : 0));   // 4 waves x 2 tile buffers (ROWP pitch)
    static LdsCache lds_cache = {};
    if (ensure_lds(reinterpret_cast<const void*>(&h2o_logits_kernel<DT, D>), lds, lds_cache) != 0) return KVC_ERR_HIP;
    const bool chunked = a.s_rows < L;
    const bool parts = a.part != nullptr && (chunked || (wide && L >= 512));    // 256-row blocks combined afterwards
    if (chunked && !a.part) return KVC_ERR_WORKSPACE;
    for (int row0 = 0; row0 < L; row0 += a.s_rows) {
        a.row0 = row0;
        a.rows = L - row0 < a.s_rows ? L - row0 : a.s_rows;
        const int row_tiles = (a.rows + 31) / 32;
        hipLaunchKernelGGL((h2o_logits_kernel<DT, D>), dim3((unsigned)((row_tiles + 3) / 4), (unsigned)heads), dim3(256), lds, st, a);
        const unsigned n_blk_c = (unsigned)((a.rows + 255) / 256);
        if constexpr (DT != KVC_FP32) {
            if (wide) {
                hipLaunchKernelGGL((h2o_rowsum_wide_kernel<DT>), dim3((unsigned)((a.rows + 127) / 128), (unsigned)heads), dim3(256), 0, st, a);
                if (parts) hipLaunchKernelGGL((h2o_colpart_kernel<DT>), dim3((unsigned)((n / 2 + 255) / 256), (unsigned)heads, n_blk_c), dim3(256), 0, st, a);
                else hipLaunchKernelGGL((h2o_colsum_wide_kernel<DT>), dim3((unsigned)((n / 2 + 255) / 256), (unsigned)heads), dim3(256), 0, st, a);
            }
        }
        if (!wide) {
            hipLaunchKernelGGL((h2o_rowsum_kernel<DT>), dim3((unsigned)((a.rows + 15) / 16), (unsigned)heads), dim3(256), 0, st, a);
            if (parts) hipLaunchKernelGGL((h2o_colpart1_kernel<DT>), dim3((unsigned)((n + 255) / 256), (unsigned)heads, n_blk_c), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((h2o_colsum_kernel<DT>), dim3((unsigned)((n + 255) / 256), (unsigned)heads), dim3(256), 0, st, a);
        }
    }
    if (parts) {
        if (wide) {
            if constexpr (DT != KVC_FP32)
                hipLaunchKernelGGL((h2o_colcomb_kernel<DT>), dim3((unsigned)((n / 2 + 255) / 256), (unsigned)heads), dim3(256), 0, st, a);
        } else {
            hipLaunchKernelGGL((h2o_colcomb1_kernel<DT>), dim3((unsigned)((n + 255) / 256), (unsigned)heads), dim3(256), 0, st, a);
        }
    }
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
