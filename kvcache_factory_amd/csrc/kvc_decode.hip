// kvc_decode.hip — SURVEY §8(f) N1: the decode step over the compacted cache.
//
// Reference (llama_model.py:287-289, :306-313): the new token's K/V are repeat_kv-expanded to H_q heads, `torch.cat`-ed
// onto the [1, H_q, cap + t, D] cache, and SDPA attends over the result — per layer and token: two expansions, two whole-
// cache copies and an attention over tensors that hold every decoded token H_q / H_kv times.
//
// Here the cache has two parts:
//   prefix  [b][H_q][P][D]   the compressed prompt exactly as kvc_compress wrote it (P = k + W rows per QUERY head: each
//                            query head selected its own rows, pyramidkv_utils.py:334-346) — never touched again;
//   tail    [b][H_kv][cap][D] the decoded tokens, stored ONCE per KV head (the group's query heads share them) and
//                            appended in place.
// kvc_decode_step = decode_append_kernel (the step's K/V rows -> tail) + decode_attention_kernel (one workgroup per
// (batch, query head, new token): logits over its prefix rows and its KV head's tail rows, online softmax, weighted V
// sum, fp32 accumulation; token i of the step sees tail rows < tail_rows + i + 1 — the causal rule of :306-313).
// The output equals SDPA over the reference-shaped cache within the dtype's rounding (tests: decode logits of a model
// run on this cache vs the oracle-built 32-head cache).  Latency-bound by design (a decode step touches (P + T) * D * 2
// bytes per head): what matters is that nothing else is copied or expanded.
#include "kvc_common.h"
#include "kvc_launch.h"

namespace kvc {

template <int DT> __device__ __forceinline__ void widen_chunk(const uint4& v, float* x) {
    if constexpr (DT == KVC_FP32) {
        x[0] = u2f(v.x); x[1] = u2f(v.y); x[2] = u2f(v.z); x[3] = u2f(v.w);
    } else {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[2 * i] = Dt<DT>::ld((uint16_t)(w[i] & 0xffffu));
            x[2 * i + 1] = Dt<DT>::ld((uint16_t)(w[i] >> 16));
        }
    }
}

// grid = (rows = new_rows, b * H_kv, 2 {K, V}), block = 64: 16-byte pieces of one row
__global__ __launch_bounds__(64) void decode_append_kernel(const DecodeArgs a) {
    const int i = blockIdx.x, bg = blockIdx.y, which = blockIdx.z;
    const int b = bg / a.n_kv_heads, g = bg % a.n_kv_heads;
    const int pieces = a.row_bytes >> 4;
    const char* src = reinterpret_cast<const char*>(which ? a.v_new : a.k_new) +
        ((int64_t)b * a.new_stride_b + (int64_t)g * a.new_stride_h + (int64_t)i * a.new_stride_l) * a.esize;
    char* dst = reinterpret_cast<char*>(which ? a.v_tail : a.k_tail) +
        ((int64_t)b * a.tail_stride_b + (int64_t)g * a.tail_stride_h) * a.esize + (int64_t)(a.tail_rows + i) * a.row_bytes;
    for (int c = threadIdx.x; c < pieces; c += 64)
        reinterpret_cast<uint4*>(dst)[c] = reinterpret_cast<const uint4*>(src)[c];
}

// grid = (b * H_q, new_rows, splits), block = 256 (4 waves).  LPR lanes share a row (16 bytes each); a block iteration takes
// UNR * 4 * RPW rows with every load of the iteration requested before the first is used (round 2: one row group per
// iteration and one workgroup per head — a 2 056-row cache took 80 us, all of it memory latency; tools/decode_timing.py).
// splits > 1: the rows of a head are divided among `splits` workgroups which leave (m, l, acc[D]) in the workspace, and
// decode_combine_kernel merges them (flash-decoding).  ThinK (pruned_rows > 0): the first rows' keys are read from k_pruned
// (pruned_dim kept channels per row, zero-padded to 16 bytes) against the query's kept channels (chan[]); their values are
// the leading rows of v_prefix.
template <int DT, int D>
__global__ __launch_bounds__(256) void decode_attention_kernel(const DecodeArgs a) {
    typedef typename Dt<DT>::raw raw;
    constexpr int ES = Dt<DT>::esize, EPL = 16 / ES, LPR = D / EPL;        // elements per lane, lanes per row
    constexpr int RPW = 64 / LPR, RPI = 4 * RPW, UNR = 4;                   // rows per wave / per block pass; passes per iteration
    __shared__ float sm_m[4 * RPW], sm_l[4 * RPW];
    __shared__ float sm_acc[4 * RPW][D];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, slot = lane / LPR, c = lane % LPR;
    const int hb = blockIdx.x, b = hb / a.n_q_heads, h = hb % a.n_q_heads, g = h / a.group;
    const int i = blockIdx.y;                                                // token of this step
    const int PA = a.pruned_rows, P = PA + a.prefix_rows, T = a.tail_rows + i + 1;   // visible rows: pruned, prefix, then tail
    const int R = P + T;
    const int r_begin = blockIdx.z * a.rows_per_split, r_end = min(R, r_begin + a.rows_per_split);
    const float scale = a.scaling * 1.44269504088896340736f;                 // logits in the log2 domain
    float qf[EPL], qa[EPL];
    {
        const char* qrow = reinterpret_cast<const char*>(a.q) +
            ((int64_t)b * a.q_stride_b + (int64_t)h * a.q_stride_h + (int64_t)i * a.q_stride_l) * ES;
        widen_chunk<DT>(reinterpret_cast<const uint4*>(qrow)[c], qf);
#pragma unroll
        for (int e = 0; e < EPL; ++e) { qf[e] *= scale; qa[e] = 0.0f; }
        if (PA > 0) {                                                        // the query's kept channels, in k_pruned's column order
            const int16_t* ch = a.chan + (int64_t)b * a.chan_stride_b + (int64_t)h * a.chan_stride_h;
            const raw* qr = reinterpret_cast<const raw*>(qrow);
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int col = c * EPL + e;
                const int src = col < a.pruned_dim ? (int)ch[col] : -1;
                qa[e] = src >= 0 ? Dt<DT>::ld(qr[src]) * scale : 0.0f;
            }
        }
    }
    const char* ka = reinterpret_cast<const char*>(a.k_pruned) + ((int64_t)b * a.pruned_stride_b + (int64_t)h * a.pruned_stride_h) * ES;
    const char* kp = reinterpret_cast<const char*>(a.k_prefix) + ((int64_t)b * a.prefix_stride_b + (int64_t)h * a.prefix_stride_h) * ES;
    const char* vp = reinterpret_cast<const char*>(a.v_prefix) + ((int64_t)b * a.vprefix_stride_b + (int64_t)h * a.vprefix_stride_h) * ES;
    const char* kt = reinterpret_cast<const char*>(a.k_tail) + ((int64_t)b * a.tail_stride_b + (int64_t)g * a.tail_stride_h) * ES;
    const char* vt = reinterpret_cast<const char*>(a.v_tail) + ((int64_t)b * a.tail_stride_b + (int64_t)g * a.tail_stride_h) * ES;
    constexpr int ROWB = D * ES;
    const int rowb_a = a.pruned_dim * ES, lanes_a = rowb_a >> 4;
    float m = -1.0e30f, l = 0.0f, acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.0f;
    for (int r0 = r_begin; r0 < r_end; r0 += RPI * UNR) {
        uint4 kv[UNR], vv[UNR];
        bool valid[UNR], pruned[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int r = r0 + u * RPI + wave * RPW + slot;
            valid[u] = r < r_end;
            pruned[u] = r < PA;
            const char* krow = r < PA ? ka + (int64_t)r * rowb_a : (r < P ? kp + (int64_t)(r - PA) * ROWB : kt + (int64_t)(r - P) * ROWB);
            const char* vrow = r < P ? vp + (int64_t)r * ROWB : vt + (int64_t)(r - P) * ROWB;
            const bool kload = valid[u] && (!pruned[u] || c < lanes_a);
            kv[u] = kload ? reinterpret_cast<const uint4*>(krow)[c] : make_uint4(0, 0, 0, 0);
            vv[u] = valid[u] ? reinterpret_cast<const uint4*>(vrow)[c] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float kf[EPL], vf[EPL];
            widen_chunk<DT>(kv[u], kf);
            widen_chunk<DT>(vv[u], vf);
            float s = 0.0f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) s = __builtin_fmaf(pruned[u] ? qa[e] : qf[e], kf[e], s);
            // sum over the LPR lanes of the row (LPR is a power of two <= 32)
            if constexpr (LPR > 1) s += xor_lane<1>(s);
            if constexpr (LPR > 2) s += xor_lane<2>(s);
            if constexpr (LPR > 4) s += xor_lane<4>(s);
            if constexpr (LPR > 8) s += xor_lane<8>(s);
            if constexpr (LPR > 16) s += xor_lane<16>(s);
            if (valid[u]) {
                const float mn = s > m ? s : m;
                const float f = __builtin_amdgcn_exp2f(m - mn), p = __builtin_amdgcn_exp2f(s - mn);
                l = __builtin_fmaf(l, f, p);
#pragma unroll
                for (int e = 0; e < EPL; ++e) acc[e] = __builtin_fmaf(acc[e], f, p * vf[e]);
                m = mn;
            }
        }
    }
    // merge the 4 * RPW row slots of the block
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
        if (a.splits == 1) {
            raw* out = reinterpret_cast<raw*>(a.out) + ((int64_t)b * a.out_stride_b + (int64_t)h * a.out_stride_h + (int64_t)i * a.out_stride_l);
            out[tid] = Dt<DT>::st(o / Lsum);
        } else {                                                             // (m, l, acc) of this split
            float* part = a.part + (((int64_t)hb * a.new_rows + i) * a.splits + blockIdx.z) * (D + 2);
            part[2 + tid] = o;
            if (tid == 0) { part[0] = M; part[1] = Lsum; }
        }
    }
}

// grid = (b * H_q, new_rows), block = D: the splits' partial results -> out
template <int DT, int D>
__global__ __launch_bounds__(D) void decode_combine_kernel(const DecodeArgs a) {
    typedef typename Dt<DT>::raw raw;
    const int hb = blockIdx.x, b = hb / a.n_q_heads, h = hb % a.n_q_heads, i = blockIdx.y, tid = threadIdx.x;
    const float* part = a.part + ((int64_t)hb * a.new_rows + i) * a.splits * (D + 2);
    float M = -1.0e30f;
    for (int s = 0; s < a.splits; ++s) { const float ms = part[(int64_t)s * (D + 2)]; M = ms > M ? ms : M; }
    float Lsum = 0.0f, o = 0.0f;
    for (int s = 0; s < a.splits; ++s) {
        const float* ps = part + (int64_t)s * (D + 2);
        const float f = __builtin_amdgcn_exp2f(ps[0] - M);
        Lsum = __builtin_fmaf(ps[1], f, Lsum);
        o = __builtin_fmaf(ps[2 + tid], f, o);
    }
    raw* out = reinterpret_cast<raw*>(a.out) + ((int64_t)b * a.out_stride_b + (int64_t)h * a.out_stride_h + (int64_t)i * a.out_stride_l);
    out[tid] = Dt<DT>::st(o / Lsum);
}

// How many workgroups share a head's rows: enough to put ~4 workgroups on every CU, never less than 64 rows each.
int decode_splits(int bsz, int n_q_heads, int new_rows, int rows) {
    const int heads = bsz * n_q_heads * new_rows;
    int s = (1024 + heads - 1) / heads;
    const int by_rows = (rows + 63) / 64;
    if (s > by_rows) s = by_rows;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}
size_t decode_part_bytes(int bsz, int n_q_heads, int new_rows, int head_dim, int splits) {
    return splits <= 1 ? 0 : (size_t)bsz * n_q_heads * new_rows * splits * (head_dim + 2) * sizeof(float);
}

int launch_decode_step(DecodeArgs a, int dtype, int head_dim, hipStream_t st) {
    hipLaunchKernelGGL(decode_append_kernel, dim3((unsigned)a.new_rows, (unsigned)(a.bsz * a.n_kv_heads), 2), dim3(64), 0, st, a);
    const int rows = a.pruned_rows + a.prefix_rows + a.tail_rows + a.new_rows;
    if (a.splits < 1 || a.part == nullptr) a.splits = 1;
    a.rows_per_split = a.splits == 1 ? rows : ((rows + a.splits - 1) / a.splits + 63) / 64 * 64;
    if (a.splits > 1) a.splits = (rows + a.rows_per_split - 1) / a.rows_per_split;          // no empty split
    const dim3 grid((unsigned)(a.bsz * a.n_q_heads), (unsigned)a.new_rows, (unsigned)a.splits), cgrid((unsigned)(a.bsz * a.n_q_heads), (unsigned)a.new_rows);
#define KVC_CASE(DT_, D_) if (dtype == DT_ && head_dim == D_) { hipLaunchKernelGGL((decode_attention_kernel<DT_, D_>), grid, dim3(256), 0, st, a); \
        if (a.splits > 1) hipLaunchKernelGGL((decode_combine_kernel<DT_, D_>), cgrid, dim3(D_), 0, st, a); return 0; }
    KVC_CASE(KVC_BF16, 128) KVC_CASE(KVC_BF16, 64) KVC_CASE(KVC_FP16, 128) KVC_CASE(KVC_FP16, 64)
    KVC_CASE(KVC_FP32, 128) KVC_CASE(KVC_FP32, 64)
#undef KVC_CASE
    return KVC_ERR_UNSUPPORTED;
}

}  // namespace kvc
