// kvc_launch.h — argument blocks and host launch functions shared by the kernel files and kvc_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kvc {

// One launch of every kernel can serve up to KVC_MAX_ITEMS independent "items" (the layers of one prompt:
// kvc_compress_batch) — same shapes, per-item tensors and per-item k.  Pointer tables travel in the kernel arguments
// (no device-side copies); blockIdx.z (or .y for the per-head kernels) is the item.  kvc_compress is the 1-item case.
constexpr int KVC_MAX_ITEMS = 32;
struct PtrTable { const void* p[KVC_MAX_ITEMS]; };
struct IntTable { int v[KVC_MAX_ITEMS]; };
struct I64Table { int64_t v[KVC_MAX_ITEMS]; };

struct ScoreView {         // one item's pointers, resolved at kernel entry
    const void* q; const void* k;
    void* logits; float* pmax; float* rowmax; float* rowsum; void* scores;
    float* part16;         // [bsz*Hq][W/16][n_pad] fp32: in-order sums of 16 window rows (softmax_rows16_kernel), or null
};

struct ScoreArgs {
    PtrTable q, k, scores; // per item; scores [bsz*Hq][L-W] dtype
    char* ws;              // workspace of item 0; item i lives ws_item_stride bytes further
    int64_t ws_item_stride;
    // byte offsets inside an item's workspace:
    int64_t off_logits;    // [bsz*Hq][L][W] dtype
    int64_t off_pmax;      // [bsz*Hq][n_tiles][W] fp32
    int64_t off_rowmax;    // [bsz*Hq][W] fp32
    int64_t off_rowsum;    // [bsz*Hq][W] fp32
    int64_t off_part16;    // [bsz*Hq][W/16][n_pad] fp32, or -1 (window not a multiple of 16, fp32)
    int n_items;
    int64_t q_stride_b, q_stride_h, q_stride_l;
    int64_t k_stride_b, k_stride_h, k_stride_l;
    int bsz, n_q_heads, n_kv_heads, group, q_len, window;
    int n_tiles, kernel_size, pooling;
    int fast_dot;          // 1: packed bf16/fp16 MFMA scan (tolerance mode), 0: exact f32 fmaf-chain MFMA
    int window_mean;       // 1: the window rows are averaged (AdaKV / HeadKV, .mean(dim=-2)) instead of summed
    int stage_mask;        // bits 0-2: 0 = all, else bit0 logits, bit1 rowsum, bit2 pool (profiling aid); bit3 / bit4: force the split / fused softmax path
    float sqrt_d;
    unsigned long long* dbg;   // diagnostic stamps (KVC_STAMPS builds only), else null
};

struct H2OArgs {
    const void* q; const void* k;
    void* S;               // [bsz*Hq][s_rows][L] dtype logits of the query rows [row0, row0 + rows) (exact mode)
    int row0, rows, s_rows;    // the chunk of query rows in flight and the row capacity of S per head (launcher-set)
    int fast;              // 1: dot_mode mfma16 — two recomputing passes on the bf16/fp16 MFMA, no S at all
    void* kt;              // exact mode, fused form: [bsz*Hkv][ceil(L/16)][D/32][64 lanes][16 B], K as the MFMA B operand is loaded
    int legacy;            // 1: round 2's materialising exact kernels (debug_stage_mask bit 11)
    float* rowmax;         // [bsz*Hq][L]   (fast mode: -max * log2(e))
    float* rinv;           // [bsz*Hq][L]
    float* part;           // [bsz*Hq][n_blk + 1][n_pad] column sums of each 256-row block (+ the leftover rows), or null
    void* scores;          // [bsz*Hq][L-W] dtype
    int64_t q_stride_b, q_stride_h, q_stride_l;
    int64_t k_stride_b, k_stride_h, k_stride_l;
    int bsz, n_q_heads, n_kv_heads, group, q_len, window;
    float sqrt_d;
};

struct GatherArgs {
    PtrTable src, out, idx;                           // per item; idx entry may be null (identity)
    IntTable k;                                       // per item
    int64_t stride_b, stride_h, stride_l;             // elements
    I64Table out_head_bytes;                          // per item: bytes between heads of `out`; 0 = dense ((k + window) rows)
    int bsz, n_q_heads, group, q_len, window, k_max, row_bytes, esize, n_items;
};

struct SelectArgs {
    PtrTable scores;       // per item: [bsz*Hq][n] dtype
    PtrTable idx;          // per item: int64 [bsz*Hq][k_item]
    IntTable k;            // per item
    int n, k_max, heads;   // heads = bsz*Hq
    int n_items;
    int pow2;              // next power of two >= k_max
    int fuse;              // 1: also gather K (gk) and V (gv) rows of the head in the same workgroup
    int members_only;      // exact mode, large partial_sort heaps: the caller needs the selected SET only (AdaKV's per-head counts)
    GatherArgs gk, gv;
};

struct GatherPair { GatherArgs t[2]; int count; };    // K and V compacted by one launch

struct RaggedSortArgs {    // sort_prefix_kernel (AdaKV / HeadKV)
    const void* scores;    // [heads][n] dtype
    const int32_t* caps;   // [heads] device: how many leading indices of the descending sort each head keeps
    int64_t* idx_out;      // [heads][out_stride]
    int n, heads, out_stride;
    int want_fixed = 0;    // caps == nullptr: leading indices kept by every head (0 = the whole sort)
    int ascending = 0;     // 1: torch-CPU's ascending sort (L2Norm's argsort, pyramidkv_utils.py:420)
};

struct MergeArgs {         // merge_kv pivot merge (kvc_merge.hip)
    const void* k; const void* v;
    const int64_t* idx;    // [bsz * n_q_heads][n_keep]: the select stage's indices
    void* k_out; void* v_out;
    int32_t* mask;         // [bsz][q_len] workspace
    int32_t* drop;         // [bsz][q_len]
    int32_t* drop_len;     // [bsz]
    float* kn;             // [bsz * n_q_heads][rows][128] normalised kept keys
    int32_t* pivot;        // [bsz * n_q_heads][q_len]
    int64_t k_stride_b, k_stride_h, k_stride_l, v_stride_b, v_stride_h, v_stride_l, out_stride_h;
    int bsz, n_q_heads, group, q_len, window, n_keep, rows;      // rows = n_keep + window
};

struct ThinkArgs {         // ThinK channel pruning (kvc_think.hip)
    const void* q;         // [bsz][n_heads][q_len][head_dim] by strides: the last 32 rows are read
    const void* kc;        // compressed keys [bsz * n_heads][rows][head_dim], kc_stride_h elements between heads, rows dense
    void* pruned;          // [bsz * n_heads][rows - recent][head_dim - n_prune]
    uint8_t* keep;         // [bsz * n_heads][head_dim]: 1 = channel kept
    void* scores;          // [bsz * n_heads][head_dim] dtype or null
    int64_t q_stride_b, q_stride_h, q_stride_l, kc_stride_h;
    int bsz, n_heads, q_len, head_dim, rows, recent, n_prune;
};

struct L2NormArgs {        // l2norm_kernel (L2NormCluster, pyramidkv_utils.py:419)
    const void* k;         // [bsz][n_kv_heads][q_len][head_dim] by strides (elements)
    void* norms;           // [bsz][n_q_heads][q_len] dtype: the norm of KV head h / group in every one of its query-head rows
    int64_t stride_b, stride_h, stride_l;
    int bsz, n_q_heads, n_kv_heads, q_len, head_dim;
};

struct RaggedArgs {        // ragged_head_stats_kernel / ragged_budget_kernel (AdaKV budgets)
    const void* scores;    // [bsz][H][n] dtype (pooled window-mean scores)
    float* ratio;          // [bsz][H] fp32 scratch: sum(top base) / sum(all) per head, dtype-rounded
    int32_t* caps;         // [bsz][H] out: rows each head keeps (without the window)
    int32_t* flags;        // [bsz][2] out: [0] cross-head tie at the global threshold, [1] slots given to threshold-valued scores
    int n, heads_per_batch, base, floor_capacity, normalize;
    float mix;             // (float)(1 - floor_ratio)
};

struct RaggedGatherArgs {  // ragged_gather_kernel: K and V rows -> the flat cache
    const void* k; const void* v; void* k_flat; void* v_flat;
    const int64_t* idx;    // [bsz*H][idx_stride] sorted indices (sort_prefix_kernel)
    const int32_t* caps;   // [bsz*H]
    const int64_t* seg_off;    // [bsz*H] first row of each head's segment in the flat cache
    int64_t k_stride_b, k_stride_h, k_stride_l, v_stride_b, v_stride_h, v_stride_l;
    int idx_stride, n_q_heads, group, q_len, window, row_bytes, esize;
};

struct RaggedDecodeArgs {  // one decode step over the ragged cache
    const void* q; const void* k_new; const void* v_new; void* k_flat; void* v_flat; void* out;
    const int64_t* seg_off; const int32_t* seg_len;      // [bsz*H]: segment start (rows) and rows after the prefill
    int64_t q_stride_b, q_stride_h, q_stride_l, new_stride_b, new_stride_h, new_stride_l, out_stride_b, out_stride_h, out_stride_l;
    int n_q_heads, group, appended, new_rows, esize, row_bytes;
    float scaling;
};

struct DecodeArgs {        // one decode step over the compacted cache (kvc_decode.hip); strides in elements
    const void* q; const void* k_new; const void* v_new;
    const void* k_prefix; const void* v_prefix;       // [b][Hq][prefix rows][D]
    void* k_tail; void* v_tail;                       // [b][Hkv][capacity][D], rows [0, tail_rows) filled
    void* out;                                        // attention output, row (b, h, i) at out + b*sb + h*sh + i*sl
    int64_t q_stride_b, q_stride_h, q_stride_l;
    int64_t new_stride_b, new_stride_h, new_stride_l;
    int64_t prefix_stride_b, prefix_stride_h, tail_stride_b, tail_stride_h;
    int64_t out_stride_b, out_stride_h, out_stride_l;
    int bsz, n_q_heads, n_kv_heads, group, prefix_rows, tail_rows, new_rows, esize, row_bytes;
    float scaling;
    // kvc_decode_step2: values with their own strides, ThinK's channel-pruned leading rows, rows split over workgroups
    int64_t vprefix_stride_b, vprefix_stride_h;
    const void* k_pruned; const int16_t* chan;        // [b][Hq][pruned_rows][pruned_dim], [b][Hq][pruned_dim]
    int64_t pruned_stride_b, pruned_stride_h, chan_stride_b, chan_stride_h;
    int pruned_rows, pruned_dim;
    float* part; int splits, rows_per_split;          // workspace: (m, l, acc[D]) per (b, h, token, split)
};

// Raise a kernel's dynamic-LDS limit above 64 KB once per (kernel instantiation, device): `cache` is a function-local
// static array owned by the caller.  Never called again on the launch path, so launches stay graph-capturable.
struct LdsCache { size_t ok[16]; };
inline int ensure_lds(const void* func, size_t lds, LdsCache& cache) {
    if (lds <= 64 * 1024) return 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 15;
    if (lds <= cache.ok[dev]) return 0;
    if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    cache.ok[dev] = lds;
    return 0;
}

int launch_scores(const ScoreArgs& a, int dtype, int head_dim, hipStream_t st);
int launch_h2o_scores(const H2OArgs& a, int dtype, int head_dim, hipStream_t st);
int h2o_chunk_rows(int heads, int L, int esize);
bool h2o_fused_eligible(int dtype, int L, int legacy);   // exact mode: the one-kernel form applies (workspace: permuted K instead of S)   // query rows of the exact mode's logit matrix held at a time
int launch_select(const SelectArgs& a, int dtype, int tie_mode, hipStream_t st);
size_t select_lds_bytes(int k);
int launch_select_exact(const SelectArgs& a, int dtype, void* scratch, hipStream_t st);   // tie_mode torch_cpu
size_t select_exact_scratch_bytes(int heads, int n, int k);     // per item
int launch_gather(const GatherPair& p, hipStream_t st);
struct CamArgs {           // CAM (kvc_cam.hip)
    const void* logits; const float* rowmax; const float* rowsum;      // the scoring stage's workspace regions
    void* colmean;            // [b*Hq][L] dtype
    void* prob;               // [b*Hq][L - W - start] dtype
    void* v; const uint8_t* mask;                                       // merge: values in place, draws
    int64_t v_stride_b, v_stride_h, v_stride_l;
    int bsz, n_q_heads, q_len, window, start;
};
int launch_cam_prob(const CamArgs& a, int dtype, hipStream_t st);
int launch_cam_merge(const CamArgs& a, int dtype, int head_dim, hipStream_t st);
int launch_decode_step(DecodeArgs a, int dtype, int head_dim, hipStream_t st);
int decode_splits(int bsz, int n_q_heads, int new_rows, int rows);
size_t decode_part_bytes(int bsz, int n_q_heads, int new_rows, int head_dim, int splits);
int launch_sort_prefix(const RaggedSortArgs& a, int dtype, void* scratch, hipStream_t st);
int launch_l2norm(const L2NormArgs& a, int dtype, hipStream_t st);
int launch_merge(const MergeArgs& a, int dtype, hipStream_t st);
int launch_think(const ThinkArgs& a, int dtype, hipStream_t st);
size_t sort_prefix_scratch_bytes(int heads, int n);
int launch_ragged_plan(const RaggedArgs& a, int dtype, int bsz, hipStream_t st);
int launch_ragged_head_stats(const RaggedArgs& a, int dtype, int bsz, hipStream_t st);
int launch_ragged_flatten(const RaggedArgs& a, int dtype, int bsz, const int64_t* idx, void* flat, hipStream_t st);
int launch_ragged_recount(const RaggedArgs& a, int bsz, const int64_t* top, hipStream_t st);
int launch_ragged_gather(const RaggedGatherArgs& a, int rows_max, int heads, hipStream_t st);
int launch_ragged_decode(const RaggedDecodeArgs& a, int dtype, int head_dim, int heads, hipStream_t st);
static constexpr int kFuseGatherMaxK = 512;   // select_kernel gathers the rows itself up to this k

}  // namespace kvc
