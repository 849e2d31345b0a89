/* kvc.h — C-ABI of the MI355X (gfx950) KV-cache compression hot path.
 *
 * Drop-in boundary for ONE path of assassin808/KVCache-Factory (reference @ 2025-02-18): the
 * post-prefill scoring + eviction step behind `kv_cluster.update_kv(key_states, query_states,
 * value_states, attention_mask, num_key_value_groups)`:
 *     SnapKVCluster.update_kv          pyramidkv/pyramidkv_utils.py:306-347
 *     PyramidKVCluster.update_kv       pyramidkv/pyramidkv_utils.py:197-283
 *     H2OKVCluster.update_kv           pyramidkv/pyramidkv_utils.py:533-575
 *     StreamingLLMKVCluster.update_kv  pyramidkv/pyramidkv_utils.py:595-620
 * The reference has no C/FFI interface on this path (it is ~60 lines of torch ops per method);
 * these entry points are what a ctypes binding of `update_kv` binds (INTEGRATION.md shows the
 * stub).  Implemented in kvcache_factory_amd/csrc/ (hand-written HIP, libkvc_hip.so).
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer (HBM), 16-byte aligned; nothing is copied to the host;
 *   - the library never allocates device memory, never synchronises the stream and never throws:
 *     all kernels are enqueued on `hip_stream` (a hipStream_t, NULL = default stream);
 *   - tensors are indexed [b][h][l][d] with d contiguous and element strides given in the params;
 *   - outputs: k_out/v_out [bsz][n_q_heads][k+window][head_dim] in the input dtype, dense or with a caller-chosen
 *     head stride (kvc_params.out_stride_h: a decode cache with spare rows per head),
 *     idx_out [bsz][n_q_heads][k] int64 (value-descending order, like torch.topk), scores_out
 *     [bsz][n_q_heads][q_len-window] in the input dtype (the pooled scores of :328-333);
 *   - return value 0 on success, negative kvc_status otherwise; kvc_last_error() gives the text.
 */
#ifndef KVC_H_
#define KVC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KVC_VERSION 5

typedef enum kvc_status {
    KVC_OK = 0,
    KVC_ERR_INVALID = -1,      /* the reference would assert / raise (pyramidkv_utils.py:289,309,333) */
    KVC_ERR_UNSUPPORTED = -2,  /* valid for the reference, not implemented by this library (documented limits) */
    KVC_ERR_WORKSPACE = -3,    /* workspace missing, misaligned or smaller than kvc_workspace_bytes() */
    KVC_ERR_ALIGNMENT = -4,    /* a pointer or stride is not 16-byte aligned */
    KVC_ERR_HIP = -5           /* a HIP runtime call failed (message carries hipGetErrorString) */
} kvc_status;

typedef enum kvc_method {      /* which update_kv is being replaced */
    KVC_SNAPKV = 0,            /* pyramidkv_utils.py:306-347 */
    KVC_PYRAMIDKV = 1,         /* :197-283 — same kernels, caller passes the per-layer k (kvc_pyramid_k) */
    KVC_H2O = 2,               /* :533-575 — all q_len query rows, no pooling */
    KVC_STREAMINGLLM = 3,      /* :595-620 — no scoring: first k + last window tokens */
    KVC_ADAKV = 4,             /* :622-757 — window-MEAN scores, per-head budgets from a global top-(H*base), ragged cache:
                                  kvc_ragged_* entry points only (k = base capacity = max_capacity_prompt - W) */
    KVC_HEADKV = 5,            /* :760-878 — the same with per-head budgets given by the caller */
    KVC_L2NORM = 6             /* :394-429 — L2NormCluster: the k smallest-norm keys per head, ascending; no query, no window:
                                  kvc_l2norm_* entry points only (k = rows kept = max_capacity_prompt) */
} kvc_method;

typedef enum kvc_dtype { KVC_BF16 = 0, KVC_FP16 = 1, KVC_FP32 = 2 } kvc_dtype;
typedef enum kvc_pooling { KVC_POOL_NONE = 0, KVC_POOL_AVG = 1, KVC_POOL_MAX = 2 } kvc_pooling;

typedef enum kvc_tie_mode {
    KVC_TIES_TORCH_CPU = 0,    /* membership and order of equal scores exactly as torch-CPU topk
                                  (libstdc++ partial_sort when k*64<=n, else nth_element+sort) */
    KVC_TIES_CANONICAL = 1     /* value descending, index ascending (fast path) */
} kvc_tie_mode;

typedef enum kvc_dot_mode {
    KVC_DOT_EXACT = 0,         /* window QK^T as the d-ascending fp32 fmaf chain (f32-input MFMA): bit-identical to the CPU
                                  oracle, and the arithmetic the reference's fp16/fp32 GEMM was measured to use */
    KVC_DOT_MFMA16 = 1         /* packed bf16/fp16 16-deep MFMA (16x less matrix time; fp32 inputs fall back to EXACT).
                                  Tolerance mode: a logit may differ from EXACT by 1 unit in the last place (~1e-4 of
                                  the entries), which can move a candidate across the top-k threshold.  H2O: two
                                  recomputing passes with no logit matrix (workspace: two floats per query row), hardware
                                  exp and fp32 sums in tile order — pooled scores within 1 unit in the last place. */
} kvc_dot_mode;

typedef struct kvc_params {
    int32_t method;            /* kvc_method */
    int32_t dtype;             /* kvc_dtype of q, k, v and of every non-index output */
    int32_t bsz;               /* batch (the reference runs bsz = 1, README.md:29; code is batch-agnostic) */
    int32_t n_q_heads;         /* heads of q and of the outputs (32 for Llama-3-8B / Mistral-7B) */
    int32_t n_kv_heads;        /* heads present in k and v: n_q_heads when the caller passes the
                                  repeat_kv()-expanded tensors like the reference (llama_model.py:277-278),
                                  or the model's KV heads (8) to read each KV head once.  Query head h
                                  reads kv head h / (n_q_heads / n_kv_heads). */
    int32_t q_len;             /* L: key_states.shape[-2] == query_states.shape[-2] (:309) */
    int32_t head_dim;          /* D: 64 or 128 */
    int32_t window;            /* W = window_size: the last W queries score, the last W tokens are kept */
    int32_t k;                 /* prefix tokens kept per head: max_capacity_prompt - W (:334), or the
                                  PyramidKV per-layer budget (:214); 0 <= k <= L - W */
    int32_t kernel_size;       /* pooling kernel (odd; padding = kernel_size/2, stride 1) (:328-333) */
    int32_t pooling;           /* kvc_pooling; ignored for H2O (no pooling, :555-561) and StreamingLLM */
    int32_t tie_mode;          /* kvc_tie_mode */
    int32_t debug_stage_mask;  /* testing / profiling aid, 0 in production.  Bits 0-2 (kvc_scores, kvc_compress_batch): only the
                                  scoring kernels named by bit0 = K-scan (logits), bit1 = row-sum, bit2 = pool are
                                  enqueued; kvc_compress requires them 0.  Bit3 / bit4: force the two-kernel / the
                                  one-workgroup-per-head form of the softmax+pool stage (identical results; the
                                  library picks by the number of heads x items otherwise).  Bit5 / bit6
                                  (kvc_compress_batch, timing only): skip the scoring kernels / skip scoring and selection
                                  and reuse what the previous identical call left in the workspace and the outputs.  Bit7
                                  (H2O exact mode, testing): hold 512 query rows of the logit matrix at a time instead of
                                  as many as fit in 1 GiB (identical results).  Bit8 (measurement): the softmax stage reloads the
                                  logits in its second pass instead of keeping the exponentials in registers (identical results).
                                  Bit9 (measurement): G * W == 128 query rows per KV head scanned by the one-M-tile-at-a-time
                                  kernel instead of the four-waves-share-a-tile one (identical results).  Bit10 (measurement): at window 8 the
                                  one-workgroup-per-head softmax stage without its chain-wave / worker-wave split (identical results).
                                  Bit11 (H2O exact mode, testing): round 2's kernels that write the logit matrix to the
                                  workspace instead of the fused one that keeps 16 query rows of it in registers (identical results).
                                  Bit12 (measurement): at windows 16 / 32 / 64 the one-workgroup-per-head softmax stage that reads the
                                  logits twice instead of the 16-rows-per-workgroup form that reads them once (identical results). */
    int32_t dot_mode;          /* kvc_dot_mode */
    int64_t q_stride_b, q_stride_h, q_stride_l;   /* element strides of q[b][h][l][:] */
    int64_t k_stride_b, k_stride_h, k_stride_l;
    int64_t v_stride_b, v_stride_h, v_stride_l;
    int64_t out_stride_h;      /* elements between consecutive heads of k_out / v_out ([b][h_q][row][D], rows dense).
                                  0 = dense, (k + window) * head_dim.  Larger: the caller's decode cache with spare rows
                                  per head ([b][h_q][capacity][D]) — K' / V' are written straight into their final home
                                  (the reference cats them into a fresh tensor, llama_model.py:286).  A multiple of 8
                                  (16-bit) or 4 (fp32) elements, >= (largest k + window) * head_dim. */
} kvc_params;

/* Library / ABI version (== KVC_VERSION). */
int kvc_version(void);

/* Thread-local text of the last error returned on this thread ("" if none). */
const char* kvc_last_error(void);

/* Bytes of device scratch kvc_compress / kvc_scores / kvc_select need for these params
 * (0 for StreamingLLM).  Returns 0 and sets the error text if the params are invalid. */
size_t kvc_workspace_bytes(const kvc_params* p);

/* The whole update_kv: scores (A1-A5, or A10 for H2O) -> per-head top-k (A7) -> gather + window
 * tail (A8), for the method in p->method.  Replaces the body of
 *   SnapKVCluster.update_kv :317-346 / PyramidKVCluster.update_kv :220-283 /
 *   H2OKVCluster.update_kv :544-575 / StreamingLLMKVCluster.update_kv :607-620.
 * q: query tensor base: row l of head h is read at q + b*q_stride_b + h*q_stride_h + l*q_stride_l.  SnapKV / PyramidKV
 *    dereference rows L-W .. L-1 ONLY, so a caller that kept just the window rows ([b][h][W][D], as a layer-batching
 *    host does) passes window_base - (L-W)*q_stride_l; H2O reads every row; StreamingLLM ignores q (may be NULL).
 * idx_out and scores_out may be NULL (scores_out is never written for StreamingLLM). */
int kvc_compress(const kvc_params* p, const void* q, const void* k, const void* v,
                 void* k_out, void* v_out, int64_t* idx_out, void* scores_out,
                 void* workspace, size_t workspace_bytes, void* hip_stream);

/* Batched form: n_items independent update_kv calls of IDENTICAL shape/dtype/strides (described by *p; p->k is
 * ignored) and per-item budgets k_per_item[i] — e.g. the 32 layers of one prompt (the reference runs them as 32
 * sequential calls, llama_model.py:285; no call reads another's result, and the prefill attention of every layer runs
 * on the UNcompressed K/V, llama_model.py:306-313, so a host may collect the layers and compress them after the last
 * one).  Every kernel is launched once per chunk of 32 items (n_items may be larger; chunks reuse the workspace), so the
 * small per-call kernels stop being launch-latency bound.  The arrays are HOST arrays of DEVICE
 * pointers (they are copied into the kernel arguments; nothing is staged on the device).  idx_out / scores_out may be
 * NULL, as may their entries.  Results are identical to n_items kvc_compress calls.
 * out_stride_h_per_item: NULL (every item uses p->out_stride_h) or n_items head strides in elements, one per item,
 * 0 = dense — each layer's decode cache sized for its own budget (PyramidKV's layers differ by 14x at 8k).
 * Workspace: kvc_workspace_bytes_batch(). */
size_t kvc_workspace_bytes_batch(const kvc_params* p, int n_items, const int32_t* k_per_item);
int kvc_compress_batch(const kvc_params* p, int n_items, const int32_t* k_per_item,
                       const void* const* q, const void* const* k, const void* const* v,
                       void* const* k_out, void* const* v_out, int64_t* const* idx_out, void* const* scores_out,
                       const int64_t* out_stride_h_per_item,
                       void* workspace, size_t workspace_bytes, void* hip_stream);

/* Stage entry points (same kernels kvc_compress enqueues), exposed so parity can be pinned stage by stage. */

/* A1-A5 (A10 for H2O): pooled scores [bsz][n_q_heads][L-W] in dtype (:317-333 / :544-561). */
int kvc_scores(const kvc_params* p, const void* q, const void* k, void* scores_out,
               void* workspace, size_t workspace_bytes, void* hip_stream);

/* A7: idx_out[b][h][0..k) = topk(scores[b][h][0..L-W), k).indices (:334).
 * Workspace: kvc_select_workspace_bytes(p) bytes, 256-byte aligned (0 for most shapes: only tie_mode torch_cpu in its
 * nth_element regime, k*64 > L-W, keeps position lists — and arrays that do not fit in LDS — there). */
size_t kvc_select_workspace_bytes(const kvc_params* p);
int kvc_select(const kvc_params* p, const void* scores, int64_t* idx_out,
               void* workspace, size_t workspace_bytes, void* hip_stream);

/* A8 for one tensor: out[b][h] = cat(src[b][h/g][idx[b][h][:]], src[b][h/g][L-W:]) (:341-346).
 * idx == NULL means idx[t] = t (StreamingLLM, :607-608).  Strides are those of `src`. */
int kvc_gather(const kvc_params* p, const void* src, int64_t stride_b, int64_t stride_h, int64_t stride_l,
               const int64_t* idx, void* out, void* hip_stream);

/* PyramidKV per-layer budget, host arithmetic only (:205-215, beta = 20 at :174).
 * Returns the k for `layer_idx`; -1 means "pass-through" (q_len < cap, :218). */
int64_t kvc_pyramid_k(int64_t max_capacity_prompt, int64_t window, int64_t q_len,
                      int64_t layer_idx, int64_t num_hidden_layers, int64_t beta);

/* ---- decode side: one decode step over the compacted cache -------------------------------------------
 * Replaces, per layer and step, the reference's `past_key_value.update` else-branch (llama_model.py:287-289: repeat_kv of
 * the new K/V to H_q heads + torch.cat onto the [1, H_q, cap + t, D] cache) and the SDPA over the result (:306-313).
 * The cache is kept in two parts: the PREFIX [b][H_q][prefix_rows][D] is what kvc_compress wrote (k + W rows per query
 * head; it may live in a buffer with a larger head stride) and is never touched again; the TAIL [b][H_kv][capacity][D]
 * holds the decoded tokens ONCE per KV head.  The call appends the step's `new_rows` K/V rows to the tail in place
 * (rows tail_rows ..) and writes, for every query head and new token i, softmax(q K^T * scaling) V over the head's prefix
 * rows and its KV head's tail rows [0, tail_rows + i + 1) — fp32 accumulation, output in dtype.  All strides in elements;
 * rows are dense (head_dim elements); head_dim 64 or 128; capacity >= tail_rows + new_rows is the caller's business. */
typedef struct kvc_decode_params {
    int32_t dtype;             /* kvc_dtype of every tensor */
    int32_t bsz, n_q_heads, n_kv_heads, head_dim;
    int32_t prefix_rows;       /* rows per query head in the prefix (k + W) */
    int32_t tail_rows;         /* rows already in the tail before this call */
    int32_t new_rows;          /* tokens of this step (rows of q, k_new, v_new) */
    float scaling;             /* softmax scale, 1 / sqrt(head_dim) for Llama / Mistral */
    int32_t reserved;
    int64_t q_stride_b, q_stride_h, q_stride_l;             /* q[b][h_q][i][:] */
    int64_t new_stride_b, new_stride_h, new_stride_l;       /* k_new / v_new [b][h_kv][i][:] */
    int64_t prefix_stride_b, prefix_stride_h;               /* k_prefix / v_prefix [b][h_q][row][:] */
    int64_t tail_stride_b, tail_stride_h;                   /* k_tail / v_tail [b][h_kv][row][:] */
    int64_t out_stride_b, out_stride_h, out_stride_l;       /* out[b][h_q][i][:] */
} kvc_decode_params;

int kvc_decode_step(const kvc_decode_params* p, const void* q, const void* k_new, const void* v_new,
                    const void* k_prefix, const void* v_prefix, void* k_tail, void* v_tail, void* out, void* hip_stream);

/* kvc_decode_step2 (ABI 5): the same step with
 *   - a workspace, so that the rows of one head are shared by several workgroups and merged afterwards (a decode step over a
 *     2 056-row cache reads 1 MB per head: with one workgroup per head 7/8 of the chip idles — 80 us -> see profiles/r03_decode*);
 *     kvc_decode_workspace_bytes(p, ext) says how much; less (or NULL) only reduces the number of splits;
 *   - optionally (ext != NULL) separate strides for v_prefix and ThinK's channel-pruned leading rows
 *     (llama_model_think.py:160-200, cache_utils_think.py:390-424): the first `pruned_rows` rows of a head have their keys in
 *     k_pruned [b][h_q][pruned_rows][pruned_dim] — only the kept channels, in ascending channel order, zero-padded to a multiple of
 *     16 bytes — and are scored against the query's channels chan[b][h_q][0..pruned_dim) (int16, -1 = padding column); their
 *     values are rows [0, pruned_rows) of v_prefix, the full-channel prefix keys k_prefix[0..prefix_rows) pair with
 *     v_prefix rows [pruned_rows, pruned_rows + prefix_rows). */
typedef struct kvc_decode_ext {
    int64_t v_prefix_stride_b, v_prefix_stride_h;           /* v_prefix [b][h_q][row][:]; 0 = the same as prefix_stride_* */
    int32_t pruned_rows, pruned_dim;
    int64_t pruned_stride_b, pruned_stride_h;               /* k_pruned [b][h_q][row][pruned_dim] */
    int64_t chan_stride_b, chan_stride_h;                   /* chan [b][h_q][pruned_dim] (elements of int16) */
} kvc_decode_ext;
size_t kvc_decode_workspace_bytes(const kvc_decode_params* p, const kvc_decode_ext* ext /* nullable */);
int kvc_decode_step2(const kvc_decode_params* p, const kvc_decode_ext* ext /* nullable */, const void* q, const void* k_new, const void* v_new,
                     const void* k_prefix, const void* v_prefix, void* k_tail, void* v_tail,
                     const void* k_pruned /* nullable */, const int16_t* chan /* nullable */, void* out,
                     void* workspace /* nullable */, size_t workspace_bytes, void* hip_stream);

/* ---- AdaKV / HeadKV: per-head budgets, ragged ("flattened") cache -----------------------------------
 * Replaces AdaKVCluster.update_kv (pyramidkv_utils.py:674-757), HeadKVCluster.update_kv (:813-878) and, at decode, the
 * repo's native update_flatten_view (csrc/csrc/cuda_api.cu:12-85) + flash_attn_varlen_func (llama_model.py:2363-2390).
 * p->method = KVC_ADAKV or KVC_HEADKV, p->k = base capacity (max_capacity_prompt - window), tie order of the per-head sort
 * exactly torch-CPU's (libstdc++ std::sort).  Three calls, all enqueue-only:
 *   kvc_ragged_plan     scores (window mean + pooling, :647-672) into the workspace; AdaKV: caps_out[b][h] = rows head h keeps
 *                       (device int32; flags_out[b][0] = 1 if values equal to the global threshold span several heads and
 *                       had to be shared out in flattened order — the one place where torch's partial_sort tie order is not
 *                       reproduced).  HeadKV: scores only (the caller owns the capacities).
 *   — the host reads the capacities (the reference's own metadata are host integers, :721-741) and lays out the flat cache:
 *     seg_off[b*H + h] = first row of head h, any spare rows behind a head are for in-place decode appends —
 *   kvc_ragged_compact  per head the first caps[h] indices of its descending sort + the W window rows, K and V, into
 *                       k_flat / v_flat at seg_off[h]; idx_out (nullable) receives the kept indices [b*H][idx_stride].
 *   kvc_ragged_decode_step  appends the step's K/V rows behind every head's segment (once per QUERY head, like the
 *                       reference's flattened cache) and attends over the segment: rows [0, seg_len[h] + appended + i + 1).
 * Workspace: kvc_ragged_workspace_bytes(p) for plan + compact (the same buffer, same stream). */
size_t kvc_ragged_workspace_bytes(const kvc_params* p);
int kvc_ragged_plan(const kvc_params* p, const void* q, const void* k, float floor_ratio, int normalize,
                    int32_t* caps_out, int32_t* flags_out, void* scores_out, void* workspace, size_t workspace_bytes, void* hip_stream);
int kvc_ragged_compact(const kvc_params* p, const void* k, const void* v, const int32_t* caps, const int64_t* seg_off, int rows_max,
                       void* k_flat, void* v_flat, int64_t* idx_out, int64_t idx_stride,
                       void* workspace, size_t workspace_bytes, void* hip_stream);
typedef struct kvc_ragged_decode_params {
    int32_t dtype, bsz, n_q_heads, n_kv_heads, head_dim;
    int32_t appended;          /* rows appended to every segment since the prefill (before this call) */
    int32_t new_rows;          /* tokens of this step */
    float scaling;
    int64_t q_stride_b, q_stride_h, q_stride_l;             /* q[b][h_q][i][:] */
    int64_t new_stride_b, new_stride_h, new_stride_l;       /* k_new / v_new [b][h_kv][i][:] */
    int64_t out_stride_b, out_stride_h, out_stride_l;       /* out[b][h_q][i][:] */
    int32_t slack_rows;        /* ABI 5: spare rows behind every segment (0 = not checked): appended + new_rows beyond it is refused */
    int32_t reserved;
} kvc_ragged_decode_params;
int kvc_ragged_decode_step(const kvc_ragged_decode_params* p, const void* q, const void* k_new, const void* v_new,
                           void* k_flat, void* v_flat, const int64_t* seg_off, const int32_t* seg_len, void* out, void* hip_stream);

/* ---- CAMKVCluster.update_kv (pyramidkv_utils.py:431-513), split at its random draw (ABI 5) --------------
 * The reference scores like SnapKV without pooling (:461-476), then walks the tokens s = start_budget .. q_len - window - 1 in
 * a Python loop (:487-502): merge_prob = colmean[s] / max(max(colmean[:start_budget]), max(colmean[s : s + window])) per head
 * (colmean = torch.mean of the window rows' probabilities), NaN -> 0, inf -> 1, clamped to [0, 1];
 * merge_mask = torch.bernoulli(merge_prob); value[s + 1 .. s + window] += value[s] * merge_mask / window, in place, in dtype
 * arithmetic; finally top-k on the window sums and gather of the keys and of the MERGED values.  colmean does not depend on
 * the merged values, so:
 *   kvc_cam_plan   p->method = KVC_SNAPKV, p->pooling = KVC_POOL_NONE: scores_out [bsz*H_q][q_len - window] (the top-k input)
 *                  and prob_out [bsz*H_q][q_len - window - start_budget] (dtype; token s at column s - start_budget);
 *   (caller)       draws merge_mask = bernoulli(prob) — torch's CPU generator stream in the reference; any draw here;
 *   kvc_cam_merge  v [bsz][H_q][q_len][head_dim] (strides p->v_stride_*, one copy per QUERY head: the masks differ per head),
 *                  updated in place exactly as the loop does given merge_mask (uint8, same shape as prob_out);
 *   kvc_select + kvc_gather (keys from k, values from the merged v) finish the job.
 * start_budget = ceil(start_budget_ratio * q_len) (:479). */
size_t kvc_cam_workspace_bytes(const kvc_params* p);
int kvc_cam_plan(const kvc_params* p, const void* q, const void* k, int start_budget, void* scores_out, void* prob_out,
                 void* workspace, size_t workspace_bytes, void* hip_stream);
int kvc_cam_merge(const kvc_params* p, void* v, const uint8_t* merge_mask, int start_budget, void* hip_stream);

/* ---- L2NormCluster.update_kv (pyramidkv_utils.py:394-429) ------------------------------------------
 * The k = max_capacity_prompt keys of every head with the SMALLEST L2 norm, in ascending norm order, and their values:
 *   token_norms = torch.norm(key_states, p=2, dim=-1)  (:419)  -> argsort ascending (:420) -> gather, first k rows (:423-427).
 * p->method = KVC_L2NORM; p->k = rows kept (1 <= k <= q_len); window, kernel_size, pooling, tie_mode, dot_mode and the q
 * strides are ignored (the reference's cluster takes neither a query nor a window).  The caller decides pass-through
 * (q_len < max_capacity_prompt, or layer_idx in skip_layers: :413-416) and does not call.
 * Norm arithmetic = torch-CPU's (8 fp32 accumulators by d % 8, added 0..7, fp32 sqrt, one rounding to the dtype);
 * order of equal norms = torch-CPU's sort (libstdc++ std::sort on (value, index), ascending).
 * k, v: [b][n_kv_heads][L][D] by the k_/v_ strides; k_out, v_out: [b][n_q_heads][k][D] (p->out_stride_h elements between
 * heads, 0 = dense) — query heads of one KV group receive identical rows, as the reference's expanded tensors do.
 * idx_out [b*n_q_heads][k] int64 and norms_out [b][n_q_heads][L] dtype may be NULL.
 * Workspace: kvc_l2norm_workspace_bytes(p), 256-byte aligned. */
size_t kvc_l2norm_workspace_bytes(const kvc_params* p);
int kvc_l2norm_compress(const kvc_params* p, const void* k, const void* v, void* k_out, void* v_out, int64_t* idx_out,
                        void* norms_out, void* workspace, size_t workspace_bytes, void* hip_stream);

/* ---- merge_kv(key_states, value_states, indices, window_size, "pivot") (pyramidkv_utils.py:119-170) ------
 * What SnapKV / PyramidKV / H2O update_kv return instead of the plain gather when their cluster was built with merge="pivot"
 * (:337-339): every token selected by NO head ("dropped", the window included) is averaged into the kept key it is most
 * similar to (cosine, first maximum) and its value into the value row of the same number; kept rows become the mean of
 * themselves and what was merged into them.  idx: [b*n_q_heads][p->k] int64 on the device — the indices of the method's
 * top-k (kvc_select / kvc_compress idx_out).  Outputs [b][n_q_heads][k + window][D]: k_out in the reference's key order
 * [window rows, selected rows], v_out in its value order [selected rows, window rows] (the reference's two cats differ, :145 / :147,
 * and the pivot found on the key order is applied to both).  head_dim must be 128 (hard-coded in the reference, :149);
 * 16-bit dtypes.  bsz > 1: the union of selections is taken per batch row (the reference flattens its isin over the batch
 * too, :131-134, and is only ever run with bsz = 1, README.md:29).  pivot_out ([b*n_q_heads][q_len] int32: row [0, drop_len) = pivot of every dropped token, ascending position)
 * and drop_len_out ([b] int32) may be NULL.  Arithmetic: see csrc/kvc_merge.hip.  Workspace: kvc_merge_workspace_bytes(p). */
size_t kvc_merge_workspace_bytes(const kvc_params* p);
int kvc_merge_pivot(const kvc_params* p, const void* k, const void* v, const int64_t* idx, void* k_out, void* v_out,
                    int32_t* pivot_out, int32_t* drop_len_out, void* workspace, size_t workspace_bytes, void* hip_stream);

/* ---- ThinK channel pruning, key_pruner_query_driven (pyramidkv_utils.py:13-26) ---------------------------
 * What SnapKVCluster.update_think (:349-392) does after its compression: per head the n_prune = int(head_dim * ratio) channels
 * with the smallest mean(q^2 over the last 32 query rows) * mean(k'^2 over the compressed keys) are removed from the compressed
 * keys except the last `recent` rows.  kc: the compressed keys [b*n_heads][rows][head_dim] (rows dense, kc_stride_h elements
 * between heads — kvc_compress's k_out with its out_stride_h); q: the prompt's queries, rows q_len-32 .. q_len-1 are read.
 * Outputs: pruned [b*n_heads][rows - recent][head_dim - n_prune] (kept channels in order), keep [b*n_heads][head_dim] bytes
 * (1 = kept: the reference's returned ~mask), channel_scores [b*n_heads][head_dim] dtype or NULL.  The recent rows stay where
 * they are (kc rows rows-recent ..).  head_dim 64 or 128, q_len >= 32.  Same arithmetic and tie order as torch-CPU
 * (csrc/kvc_think.hip).  No workspace. */
typedef struct kvc_think_params {
    int32_t dtype, bsz, n_heads, q_len, head_dim;
    int32_t rows, recent, n_prune;
    int64_t q_stride_b, q_stride_h, q_stride_l;     /* elements */
    int64_t kc_stride_h;                            /* elements between heads of kc (>= rows * head_dim) */
} kvc_think_params;
int kvc_think_prune(const kvc_think_params* p, const void* q, const void* kc, void* pruned, uint8_t* keep, void* channel_scores,
                    void* hip_stream);

/* Debug/parity aid: byte offsets inside the workspace of the intermediates kvc_scores leaves behind.
 * offs[0]=logits [b][h][L][W] dtype, offs[1]=row max [b][h][W] f32, offs[2]=row sum [b][h][W] f32.
 * Returns KVC_OK or an error. */
int kvc_workspace_layout(const kvc_params* p, size_t offs[3]);

#ifdef __cplusplus
}
#endif
#endif /* KVC_H_ */
