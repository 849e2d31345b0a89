// kvc_api.hip — the C-ABI of include/kvc.h: validation, workspace carving and kernel enqueue.
// Nothing here allocates device memory, synchronises, or touches the host copy of any tensor.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "kvc_common.h"
#include "kvc_launch.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int esize_of(int dtype) { return dtype == KVC_FP32 ? 4 : 2; }
inline bool scoring(int method) { return method == KVC_SNAPKV || method == KVC_PYRAMIDKV || method == KVC_H2O || method == KVC_ADAKV || method == KVC_HEADKV; }
inline bool ragged(int method) { return method == KVC_ADAKV || method == KVC_HEADKV; }

struct Layout {
    size_t logits, pmax, psum, rowmax, rowsum, scores, idx, exact, total;   // H2O: logits = S [L][L], rowmax/rowsum = [L] per head
    int n_tiles, n_chunks;
    int h2o_rows;      // H2O exact mode: query rows of S in flight; 0 in fast mode (no S)
    bool has_part16;   // the psum region holds the window methods' 16-row partial sums
};

// Validation shared by every entry point.  `need_q`: the call reads q/k for scoring.
int validate(const kvc_params* p, bool need_scores) {
    if (!p) return fail(KVC_ERR_INVALID, "params is NULL");
    if (p->method < KVC_SNAPKV || p->method > KVC_HEADKV) return fail(KVC_ERR_INVALID, "unknown method %d", p->method);
    if (p->dtype < KVC_BF16 || p->dtype > KVC_FP32) return fail(KVC_ERR_INVALID, "unknown dtype %d", p->dtype);
    if (p->bsz < 1 || p->n_q_heads < 1 || p->n_kv_heads < 1) return fail(KVC_ERR_INVALID, "bsz / head counts must be positive");
    if (p->n_q_heads % p->n_kv_heads) return fail(KVC_ERR_INVALID, "n_q_heads %d not a multiple of n_kv_heads %d", p->n_q_heads, p->n_kv_heads);
    if (p->window < 1) return fail(KVC_ERR_INVALID, "window must be >= 1");
    if (p->q_len <= p->window) return fail(KVC_ERR_INVALID, "q_len %d must exceed window %d", p->q_len, p->window);
    if (p->k < 0 || p->k > p->q_len - p->window)
        return fail(KVC_ERR_INVALID, "k=%d outside [0, q_len-window=%d] (reference: topk k out of range)", p->k, p->q_len - p->window);
    const int es = esize_of(p->dtype);
    if (p->head_dim < 1 || (p->head_dim * es) % 16) return fail(KVC_ERR_UNSUPPORTED, "head_dim*esize must be a multiple of 16 bytes");
    if (p->head_dim * es > 4096) return fail(KVC_ERR_UNSUPPORTED, "head_dim too large");
    if (p->tie_mode != KVC_TIES_TORCH_CPU && p->tie_mode != KVC_TIES_CANONICAL) return fail(KVC_ERR_INVALID, "unknown tie_mode %d", p->tie_mode);
    if (p->dot_mode != KVC_DOT_EXACT && p->dot_mode != KVC_DOT_MFMA16) return fail(KVC_ERR_INVALID, "unknown dot_mode %d", p->dot_mode);
    if (need_scores && scoring(p->method)) {
        if (p->head_dim != 64 && p->head_dim != 128) return fail(KVC_ERR_UNSUPPORTED, "scoring kernels are built for head_dim 64 and 128, got %d", p->head_dim);
        if (p->window > 64) return fail(KVC_ERR_UNSUPPORTED, "scoring window %d > 64 not built", p->window);
        if (p->method != KVC_H2O) {
            if (p->pooling != KVC_POOL_AVG && p->pooling != KVC_POOL_MAX && p->pooling != KVC_POOL_NONE)
                return fail(KVC_ERR_INVALID, "Pooling method not supported");   // pyramidkv_utils.py:333
            if (p->pooling != KVC_POOL_NONE) {
                if (p->kernel_size < 1 || p->kernel_size > 63) return fail(KVC_ERR_UNSUPPORTED, "kernel_size %d outside [1,63]", p->kernel_size);
                if ((p->kernel_size & 1) == 0)
                    return fail(KVC_ERR_UNSUPPORTED, "even kernel_size %d: the reference's pooled length becomes n+1 and its gather can index out of range", p->kernel_size);
            }
        }
    }
    return KVC_OK;
}

int check_strides(const char* what, int es, int64_t sb, int64_t sh, int64_t sl, const void* ptr) {
    if (((uintptr_t)ptr) % 16) return fail(KVC_ERR_ALIGNMENT, "%s pointer not 16-byte aligned", what);
    if ((sb * es) % 16 || (sh * es) % 16 || (sl * es) % 16) return fail(KVC_ERR_ALIGNMENT, "%s strides not multiples of 16 bytes", what);
    return KVC_OK;
}

Layout carve(const kvc_params* p, int n_items = 1) {
    Layout l;
    std::memset(&l, 0, sizeof(l));
    const size_t es = (size_t)esize_of(p->dtype);
    const size_t heads = (size_t)p->bsz * p->n_q_heads, L = (size_t)p->q_len, W = (size_t)p->window, n = L - W;
    // logits_kernel: 4 waves per workgroup, each wave walks 32-key tiles.  A call owns few KV heads: spread each head
    // over many workgroups (one tile per wave at 8k).  A batch has parallelism to spare: pick the workgroups per head so
    // that the grid is a whole number of "rounds" of the chip (3 workgroups x 256 CUs resident at once) with as many
    // tiles per wave as that allows — fewer prologues (Q image, first-tile latency), no half-empty last round
    // (measured at C2 x 32 layers: 4 tiles per wave 199-204 us, 11: 189 us, 21 = one round: 187 us, 32: 196 us).
    {
        const int tiles32 = (int)((L + 31) / 32);
        const int64_t units = (int64_t)n_items * p->bsz * p->n_kv_heads;        // workgroups per "group" index
        const int64_t slots = 3 * 256;
        int best_g = 1;
        double best = 1e30;
        const int g_max = (tiles32 + 3) / 4;
        for (int g = 1; g <= g_max; ++g) {
            const int64_t rounds = (g * units + slots - 1) / slots;
            const int tpw = (tiles32 + 4 * g - 1) / (4 * g);
            const double cost = (double)rounds * (tpw + 1.5);                   // 1.5 tiles' worth of prologue per workgroup
            if (cost < best - 1e-9) { best = cost; best_g = g; }
        }
        l.n_tiles = best_g;                                // workgroups (4 waves each) per KV head == tile maxima per row
    }
    l.n_chunks = (int)((L + 255) / 256);
    size_t off = 0;
    if (scoring(p->method)) {
        const size_t R = p->method == KVC_H2O ? L : W;     // query rows that score
        const bool h2o_fast = p->method == KVC_H2O && p->dot_mode == KVC_DOT_MFMA16 && p->dtype != KVC_FP32;
        l.h2o_rows = (p->method == KVC_H2O && !h2o_fast) ? kvc::h2o_chunk_rows((int)heads, (int)L, (int)es) : 0;
        if (l.h2o_rows > 512 && (p->debug_stage_mask & 128)) l.h2o_rows = 512;      // testing aid: several chunks at small L
        // logits: [h][L][W] for the window methods; H2O exact: the chunk [h][h2o_rows][L]; H2O fast: nothing
        // (H2O exact, fused form: the same region holds the permuted copy of K, [bsz * Hkv][L][D])
        const size_t h2o_kt = (size_t)p->bsz * p->n_kv_heads * ((L + 15) / 16 * 16) * p->head_dim * es;
        const size_t h2o_s = heads * (size_t)l.h2o_rows * L * es;
        const bool h2o_fused = p->method == KVC_H2O && !h2o_fast && kvc::h2o_fused_eligible(p->dtype, (int)L, (p->debug_stage_mask & 2048) ? 1 : 0);
        l.logits = off; off = align_up(off + (p->method == KVC_H2O ? (h2o_fast ? 0 : (h2o_fused ? h2o_kt : h2o_s)) : heads * L * R * es), 256);
        l.pmax = off;   off = align_up(off + (p->method == KVC_H2O ? 0 : heads * (size_t)l.n_tiles * W * 4), 256);
        // H2O: column sums of every 256-row block (+ one slot for the leftover rows), fp32, columns padded to even
        // window methods at W = 16 / 32 / 64, 16-bit: the 16-row partial window sums of softmax_rows16_kernel, fp32
        l.has_part16 = p->method != KVC_H2O && p->dtype != KVC_FP32 && (W == 16 || W == 32 || W == 64);
        l.psum = off;   off = align_up(off + ((p->method == KVC_H2O && !h2o_fast) ? heads * (size_t)(l.n_chunks + 1) * ((n + 1) & ~(size_t)1) * 4
                                              : (l.has_part16 ? heads * (W / 16) * ((n + 1) & ~(size_t)1) * 4 : 0)), 256);
        l.rowmax = off; off = align_up(off + heads * R * 4, 256);
        l.rowsum = off; off = align_up(off + heads * R * 4, 256);
        l.scores = off; off = align_up(off + heads * n * es, 256);
        l.idx = off;    off = align_up(off + heads * (size_t)p->k * 8, 256);
        l.exact = off;     // (the exact tie mode's scratch follows the per-item regions: exact_scratch_total)
    }
    l.total = off;
    return l;
}

int check_ws(const Layout& l, void* ws, size_t bytes) {
    if (l.total == 0) return KVC_OK;
    if (!ws) return fail(KVC_ERR_WORKSPACE, "workspace is NULL, need %zu bytes", l.total);
    if (((uintptr_t)ws) % 256) return fail(KVC_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    if (bytes < l.total) return fail(KVC_ERR_WORKSPACE, "workspace too small: %zu < %zu", bytes, l.total);
    return KVC_OK;
}

int hip_ok(const char* where) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(KVC_ERR_HIP, "%s: %s", where, hipGetErrorString(e));
    return KVC_OK;
}

// One "item" = one update_kv (one layer).  kvc_compress is the 1-item case of the batched entry point.
struct Items {
    int n;
    const void* q[kvc::KVC_MAX_ITEMS]; const void* k[kvc::KVC_MAX_ITEMS]; const void* v[kvc::KVC_MAX_ITEMS];
    void* k_out[kvc::KVC_MAX_ITEMS]; void* v_out[kvc::KVC_MAX_ITEMS];
    void* scores[kvc::KVC_MAX_ITEMS]; int64_t* idx[kvc::KVC_MAX_ITEMS];
    int keep[kvc::KVC_MAX_ITEMS];
    int64_t out_stride_h[kvc::KVC_MAX_ITEMS];       // elements between output heads, 0 = dense
    int k_max;
};

// scores for every item (pointers in it.scores must be set).  Workspace of item i = ws + i * l.total.
int enqueue_scores(const kvc_params* p, const Layout& l, const Items& it, char* ws, hipStream_t st) {
    if (p->method == KVC_H2O) {
        for (int i = 0; i < it.n; ++i) {
            char* w = ws + (size_t)i * l.total;
            kvc::H2OArgs h;
            h.q = it.q[i]; h.k = it.k[i];
            h.S = w + l.logits;
            h.row0 = 0; h.rows = p->q_len; h.s_rows = l.h2o_rows;
            h.fast = l.h2o_rows == 0 ? 1 : 0;
            h.kt = w + l.logits;
            h.legacy = (p->debug_stage_mask & 2048) ? 1 : 0;
            h.rowmax = reinterpret_cast<float*>(w + l.rowmax);
            h.rinv = reinterpret_cast<float*>(w + l.rowsum);
            h.part = reinterpret_cast<float*>(w + l.psum);
            h.scores = it.scores[i];
            h.q_stride_b = p->q_stride_b; h.q_stride_h = p->q_stride_h; h.q_stride_l = p->q_stride_l;
            h.k_stride_b = p->k_stride_b; h.k_stride_h = p->k_stride_h; h.k_stride_l = p->k_stride_l;
            h.bsz = p->bsz; h.n_q_heads = p->n_q_heads; h.n_kv_heads = p->n_kv_heads; h.group = p->n_q_heads / p->n_kv_heads;
            h.q_len = p->q_len; h.window = p->window;
            h.sqrt_d = (float)std::sqrt((double)p->head_dim);
            const int rc = kvc::launch_h2o_scores(h, p->dtype, p->head_dim, st);
            if (rc) return fail(rc, "no H2O kernel for dtype %d head_dim %d", p->dtype, p->head_dim);
        }
        return hip_ok("h2o scores launch");
    }
    kvc::ScoreArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int i = 0; i < it.n; ++i) { a.q.p[i] = it.q[i]; a.k.p[i] = it.k[i]; a.scores.p[i] = it.scores[i]; }
    a.n_items = it.n;
    a.ws = ws; a.ws_item_stride = (int64_t)l.total;
    a.off_logits = (int64_t)l.logits; a.off_pmax = (int64_t)l.pmax;
    a.off_rowmax = (int64_t)l.rowmax; a.off_rowsum = (int64_t)l.rowsum;
    a.off_part16 = l.has_part16 ? (int64_t)l.psum : -1;
    a.q_stride_b = p->q_stride_b; a.q_stride_h = p->q_stride_h; a.q_stride_l = p->q_stride_l;
    a.k_stride_b = p->k_stride_b; a.k_stride_h = p->k_stride_h; a.k_stride_l = p->k_stride_l;
    a.bsz = p->bsz; a.n_q_heads = p->n_q_heads; a.n_kv_heads = p->n_kv_heads;
    a.group = p->n_q_heads / p->n_kv_heads;
    a.q_len = p->q_len; a.window = p->window;
    a.n_tiles = l.n_tiles;
    a.kernel_size = p->kernel_size; a.pooling = p->pooling;
    a.stage_mask = p->debug_stage_mask;
    a.fast_dot = (p->dot_mode == KVC_DOT_MFMA16 && p->dtype != KVC_FP32) ? 1 : 0;
    a.window_mean = ragged(p->method) ? 1 : 0;              // calcul_attn_sore (:660): .mean(dim=-2)
    a.dbg = nullptr;
#if defined(KVC_STAMPS)
    a.dbg = reinterpret_cast<unsigned long long*>(it.scores[0]);   // diagnostic build: stamps land in scores_out
    if ((p->debug_stage_mask & 7) == 6 && it.idx[0]) a.dbg = reinterpret_cast<unsigned long long*>(it.idx[0]);   // softmax stamps: item 0's idx_out
#endif
    a.sqrt_d = (float)std::sqrt((double)p->head_dim);   // math.sqrt(head_dim) -> fp32 (pyramidkv_utils.py:317)
    const int rc = kvc::launch_scores(a, p->dtype, p->head_dim, st);
    if (rc) return fail(rc, "no scoring kernel for dtype %d head_dim %d", p->dtype, p->head_dim);
    return hip_ok("scores launch");
}

// which: 0 = K (it.k -> it.k_out), 1 = V.  use_idx: gather through it.idx (else identity indices, StreamingLLM).
kvc::GatherArgs gather_args(const kvc_params* p, const Items& it, int which, bool use_idx) {
    kvc::GatherArgs g;
    std::memset(&g, 0, sizeof(g));
    for (int i = 0; i < it.n; ++i) {
        g.src.p[i] = which ? it.v[i] : it.k[i];
        g.out.p[i] = which ? it.v_out[i] : it.k_out[i];
        g.idx.p[i] = use_idx ? it.idx[i] : nullptr;
        g.k.v[i] = it.keep[i];
        g.out_head_bytes.v[i] = it.out_stride_h[i] * esize_of(p->dtype);
    }
    g.n_items = it.n; g.k_max = it.k_max;
    g.stride_b = which ? p->v_stride_b : p->k_stride_b;
    g.stride_h = which ? p->v_stride_h : p->k_stride_h;
    g.stride_l = which ? p->v_stride_l : p->k_stride_l;
    g.bsz = p->bsz; g.n_q_heads = p->n_q_heads; g.group = p->n_q_heads / p->n_kv_heads;
    g.q_len = p->q_len; g.window = p->window;
    g.esize = esize_of(p->dtype);
    g.row_bytes = p->head_dim * g.esize;
    return g;
}

// A7 for every item (+ A8 in the same kernel when `fuse`).  exact_scratch: n_items regions for the exact tie mode.
int enqueue_select(const kvc_params* p, const Items& it, bool fuse, void* exact_scratch, hipStream_t st, bool members_only = false) {
    const int n_sel = p->q_len - p->window;
    if (p->tie_mode == KVC_TIES_CANONICAL) {
        if (it.k_max > 16384) return fail(KVC_ERR_UNSUPPORTED, "tie_mode canonical: k=%d > 16384 (the LDS sort of the selected set) not built", it.k_max);
        if (n_sel > 131072) return fail(KVC_ERR_UNSUPPORTED, "tie_mode canonical: more than 131072 candidates per head not built (torch_cpu has no such limit)");
    } else if ((int64_t)it.k_max * 64 <= (int64_t)n_sel && it.k_max > 18000) {
        return fail(KVC_ERR_UNSUPPORTED, "tie_mode torch_cpu: a partial_sort heap of k=%d does not fit in LDS", it.k_max);
    }
    kvc::SelectArgs s;
    std::memset(&s, 0, sizeof(s));
    for (int i = 0; i < it.n; ++i) { s.scores.p[i] = it.scores[i]; s.idx.p[i] = it.idx[i]; s.k.v[i] = it.keep[i]; }
    s.n_items = it.n;
    s.n = p->q_len - p->window; s.k_max = it.k_max; s.heads = p->bsz * p->n_q_heads;
    s.pow2 = 1;
    while (s.pow2 < s.k_max) s.pow2 <<= 1;
    if (fuse) { s.fuse = 1; s.gk = gather_args(p, it, 0, true); s.gv = gather_args(p, it, 1, true); }
    s.members_only = members_only ? 1 : 0;
    if (p->tie_mode == KVC_TIES_TORCH_CPU) {
#if defined(KVC_STAMPS)
        if (p->debug_stage_mask & 32) s.fuse = 2;            // diag build: phase stamps over the index output (no gather follows)
#endif
        const int rc = kvc::launch_select_exact(s, p->dtype, exact_scratch, st);
        if (rc == KVC_ERR_WORKSPACE) return fail(rc, "tie_mode torch_cpu at n=%d needs the workspace (kvc_workspace_bytes)", s.n);
        if (rc) return fail(rc, "exact select launch failed");
        return hip_ok("exact select launch");
    }
    const int rc = kvc::launch_select(s, p->dtype, p->tie_mode, st);
    if (rc == KVC_ERR_UNSUPPORTED) return fail(rc, "tie_mode %d not built for this shape", p->tie_mode);
    if (rc) return fail(rc, "select launch failed");
    return hip_ok("select launch");
}

int enqueue_gather(const kvc::GatherArgs* g0, const kvc::GatherArgs* g1, hipStream_t st) {
    kvc::GatherPair pr;
    pr.t[0] = *g0;
    pr.count = 1;
    if (g1) { pr.t[1] = *g1; pr.count = 2; } else { pr.t[1] = *g0; }
    kvc::launch_gather(pr, st);
    return hip_ok("gather launch");
}

// Bytes of the (shared, after the per-item regions) scratch the exact tie mode needs for all items.
size_t exact_scratch_total(const kvc_params* p, int n_items, const int* keep) {
    if (p->tie_mode != KVC_TIES_TORCH_CPU || !scoring(p->method)) return 0;
    size_t worst = 0;
    for (int i = 0; i < n_items; ++i) {
        const size_t b = kvc::select_exact_scratch_bytes(p->bsz * p->n_q_heads, p->q_len - p->window, keep[i]);
        if (b > worst) worst = b;
    }
    return align_up(worst * (size_t)n_items, 256);
}

// The whole pipeline for it.n items of identical shape (p) and per-item k.
int run_items(const kvc_params* p, Items& it, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const int es = esize_of(p->dtype);
    if (int rc = check_strides("k", es, p->k_stride_b, p->k_stride_h, p->k_stride_l, it.k[0])) return rc;
    if (int rc = check_strides("v", es, p->v_stride_b, p->v_stride_h, p->v_stride_l, it.v[0])) return rc;
    it.k_max = 0;
    for (int i = 0; i < it.n; ++i) {
        if (!it.k[i] || !it.v[i] || !it.k_out[i] || !it.v_out[i]) return fail(KVC_ERR_INVALID, "k, v, k_out and v_out must be non-NULL (item %d)", i);
        if (((uintptr_t)it.k[i]) % 16 || ((uintptr_t)it.v[i]) % 16 || ((uintptr_t)it.k_out[i]) % 16 || ((uintptr_t)it.v_out[i]) % 16)
            return fail(KVC_ERR_ALIGNMENT, "k / v / k_out / v_out not 16-byte aligned (item %d)", i);
        if (it.keep[i] < 0 || it.keep[i] > p->q_len - p->window) return fail(KVC_ERR_INVALID, "k=%d outside [0, q_len-window] (item %d)", it.keep[i], i);
        if (it.keep[i] > it.k_max) it.k_max = it.keep[i];
    }
    for (int i = 0; i < it.n; ++i) {
        const int64_t os = it.out_stride_h[i];
        if (os == 0) continue;
        if (os < (int64_t)(it.keep[i] + p->window) * p->head_dim)
            return fail(KVC_ERR_INVALID, "out_stride_h %lld smaller than (k + window) * head_dim = %lld (item %d)", (long long)os,
                        (long long)(it.keep[i] + p->window) * p->head_dim, i);
        if ((os * es) % 16) return fail(KVC_ERR_ALIGNMENT, "out_stride_h not a multiple of 16 bytes (item %d)", i);
    }
    if (!scoring(p->method)) {
        for (int i = 0; i < it.n; ++i)
            if (it.idx[i] && it.keep[i] > 0)
                return fail(KVC_ERR_UNSUPPORTED, "StreamingLLM: idx_out must be NULL (indices are arange(k), pyramidkv_utils.py:607)");
        const kvc::GatherArgs gk = gather_args(p, it, 0, false), gv = gather_args(p, it, 1, false);
        return enqueue_gather(&gk, &gv, st);
    }
    kvc_params pk = *p;
    pk.k = it.k_max;
    const Layout l = carve(&pk, it.n);
    const size_t exact = exact_scratch_total(p, it.n, it.keep);
    const size_t need = l.total * (size_t)it.n + exact;
    if (!workspace) return fail(KVC_ERR_WORKSPACE, "workspace is NULL, need %zu bytes", need);
    if (((uintptr_t)workspace) % 256) return fail(KVC_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    if (workspace_bytes < need) return fail(KVC_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, need);
    char* ws = static_cast<char*>(workspace);
    bool any_zero = false;
    for (int i = 0; i < it.n; ++i) {
        if (!it.q[i]) return fail(KVC_ERR_INVALID, "q must be non-NULL for scoring methods (item %d)", i);
        if (((uintptr_t)it.q[i]) % 16) return fail(KVC_ERR_ALIGNMENT, "q pointer not 16-byte aligned (item %d)", i);
        if (it.idx[i] && ((uintptr_t)it.idx[i]) % 16) return fail(KVC_ERR_ALIGNMENT, "idx_out not 16-byte aligned");
        if (it.scores[i] && ((uintptr_t)it.scores[i]) % 16) return fail(KVC_ERR_ALIGNMENT, "scores_out not 16-byte aligned");
        char* w = ws + (size_t)i * l.total;
        if (!it.scores[i]) it.scores[i] = w + l.scores;
        if (!it.idx[i]) it.idx[i] = reinterpret_cast<int64_t*>(w + l.idx);
        any_zero = any_zero || it.keep[i] == 0;
    }
    if (int rc = check_strides("q", es, p->q_stride_b, p->q_stride_h, p->q_stride_l, it.q[0])) return rc;
    if (any_zero && it.n > 1) return fail(KVC_ERR_UNSUPPORTED, "a batch needs k >= 1 for every item");
    const int dbg = p->debug_stage_mask;
    if (!(dbg & 96))                                         // profiling aid bits 5 / 6: scores (and indices) of the previous run
        if (int rc = enqueue_scores(p, l, it, ws, st)) return rc;
    if ((dbg & 7) != 0) return KVC_OK;                       // profiling aid: only the selected scoring kernels
    if (dbg & 64) {                                          // ... only the gather
        const kvc::GatherArgs gk = gather_args(p, it, 0, true), gv = gather_args(p, it, 1, true);
        return enqueue_gather(&gk, &gv, st);
    }
    if (it.k_max == 0) {                                     // nothing to select: only the window tail is kept
        const kvc::GatherArgs gk = gather_args(p, it, 0, false), gv = gather_args(p, it, 1, false);
        return enqueue_gather(&gk, &gv, st);
    }
    const bool fuse = it.k_max <= kvc::kFuseGatherMaxK && p->tie_mode == KVC_TIES_CANONICAL;
    if (int rc = enqueue_select(p, it, fuse, ws + l.total * (size_t)it.n, st)) return rc;
    if (fuse) return KVC_OK;
#if defined(KVC_STAMPS)
    if (dbg & 32) return KVC_OK;                             // the index output holds phase stamps in this build: no gather
#endif
    const kvc::GatherArgs gk = gather_args(p, it, 0, true), gv = gather_args(p, it, 1, true);
    return enqueue_gather(&gk, &gv, st);
}

}  // namespace

extern "C" {

__attribute__((visibility("default"))) int kvc_version(void) { return KVC_VERSION; }

__attribute__((visibility("default"))) const char* kvc_last_error(void) { return g_err; }

__attribute__((visibility("default"))) size_t kvc_workspace_bytes(const kvc_params* p) {
    if (validate(p, true) != KVC_OK) return 0;
    g_err[0] = 0;
    return carve(p).total + exact_scratch_total(p, 1, &p->k);
}

__attribute__((visibility("default"))) size_t kvc_workspace_bytes_batch(const kvc_params* p, int n_items, const int32_t* k_per_item) {
    if (validate(p, true) != KVC_OK) return 0;
    if (n_items < 1 || !k_per_item) { fail(KVC_ERR_INVALID, "n_items must be >= 1 and k_per_item non-NULL"); return 0; }
    // kvc_compress_batch runs chunks of KVC_MAX_ITEMS items one after the other in the same workspace: size the largest
    size_t need = 0;
    for (int base = 0; base < n_items; base += kvc::KVC_MAX_ITEMS) {
        const int n = n_items - base < kvc::KVC_MAX_ITEMS ? n_items - base : kvc::KVC_MAX_ITEMS;
        kvc_params pk = *p;
        pk.k = 0;
        for (int i = 0; i < n; ++i) {
            const int kk = k_per_item[base + i];
            if (kk < 0 || kk > p->q_len - p->window) { fail(KVC_ERR_INVALID, "k=%d out of range (item %d)", kk, base + i); return 0; }
            if (kk > pk.k) pk.k = kk;
        }
        const size_t b = carve(&pk, n).total * (size_t)n + exact_scratch_total(p, n, k_per_item + base);
        if (b > need) need = b;
    }
    g_err[0] = 0;
    return need;
}

__attribute__((visibility("default"))) int kvc_workspace_layout(const kvc_params* p, size_t offs[3]) {
    if (int rc = validate(p, true)) return rc;
    if (!offs) return fail(KVC_ERR_INVALID, "offs is NULL");
    const Layout l = carve(p);
    offs[0] = l.logits; offs[1] = l.rowmax; offs[2] = l.rowsum;
    return KVC_OK;
}

__attribute__((visibility("default"))) int64_t kvc_pyramid_k(int64_t cap, int64_t W, int64_t q_len, int64_t layer_idx,
                                                              int64_t n_layers, int64_t beta) {
    // pyramidkv_utils.py:205-215 (Python floor division restated for possibly negative operands)
    int64_t min_num = (cap - W) / beta;
    if (((cap - W) % beta != 0) && (((cap - W) < 0) != (beta < 0))) --min_num;
    int64_t max_num = (cap - W) * 2 - min_num;
    if (max_num >= q_len - W) { max_num = q_len - W; min_num = (cap - W) * 2 - max_num; }
    const int64_t num = max_num - min_num, den = n_layers - 1;
    int64_t steps = 0;
    if (den != 0) { steps = num / den; if ((num % den != 0) && ((num < 0) != (den < 0))) --steps; }
    if (q_len < cap) return -1;                       // :218 pass-through
    if (q_len < (cap - W) * 2) return cap - W;        // :220 SnapKV budget
    return max_num - layer_idx * steps;               // :252
}

__attribute__((visibility("default"))) int kvc_scores(const kvc_params* p, const void* q, const void* k, void* scores_out,
                                                      void* workspace, size_t workspace_bytes, void* hip_stream) {
    if (int rc = validate(p, true)) return rc;
    if (!scoring(p->method)) return fail(KVC_ERR_INVALID, "StreamingLLM has no scores");
    if (!q || !k || !scores_out) return fail(KVC_ERR_INVALID, "q, k and scores_out must be non-NULL");
    const int es = esize_of(p->dtype);
    if (int rc = check_strides("q", es, p->q_stride_b, p->q_stride_h, p->q_stride_l, q)) return rc;
    if (int rc = check_strides("k", es, p->k_stride_b, p->k_stride_h, p->k_stride_l, k)) return rc;
    if (((uintptr_t)scores_out) % 16) return fail(KVC_ERR_ALIGNMENT, "scores_out not 16-byte aligned");
    const Layout l = carve(p);
    if (int rc = check_ws(l, workspace, workspace_bytes)) return rc;
    Items it;
    std::memset(&it, 0, sizeof(it));
    it.n = 1; it.q[0] = q; it.k[0] = k; it.scores[0] = scores_out;
    return enqueue_scores(p, l, it, static_cast<char*>(workspace), static_cast<hipStream_t>(hip_stream));
}

__attribute__((visibility("default"))) size_t kvc_select_workspace_bytes(const kvc_params* p) {
    if (validate(p, false) != KVC_OK) return 0;
    g_err[0] = 0;
    if (p->tie_mode != KVC_TIES_TORCH_CPU || p->k == 0) return 0;
    return align_up(kvc::select_exact_scratch_bytes(p->bsz * p->n_q_heads, p->q_len - p->window, p->k), 256);
}

__attribute__((visibility("default"))) int kvc_select(const kvc_params* p, const void* scores, int64_t* idx_out,
                                                      void* workspace, size_t workspace_bytes, void* hip_stream) {
    if (int rc = validate(p, false)) return rc;
    if (!scores || !idx_out) return fail(KVC_ERR_INVALID, "scores and idx_out must be non-NULL");
    if (((uintptr_t)scores) % 16 || ((uintptr_t)idx_out) % 16) return fail(KVC_ERR_ALIGNMENT, "scores / idx_out not 16-byte aligned");
    if (p->k == 0) return KVC_OK;
    void* scratch = nullptr;
    if (p->tie_mode == KVC_TIES_TORCH_CPU) {
        const size_t need = kvc::select_exact_scratch_bytes(p->bsz * p->n_q_heads, p->q_len - p->window, p->k);
        if (need) {
            if (!workspace || workspace_bytes < need || ((uintptr_t)workspace) % 256)
                return fail(KVC_ERR_WORKSPACE, "kvc_select with tie_mode torch_cpu needs %zu bytes of 256-byte aligned workspace", need);
            scratch = workspace;
        }
    }
    Items it;
    std::memset(&it, 0, sizeof(it));
    it.n = 1; it.scores[0] = const_cast<void*>(scores); it.idx[0] = idx_out; it.keep[0] = p->k; it.k_max = p->k;
    return enqueue_select(p, it, false, scratch, static_cast<hipStream_t>(hip_stream));
}

__attribute__((visibility("default"))) int kvc_gather(const kvc_params* p, const void* src, int64_t stride_b, int64_t stride_h,
                                                      int64_t stride_l, const int64_t* idx, void* out, void* hip_stream) {
    if (int rc = validate(p, false)) return rc;
    if (!src || !out) return fail(KVC_ERR_INVALID, "src and out must be non-NULL");
    if (int rc = check_strides("src", esize_of(p->dtype), stride_b, stride_h, stride_l, src)) return rc;
    if (((uintptr_t)out) % 16) return fail(KVC_ERR_ALIGNMENT, "out not 16-byte aligned");
    Items it;
    std::memset(&it, 0, sizeof(it));
    it.n = 1; it.k[0] = src; it.k_out[0] = out; it.idx[0] = const_cast<int64_t*>(idx); it.keep[0] = p->k; it.k_max = p->k;
    it.out_stride_h[0] = p->out_stride_h;
    kvc_params ps = *p;
    ps.k_stride_b = stride_b; ps.k_stride_h = stride_h; ps.k_stride_l = stride_l;
    const kvc::GatherArgs g = gather_args(&ps, it, 0, idx != nullptr);
    return enqueue_gather(&g, nullptr, static_cast<hipStream_t>(hip_stream));
}

namespace {
int decode_check(const kvc_decode_params* p, const kvc_decode_ext* x) {
    if (!p) return fail(KVC_ERR_INVALID, "params is NULL");
    if (p->dtype < KVC_BF16 || p->dtype > KVC_FP32) return fail(KVC_ERR_INVALID, "unknown dtype %d", p->dtype);
    if (p->bsz < 1 || p->n_q_heads < 1 || p->n_kv_heads < 1 || p->n_q_heads % p->n_kv_heads)
        return fail(KVC_ERR_INVALID, "bsz / head counts invalid");
    if (p->head_dim != 64 && p->head_dim != 128) return fail(KVC_ERR_UNSUPPORTED, "decode kernels are built for head_dim 64 and 128, got %d", p->head_dim);
    if (p->prefix_rows < 0 || p->tail_rows < 0 || p->new_rows < 1) return fail(KVC_ERR_INVALID, "row counts invalid");
    if (x) {
        if (x->pruned_rows < 0) return fail(KVC_ERR_INVALID, "pruned_rows < 0");
        if (x->pruned_rows > 0 && (x->pruned_dim < 1 || x->pruned_dim > p->head_dim || (x->pruned_dim * esize_of(p->dtype)) % 16))
            return fail(KVC_ERR_INVALID, "pruned_dim must be 1..head_dim and a multiple of 16 bytes (pad the kept channels with zero columns), got %d", x->pruned_dim);
    }
    return KVC_OK;
}
int decode_run(const kvc_decode_params* p, const kvc_decode_ext* x, const void* q, const void* k_new, const void* v_new,
               const void* k_prefix, const void* v_prefix, void* k_tail, void* v_tail, const void* k_pruned, const int16_t* chan,
               void* out, void* workspace, size_t workspace_bytes, void* hip_stream) {
    if (int rc = decode_check(p, x)) return rc;
    const int pruned = x ? x->pruned_rows : 0;
    if (!q || !k_new || !v_new || !k_tail || !v_tail || !out || (p->prefix_rows > 0 && !k_prefix) || (p->prefix_rows + pruned > 0 && !v_prefix))
        return fail(KVC_ERR_INVALID, "q, k_new, v_new, the tail, out (and the prefix when it has rows) must be non-NULL");
    if (pruned > 0 && (!k_pruned || !chan)) return fail(KVC_ERR_INVALID, "pruned_rows > 0 needs k_pruned and chan");
    const int es = esize_of(p->dtype);
    const void* ptrs[] = {q, k_new, v_new, k_prefix, v_prefix, k_tail, v_tail, pruned > 0 ? k_pruned : nullptr};
    for (const void* ptr : ptrs) if (((uintptr_t)ptr) % 16) return fail(KVC_ERR_ALIGNMENT, "a pointer is not 16-byte aligned");
    const int64_t strides[] = {p->q_stride_b, p->q_stride_h, p->q_stride_l, p->new_stride_b, p->new_stride_h, p->new_stride_l,
                               p->prefix_stride_b, p->prefix_stride_h, p->tail_stride_b, p->tail_stride_h,
                               x ? x->v_prefix_stride_b : 0, x ? x->v_prefix_stride_h : 0, pruned > 0 ? x->pruned_stride_b : 0, pruned > 0 ? x->pruned_stride_h : 0};
    for (int64_t sd : strides) if ((sd * es) % 16) return fail(KVC_ERR_ALIGNMENT, "a stride is not a multiple of 16 bytes");
    kvc::DecodeArgs a;
    std::memset(&a, 0, sizeof(a));
    a.q = q; a.k_new = k_new; a.v_new = v_new; a.k_prefix = k_prefix; a.v_prefix = v_prefix; a.k_tail = k_tail; a.v_tail = v_tail; a.out = out;
    a.q_stride_b = p->q_stride_b; a.q_stride_h = p->q_stride_h; a.q_stride_l = p->q_stride_l;
    a.new_stride_b = p->new_stride_b; a.new_stride_h = p->new_stride_h; a.new_stride_l = p->new_stride_l;
    a.prefix_stride_b = p->prefix_stride_b; a.prefix_stride_h = p->prefix_stride_h;
    a.vprefix_stride_b = (x && x->v_prefix_stride_h) ? x->v_prefix_stride_b : p->prefix_stride_b;
    a.vprefix_stride_h = (x && x->v_prefix_stride_h) ? x->v_prefix_stride_h : p->prefix_stride_h;
    a.tail_stride_b = p->tail_stride_b; a.tail_stride_h = p->tail_stride_h;
    a.out_stride_b = p->out_stride_b; a.out_stride_h = p->out_stride_h; a.out_stride_l = p->out_stride_l;
    a.bsz = p->bsz; a.n_q_heads = p->n_q_heads; a.n_kv_heads = p->n_kv_heads; a.group = p->n_q_heads / p->n_kv_heads;
    a.prefix_rows = p->prefix_rows; a.tail_rows = p->tail_rows; a.new_rows = p->new_rows;
    a.esize = es; a.row_bytes = p->head_dim * es; a.scaling = p->scaling;
    if (pruned > 0) {
        a.k_pruned = k_pruned; a.chan = chan; a.pruned_rows = pruned; a.pruned_dim = x->pruned_dim;
        a.pruned_stride_b = x->pruned_stride_b; a.pruned_stride_h = x->pruned_stride_h;
        a.chan_stride_b = x->chan_stride_b; a.chan_stride_h = x->chan_stride_h;
    }
    a.splits = 1;
    if (workspace) {                                                  // as many splits as the workspace handed over can hold
        const int rows = pruned + p->prefix_rows + p->tail_rows + p->new_rows;
        int s = kvc::decode_splits(p->bsz, p->n_q_heads, p->new_rows, rows);
        while (s > 1 && kvc::decode_part_bytes(p->bsz, p->n_q_heads, p->new_rows, p->head_dim, s) > workspace_bytes) --s;
        if (((uintptr_t)workspace) % 16) return fail(KVC_ERR_ALIGNMENT, "workspace is not 16-byte aligned");
        a.splits = s; a.part = static_cast<float*>(workspace);
    }
    if (int rc = kvc::launch_decode_step(a, p->dtype, p->head_dim, static_cast<hipStream_t>(hip_stream)))
        return fail(rc, "no decode kernel for dtype %d head_dim %d", p->dtype, p->head_dim);
    return hip_ok("decode step launch");
}
}  // namespace

__attribute__((visibility("default"))) int kvc_decode_step(const kvc_decode_params* p, const void* q, const void* k_new, const void* v_new,
                                                           const void* k_prefix, const void* v_prefix, void* k_tail, void* v_tail,
                                                           void* out, void* hip_stream) {
    return decode_run(p, nullptr, q, k_new, v_new, k_prefix, v_prefix, k_tail, v_tail, nullptr, nullptr, out, nullptr, 0, hip_stream);
}

__attribute__((visibility("default"))) size_t kvc_decode_workspace_bytes(const kvc_decode_params* p, const kvc_decode_ext* ext) {
    if (decode_check(p, ext)) return 0;
    const int rows = (ext ? ext->pruned_rows : 0) + p->prefix_rows + p->tail_rows + p->new_rows;
    return kvc::decode_part_bytes(p->bsz, p->n_q_heads, p->new_rows, p->head_dim, kvc::decode_splits(p->bsz, p->n_q_heads, p->new_rows, rows));
}

__attribute__((visibility("default"))) int kvc_decode_step2(const kvc_decode_params* p, const kvc_decode_ext* ext, const void* q, const void* k_new,
                                                            const void* v_new, const void* k_prefix, const void* v_prefix, void* k_tail, void* v_tail,
                                                            const void* k_pruned, const int16_t* chan, void* out, void* workspace,
                                                            size_t workspace_bytes, void* hip_stream) {
    return decode_run(p, ext, q, k_new, v_new, k_prefix, v_prefix, k_tail, v_tail, k_pruned, chan, out, workspace, workspace_bytes, hip_stream);
}

// ---- AdaKV / HeadKV -------------------------------------------------------------------------------------------------------
namespace {
struct RaggedLayout { size_t ratio, idx, sort, flat, top, sel, total; };
RaggedLayout ragged_carve(const kvc_params* p, const Layout& l) {
    RaggedLayout r;
    const size_t heads = (size_t)p->bsz * p->n_q_heads, n = (size_t)(p->q_len - p->window);
    size_t off = l.total;
    r.ratio = off; off = align_up(off + heads * 4, 256);
    r.idx = off;   off = align_up(off + heads * n * 8, 256);
    r.sort = off;  off = align_up(off + kvc::sort_prefix_scratch_bytes((int)heads, (int)n), 256);
    r.flat = r.top = r.sel = off;
    if (p->method == KVC_ADAKV && p->tie_mode == KVC_TIES_TORCH_CPU) {        // the reference's own flattened top-(H*base)
        const size_t n_tot = (size_t)p->n_q_heads * n, k_tot = (size_t)p->n_q_heads * (size_t)p->k;
        r.flat = off; off = align_up(off + (size_t)p->bsz * n_tot * esize_of(p->dtype), 256);
        r.top = off;  off = align_up(off + (size_t)p->bsz * k_tot * 8, 256);
        r.sel = off;  off = align_up(off + kvc::select_exact_scratch_bytes(p->bsz, (int)n_tot, (int)k_tot), 256);
    }
    r.total = off;
    return r;
}
int ragged_check(const kvc_params* p) {
    if (int rc = validate(p, true)) return rc;
    if (!ragged(p->method)) return fail(KVC_ERR_INVALID, "kvc_ragged_*: method must be KVC_ADAKV or KVC_HEADKV");
    if (p->n_q_heads > 256) return fail(KVC_ERR_UNSUPPORTED, "more than 256 heads");
    return KVC_OK;
}
}  // namespace

__attribute__((visibility("default"))) size_t kvc_ragged_workspace_bytes(const kvc_params* p) {
    if (ragged_check(p) != KVC_OK) return 0;
    g_err[0] = 0;
    kvc_params pk = *p;
    pk.k = 0;
    return ragged_carve(p, carve(&pk)).total;
}

__attribute__((visibility("default"))) int kvc_ragged_plan(const kvc_params* p, const void* q, const void* k, float floor_ratio,
                                                           int normalize, int32_t* caps_out, int32_t* flags_out, void* scores_out,
                                                           void* workspace, size_t workspace_bytes, void* hip_stream) {
    if (int rc = ragged_check(p)) return rc;
    if (!q || !k) return fail(KVC_ERR_INVALID, "q and k must be non-NULL");
    const int es = esize_of(p->dtype);
    if (int rc = check_strides("q", es, p->q_stride_b, p->q_stride_h, p->q_stride_l, q)) return rc;
    if (int rc = check_strides("k", es, p->k_stride_b, p->k_stride_h, p->k_stride_l, k)) return rc;
    kvc_params pk = *p;
    pk.k = 0;
    const Layout l = carve(&pk);
    const RaggedLayout r = ragged_carve(p, l);
    if (!workspace || ((uintptr_t)workspace) % 256 || workspace_bytes < r.total)
        return fail(KVC_ERR_WORKSPACE, "kvc_ragged_plan needs %zu bytes of 256-byte aligned workspace", r.total);
    char* ws = static_cast<char*>(workspace);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    Items it;
    std::memset(&it, 0, sizeof(it));
    it.n = 1; it.q[0] = q; it.k[0] = k; it.scores[0] = ws + l.scores;
    if (int rc = enqueue_scores(&pk, l, it, ws, st)) return rc;
    const size_t heads = (size_t)p->bsz * p->n_q_heads, n = (size_t)(p->q_len - p->window);
    if (scores_out) {
        if (hipMemcpyAsync(scores_out, ws + l.scores, heads * n * es, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return fail(KVC_ERR_HIP, "copy of the scores failed");
    }
    if (p->method == KVC_ADAKV) {
        if (!caps_out) return fail(KVC_ERR_INVALID, "caps_out must be non-NULL for AdaKV");
        if (p->k < 1 || (size_t)p->k > n) return fail(KVC_ERR_INVALID, "base capacity k=%d outside [1, q_len - window]", p->k);
        if (!(floor_ratio >= 0.0f && floor_ratio <= 1.0f)) return fail(KVC_ERR_INVALID, "floor_ratio outside [0, 1]");
        kvc::RaggedArgs a;
        std::memset(&a, 0, sizeof(a));
        a.scores = ws + l.scores;
        a.ratio = reinterpret_cast<float*>(ws + r.ratio);
        a.caps = caps_out;
        a.flags = flags_out ? flags_out : reinterpret_cast<int32_t*>(ws + r.sort);     // (scratch slot when not wanted)
        a.n = (int)n; a.heads_per_batch = p->n_q_heads; a.base = p->k; a.normalize = normalize ? 1 : 0;
        a.floor_capacity = (int)((double)p->k * (double)floor_ratio);                 // int(base_capacity * floor_ratio), :632
        a.mix = (float)(1.0 - (double)floor_ratio);
        if (p->tie_mode == KVC_TIES_TORCH_CPU) {
            // exactly the reference: every head's whole descending sort, the normalised values in that order, torch-CPU's
            // top-(H*base) over the flattened array (libstdc++ partial_sort / nth_element: one wave walks it), counts per head
            const int64_t n_tot = (int64_t)p->n_q_heads * (int64_t)n, k_tot = (int64_t)p->n_q_heads * p->k;
            if (n_tot > 0x7fffffff) return fail(KVC_ERR_UNSUPPORTED, "flattened score array too long");
            kvc::RaggedSortArgs sa;
            sa.scores = ws + l.scores; sa.caps = nullptr; sa.idx_out = reinterpret_cast<int64_t*>(ws + r.idx);
            sa.n = (int)n; sa.heads = (int)heads; sa.out_stride = (int)n;
            if (int rc = kvc::launch_sort_prefix(sa, p->dtype, ws + r.sort, st)) return fail(rc, "sort launch failed");
            if (a.normalize) kvc::launch_ragged_head_stats(a, p->dtype, p->bsz, st);
            kvc::launch_ragged_flatten(a, p->dtype, p->bsz, sa.idx_out, ws + r.flat, st);
            kvc_params ps = *p;
            ps.bsz = p->bsz; ps.n_q_heads = 1; ps.n_kv_heads = 1; ps.q_len = (int)n_tot + p->window; ps.k = (int)k_tot;
            Items sel;
            std::memset(&sel, 0, sizeof(sel));
            sel.n = 1; sel.scores[0] = ws + r.flat; sel.idx[0] = reinterpret_cast<int64_t*>(ws + r.top); sel.keep[0] = (int)k_tot; sel.k_max = (int)k_tot;
            // (only WHICH elements torch's top-(H*base) holds matters here, not their order: the final __sort_heap — a fifth of the
            // walk — is skipped)
            if (int rc = enqueue_select(&ps, sel, false, ws + r.sel, st, true)) return rc;
            kvc::launch_ragged_recount(a, p->bsz, reinterpret_cast<const int64_t*>(ws + r.top), st);
        } else if (int rc = kvc::launch_ragged_plan(a, p->dtype, p->bsz, st)) {
            return fail(rc, "ragged plan launch failed");
        }
    }
    return hip_ok("ragged plan launch");
}

__attribute__((visibility("default"))) int kvc_ragged_compact(const kvc_params* p, const void* k, const void* v, const int32_t* caps,
                                                              const int64_t* seg_off, int rows_max, void* k_flat, void* v_flat,
                                                              int64_t* idx_out, int64_t idx_stride, void* workspace,
                                                              size_t workspace_bytes, void* hip_stream) {
    if (int rc = ragged_check(p)) return rc;
    if (!k || !v || !caps || !seg_off || !k_flat || !v_flat) return fail(KVC_ERR_INVALID, "k, v, caps, seg_off and the flat buffers must be non-NULL");
    const int es = esize_of(p->dtype);
    if (int rc = check_strides("k", es, p->k_stride_b, p->k_stride_h, p->k_stride_l, k)) return rc;
    if (int rc = check_strides("v", es, p->v_stride_b, p->v_stride_h, p->v_stride_l, v)) return rc;
    if (((uintptr_t)k_flat) % 16 || ((uintptr_t)v_flat) % 16) return fail(KVC_ERR_ALIGNMENT, "flat buffers not 16-byte aligned");
    kvc_params pk = *p;
    pk.k = 0;
    const Layout l = carve(&pk);
    const RaggedLayout r = ragged_carve(p, l);
    if (!workspace || ((uintptr_t)workspace) % 256 || workspace_bytes < r.total)
        return fail(KVC_ERR_WORKSPACE, "kvc_ragged_compact needs %zu bytes of 256-byte aligned workspace", r.total);
    const int heads = p->bsz * p->n_q_heads, n = p->q_len - p->window;
    if (rows_max < p->window || rows_max > n + p->window) return fail(KVC_ERR_INVALID, "rows_max %d outside [window, q_len]", rows_max);
    if (idx_out && idx_stride < rows_max - p->window) return fail(KVC_ERR_INVALID, "idx_stride smaller than the largest capacity");
    char* ws = static_cast<char*>(workspace);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    kvc::RaggedSortArgs sa;
    sa.scores = ws + l.scores; sa.caps = caps;
    sa.idx_out = idx_out ? idx_out : reinterpret_cast<int64_t*>(ws + r.idx);
    sa.n = n; sa.heads = heads; sa.out_stride = idx_out ? (int)idx_stride : n;
    // (AdaKV with torch_cpu ties: kvc_ragged_plan left every head's whole sort in the workspace — its prefix is the answer)
    const bool presorted = p->method == KVC_ADAKV && p->tie_mode == KVC_TIES_TORCH_CPU && !idx_out;
    if (!presorted)
        if (int rc = kvc::launch_sort_prefix(sa, p->dtype, ws + r.sort, st)) return fail(rc, "sort prefix launch failed");
    kvc::RaggedGatherArgs g;
    std::memset(&g, 0, sizeof(g));
    g.k = k; g.v = v; g.k_flat = k_flat; g.v_flat = v_flat;
    g.idx = sa.idx_out; g.idx_stride = sa.out_stride; g.caps = caps; g.seg_off = seg_off;
    g.k_stride_b = p->k_stride_b; g.k_stride_h = p->k_stride_h; g.k_stride_l = p->k_stride_l;
    g.v_stride_b = p->v_stride_b; g.v_stride_h = p->v_stride_h; g.v_stride_l = p->v_stride_l;
    g.n_q_heads = p->n_q_heads; g.group = p->n_q_heads / p->n_kv_heads; g.q_len = p->q_len; g.window = p->window;
    g.esize = es; g.row_bytes = p->head_dim * es;
    kvc::launch_ragged_gather(g, rows_max, heads, st);
    return hip_ok("ragged compact launch");
}

// ---- L2NormCluster.update_kv (pyramidkv_utils.py:394-429) ---------------------------------------------
namespace {
struct L2Layout { size_t norms, idx, sort, total; };
L2Layout l2norm_carve(const kvc_params* p) {
    L2Layout l;
    const size_t heads = (size_t)p->bsz * p->n_q_heads, L = (size_t)p->q_len;
    size_t off = 0;
    l.norms = off; off = align_up(off + heads * L * esize_of(p->dtype), 256);
    l.idx = off;   off = align_up(off + heads * (size_t)p->k * 8, 256);
    l.sort = off;  off = align_up(off + kvc::sort_prefix_scratch_bytes((int)heads, (int)L), 256);
    l.total = off;
    return l;
}
int l2norm_check(const kvc_params* p) {
    if (!p) return fail(KVC_ERR_INVALID, "params is NULL");
    if (p->method != KVC_L2NORM) return fail(KVC_ERR_INVALID, "kvc_l2norm_*: method must be KVC_L2NORM");
    if (p->dtype < KVC_BF16 || p->dtype > KVC_FP32) return fail(KVC_ERR_INVALID, "unknown dtype %d", p->dtype);
    if (p->bsz < 1 || p->n_q_heads < 1 || p->n_kv_heads < 1) return fail(KVC_ERR_INVALID, "bsz / head counts must be positive");
    if (p->n_q_heads % p->n_kv_heads) return fail(KVC_ERR_INVALID, "n_q_heads %d not a multiple of n_kv_heads %d", p->n_q_heads, p->n_kv_heads);
    if (p->q_len < 1) return fail(KVC_ERR_INVALID, "q_len must be positive");
    if (p->k < 1 || p->k > p->q_len) return fail(KVC_ERR_INVALID, "k=%d outside [1, q_len=%d] (rows kept = max_capacity_prompt)", p->k, p->q_len);
    const int es = esize_of(p->dtype);
    if (p->head_dim < 1 || (p->head_dim * es) % 32) return fail(KVC_ERR_UNSUPPORTED, "head_dim*esize must be a multiple of 32 bytes");
    if (p->head_dim * es > 4096) return fail(KVC_ERR_UNSUPPORTED, "head_dim too large");
    return KVC_OK;
}
}  // namespace

__attribute__((visibility("default"))) size_t kvc_l2norm_workspace_bytes(const kvc_params* p) {
    if (l2norm_check(p) != KVC_OK) return 0;
    g_err[0] = 0;
    return l2norm_carve(p).total;
}

__attribute__((visibility("default"))) int kvc_l2norm_compress(const kvc_params* p, const void* k, const void* v, void* k_out, void* v_out,
                                                               int64_t* idx_out, void* norms_out, void* workspace, size_t workspace_bytes,
                                                               void* hip_stream) {
    if (int rc = l2norm_check(p)) return rc;
    if (!k || !v || !k_out || !v_out) return fail(KVC_ERR_INVALID, "k, v, k_out and v_out must be non-NULL");
    const int es = esize_of(p->dtype);
    if (int rc = check_strides("k", es, p->k_stride_b, p->k_stride_h, p->k_stride_l, k)) return rc;
    if (int rc = check_strides("v", es, p->v_stride_b, p->v_stride_h, p->v_stride_l, v)) return rc;
    if (((uintptr_t)k_out) % 16 || ((uintptr_t)v_out) % 16) return fail(KVC_ERR_ALIGNMENT, "outputs not 16-byte aligned");
    if (p->out_stride_h && (p->out_stride_h < (int64_t)p->k * p->head_dim || (p->out_stride_h * es) % 16))
        return fail(KVC_ERR_INVALID, "out_stride_h smaller than k * head_dim, or not a multiple of 16 bytes");
    const L2Layout l = l2norm_carve(p);
    if (!workspace || ((uintptr_t)workspace) % 256 || workspace_bytes < l.total)
        return fail(KVC_ERR_WORKSPACE, "kvc_l2norm_compress needs %zu bytes of 256-byte aligned workspace", l.total);
    char* ws = static_cast<char*>(workspace);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const int heads = p->bsz * p->n_q_heads;
    kvc::L2NormArgs na;
    na.k = k; na.norms = norms_out ? norms_out : ws + l.norms;
    na.stride_b = p->k_stride_b; na.stride_h = p->k_stride_h; na.stride_l = p->k_stride_l;
    na.bsz = p->bsz; na.n_q_heads = p->n_q_heads; na.n_kv_heads = p->n_kv_heads; na.q_len = p->q_len; na.head_dim = p->head_dim;
    if (int rc = kvc::launch_l2norm(na, p->dtype, st)) return fail(rc, "l2norm launch failed");
    kvc::RaggedSortArgs sa;
    sa.scores = na.norms; sa.caps = nullptr;
    sa.idx_out = idx_out ? idx_out : reinterpret_cast<int64_t*>(ws + l.idx);
    sa.n = p->q_len; sa.heads = heads; sa.out_stride = p->k; sa.want_fixed = p->k; sa.ascending = 1;
    if (int rc = kvc::launch_sort_prefix(sa, p->dtype, ws + l.sort, st)) return fail(rc, "sort prefix launch failed");
    kvc::GatherArgs g[2];
    for (int which = 0; which < 2; ++which) {
        std::memset(&g[which], 0, sizeof(g[which]));
        g[which].src.p[0] = which ? v : k;
        g[which].out.p[0] = which ? v_out : k_out;
        g[which].idx.p[0] = sa.idx_out;
        g[which].k.v[0] = p->k;
        g[which].out_head_bytes.v[0] = p->out_stride_h * es;
        g[which].n_items = 1; g[which].k_max = p->k;
        g[which].stride_b = which ? p->v_stride_b : p->k_stride_b;
        g[which].stride_h = which ? p->v_stride_h : p->k_stride_h;
        g[which].stride_l = which ? p->v_stride_l : p->k_stride_l;
        g[which].bsz = p->bsz; g[which].n_q_heads = p->n_q_heads; g[which].group = p->n_q_heads / p->n_kv_heads;
        g[which].q_len = p->q_len; g[which].window = 0;
        g[which].esize = es; g[which].row_bytes = p->head_dim * es;
    }
    return enqueue_gather(&g[0], &g[1], st);
}

// ---- merge_kv(..., "pivot") (pyramidkv_utils.py:119-170) ----------------------------------------------------
namespace {
struct MergeLayout { size_t mask, drop, drop_len, kn, pivot, total; };
MergeLayout merge_carve(const kvc_params* p) {
    MergeLayout l;
    const size_t heads = (size_t)p->bsz * p->n_q_heads, L = (size_t)p->q_len, R = (size_t)p->k + p->window;
    size_t off = 0;
    l.mask = off;     off = align_up(off + (size_t)p->bsz * L * 4, 256);
    l.drop = off;     off = align_up(off + (size_t)p->bsz * L * 4, 256);
    l.drop_len = off; off = align_up(off + (size_t)p->bsz * 4, 256);
    l.kn = off;       off = align_up(off + heads * R * 128 * 4, 256);
    l.pivot = off;    off = align_up(off + heads * L * 4, 256);
    l.total = off;
    return l;
}
int merge_check(const kvc_params* p) {
    if (int rc = validate(p, false)) return rc;
    if (p->method != KVC_SNAPKV && p->method != KVC_PYRAMIDKV && p->method != KVC_H2O && p->method != KVC_STREAMINGLLM)
        return fail(KVC_ERR_INVALID, "kvc_merge_pivot: merge_kv is reached from SnapKV / PyramidKV / H2O / StreamingLLM update_kv only");
    if (p->head_dim != 128) return fail(KVC_ERR_UNSUPPORTED, "merge_kv hard-codes head_dim 128 (pyramidkv_utils.py:149), got %d", p->head_dim);
    if (p->dtype == KVC_FP32) return fail(KVC_ERR_UNSUPPORTED, "merge_kv: fp32 not built (torch's fp32 GEMM order is not restated)");
    if (p->k < 1) return fail(KVC_ERR_INVALID, "merge_kv needs k >= 1");
    return KVC_OK;
}
}  // namespace

__attribute__((visibility("default"))) size_t kvc_merge_workspace_bytes(const kvc_params* p) {
    if (merge_check(p) != KVC_OK) return 0;
    g_err[0] = 0;
    return merge_carve(p).total;
}

__attribute__((visibility("default"))) int kvc_merge_pivot(const kvc_params* p, const void* k, const void* v, const int64_t* idx,
                                                           void* k_out, void* v_out, int32_t* pivot_out, int32_t* drop_len_out,
                                                           void* workspace, size_t workspace_bytes, void* hip_stream) {
    if (int rc = merge_check(p)) return rc;
    if (!k || !v || !idx || !k_out || !v_out) return fail(KVC_ERR_INVALID, "k, v, idx, k_out and v_out must be non-NULL");
    const int es = esize_of(p->dtype);
    if (int rc = check_strides("k", es, p->k_stride_b, p->k_stride_h, p->k_stride_l, k)) return rc;
    if (int rc = check_strides("v", es, p->v_stride_b, p->v_stride_h, p->v_stride_l, v)) return rc;
    if (((uintptr_t)k_out) % 16 || ((uintptr_t)v_out) % 16) return fail(KVC_ERR_ALIGNMENT, "outputs not 16-byte aligned");
    const int64_t rows = (int64_t)p->k + p->window;
    if (p->out_stride_h && (p->out_stride_h < rows * p->head_dim || (p->out_stride_h * es) % 16))
        return fail(KVC_ERR_INVALID, "out_stride_h smaller than (k + window) * head_dim, or not a multiple of 16 bytes");
    const MergeLayout l = merge_carve(p);
    if (!workspace || ((uintptr_t)workspace) % 256 || workspace_bytes < l.total)
        return fail(KVC_ERR_WORKSPACE, "kvc_merge_pivot needs %zu bytes of 256-byte aligned workspace", l.total);
    char* ws = static_cast<char*>(workspace);
    kvc::MergeArgs a;
    std::memset(&a, 0, sizeof(a));
    a.k = k; a.v = v; a.idx = idx; a.k_out = k_out; a.v_out = v_out;
    a.mask = reinterpret_cast<int32_t*>(ws + l.mask);
    a.drop = reinterpret_cast<int32_t*>(ws + l.drop);
    a.drop_len = drop_len_out ? drop_len_out : reinterpret_cast<int32_t*>(ws + l.drop_len);
    a.kn = reinterpret_cast<float*>(ws + l.kn);
    a.pivot = pivot_out ? pivot_out : reinterpret_cast<int32_t*>(ws + l.pivot);
    a.k_stride_b = p->k_stride_b; a.k_stride_h = p->k_stride_h; a.k_stride_l = p->k_stride_l;
    a.v_stride_b = p->v_stride_b; a.v_stride_h = p->v_stride_h; a.v_stride_l = p->v_stride_l;
    a.out_stride_h = p->out_stride_h;
    a.bsz = p->bsz; a.n_q_heads = p->n_q_heads; a.group = p->n_q_heads / p->n_kv_heads; a.q_len = p->q_len;
    a.window = p->window; a.n_keep = p->k; a.rows = (int)rows;
    if (int rc = kvc::launch_merge(a, p->dtype, static_cast<hipStream_t>(hip_stream))) return fail(rc, "merge launch failed");
    return hip_ok("merge launch");
}

// ---- ThinK channel pruning (pyramidkv_utils.py:13-26) -------------------------------------------------------
__attribute__((visibility("default"))) int kvc_think_prune(const kvc_think_params* p, const void* q, const void* kc, void* pruned,
                                                           uint8_t* keep, void* channel_scores, void* hip_stream) {
    if (!p || !q || !kc || !keep) return fail(KVC_ERR_INVALID, "params, q, kc and keep must be non-NULL");
    if (!pruned && p->rows > p->recent) return fail(KVC_ERR_INVALID, "pruned must be non-NULL");
    if (p->dtype < KVC_BF16 || p->dtype > KVC_FP32) return fail(KVC_ERR_INVALID, "unknown dtype %d", p->dtype);
    if (p->bsz < 1 || p->n_heads < 1) return fail(KVC_ERR_INVALID, "bsz / n_heads must be positive");
    if (p->head_dim != 64 && p->head_dim != 128) return fail(KVC_ERR_UNSUPPORTED, "ThinK pruning is built for head_dim 64 and 128, got %d", p->head_dim);
    if (p->q_len < 32) return fail(KVC_ERR_INVALID, "q_len %d < 32 (the reference reads the last 32 query rows, :17)", p->q_len);
    if (p->rows < 1 || p->recent < 0 || p->recent > p->rows) return fail(KVC_ERR_INVALID, "recent=%d outside [0, rows=%d]", p->recent, p->rows);
    if (p->n_prune < 0 || p->n_prune > p->head_dim) return fail(KVC_ERR_INVALID, "n_prune=%d outside [0, head_dim]", p->n_prune);
    if (p->kc_stride_h < (int64_t)p->rows * p->head_dim) return fail(KVC_ERR_INVALID, "kc_stride_h smaller than rows * head_dim");
    kvc::ThinkArgs a;
    std::memset(&a, 0, sizeof(a));
    a.q = q; a.kc = kc; a.pruned = pruned; a.keep = keep; a.scores = channel_scores;
    a.q_stride_b = p->q_stride_b; a.q_stride_h = p->q_stride_h; a.q_stride_l = p->q_stride_l; a.kc_stride_h = p->kc_stride_h;
    a.bsz = p->bsz; a.n_heads = p->n_heads; a.q_len = p->q_len; a.head_dim = p->head_dim;
    a.rows = p->rows; a.recent = p->recent; a.n_prune = p->n_prune;
    if (int rc = kvc::launch_think(a, p->dtype, static_cast<hipStream_t>(hip_stream))) return fail(rc, "think launch failed");
    return hip_ok("think launch");
}

// ---- CAM (pyramidkv_utils.py:431-513), split at its random draw --------------------------------------------
namespace {
int cam_check(const kvc_params* p, int start_budget) {
    if (int rc = validate(p, true)) return rc;
    if (p->method != KVC_SNAPKV || p->pooling != KVC_POOL_NONE)
        return fail(KVC_ERR_INVALID, "kvc_cam_*: method must be KVC_SNAPKV with KVC_POOL_NONE (CAM scores without pooling, :468-476)");
    if (start_budget < 1 || start_budget + p->window >= p->q_len)
        return fail(KVC_ERR_INVALID, "start_budget %d: the merge loop needs 1 <= start_budget and start_budget + window < q_len", start_budget);
    return KVC_OK;
}
size_t cam_colmean_off(const Layout& l) { return align_up(l.total, 256); }
}  // namespace

__attribute__((visibility("default"))) size_t kvc_cam_workspace_bytes(const kvc_params* p) {
    if (validate(p, true)) return 0;
    const Layout l = carve(p);
    return cam_colmean_off(l) + align_up((size_t)p->bsz * p->n_q_heads * p->q_len * esize_of(p->dtype), 256);
}

__attribute__((visibility("default"))) int kvc_cam_plan(const kvc_params* p, const void* q, const void* k, int start_budget, void* scores_out,
                                                        void* prob_out, void* workspace, size_t workspace_bytes, void* hip_stream) {
    if (int rc = cam_check(p, start_budget)) return rc;
    if (!q || !k || !scores_out || !prob_out) return fail(KVC_ERR_INVALID, "q, k, scores_out and prob_out must be non-NULL");
    const int es = esize_of(p->dtype);
    if (int rc = check_strides("q", es, p->q_stride_b, p->q_stride_h, p->q_stride_l, q)) return rc;
    if (int rc = check_strides("k", es, p->k_stride_b, p->k_stride_h, p->k_stride_l, k)) return rc;
    const Layout l = carve(p);
    if (!workspace || ((uintptr_t)workspace) % 256 || workspace_bytes < kvc_cam_workspace_bytes(p))
        return fail(KVC_ERR_WORKSPACE, "workspace NULL, misaligned or smaller than kvc_cam_workspace_bytes");
    char* ws = static_cast<char*>(workspace);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    Items it;
    std::memset(&it, 0, sizeof(it));
    it.n = 1; it.q[0] = q; it.k[0] = k; it.scores[0] = scores_out;
    if (int rc = enqueue_scores(p, l, it, ws, st)) return rc;            // window-row sums, no pooling: the reference's attn_cache
    kvc::CamArgs a;
    std::memset(&a, 0, sizeof(a));
    a.logits = ws + l.logits; a.rowmax = reinterpret_cast<const float*>(ws + l.rowmax); a.rowsum = reinterpret_cast<const float*>(ws + l.rowsum);
    a.colmean = ws + cam_colmean_off(l); a.prob = prob_out;
    a.bsz = p->bsz; a.n_q_heads = p->n_q_heads; a.q_len = p->q_len; a.window = p->window; a.start = start_budget;
    if (int rc = kvc::launch_cam_prob(a, p->dtype, st)) return fail(rc, "no CAM kernel for dtype %d", p->dtype);
    return hip_ok("cam plan launch");
}

__attribute__((visibility("default"))) int kvc_cam_merge(const kvc_params* p, void* v, const uint8_t* merge_mask, int start_budget, void* hip_stream) {
    if (int rc = cam_check(p, start_budget)) return rc;
    if (!v || !merge_mask) return fail(KVC_ERR_INVALID, "v and merge_mask must be non-NULL");
    if (p->head_dim > 256) return fail(KVC_ERR_UNSUPPORTED, "head_dim > 256");
    kvc::CamArgs a;
    std::memset(&a, 0, sizeof(a));
    a.v = v; a.mask = merge_mask;
    a.v_stride_b = p->v_stride_b; a.v_stride_h = p->v_stride_h; a.v_stride_l = p->v_stride_l;
    a.bsz = p->bsz; a.n_q_heads = p->n_q_heads; a.q_len = p->q_len; a.window = p->window; a.start = start_budget;
    if (int rc = kvc::launch_cam_merge(a, p->dtype, p->head_dim, static_cast<hipStream_t>(hip_stream))) return fail(rc, "no CAM kernel for dtype %d", p->dtype);
    return hip_ok("cam merge launch");
}

__attribute__((visibility("default"))) int kvc_ragged_decode_step(const kvc_ragged_decode_params* p, const void* q, const void* k_new,
                                                                  const void* v_new, void* k_flat, void* v_flat, const int64_t* seg_off,
                                                                  const int32_t* seg_len, void* out, void* hip_stream) {
    if (!p) return fail(KVC_ERR_INVALID, "params is NULL");
    if (p->dtype < KVC_BF16 || p->dtype > KVC_FP32) return fail(KVC_ERR_INVALID, "unknown dtype %d", p->dtype);
    if (p->bsz < 1 || p->n_q_heads < 1 || p->n_kv_heads < 1 || p->n_q_heads % p->n_kv_heads) return fail(KVC_ERR_INVALID, "bsz / head counts invalid");
    if (p->head_dim != 64 && p->head_dim != 128) return fail(KVC_ERR_UNSUPPORTED, "decode kernels are built for head_dim 64 and 128, got %d", p->head_dim);
    if (p->appended < 0 || p->new_rows < 1) return fail(KVC_ERR_INVALID, "row counts invalid");
    if (!q || !k_new || !v_new || !k_flat || !v_flat || !seg_off || !seg_len || !out) return fail(KVC_ERR_INVALID, "a pointer is NULL");
    const int es = esize_of(p->dtype);
    {   // the kernels move 16-byte pieces, like kvc_decode_step's
        const void* ptrs[] = {q, k_new, v_new, k_flat, v_flat};
        for (const void* ptr : ptrs) if (((uintptr_t)ptr) % 16) return fail(KVC_ERR_ALIGNMENT, "a pointer is not 16-byte aligned");
        const int64_t strides[] = {p->q_stride_b, p->q_stride_h, p->q_stride_l, p->new_stride_b, p->new_stride_h, p->new_stride_l};
        for (int64_t sd : strides) if ((sd * es) % 16) return fail(KVC_ERR_ALIGNMENT, "a stride is not a multiple of 16 bytes");
        if (p->slack_rows > 0 && p->appended + p->new_rows > p->slack_rows)
            return fail(KVC_ERR_INVALID, "appended %d + new_rows %d exceed the %d spare rows behind every segment", p->appended, p->new_rows, p->slack_rows);
    }
    kvc::RaggedDecodeArgs a;
    std::memset(&a, 0, sizeof(a));
    a.q = q; a.k_new = k_new; a.v_new = v_new; a.k_flat = k_flat; a.v_flat = v_flat; a.out = out; a.seg_off = seg_off; a.seg_len = seg_len;
    a.q_stride_b = p->q_stride_b; a.q_stride_h = p->q_stride_h; a.q_stride_l = p->q_stride_l;
    a.new_stride_b = p->new_stride_b; a.new_stride_h = p->new_stride_h; a.new_stride_l = p->new_stride_l;
    a.out_stride_b = p->out_stride_b; a.out_stride_h = p->out_stride_h; a.out_stride_l = p->out_stride_l;
    a.n_q_heads = p->n_q_heads; a.group = p->n_q_heads / p->n_kv_heads; a.appended = p->appended; a.new_rows = p->new_rows;
    a.esize = es; a.row_bytes = p->head_dim * es; a.scaling = p->scaling;
    if (int rc = kvc::launch_ragged_decode(a, p->dtype, p->head_dim, p->bsz * p->n_q_heads, static_cast<hipStream_t>(hip_stream)))
        return fail(rc, "no ragged decode kernel for dtype %d head_dim %d", p->dtype, p->head_dim);
    return hip_ok("ragged decode launch");
}

__attribute__((visibility("default"))) int kvc_compress(const kvc_params* p, const void* q, const void* k, const void* v,
                                                        void* k_out, void* v_out, int64_t* idx_out, void* scores_out,
                                                        void* workspace, size_t workspace_bytes, void* hip_stream) {
    if (int rc = validate(p, true)) return rc;
    if ((p->debug_stage_mask & 7) != 0) return fail(KVC_ERR_INVALID, "debug_stage_mask bits 0-2 are only honoured by kvc_scores / kvc_compress_batch");
    Items it;
    std::memset(&it, 0, sizeof(it));
    it.n = 1;
    it.q[0] = q; it.k[0] = k; it.v[0] = v; it.k_out[0] = k_out; it.v_out[0] = v_out;
    it.idx[0] = idx_out; it.scores[0] = scores_out; it.keep[0] = p->k; it.out_stride_h[0] = p->out_stride_h;
    if (ragged(p->method)) return fail(KVC_ERR_INVALID, "AdaKV / HeadKV produce a ragged cache: use the kvc_ragged_* entry points");
    return run_items(p, it, workspace, workspace_bytes, static_cast<hipStream_t>(hip_stream));
}

__attribute__((visibility("default"))) int kvc_compress_batch(const kvc_params* p, int n_items, const int32_t* k_per_item,
                                                              const void* const* q, const void* const* k, const void* const* v,
                                                              void* const* k_out, void* const* v_out, int64_t* const* idx_out,
                                                              void* const* scores_out, const int64_t* out_stride_h_per_item,
                                                              void* workspace, size_t workspace_bytes, void* hip_stream) {
    if (int rc = validate(p, true)) return rc;
    if (n_items < 1 || !k_per_item || !k || !v || !k_out || !v_out) return fail(KVC_ERR_INVALID, "n_items >= 1 and the pointer arrays must be non-NULL");
    if (scoring(p->method) && !q) return fail(KVC_ERR_INVALID, "q array must be non-NULL for scoring methods");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    // chunks of KVC_MAX_ITEMS items share one launch of each kernel; each chunk uses the same workspace (stream-ordered)
    for (int base = 0; base < n_items; base += kvc::KVC_MAX_ITEMS) {
        Items it;
        std::memset(&it, 0, sizeof(it));
        it.n = n_items - base < kvc::KVC_MAX_ITEMS ? n_items - base : kvc::KVC_MAX_ITEMS;
        for (int i = 0; i < it.n; ++i) {
            it.q[i] = q ? q[base + i] : nullptr; it.k[i] = k[base + i]; it.v[i] = v[base + i];
            it.k_out[i] = k_out[base + i]; it.v_out[i] = v_out[base + i];
            it.idx[i] = idx_out ? idx_out[base + i] : nullptr;
            it.scores[i] = scores_out ? scores_out[base + i] : nullptr;
            it.keep[i] = k_per_item[base + i];
            it.out_stride_h[i] = out_stride_h_per_item ? out_stride_h_per_item[base + i] : p->out_stride_h;
        }
        if (int rc = run_items(p, it, workspace, workspace_bytes, st)) return rc;
    }
    return KVC_OK;
}

}  // extern "C"
