// kvc_gather.hip — A8: compaction of K / V from q_len rows down to k + window rows per head
// (pyramidkv_utils.py:335-346: gather(dim=2, index) of the prefix + cat with the last W tokens).
// Pure copy, HBM-bound: every 16-byte lane moves one chunk of one row; rows of a head are
// contiguous in the output, so stores are full lines and loads are whole rows (256 B at D=128 bf16).
#include "kvc_common.h"
#include "kvc_launch.h"

namespace kvc {

__global__ __launch_bounds__(256) void gather_kernel(const GatherPair pr) {
    const int which = blockIdx.z % pr.count, item = blockIdx.z / pr.count;
    const GatherArgs& a = pr.t[which];
    const int k = a.k.v[item];
    const int lanes_per_row = a.row_bytes >> 4;
    const int rows_per_block = 256 / lanes_per_row;
    const int r = threadIdx.x / lanes_per_row, c = threadIdx.x % lanes_per_row;
    const int cap = k + a.window;
    const int t = blockIdx.x * rows_per_block + r;
    if (r >= rows_per_block || t >= cap) return;
    const int hb = blockIdx.y, b = hb / a.n_q_heads, h = hb % a.n_q_heads;
    int64_t srow;
    const int64_t* idx = reinterpret_cast<const int64_t*>(a.idx.p[item]);
    if (t < k) srow = idx ? idx[(int64_t)hb * k + t] : (int64_t)t;
    else srow = (int64_t)(a.q_len - a.window) + (t - k);
    const int64_t esz = a.esize;
    const char* src = reinterpret_cast<const char*>(a.src.p[item]) +
        ((int64_t)b * a.stride_b + (int64_t)(h / a.group) * a.stride_h + srow * a.stride_l) * esz + c * 16;
    const int64_t ohb = a.out_head_bytes.v[item];
    const int64_t head_bytes = ohb ? ohb : (int64_t)cap * a.row_bytes;
    char* dst = reinterpret_cast<char*>(const_cast<void*>(a.out.p[item])) + (int64_t)hb * head_bytes + (int64_t)t * a.row_bytes + c * 16;
    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
}

int launch_gather(const GatherPair& p, hipStream_t st) {
    const GatherArgs& a = p.t[0];
    const int lanes_per_row = a.row_bytes / 16;
    const int rows_per_block = 256 / lanes_per_row;
    const int cap = a.k_max + a.window;
    dim3 grid((unsigned)((cap + rows_per_block - 1) / rows_per_block), (unsigned)(a.bsz * a.n_q_heads), (unsigned)(p.count * a.n_items));
    hipLaunchKernelGGL(gather_kernel, grid, dim3(256), 0, st, p);
    return 0;
}

}  // namespace kvc
