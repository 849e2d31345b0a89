// Probe for the lane-parallel libstdc++ heaps of kvc_select_exact.hip (partial_sort regime, k <= 128): every head's
// indices against the host's std::partial_sort with torch's comparator (the reference order, ties included), and the
// latency of one wave / of 1 024 waves for the round-2 heap (WaveHeap: LDS exchange inside the step) and the round-3
// heap (WaveHeapL: locally maintained child caches).  Usage: heap_probe [n] [k] [distinct_values]
#define KVC_HEAP_STAMPS 1
#include "../kvcache_factory_amd/csrc/kvc_select_exact.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <queue>
#include <vector>
using namespace kvc;
template <int WHICH>
__global__ __launch_bounds__(64) void k(const uint16_t* s, int n, int kk, int64_t* out, unsigned long long* t) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (WHICH == 0) partial_sort_wave<KVC_BF16, uint32_t>(s + (size_t)blockIdx.x * n, n, kk, reinterpret_cast<uint32_t*>(smem), out + (size_t)blockIdx.x * kk);
    else partial_sort_waveL<KVC_BF16>(s + (size_t)blockIdx.x * n, n, kk, reinterpret_cast<uint32_t*>(smem), out + (size_t)blockIdx.x * kk);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
// N dependent steps in a plain loop (every value replaces the root), with parts of the step left out: V = 0 the whole step,
// 1 without the v_readlane of the root's child, 2 without the LDS exchange, 3 with neither
template <int V>
__global__ __launch_bounds__(64) void steps(const uint16_t* s, int n, int kk, int iters, unsigned long long* t, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* lds = reinterpret_cast<uint32_t*>(smem);
    typedef HeapNode<uint32_t> N;
    const int lane = threadIdx.x;
    WaveHeapL H;
    H.init(lds, lane < kk ? N::make(Key<KVC_BF16>::of(s[lane]), lane) : WaveHeapL::SENT, lane + 64 < kk ? N::make(Key<KVC_BF16>::of(s[lane + 64]), lane + 64) : WaveHeapL::SENT);
    H.make_heap(kk);
    uint32_t key = N::key(N::rdlane(H.lo, 0)), rootor = N::rdlane(H.lo, 0) | 0xffffu;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < iters; ++i) {
        key += (i & 3) == 0;
        H.template step<true, V>(N::make(key, i & 0xffff), rootor);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (lane == 0) t[0] = t1 - t0;
    sink[lane] = H.lo + H.xl + H.xr + H.xc + H.xcc + H.c0 + rootor;
}
static uint32_t key_of(uint16_t r) { return (r & 0x8000u) ? (uint32_t)(~r & 0xffffu) : (uint32_t)(r | 0x8000u); }
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 7992, kk = argc > 2 ? atoi(argv[2]) : 120, distinct = argc > 3 ? atoi(argv[3]) : 260, heads = 1024;
    std::vector<uint16_t> h((size_t)n * heads);
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) {
        const int head = (int)(i / n);
        const int d = head % 7 == 3 ? 3 : (head % 7 == 5 ? 40000 : distinct);      // a few heads with huge plateaus / nearly tie-free
        float f = (float)(rand() % d) / (float)d * 0.01f; uint32_t u; memcpy(&u, &f, 4); h[i] = (uint16_t)(u >> 16);
    }
    uint16_t* ds; int64_t* dout; unsigned long long* dt;
    (void)hipMalloc(&ds, h.size() * 2); (void)hipMalloc(&dout, (size_t)heads * kk * 8); (void)hipMalloc(&dt, heads * 8);
    (void)hipMemcpy(ds, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    // the reference order: std::partial_sort on (key, index) with a key-only comparator (torch-CPU topk, k * 64 <= n)
    std::vector<int64_t> want((size_t)heads * kk);
    long ops0 = 0;
    {   // replacements std::partial_sort performs on head 0 (min-heap of the k largest keys; strict >)
        std::priority_queue<uint32_t, std::vector<uint32_t>, std::greater<uint32_t>> pq;
        for (int i = 0; i < n; ++i) { const uint32_t key = key_of(h[i]); if (i < kk) pq.push(key); else if (key > pq.top()) { pq.pop(); pq.push(key); ++ops0; } }
    }
    for (int hd = 0; hd < heads; ++hd) {
        std::vector<std::pair<uint32_t, int>> q(n);
        for (int i = 0; i < n; ++i) q[i] = {key_of(h[(size_t)hd * n + i]), i};
        std::partial_sort(q.begin(), q.begin() + kk, q.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
        for (int i = 0; i < kk; ++i) want[(size_t)hd * kk + i] = q[i].second;
    }
    for (int which : {0, 1}) {
        for (int grid : {1, 1024}) {
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipMemset(dout, 0xff, (size_t)heads * kk * 8);
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                (void)hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 20480, 0, ds, n, kk, dout, dt);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 20480, 0, ds, n, kk, dout, dt);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> t(grid);
                std::vector<int64_t> got((size_t)grid * kk);
                (void)hipMemcpy(t.data(), dt, grid * 8, hipMemcpyDeviceToHost);
                (void)hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost);
                long bad = 0; int bad_heads = 0;
                for (int hd = 0; hd < grid; ++hd) { long b = 0; for (int i = 0; i < kk; ++i) b += got[(size_t)hd * kk + i] != want[(size_t)hd * kk + i]; bad += b; bad_heads += b != 0; }
                unsigned long long mx = 0, sum = 0; for (auto v : t) { mx = v > mx ? v : mx; sum += v; }
                if (which == 1 && rep == 1) {
                    unsigned long long st[8];
                    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_heap_stamps), sizeof(st));
                    printf("    head 0 cycles: first vector + init %llu | make_heap (%d steps) %llu | land + heap_select %llu (%llu replacements) | sort_heap (%d pops) %llu | output %llu\n",
                           st[1] - st[0], kk / 2, st[2] - st[1], st[3] - st[2], (unsigned long long)ops0, kk - 1, st[4] - st[3], st[5] - st[4]);
                }
                printf("%s grid %4d: kernel %7.1f us; s_memtime ticks per wave: mean %.0f max %llu; wrong indices %ld in %d heads\n",
                       which ? "WaveHeapL" : "WaveHeap ", grid, ms * 1000, (double)sum / grid, mx, bad, bad_heads);
            }
        }
    }
    {
        uint32_t* sink; (void)hipMalloc(&sink, 256);
        const int iters = 2000;
        const char* what[] = {"whole step", "no v_readlane of the root's child", "no LDS exchange", "neither"};
        for (int v = 0; v < 4; ++v) {
            for (int rep = 0; rep < 2; ++rep) {
                if (v == 0) hipLaunchKernelGGL(steps<0>, dim3(1), dim3(64), 20480, 0, ds, n, kk, iters, dt, sink);
                if (v == 1) hipLaunchKernelGGL(steps<1>, dim3(1), dim3(64), 20480, 0, ds, n, kk, iters, dt, sink);
                if (v == 2) hipLaunchKernelGGL(steps<2>, dim3(1), dim3(64), 20480, 0, ds, n, kk, iters, dt, sink);
                if (v == 3) hipLaunchKernelGGL(steps<3>, dim3(1), dim3(64), 20480, 0, ds, n, kk, iters, dt, sink);
                (void)hipDeviceSynchronize();
            }
            unsigned long long c; (void)hipMemcpy(&c, dt, 8, hipMemcpyDeviceToHost);
            printf("    %d steps in a plain loop, %-36s: %.1f cycles per step\n", iters, what[v], (double)c / iters);
        }
    }
    return 0;
}
