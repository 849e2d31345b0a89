// Latency probe for the lane-parallel libstdc++ heap (kvc_select_exact.hip): one wave runs partial_sort_wave on a row
// of 7992 bf16 scores; s_memtime around the whole call and around the three phases (diagnostic counters).
#include "../kvcache_factory_amd/csrc/kvc_select_exact.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <queue>
using namespace kvc;
__global__ __launch_bounds__(64) void k(const uint16_t* s, int n, int kk, int64_t* out, unsigned long long* t) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    partial_sort_wave<KVC_BF16, uint32_t>(s + (size_t)blockIdx.x * n, n, kk, reinterpret_cast<uint32_t*>(smem), out + (size_t)blockIdx.x * kk);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
int main(int argc, char** argv) {
    const int n = 7992, kk = 120, heads = 1024;
    std::vector<uint16_t> h((size_t)n * heads);
    srand(1);
    for (auto& x : h) { float f = (float)(rand() % 997) / 997.0f * 0.01f; uint32_t u; memcpy(&u, &f, 4); x = (uint16_t)(u >> 16); }
    uint16_t* ds; int64_t* dout; unsigned long long* dt;
    (void)hipMalloc(&ds, h.size() * 2); (void)hipMalloc(&dout, (size_t)heads * kk * 8); (void)hipMalloc(&dt, heads * 8);
    (void)hipMemcpy(ds, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    {   // replacements std::partial_sort performs on head 0 (min-heap of the k largest keys; strict >)
        std::priority_queue<uint16_t, std::vector<uint16_t>, std::greater<uint16_t>> pq;
        int ops = 0;
        for (int i = 0; i < n; ++i) { if (i < kk) pq.push(h[i]); else if (h[i] > pq.top()) { pq.pop(); pq.push(h[i]); ++ops; } }
        printf("head 0: %d tail replacements + %d make_heap + %d sort_heap steps\n", ops, kk / 2, kk - 1);
    }
    for (int grid : {1, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(64), 4096, 0, ds, n, kk, dout, dt);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> t(grid);
            (void)hipMemcpy(t.data(), dt, grid * 8, hipMemcpyDeviceToHost);
            unsigned long long mx = 0, sum = 0; for (auto v : t) { mx = v > mx ? v : mx; sum += v; }
            printf("grid %4d: kernel %.1f us; shader cycles per wave (s_memtime): mean %.0f max %llu\n", grid, ms * 1000, (double)sum / grid, mx);
        }
    }
    return 0;
}
