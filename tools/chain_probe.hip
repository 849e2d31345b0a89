// How long does a chain of dependent fp32 adds take on one wave of gfx950 — operands in registers, and fetched from LDS eight at
// a time one batch ahead (the torch-order row sums of softmax_pool_kernel)?  Alone on its SIMD and with 3 other waves on it.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* out, unsigned long long* cyc, int reps, int mode) {
    __shared__ float s[16 * 1040];
    for (int i = threadIdx.x; i < 16 * 1040; i += blockDim.x) s[i] = (float)(i % 7) * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.0f;
    unsigned long long t0 = 0, t1 = 0;
    if (wave < 2 || mode == 2) {
        const float* row = s + (wave & 7) * 1040 + (lane & 15);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        for (int r = 0; r < reps; ++r) {
            if (mode == 0) {
                float v = (float)lane;
#pragma unroll
                for (int i = 0; i < 64; ++i) { acc = acc + v; asm volatile("" : "+v"(acc)); }
            } else {
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
                asm volatile("" : "+v"(acc));
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    }
    out[threadIdx.x] = acc;
    if (lane == 0) cyc[wave] = t1 - t0;
}
int main() {
    float* o; unsigned long long* c; unsigned long long h[16];
    (void)hipMalloc(&o, 4096); (void)hipMalloc(&c, 128);
    for (int mode = 0; mode < 3; ++mode) {
        for (int threads : {128, 1024}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, o, c, 100, mode);
            (void)hipMemcpy(h, c, 128, hipMemcpyDeviceToHost);
            printf("mode %d (%s) block %4d: wave0 %.1f ticks per 64-add chain\n", mode,
                   mode == 0 ? "register operands" : mode == 1 ? "LDS operands, 2 chain waves" : "LDS operands, every wave chains", threads, (double)h[0] / 100.0);
        }
    }
    return 0;
}
