// Does VALU work issue in the shadow of a dependent f32 MFMA chain on gfx950?  Times, per wave configuration:
//   MFMA-only chain, VALU-only (NV v_fma per step), and both interleaved.  If interleaved ~= max(...) they overlap;
//   if ~= sum(...) they serialise.  Also the same for the packed-bf16 32x32x16 MFMA (8 passes).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NV, bool BF16>
__global__ __launch_bounds__(256) void k(float* out, int steps, float a0, float b0) {
    f32x16 acc = {0};
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a0 + i + threadIdx.x;
    float a = a0, b = b0;
    s16x8 pa = {1, 2, 3, 4, 5, 6, 7, 8}, pb = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int s = 0; s < steps; ++s) {
        if (MODE & 1) {
            if constexpr (BF16) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], b0, a0);
        }
        if ((MODE & 3) == 3) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE, int NV, bool BF16>
float run(int blocks, int steps, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV, BF16>), dim3(blocks), dim3(256), 0, 0, d, steps, 1.0f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV, BF16>), dim3(blocks), dim3(256), 0, 0, d, steps, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f;
}
template <int NV, bool BF16> void row(int blocks, int steps, float* d) {
    const float m = run<1, NV, BF16>(blocks, steps, d), v = run<2, NV, BF16>(blocks, steps, d), b = run<3, NV, BF16>(blocks, steps, d);
    printf("%s waves/SIMD=%d NV=%2d: mfma %.1f us  valu %.1f us  both %.1f us  (sum %.1f, max %.1f)\n", BF16 ? "bf16x16" : "f32x2  ",
           blocks / 256, NV, m, v, b, m + v, m > v ? m : v);
}
int main() {
    float* d; hipMalloc(&d, 256 * 12 * 256 * 4);
    const int steps = 20000;
    for (int w = 1; w <= 3; w += 2) {
        row<4, false>(256 * w, steps, d); row<8, false>(256 * w, steps, d); row<12, false>(256 * w, steps, d); row<16, false>(256 * w, steps, d);
        row<4, true>(256 * w, steps, d); row<8, true>(256 * w, steps, d);
    }
    return 0;
}
