// mfma_peak_probe.hip — what f32-input MFMA rate does an MI355X SUSTAIN (all CUs busy for milliseconds, operands that are
// not constants)?  The exact-mode kernels are priced against 157.3 TFLOP/s (256 CUs x 256 flop/cycle x 2.4 GHz); this
// measures the rate of a pure dependent-chain kernel so that their fractions can be read against what the chip delivers.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_peak_probe mfma_peak_probe.hip && ./mfma_peak_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int SHAPE, int CHAINS>
__global__ __launch_bounds__(1024) void k(float* out, const float* in, int steps, uint64_t* clk) {
    const float a = in[threadIdx.x], b = in[threadIdx.x + 1024];
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    float r = 0.0f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0, 0, 0, 0};
        for (int s = 0; s < steps; ++s)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + c, b, acc[c], 0, 0, 0);
        for (int c = 0; c < CHAINS; ++c) r += acc[c][0] + acc[c][3];
    } else {
        f32x16 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.0f;
        for (int s = 0; s < steps; ++s)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + c, b, acc[c], 0, 0, 0);
        for (int c = 0; c < CHAINS; ++c) r += acc[c][0] + acc[c][15];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
template <int SHAPE, int CHAINS> void run(const char* name, int threads, int wgs_per_cu, int steps, float* out, float* in, uint64_t* clk) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL((k<SHAPE, CHAINS>), dim3(grid), dim3(threads), 0, 0, out, in, steps / 8, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, CHAINS>), dim3(grid), dim3(threads), 0, 0, out, in, steps, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)grid * (threads / 64) * (double)steps * CHAINS * (SHAPE == 16 ? 2048.0 : 4096.0);
    printf("%-44s %8.3f ms  %7.1f TFLOP/s   s_memtime %.0f MHz, wall_clock64 %.0f MHz\n", name, ms, flops / ms * 1e-9, h[0] / ms * 1e-3, h[1] / ms * 1e-3);
}
int main() {
    float *out, *in; uint64_t* clk;
    hipMalloc(&out, 256 * 4 * 1024 * 4); hipMalloc(&in, 8192); hipMalloc(&clk, 16);
    float h[2048]; for (int i = 0; i < 2048; ++i) h[i] = 0.37f * (float)((i * 2654435761u) % 1000) / 1000.0f - 0.2f;
    hipMemcpy(in, h, 8192, hipMemcpyHostToDevice);
    run<16, 1>("16x16x4 f32, 16 waves/CU, 1 chain", 1024, 1, 200000, out, in, clk);
    run<16, 2>("16x16x4 f32, 16 waves/CU, 2 chains", 1024, 1, 100000, out, in, clk);
    run<16, 1>("16x16x4 f32, 4 waves/CU, 1 chain", 256, 1, 400000, out, in, clk);
    run<16, 4>("16x16x4 f32, 4 waves/CU, 4 chains", 256, 1, 100000, out, in, clk);
    run<32, 1>("32x32x2 f32, 16 waves/CU, 1 chain", 1024, 1, 100000, out, in, clk);
    run<32, 1>("32x32x2 f32, 8 waves/CU, 1 chain", 512, 1, 200000, out, in, clk);
    run<32, 1>("32x32x2 f32, 4 waves/CU, 1 chain", 256, 1, 200000, out, in, clk);
    run<32, 2>("32x32x2 f32, 4 waves/CU, 2 chains", 256, 1, 100000, out, in, clk);
    return 0;
}
