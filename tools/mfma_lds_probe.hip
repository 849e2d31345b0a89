// Cost of non-VALU instructions inside a dependent f32 MFMA chain on gfx950 (3 waves / SIMD):
//   0: MFMA chain only (4 per step)     1: + 5 LDS reads per step (1 b128 + 4 u16_d16_hi), waited one step later
//   2: + only s_waitcnt + s_nop         3: + 1 b128 + 1 b128 (B as 16-byte read, no widening: wrong data, timing only)
//   4: + 5 LDS reads but MFMAs use constant operands (loads unused)   5: + 8 s_nop 0   6: + one SALU add chain (8 ops)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 3) void k(float* out, int steps, float a0, float b0, int sa) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<float*>(lds)[i] = 1.0f;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t aaddr = base + (threadIdx.x & 63) * 16, baddr = base + 16384 + (threadIdx.x & 31) * 272 + 2 * ((threadIdx.x >> 5) & 1);
    f32x16 acc = {0};
    f32x4 A0 = {a0, a0, a0, a0}, A1 = A0;
    uint32_t B0[4] = {0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u}, B1[4] = {0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u};
    int sacc = sa;
    for (int s = 0; s < steps; s += 2) {
#define LD(A, B)                                                                                                      \
        asm volatile("ds_read_b128 %0, %5 offset:1024\n\tds_read_u16_d16_hi %1, %6 offset:16\n\tds_read_u16_d16_hi %2, %6 offset:20\n\t" \
                     "ds_read_u16_d16_hi %3, %6 offset:24\n\tds_read_u16_d16_hi %4, %6 offset:28"                       \
                     : "=v"(A), "=v"(B[0]), "=v"(B[1]), "=v"(B[2]), "=v"(B[3]) : "v"(aaddr), "v"(baddr))
#define LD2(A, B)                                                                                                     \
        asm volatile("ds_read_b128 %0, %2 offset:1024\n\tds_read_b128 %1, %3 offset:16" : "=v"(A), "=v"(*reinterpret_cast<f32x4*>(B)) : "v"(aaddr), "v"(baddr & ~15u))
#define WAIT(N, A, B) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(A), "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]))
#define MF(A, B)                                                                                                      \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[0], __uint_as_float(B[0]), acc, 0, 0, 0);                         \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[1], __uint_as_float(B[1]), acc, 0, 0, 0);                         \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[2], __uint_as_float(B[2]), acc, 0, 0, 0);                         \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[3], __uint_as_float(B[3]), acc, 0, 0, 0)
        if constexpr (MODE == 0) { MF(A0, B0); MF(A1, B1); }
        if constexpr (MODE == 1) { LD(A1, B1); WAIT(5, A0, B0); MF(A0, B0); LD(A0, B0); WAIT(5, A1, B1); MF(A1, B1); }
        if constexpr (MODE == 2) { WAIT(5, A0, B0); MF(A0, B0); WAIT(5, A1, B1); MF(A1, B1); }
        if constexpr (MODE == 3) { LD2(A1, B1); WAIT(2, A0, B0); MF(A0, B0); LD2(A0, B0); WAIT(2, A1, B1); MF(A1, B1); }
        if constexpr (MODE == 4) {
            f32x4 X; uint32_t Y[4];
            LD(X, Y); MF(A0, B0); LD(X, Y); MF(A1, B1);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X), "+v"(Y[0]), "+v"(Y[1]), "+v"(Y[2]), "+v"(Y[3]));
        }
        if constexpr (MODE == 5) {
            MF(A0, B0); asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
            MF(A1, B1); asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
        }
        if constexpr (MODE == 6) {
            MF(A0, B0);
#pragma unroll
            for (int i = 0; i < 8; ++i) sacc = __builtin_amdgcn_readfirstlane(sacc) * 3 + 1;
            MF(A1, B1);
#pragma unroll
            for (int i = 0; i < 8; ++i) sacc = __builtin_amdgcn_readfirstlane(sacc) * 3 + 1;
        }
    }
    float r = (float)sacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> void run(float* d, int steps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(768), dim3(256), 0, 0, d, steps, 1.0f, 0.5f, 1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(768), dim3(256), 0, 0, d, steps, 1.0f, 0.5f, 1);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d: %.1f us  -> %.1f ns per 4-MFMA step per wave-slot (SIMD time per step %.1f ns)\n", MODE, ms * 1000, ms * 1e6 / steps, ms * 1e6 / steps / 3);
}
int main() {
    float* d; (void)hipMalloc(&d, 768 * 256 * 4);
    const int steps = 4000;
    run<0>(d, steps); run<1>(d, steps); run<2>(d, steps); run<3>(d, steps); run<4>(d, steps); run<5>(d, steps); run<6>(d, steps);
    return 0;
}
