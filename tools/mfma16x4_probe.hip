// mfma16x4_probe.hip — is v_mfma_f32_16x16x4_f32 a k-ordered fmaf chain (like 32x32x2, which the exact-mode kernels rely on)?
// Not part of the product; run once on the GPU box.
//   hipcc --offload-arch=gfx950 -O2 -o mfma16x4_probe mfma16x4_probe.hip && ./mfma16x4_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// A: [16][K] row-major f32, B: [K][16], D: [16][16]; K = 4 * steps
__global__ void k16(const float* A, const float* B, float* D, int steps) {
    const int l = threadIdx.x, n = l & 15, g = l >> 4;
    f32x4 c = {0, 0, 0, 0};
    for (int s = 0; s < steps; ++s)
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[n * 4 * steps + 4 * s + g], B[(4 * s + g) * 16 + n], c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[(4 * g + v) * 16 + n] = c[v];
}
static double rnd() { return (double)rand() / RAND_MAX; }
static double gauss() { double u = rnd() + 1e-12, v = rnd(); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); }
int main() {
    const int steps = 32, K = 4 * steps;
    float *dA, *dB, *dD;
    hipMalloc(&dA, 16 * K * 4); hipMalloc(&dB, 16 * K * 4); hipMalloc(&dD, 1024);
    long n = 0, bad_chain = 0, bad_pair = 0, bad_rev = 0;
    for (int t = 0; t < 200; ++t) {
        std::vector<float> A(16 * K), B(16 * K), D(256);
        for (auto& x : A) x = (float)(gauss() * ldexp(1.0, (int)(rnd() * 7) - 3));
        for (auto& x : B) x = (float)(gauss() * ldexp(1.0, (int)(rnd() * 7) - 3));
        hipMemcpy(dA, A.data(), 16 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 16 * K * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD, steps);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            float h1 = 0.0f, h2 = 0.0f, h3 = 0.0f;
            for (int k = 0; k < K; ++k) h1 = fmaf(A[i * K + k], B[k * 16 + j], h1);
            for (int k = 0; k < K; k += 2) h2 = (float)((long double)h2 + (long double)A[i * K + k] * B[k * 16 + j] + (long double)A[i * K + k + 1] * B[(k + 1) * 16 + j]);
            for (int k = 0; k < K; k += 4) for (int e = 3; e >= 0; --e) h3 = fmaf(A[i * K + k + e], B[(k + e) * 16 + j], h3);
            const float d = D[i * 16 + j];
            ++n; bad_chain += memcmp(&d, &h1, 4) != 0; bad_pair += memcmp(&d, &h2, 4) != 0; bad_rev += memcmp(&d, &h3, 4) != 0;
        }
    }
    printf("f32 16x16x4 x %d steps, n=%ld: mismatches vs k-ordered fmaf chain %ld | pairs-chained %ld | reversed-in-group chain %ld\n", steps, n, bad_chain, bad_pair, bad_rev);
    return 0;
}
