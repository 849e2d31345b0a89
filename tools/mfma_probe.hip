// mfma_probe.hip — characterises the accumulation arithmetic of the gfx950 bf16/f16 MFMAs so the
// oracle can (or cannot) restate it bit for bit.  Not part of the product; run once on the GPU box.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// A: [32][16] row-major bf16 bits, B: [16][32] row-major, C/D: [32][32] f32
__global__ void k_bf16(const uint16_t* A, const uint16_t* B, const float* C, float* D) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    s16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)A[r * 16 + 8 * h + j]; b[j] = (short)B[(8 * h + j) * 32 + r]; }
    f32x16 c;
    for (int g = 0; g < 16; ++g) c[g] = C[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r];
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int g = 0; g < 16; ++g) D[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = c[g];
}
__global__ void k_f16(const uint16_t* A, const uint16_t* B, const float* C, float* D) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    h16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        uint16_t x = A[r * 16 + 8 * h + j], y = B[(8 * h + j) * 32 + r];
        _Float16 fx, fy; memcpy(&fx, &x, 2); memcpy(&fy, &y, 2);
        a[j] = fx; b[j] = fy;
    }
    f32x16 c;
    for (int g = 0; g < 16; ++g) c[g] = C[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r];
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int g = 0; g < 16; ++g) D[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = c[g];
}

static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float h2f(uint16_t h) { _Float16 x; memcpy(&x, &h, 2); return (float)x; }
static uint16_t f2h(float f) { _Float16 x = (_Float16)f; uint16_t h; memcpy(&h, &x, 2); return h; }
static double rnd() { return (double)rand() / RAND_MAX; }
static double gauss() { double u = rnd() + 1e-12, v = rnd(); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); }
static float rz(long double x) {   // round toward zero to fp32
    float f = (float)x;            // RN
    if (fabsl((long double)f) > fabsl(x)) f = nextafterf(f, 0.0f);
    return f;
}

int main() {
    const int TRIALS = 400;
    uint16_t *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, 32 * 16 * 2); hipMalloc(&dB, 16 * 32 * 2); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
    for (int fmt = 0; fmt < 2; ++fmt) {
        for (int spread = 0; spread <= 12; spread += 6) {
            for (int cmode = 0; cmode < 2; ++cmode) {
                long n = 0, mm[8] = {0};
                for (int t = 0; t < TRIALS; ++t) {
                    std::vector<uint16_t> A(512), B(512); std::vector<float> C(1024), D(1024);
                    for (auto& x : A) { float v = (float)(gauss() * ldexp(1.0, (int)(rnd() * (spread + 1)) - spread / 2)); x = fmt ? f2h(v) : f2bf(v); }
                    for (auto& x : B) { float v = (float)(gauss() * ldexp(1.0, (int)(rnd() * (spread + 1)) - spread / 2)); x = fmt ? f2h(v) : f2bf(v); }
                    for (auto& x : C) x = cmode ? (float)(gauss() * 8.0) : 0.0f;
                    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
                    hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
                    if (fmt) hipLaunchKernelGGL(k_f16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
                    else hipLaunchKernelGGL(k_bf16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
                    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
                    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
                        float a[16], b[16];
                        for (int k = 0; k < 16; ++k) { a[k] = fmt ? h2f(A[i * 16 + k]) : bf2f(A[i * 16 + k]); b[k] = fmt ? h2f(B[k * 32 + j]) : bf2f(B[k * 32 + j]); }
                        const float c = C[i * 32 + j], d = D[i * 32 + j];
                        long double ex = c; for (int k = 0; k < 16; ++k) ex += (long double)a[k] * b[k];
                        float h0 = (float)ex;                                   // exact, one RN rounding
                        float h1 = c; for (int k = 0; k < 16; ++k) h1 = fmaf(a[k], b[k], h1);   // k-ordered fma chain
                        float h2 = rz(ex);                                      // exact, truncated
                        long double s0 = 0, s1 = 0; for (int k = 0; k < 8; ++k) { s0 += (long double)a[k] * b[k]; s1 += (long double)a[8 + k] * b[8 + k]; }
                        float h3 = (float)((long double)(float)((long double)c + s0) + s1);       // two groups of 8, RN each
                        float h4 = (float)((long double)c + (long double)(float)(s0 + s1));       // products summed+rounded, then + c
                        long double q = 0; for (int k = 0; k < 16; k += 4) { long double g4 = 0; for (int e = 0; e < 4; ++e) g4 += (long double)a[k + e] * b[k + e]; q += g4; }
                        float h5 = c; for (int k = 0; k < 16; k += 4) { long double g4 = 0; for (int e = 0; e < 4; ++e) g4 += (long double)a[k + e] * b[k + e]; h5 = (float)((long double)h5 + g4); }   // groups of 4 chained
                        float h6 = c; for (int k = 0; k < 16; k += 2) { long double g2 = (long double)a[k] * b[k] + (long double)a[k + 1] * b[k + 1]; h6 = (float)((long double)h6 + g2); }       // pairs chained
                        ++n;
                        mm[0] += memcmp(&d, &h0, 4) != 0; mm[1] += memcmp(&d, &h1, 4) != 0; mm[2] += memcmp(&d, &h2, 4) != 0;
                        mm[3] += memcmp(&d, &h3, 4) != 0; mm[4] += memcmp(&d, &h4, 4) != 0; mm[5] += memcmp(&d, &h5, 4) != 0; mm[6] += memcmp(&d, &h6, 4) != 0;
                        if (t == 0 && i == 0 && j < 2 && memcmp(&d, &h0, 4)) printf("   ex d=%a exactRN=%a chain=%a rz=%a\n", d, h0, h1, h2);
                    }
                }
                printf("%s 32x32x16 spread=2^%d c=%s n=%ld mismatches: exactRN %ld | fma-chain %ld | exactRZ %ld | 8+8 %ld | sum-then-c %ld | 4-chained %ld | 2-chained %ld\n",
                       fmt ? "f16 " : "bf16", spread, cmode ? "rand" : "0", n, mm[0], mm[1], mm[2], mm[3], mm[4], mm[5], mm[6]);
            }
        }
    }
    return 0;
}
