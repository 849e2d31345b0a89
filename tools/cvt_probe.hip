// cvt_probe.hip — does v_cvt_f16_f32 (what `(_Float16)f` compiles to) round like IEEE RNE for results that are
// fp16 subnormals?  Compared with a software RNE conversion on 4M floats.  Bring-up tool, not product.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
__global__ void k(const float* x, uint16_t* y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { _Float16 h = (_Float16)x[i]; uint16_t r; __builtin_memcpy(&r, &h, 2); y[i] = r; }
}
__global__ void k2(const float* a, const float* b, uint16_t* y, int n) {   // product then convert, like rnd(e*rinv)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float p = a[i] * b[i]; _Float16 h = (_Float16)p; float f = (float)h; _Float16 h2 = (_Float16)f; uint16_t r; __builtin_memcpy(&r, &h2, 2); y[i] = r; }
}
static uint16_t sw(float f) {
    uint32_t u; memcpy(&u, &f, 4); uint32_t sign = u & 0x80000000u; u ^= sign; uint16_t o;
    if (u >= ((127u + 16u) << 23)) o = (u > 0x7f800000u) ? 0x7e00 : 0x7c00;
    else if (u < (113u << 23)) { uint32_t magic = ((127u - 15u) + (23u - 10u) + 1u) << 23; float t, m; memcpy(&t, &u, 4); memcpy(&m, &magic, 4); t += m; uint32_t tu; memcpy(&tu, &t, 4); o = (uint16_t)(tu - magic); }
    else { uint32_t odd = (u >> 13) & 1u; u += ((uint32_t)(15 - 127) << 23) + 0xfffu; u += odd; o = (uint16_t)(u >> 13); }
    return o | (sign >> 16);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n), a(n), b(n); std::vector<uint16_t> y(n), y2(n);
    for (int i = 0; i < n; ++i) {
        uint32_t e = 127 - 27 + (rand() % 16);            // exponents 2^-27 .. 2^-12
        uint32_t u = (e << 23) | ((uint32_t)rand() & 0x7fffff); if (i & 1) u &= ~0x3ffu;   // half of them with few low bits (ties)
        memcpy(&x[i], &u, 4);
        a[i] = (float)rand() / RAND_MAX; b[i] = 1.0f / (200.0f + rand() % 300) * 0.01f;
    }
    float *dx, *da, *db; uint16_t* dy;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dy, n * 2);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, n); hipMemcpy(y.data(), dy, n * 2, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k2, dim3(n / 256), dim3(256), 0, 0, da, db, dy, n); hipMemcpy(y2.data(), dy, n * 2, hipMemcpyDeviceToHost);
    long bad = 0, bad2 = 0;
    for (int i = 0; i < n; ++i) {
        if (y[i] != sw(x[i])) { if (bad < 8) printf("cvt x=%a hw=%04x sw=%04x\n", x[i], y[i], sw(x[i])); ++bad; }
        float p = a[i] * b[i];
        if (y2[i] != sw(p)) { if (bad2 < 8) printf("mulcvt a=%a b=%a p=%a hw=%04x sw=%04x\n", a[i], b[i], p, y2[i], sw(p)); ++bad2; }
    }
    printf("cvt mismatches %ld / %d ; mul+cvt mismatches %ld / %d\n", bad, n, bad2, n);
    return 0;
}
