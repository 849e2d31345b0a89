/* Exhaustive check (all 2^32 fp32 inputs) that the kernel's 2-FMA scaling  q0 = x*rc; r = fma(-q0,c,x); q = fma(r,rc,q0)
 * with c = sqrtf(128), rc = RN(1/c) equals the IEEE division x / c whenever 2^-100 <= |x| < inf — the guard used by
 * ScaleDiv<128> in kvcache_factory_amd/csrc/kvc_score.hip (everything else takes the true division). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
int main(void) {
    const float c = sqrtf(128.0f), rc = u2f(0x3db504f3u), lo = u2f(0x0d800000u);
    uint32_t cb; memcpy(&cb, &c, 4);
    if (cb != 0x413504f3u || rc != 1.0f / c) { printf("BAD constants\n"); return 1; }
    long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long long u = 0; u < (1ll << 32); ++u) {
        const float x = u2f((uint32_t)u), ax = fabsf(x);
        if (!(ax >= lo && ax < INFINITY)) continue;
        const float ref = x / c, q0 = x * rc, r = fmaf(-q0, c, x), q = fmaf(r, rc, q0);
        if (memcmp(&ref, &q, 4)) ++bad;
    }
    printf("%s mismatches=%ld\n", bad ? "BAD" : "OK", bad);
    return bad != 0;
}
