// ds_read_u16_d16_hi on gfx950: does the 16-bit LDS load land in the HIGH half of the VGPR with the low half zero
// (either preserved from a zeroed register or cleared by the load)?  That is an exact bf16 -> fp32 widening with no VALU op.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const uint16_t* in, uint32_t* out) {
    __shared__ uint16_t s[256];
    s[threadIdx.x] = in[threadIdx.x]; s[threadIdx.x + 64] = in[threadIdx.x + 64];
    s[threadIdx.x + 128] = in[threadIdx.x + 128]; s[threadIdx.x + 192] = in[threadIdx.x + 192];
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)s + threadIdx.x * 2;
    uint32_t a = 0, b = 0xdeadbeefu, c = 0;
    asm volatile("ds_read_u16_d16_hi %0, %1" : "+v"(a) : "v"(addr) : "memory");
    asm volatile("ds_read_u16_d16_hi %0, %1 offset:128" : "+v"(b) : "v"(addr) : "memory");
    asm volatile("ds_read_u16_d16_hi %0, %1 offset:256" : "+v"(c) : "v"(addr) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("ds_read_u16_d16_hi %0, %1 offset:384" : "+v"(c) : "v"(addr) : "memory");   // second load into the same register
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b; out[128 + threadIdx.x] = c;
}
int main() {
    uint16_t h[256]; for (int i = 0; i < 256; ++i) h[i] = (uint16_t)(0x3f80 + i * 37);
    uint16_t* di; uint32_t* dout; uint32_t o[192];
    (void)hipMalloc(&di, 512); (void)hipMalloc(&dout, 768);
    (void)hipMemcpy(di, h, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    (void)hipMemcpy(o, dout, 768, hipMemcpyDeviceToHost);
    int bad_a = 0, bad_b_keep = 0, bad_b_zero = 0, bad_c = 0;
    for (int i = 0; i < 64; ++i) {
        if (o[i] != ((uint32_t)h[i] << 16)) ++bad_a;
        if (o[64 + i] != (((uint32_t)h[64 + i] << 16) | 0xbeefu)) ++bad_b_keep;
        if (o[64 + i] != ((uint32_t)h[64 + i] << 16)) ++bad_b_zero;
        if (o[128 + i] != ((uint32_t)h[192 + i] << 16)) ++bad_c;
    }
    printf("zeroed reg: %d bad; dirty reg: low half preserved? %s, zeroed? %s (sample %08x); reload into same reg: %d bad\n", bad_a,
           bad_b_keep == 0 ? "yes" : "no", bad_b_zero == 0 ? "yes" : "no", o[64], bad_c);
    return bad_a || bad_c;
}
