#include <hip/hip_runtime.h>
template <int MASK> __device__ __forceinline__ float xor_lane(float v) {
    const int x = __float_as_int(v);
    int r;
    if constexpr (MASK == 1) r = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);
    else if constexpr (MASK == 2) r = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);
    else if constexpr (MASK == 4) { r = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false); r = __builtin_amdgcn_update_dpp(r, x, 0x114, 0xF, 0xA, false); }
    else if constexpr (MASK == 8) { r = __builtin_amdgcn_update_dpp(x, x, 0x108, 0xF, 0x3, false); r = __builtin_amdgcn_update_dpp(r, x, 0x118, 0xF, 0xC, false); }
    else if constexpr (MASK == 16) { auto p = __builtin_amdgcn_permlane16_swap(x, x, false, false); r = (threadIdx.x & 16) ? p[0] : p[1]; }
    else { auto p = __builtin_amdgcn_permlane32_swap(x, x, false, false); r = (threadIdx.x & 32) ? p[0] : p[1]; }
    return __int_as_float(r);
}
__global__ void k(const float* in, float* out) {
    float v = in[threadIdx.x];
    out[threadIdx.x] = xor_lane<1>(v); out[64 + threadIdx.x] = xor_lane<2>(v); out[128 + threadIdx.x] = xor_lane<4>(v);
    out[192 + threadIdx.x] = xor_lane<8>(v); out[256 + threadIdx.x] = xor_lane<16>(v); out[320 + threadIdx.x] = xor_lane<32>(v);
}
int main() {
    float h[64], o[384]; for (int i = 0; i < 64; ++i) h[i] = (float)i;
    float *di, *dout; hipMalloc(&di, 256); hipMalloc(&dout, 1536);
    hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o, dout, 1536, hipMemcpyDeviceToHost);
    int masks[6] = {1, 2, 4, 8, 16, 32}; int bad = 0;
    for (int m = 0; m < 6; ++m) for (int i = 0; i < 64; ++i) if (o[m * 64 + i] != (float)(i ^ masks[m])) { if (bad < 10) printf("mask %d lane %d got %g\n", masks[m], i, o[m*64+i]); ++bad; }
    printf("xor_lane check: %d bad\n", bad);
    return bad != 0;
}
