// Cycles of the cross-unit dependency hops a latency-bound one-wave kernel is made of (gfx950, one wave alone on its SIMD):
// VALU -> VALU, v_cmp -> VCC -> v_cndmask, v_cmp -> SGPR -> SALU -> v_cndmask, LDS write -> read, ds_bpermute, readlane, a taken
// branch.  Each pattern is 64 dependent repetitions inside one asm block, timed with s_memtime.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define TIC(t) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
__global__ __launch_bounds__(64) void k(unsigned long long* out, uint32_t* sink, uint32_t seed) {
    __shared__ uint32_t lds[256];
    uint32_t a = threadIdx.x * seed, b = seed ^ 0x1234567u, c = threadIdx.x + 7, d = seed * 3;
    unsigned long long m = seed, m2 = ~0ull;
    lds[threadIdx.x] = a; lds[threadIdx.x + 64] = b; lds[threadIdx.x + 128] = c; lds[threadIdx.x + 192] = d;
    __syncthreads();
    unsigned long long t0, t1; int p = 0;
    uint32_t la = threadIdx.x * 4;
#define PAT(...) TIC(t0); asm volatile(__VA_ARGS__); TIC(t1); if (threadIdx.x == 0) out[p] = t1 - t0; ++p;
    // 0: empty
    PAT("" :::);
    // 1: dependent VALU chain
    PAT(".rept 64\n v_add_u32 %0, %0, %1\n .endr" : "+v"(a) : "v"(b));
    // 2: independent VALU (4 chains)
    PAT(".rept 16\n v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n .endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(la));
    // 3: v_cmp -> vcc -> v_cndmask -> v_cmp ...
    PAT(".rept 64\n v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %2, %3, vcc\n .endr" : "+v"(a) : "v"(b), "v"(c), "v"(d) : "vcc");
    // 4: v_cmp -> sgpr pair -> v_cndmask_e64
    PAT(".rept 64\n v_cmp_gt_u32 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %2, %3, s[20:21]\n .endr" : "+v"(a) : "v"(b), "v"(c), "v"(d) : "s20", "s21");
    // 5: v_cmp -> sgpr -> 1 SALU -> v_cndmask
    PAT(".rept 64\n v_cmp_gt_u32 s[20:21], %0, %1\n s_and_b64 s[20:21], s[20:21], %4\n v_cndmask_b32_e64 %0, %2, %3, s[20:21]\n .endr" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(m2) : "s20", "s21", "scc");
    // 6: v_cmp -> sgpr -> 3 SALU -> v_cndmask
    PAT(".rept 64\n v_cmp_gt_u32 s[20:21], %0, %1\n s_and_b64 s[20:21], s[20:21], %4\n s_andn2_b64 s[20:21], s[20:21], %5\n s_or_b64 s[20:21], s[20:21], %5\n v_cndmask_b32_e64 %0, %2, %3, s[20:21]\n .endr" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(m2), "s"(m) : "s20", "s21", "scc");
    // 7: dependent SALU chain
    PAT(".rept 64\n s_add_u32 s20, s20, %0\n .endr" :: "s"(seed) : "s20", "scc");
    // 8: ds_write -> ds_read -> wait -> dependent VALU
    PAT(".rept 64\n ds_write_b32 %1, %0\n ds_read_b32 %0, %1 offset:256\n s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, %2\n .endr" : "+v"(a) : "v"(la), "v"(b) : "memory");
    // 9: ds_read only -> wait -> VALU
    PAT(".rept 64\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_and_b32 %1, 0xfc, %0\n .endr" : "+v"(a), "+v"(la) :: "memory");
    la = threadIdx.x * 4;
    // 10: ds_bpermute -> wait -> VALU
    PAT(".rept 64\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, %2\n .endr" : "+v"(a) : "v"(la), "v"(b) : "memory");
    // 11: readlane -> SALU -> VALU using the SGPR
    PAT(".rept 64\n v_readlane_b32 s20, %0, 0\n s_lshr_b32 s20, s20, 1\n v_add_u32 %0, s20, %0\n .endr" : "+v"(a) :: "s20", "scc");
    // 12: taken branches
    PAT("s_mov_b32 s20, 64\n 1:\n s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" ::: "s20", "scc");
    // 13: v_cmp_sdwa -> sgpr -> v_cndmask
    PAT(".rept 64\n v_cmp_gt_u32_sdwa s[20:21], %0, %1 src0_sel:WORD_1 src1_sel:WORD_1\n v_cndmask_b32_e64 %0, %2, %3, s[20:21]\n .endr" : "+v"(a) : "v"(b), "v"(c), "v"(d) : "s20", "s21");
    // 14: v_and with SGPR operand written by v_cmp (VALU -> SGPR -> VALU as data), then v_cmp_eq_u64
    PAT(".rept 64\n v_cmp_gt_u32 s[20:21], %0, %1\n v_and_b32 %0, s20, %2\n .endr" : "+v"(a) : "v"(b), "v"(c) : "s20", "s21");
    // 15: DPP move dependent chain
    PAT(".rept 64\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n .endr" : "+v"(a));
    // 16: s_nop 0 x64
    PAT(".rept 64\n s_nop 0\n .endr" :::);
    // 17: v_cmp -> vcc -> s_and vcc -> cndmask vcc
    PAT(".rept 64\n v_cmp_gt_u32 vcc, %0, %1\n s_and_b64 vcc, vcc, %4\n v_cndmask_b32 %0, %2, %3, vcc\n .endr" : "+v"(a) : "v"(b), "v"(c), "v"(d), "s"(m2) : "vcc", "scc");
    // 18: ds_write + ds_read, 4 independent VALU before the wait
    PAT(".rept 64\n ds_write_b32 %1, %0\n ds_read_b32 %0, %1 offset:256\n v_add_u32 %3, %3, %2\n v_add_u32 %3, %3, %2\n v_add_u32 %3, %3, %2\n v_add_u32 %3, %3, %2\n s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, %2\n .endr" : "+v"(a) : "v"(la), "v"(b), "v"(c) : "memory");
    // 19: v_permlane32_swap chain
    PAT(".rept 64\n v_permlane32_swap_b32 %0, %1\n .endr" : "+v"(a), "+v"(b));
    // 20: v_readlane with SGPR lane select -> v_writelane back
    PAT(".rept 64\n v_readlane_b32 s20, %0, 5\n s_nop 3\n v_writelane_b32 %0, s20, 9\n .endr" : "+v"(a) :: "s20");
    sink[threadIdx.x] = a + b + c + d + la;
}
int main() {
    unsigned long long* d; uint32_t* s;
    (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&s, 64 * 4);
    const char* names[] = {"empty", "dependent v_add chain", "4 independent v_add chains (per instr)", "v_cmp->vcc->v_cndmask (pair)", "v_cmp->sgpr->v_cndmask_e64 (pair)",
        "v_cmp->sgpr->s_and->v_cndmask (triple)", "v_cmp->sgpr->3 SALU->v_cndmask (5)", "dependent s_add chain", "ds_write->ds_read->wait->v_add",
        "ds_read->wait->v_and (dependent address)", "ds_bpermute->wait->v_add", "v_readlane->s_lshr->v_add", "taken branch loop (3 instr)", "v_cmp_sdwa->sgpr->v_cndmask (pair)",
        "v_cmp->sgpr->v_and(sgpr data) (pair)", "dependent DPP mov chain", "s_nop 0", "v_cmp->vcc->s_and vcc->v_cndmask (triple)", "ds_write->ds_read + 4 v_add->wait->v_add", "v_permlane32_swap chain", "readlane->s_nop 3->writelane"};
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s, 12345u + rep);
        (void)hipDeviceSynchronize();
    }
    unsigned long long h[64];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 21; ++i) printf("%2d %-50s %6llu cycles / 64 = %.1f (minus empty: %.1f)\n", i, names[i], h[i], h[i] / 64.0, (double)(h[i] - h[0]) / 64.0);
    return 0;
}
