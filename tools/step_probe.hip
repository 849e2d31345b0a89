// Where do the cycles of WaveHeapL's step go?  The step's instructions, 32 repetitions in a row, on fixed registers; groups
// left out one at a time.  (One wave alone on its SIMD; operands are arbitrary: only the dependency structure matters.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define TIC(t) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
#define A_BLK "v_cmp_gt_u32_sdwa vcc, v11, v10 src0_sel:WORD_1 src1_sel:WORD_1\n v_cmp_gt_u32_e64 s[20:21], v12, s40\n v_cmp_gt_u32_e64 s[22:23], v13, s40\n v_and_b32_e32 v20, vcc_lo, v15\n s_mov_b64 s[24:25], vcc\n v_mov_b32_e32 v21, s41\n v_cmp_eq_u32_e64 s[26:27], v20, v16\n v_cndmask_b32_e64 v22, v13, v21, s[22:23]\n"
#define B_PRE "v_cmp_gt_u32_e64 vcc, v14, s40\n s_or_b32 s42, s43, 0xffff\n s_min_u32 s42, s42, s40\n s_cmp_gt_u32 s41, s44\n s_cselect_b32 s42, s42, s44\n s_cselect_b64 s[28:29], s[26:27], 0\n s_andn2_b64 s[28:29], s[28:29], s[20:21]\n s_andn2_b64 s[30:31], s[28:29], s[22:23]\n"
#define B_MOV "v_cndmask_b32_e32 v23, v14, v21, vcc\n v_cndmask_b32_e64 v23, v13, v23, s[30:31]\n v_cndmask_b32_e64 v10, v10, v23, s[24:25]\n v_cndmask_b32_e64 v11, v23, v11, s[24:25]\n v_cmp_gt_u32_sdwa s[36:37], v11, v10 src0_sel:WORD_1 src1_sel:WORD_1\n v_cndmask_b32_e64 v12, v12, v22, s[28:29]\n s_nop 0\n v_cndmask_b32_e64 v13, v11, v10, s[36:37]\n v_cndmask_b32_e64 v24, v18, v17, s[36:37]\n"
#define LDSX "ds_write_b32 v19, v13\n ds_read_b32 v14, v24\n"
#define RDL "v_readlane_b32 s43, v13, 0\n"
#define WAIT "s_waitcnt lgkmcnt(0)\n"
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s36","s37","s40","s41","s42","s43","s44","vcc","scc","memory"
__global__ __launch_bounds__(64) void k(unsigned long long* out, uint32_t seed) {
    __shared__ uint32_t lds[512];
    lds[threadIdx.x] = seed * threadIdx.x; lds[threadIdx.x + 64] = seed; lds[threadIdx.x + 128] = 3; lds[threadIdx.x + 192] = 5;
    __syncthreads();
    unsigned long long t0, t1; int p = 0;
    asm volatile("v_mov_b32 v10, %0\n v_mov_b32 v11, %1\n v_mov_b32 v12, %0\n v_mov_b32 v13, %1\n v_mov_b32 v14, %0\n v_mov_b32 v15, %1\n v_mov_b32 v16, 0\n"
                 "v_lshlrev_b32 v17, 2, %2\n v_lshlrev_b32 v18, 2, %2\n v_lshlrev_b32 v19, 2, %2\n s_mov_b32 s40, %3\n s_mov_b32 s41, %3\n s_mov_b32 s43, %3\n s_mov_b32 s44, 0\n"
                 :: "v"(seed * (threadIdx.x + 1)), "v"(seed ^ threadIdx.x), "v"(threadIdx.x), "s"(seed) : CLOB);
#define PAT(body) TIC(t0); asm volatile(".rept 32\n" body ".endr" ::: CLOB); TIC(t1); if (threadIdx.x == 0) out[p] = t1 - t0; ++p;
    PAT("");
    PAT(A_BLK);
    PAT(B_PRE B_MOV);
    PAT(A_BLK B_PRE B_MOV);
    PAT(A_BLK WAIT B_PRE B_MOV LDSX);
    PAT(A_BLK WAIT B_PRE B_MOV LDSX RDL);
    PAT(A_BLK B_MOV);
    PAT(B_PRE);
    PAT(B_MOV);
    PAT(A_BLK WAIT B_PRE B_MOV LDSX RDL "s_add_u32 s44, s44, 1\n s_cmp_lg_u32 s44, 0\n s_cbranch_scc0 1f\n 1:\n");
    PAT(A_BLK WAIT B_PRE B_MOV LDSX RDL "s_add_u32 s44, s44, 1\n s_cmp_lg_u32 s44, 0\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n");
}
int main() {
    unsigned long long* d; (void)hipMalloc(&d, 64 * 8);
    const char* names[] = {"empty", "block A (8)", "block B (17)", "A + B (25)", "A + wait + B + LDS exchange (27)", "... + v_readlane of the root's child (28)", "A + moves only (17)",
                           "B's scalar prefix (8)", "B's moves (9)", "whole + not-taken branch (31)", "whole + taken forward branch (31)"};
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 12345u + rep); (void)hipDeviceSynchronize(); }
    unsigned long long h[64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 11; ++i) printf("%2d %-48s %7llu cycles / 32 = %.1f\n", i, names[i], h[i], (double)(h[i] - h[0]) / 32.0);
    return 0;
}
