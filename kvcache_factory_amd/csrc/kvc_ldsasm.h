// kvc_ldsasm.h — hand-issued LDS operand loads for the exact f32-MFMA chains (kvc_score.hip, kvc_h2o.hip).
//
// On gfx950 the f32-input MFMA does not overlap VALU work on its SIMD (tools/mfma_valu_probe.hip), so the exact scans
// fetch their operands with loads that need no VALU instruction at all: the fp32 A fragment by ds_read_b128, the bf16 B
// elements by ds_read_u16_d16_hi, which puts the 16 bits in the high half of the VGPR and clears the low half
// (tools/d16_probe.hip) — an exact bf16 -> fp32 widening inside the load — issued one MFMA step ahead of their use.
//
// Such loads are invisible to the compiler's s_waitcnt bookkeeping (cdna_hip_programming.md 5.7 item 1): for hipcc the
// destinations are written when the statement ends; the data lands later.  What keeps that safe, as properties of the
// SOURCE:
//   (1) every destination is EARLY-CLOBBER ("=&v").  A statement holds several loads that all address through the same
//       input register; with plain "=v" hipcc may give a destination the address register itself once that is dead after
//       the statement — round 1's H2O variant compiled to `ds_read_u16_d16_hi v18, v18 offset:0xf0` followed by three
//       more loads addressed through v18: when the first load's data returns before the others have issued (a wave
//       that loses its issue slot for one LDS latency) they read through garbage.  That is the "rare, run-to-run
//       different" error recorded in round 1; tools/asm_audit.py shows it in the listing ("destination overlaps its
//       address register") and it is gone with "=&v".  -DKVC_DIAG_NO_EARLYCLOBBER rebuilds the hazard for the
//       demonstration in tools/h2o_d16_stress.py and is never set for the shipped library;
//   (2) the consumer is a wait statement that names every destination "+v": no MFMA can be scheduled above it, and it
//       prints the registers it retires so the build audit can match them with the loads';
//   (3) load and wait statements clobber "memory": the compiler's own LDS stores stay on their side of the reads.
// What the source cannot promise is register ALLOCATION — under pressure hipcc may still copy or spill a destination
// between load and wait, and such a copy would race with the LDS return.  So every build is audited:
// tools/asm_audit.py / tests/test_asm_audit.py disassemble the kernels and require that between each load and its
// wait there are only v_mfma / scalar instructions, no mention of a pending destination, matching register names and
// wait counts.
#pragma once
#include <stdint.h>

#include <type_traits>

namespace kvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#if defined(KVC_DIAG_NO_EARLYCLOBBER)
#define KVC_LD_OUT "=v"            /* diagnostic build only: reproduces round 1's hazard */
#else
#define KVC_LD_OUT "=&v"
#endif

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N)
template <int I0, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I0 < N) { f(std::integral_constant<int, I0>{}); static_for<I0 + 1, N>(f); }
}

__device__ __forceinline__ uint32_t lds_addr(const void* p) {       // byte address inside the workgroup's LDS
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

// MFMA step STI of the exact bf16 scan: A = chunk STI of the lane's fp32 fragment image (chunk-major, 1024 B apart),
// B = bf16 elements 8*STI + 2s + kh, s = 0..3, of the lane's key row (krow_a already holds + 2*kh).  5 LDS reads.
template <int STI> __device__ __forceinline__ void ld_step(f32x4& A, uint32_t (&B)[4], uint32_t arow_a, uint32_t krow_a) {
    asm volatile("ds_read_b128 %0, %5 offset:%7\n\t"
                 "ds_read_u16_d16_hi %1, %6 offset:%8\n\t"
                 "ds_read_u16_d16_hi %2, %6 offset:%9\n\t"
                 "ds_read_u16_d16_hi %3, %6 offset:%10\n\t"
                 "ds_read_u16_d16_hi %4, %6 offset:%11"
                 : KVC_LD_OUT(A), KVC_LD_OUT(B[0]), KVC_LD_OUT(B[1]), KVC_LD_OUT(B[2]), KVC_LD_OUT(B[3])
                 : "v"(arow_a), "v"(krow_a), "n"(STI * 1024), "n"(STI * 16), "n"(STI * 16 + 4), "n"(STI * 16 + 8), "n"(STI * 16 + 12)
                 : "memory");
}
// Wait until at most N LDS operations issued after this step's loads are outstanding (the LDS returns a wave's reads in
// order), i.e. until A and B have landed.  N = 5 x (younger ld_step statements in flight).
template <int N> __device__ __forceinline__ void wait_step(f32x4& A, uint32_t (&B)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%5) ; retire %0 %1 %2 %3 %4"
                 : "+v"(A), "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]) : "n"(N) : "memory");
}

// The same for a kernel that keeps its A fragment in registers (kvc_h2o.hip): B only, 4 LDS reads per step.
template <int STI> __device__ __forceinline__ void ld_b_step(uint32_t (&B)[4], uint32_t krow_a) {
    asm volatile("ds_read_u16_d16_hi %0, %4 offset:%5\n\t"
                 "ds_read_u16_d16_hi %1, %4 offset:%6\n\t"
                 "ds_read_u16_d16_hi %2, %4 offset:%7\n\t"
                 "ds_read_u16_d16_hi %3, %4 offset:%8"
                 : KVC_LD_OUT(B[0]), KVC_LD_OUT(B[1]), KVC_LD_OUT(B[2]), KVC_LD_OUT(B[3])
                 : "v"(krow_a), "n"(STI * 16), "n"(STI * 16 + 4), "n"(STI * 16 + 8), "n"(STI * 16 + 12)
                 : "memory");
}
template <int N> __device__ __forceinline__ void wait_b_step(uint32_t (&B)[4]) {   // N = 4 x (younger ld_b_step statements)
    asm volatile("s_waitcnt lgkmcnt(%4) ; retire %0 %1 %2 %3"
                 : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]) : "n"(N) : "memory");
}

}  // namespace kvc
