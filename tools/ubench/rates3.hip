// Shift-family issue rates with NON-DEGENERATE operands (rates2.hip shifts a value by itself until it is zero): 8 independent destination
// registers per thread, sources are loop-invariant registers.  Question: is there a full-rate funnel shift on gfx950?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32;
#define ITERS 2048
#define DEFK(NAME, ASM)                                                              \
  __global__ void __launch_bounds__(256) NAME(u32* out, u32 seed) {                  \
    u32 x0 = 0, x1 = 0, x2 = 0, x3 = 0, x4 = 0, x5 = 0, x6 = 0, x7 = 0;              \
    u32 t = threadIdx.x + blockIdx.x * blockDim.x;                                   \
    u32 b = (seed * 2654435761u) ^ t, c = seed * 3 + 7 + t;                          \
    for (int it = 0; it < ITERS; ++it) {                                             \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)         \
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
                     : "v"(b), "v"(c));                                              \
      }                                                                              \
    }                                                                                \
    out[t] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;                                  \
  }
#define A_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %8\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %8\n"
#define A_ASHR(i) "v_ashrrev_i32 %" #i ", 3, %8\n"
#define A_ALIGN(i) "v_alignbit_b32 %" #i ", %8, %9, 29\n"
#define A_ALIGNV(i) "v_alignbit_b32 %" #i ", %8, %9, %" #i "\n"
#define A_BFEU(i) "v_bfe_u32 %" #i ", %8, 3, 29\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %8, %9\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %8, 3, %9\n"
#define A_LSHLOR(i) "v_lshl_or_b32 %" #i ", %8, 3, %9\n"
#define A_ANDOR(i) "v_and_or_b32 %" #i ", %8, %9, %" #i "\n"
#define A_MULK(i) "v_mul_lo_u32 %" #i ", %8, 8\n"
#define A_MUL24K(i) "v_mul_u32_u24 %" #i ", %8, 8\n"
#define A_LSHRV(i) "v_lshrrev_b32 %" #i ", %9, %8\n"
#define A_LSHLV(i) "v_lshlrev_b32 %" #i ", %9, %8\n"
DEFK(k_lshr, A_LSHR) DEFK(k_lshl, A_LSHL) DEFK(k_ashr, A_ASHR) DEFK(k_align, A_ALIGN) DEFK(k_alignv, A_ALIGNV) DEFK(k_bfeu, A_BFEU) DEFK(k_xor, A_XOR)
DEFK(k_lshladd, A_LSHLADD) DEFK(k_lshlor, A_LSHLOR) DEFK(k_andor, A_ANDOR) DEFK(k_mulk, A_MULK) DEFK(k_mul24k, A_MUL24K) DEFK(k_lshrv, A_LSHRV) DEFK(k_lshlv, A_LSHLV)
#define DEFK64(NAME, ASM)                                                            \
  __global__ void __launch_bounds__(256) NAME(u32* out, u32 seed) {                  \
    uint64_t x0 = 0, x1 = 0, x2 = 0, x3 = 0;                                         \
    u32 t = threadIdx.x + blockIdx.x * blockDim.x;                                   \
    uint64_t b = ((uint64_t)(seed * 2654435761u) << 32) | (t * 40503u + 11);         \
    u32 c = seed * 3 + 7;                                                            \
    for (int it = 0; it < ITERS; ++it) {                                             \
      _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c)); \
      }                                                                              \
    }                                                                                \
    out[t] = (u32)(x0 ^ x1 ^ x2 ^ x3) ^ (u32)((x0 ^ x1 ^ x2 ^ x3) >> 32);            \
  }
#define A_LSHR64(i) "v_lshrrev_b64 %" #i ", 3, %4\n"
#define A_LSHL64(i) "v_lshlrev_b64 %" #i ", 3, %4\n"
#define A_ASHR64(i) "v_ashrrev_i64 %" #i ", 3, %4\n"
DEFK64(k_lshr64, A_LSHR64) DEFK64(k_lshl64, A_LSHL64) DEFK64(k_ashr64, A_ASHR64)
template <class K> void run(const char* name, K kern, int w, double ops_per_thread) {
  int blocks = 256 * w;
  u32* d; hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, 256>>>(d, 12345); hipDeviceSynchronize();
  hipEventRecord(e0); kern<<<blocks, 256>>>(d, 12345); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 256 * ops_per_thread;
  printf("%-24s w/SIMD=%d %8.3f ms %7.2f T lane-ops/s  cycles/wave-instr/SIMD(at 2.4 GHz)=%.2f\n", name, w, ms, ops / ms / 1e9, (ms * 1e-3 * 2.4e9) / (ops_per_thread * w));
  hipFree(d);
}
int main() {
  double n = (double)ITERS * 4 * 8, n64 = (double)ITERS * 8 * 4;
  for (int w : {4}) {
    run("v_xor_b32", k_xor, w, n); run("v_lshrrev_b32 imm", k_lshr, w, n); run("v_lshlrev_b32 imm", k_lshl, w, n); run("v_ashrrev_i32 imm", k_ashr, w, n);
    run("v_lshrrev_b32 vgpr amt", k_lshrv, w, n); run("v_lshlrev_b32 vgpr amt", k_lshlv, w, n);
    run("v_alignbit_b32 imm", k_align, w, n); run("v_alignbit_b32 vgpr amt", k_alignv, w, n); run("v_bfe_u32", k_bfeu, w, n);
    run("v_lshl_add_u32", k_lshladd, w, n); run("v_lshl_or_b32", k_lshlor, w, n); run("v_and_or_b32", k_andor, w, n);
    run("v_mul_lo_u32 x8", k_mulk, w, n); run("v_mul_u32_u24 x8", k_mul24k, w, n);
    run("v_lshrrev_b64", k_lshr64, w, n64); run("v_lshlrev_b64", k_lshl64, w, n64); run("v_ashrrev_i64", k_ashr64, w, n64);
  }
}
