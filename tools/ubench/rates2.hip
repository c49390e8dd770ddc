// VALU issue-rate microbenchmark with inline asm (no compiler folding): 8 independent chains / thread.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32;
#define ITERS 2048
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define DEFK(NAME, ASM)                                                              \
  __global__ void __launch_bounds__(256) NAME(u32* out, u32 seed) {                  \
    u32 x0, x1, x2, x3, x4, x5, x6, x7;                                              \
    u32 t = threadIdx.x + blockIdx.x * blockDim.x;                                   \
    x0 = t; x1 = t + 1; x2 = t + 2; x3 = t + 3; x4 = t + 4; x5 = t + 5; x6 = t + 6; x7 = t + 7; \
    u32 b = seed | 1, c = seed * 3 + 7;                                              \
    for (int it = 0; it < ITERS; ++it) {                                             \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)         \
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
                     : "v"(b), "v"(c));                                              \
      }                                                                              \
    }                                                                                \
    out[t] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;                                  \
  }
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define A_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x6a\n"
#define A_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
#define A_BFE(i) "v_bfe_i32 %" #i ", %" #i ", 3, 1\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define A_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define A_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 1, %8\n"
#define A_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n"
DEFK(k_xor, A_XOR) DEFK(k_and, A_AND) DEFK(k_add, A_ADD) DEFK(k_lshl, A_LSHL) DEFK(k_bitop3, A_BITOP3) DEFK(k_align, A_ALIGN)
DEFK(k_bfe, A_BFE) DEFK(k_mullo, A_MULLO) DEFK(k_mul24, A_MUL24) DEFK(k_perm, A_PERM) DEFK(k_andor, A_ANDOR)
DEFK(k_lshlor, A_LSHLOR) DEFK(k_xad, A_XAD) DEFK(k_cnd, A_CNDMASK) DEFK(k_fma, A_FMA) DEFK(k_or3, A_OR3) DEFK(k_add3, A_ADD3) DEFK(k_mov, A_MOV)
__global__ void __launch_bounds__(256) k_mad64(u32* out, u32 seed) {
  uint64_t x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3; u32 b = seed | 1, c = seed * 3 + 7;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\nv_mad_u64_u32 %1, vcc, %4, %5, %1\nv_mad_u64_u32 %2, vcc, %4, %5, %2\nv_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c) : "vcc");
    }
  }
  out[threadIdx.x + blockIdx.x * blockDim.x] = (u32)(x0 ^ x1 ^ x2 ^ x3);
}
// 64-bit forms: 4 independent 64-bit chains per thread
#define DEFK64(NAME, ASM)                                                            \
  __global__ void __launch_bounds__(256) NAME(u32* out, u32 seed) {                  \
    uint64_t x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, b = seed | 1;                 \
    u32 c = seed * 3 + 7;                                                            \
    for (int it = 0; it < ITERS; ++it) {                                             \
      _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c)); \
      }                                                                              \
    }                                                                                \
    out[threadIdx.x + blockIdx.x * blockDim.x] = (u32)(x0 ^ x1 ^ x2 ^ x3);           \
  }
#define A_LSHL64(i) "v_lshlrev_b64 %" #i ", 3, %" #i "\n"
#define A_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 3, %4\n"
#define A_LSHRADD32(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n"
#define A_ADDLSHL32(i) "v_add_lshl_u32 %" #i ", %" #i ", %8, 3\n"
#define A_PKLSHL16(i) "v_pk_lshlrev_b16 %" #i ", 3, %" #i "\n"
#define A_PKADD16(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define A_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define A_ALIGNBYTE(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, 1\n"
#define A_BFI(i) "v_bfi_b32 %" #i ", %" #i ", %8, %9\n"
#define A_XOR3B(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n"
DEFK64(k_lshl64, A_LSHL64) DEFK64(k_lshladd64, A_LSHLADD64)
DEFK(k_lshladd32, A_LSHRADD32) DEFK(k_addlshl32, A_ADDLSHL32) DEFK(k_pklshl16, A_PKLSHL16) DEFK(k_pkadd16, A_PKADD16) DEFK(k_lshr, A_LSHR)
DEFK(k_alignbyte, A_ALIGNBYTE) DEFK(k_bfi, A_BFI) DEFK(k_xor3b, A_XOR3B)
template <class K> void run(const char* name, K kern, int w, double ops_per_thread) {
  int blocks = 256 * w;
  u32* d; hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, 256>>>(d, 12345); hipDeviceSynchronize();
  hipEventRecord(e0); kern<<<blocks, 256>>>(d, 12345); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 256 * ops_per_thread;
  printf("%-16s w/SIMD=%d %8.3f ms %7.2f T lane-ops/s  cycles/wave-instr/SIMD=%.2f\n", name, w, ms, ops / ms / 1e9,
         (ms * 1e-3 * 2.4e9) / (ops_per_thread * w));
  hipFree(d);
}
int main() {
  double n = (double)ITERS * 4 * 8;
  for (int w : {2, 4}) {
    run("v_xor_b32", k_xor, w, n); run("v_and_b32", k_and, w, n); run("v_add_u32", k_add, w, n); run("v_lshlrev_b32", k_lshl, w, n);
    run("v_mov_b32", k_mov, w, n);
    run("v_bitop3_b32", k_bitop3, w, n); run("v_alignbit_b32", k_align, w, n); run("v_bfe_i32", k_bfe, w, n);
    run("v_perm_b32", k_perm, w, n); run("v_and_or_b32", k_andor, w, n); run("v_lshl_or_b32", k_lshlor, w, n); run("v_xad_u32", k_xad, w, n);
    run("v_or3_b32", k_or3, w, n); run("v_add3_u32", k_add3, w, n);
    run("v_cndmask_b32", k_cnd, w, n); run("v_mul_lo_u32", k_mullo, w, n); run("v_mul_u32_u24", k_mul24, w, n);
    run("v_fma_f32", k_fma, w, n); run("v_mad_u64_u32", k_mad64, w, (double)ITERS * 8 * 4);
    run("v_lshlrev_b64", k_lshl64, w, (double)ITERS * 8 * 4); run("v_lshl_add_u64", k_lshladd64, w, (double)ITERS * 8 * 4);
    run("v_lshl_add_u32", k_lshladd32, w, n); run("v_add_lshl_u32", k_addlshl32, w, n); run("v_pk_lshlrev_b16", k_pklshl16, w, n);
    run("v_pk_add_u16", k_pkadd16, w, n); run("v_lshrrev_b32", k_lshr, w, n); run("v_alignbyte_b32", k_alignbyte, w, n);
    run("v_bfi_b32", k_bfi, w, n); run("v_bitop3 xor3", k_xor3b, w, n);
    printf("\n");
  }
}
