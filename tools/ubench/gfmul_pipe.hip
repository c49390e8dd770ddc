// LDS-comb product with the table reads of row k-1 issued before the xors of row k (software pipelining): does it beat
// the row-at-a-time form at 8 waves/CU?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../dv-pari_amd/csrc/gf233.cuh"
using namespace dvp;
typedef uint32_t u32;
struct RowBuf { gf_u32x4 lo[8], hi[8]; };
template <int NW>
__device__ __forceinline__ void row_load(RowBuf& b, const Gf& a, const GfLds& c, int rsh, int lsh) {
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    u32 sh = (a.w[j] >> rsh) << lsh;
    u32 addr = (sh & 0x1C00u) | c.lane_base;
    b.lo[j] = *(const gf_u32x4*)(c.lds + addr);
    b.hi[j] = *(const gf_u32x4*)(c.lds + addr + 8192);
  }
}
template <int NW>
__device__ __forceinline__ void row_xor(u32* acc, const RowBuf& b) {
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    acc[j + 0] ^= b.lo[j].x; acc[j + 1] ^= b.lo[j].y; acc[j + 2] ^= b.lo[j].z; acc[j + 3] ^= b.lo[j].w;
    acc[j + 4] ^= b.hi[j].x; acc[j + 5] ^= b.hi[j].y; acc[j + 6] ^= b.hi[j].z;
    if (j + 7 < 15) acc[j + 7] ^= b.hi[j].w;
  }
}
#define BAR asm volatile("" ::: "memory")
__device__ __forceinline__ Gf gf_mul_tab_pipe(const Gf& a, const GfLds& c) {
  u32 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0;
  RowBuf A, B;
  row_load<7>(A, a, c, 20, 0);            // k = 10
  row_load<7>(B, a, c, 17, 0); BAR;       // k = 9
  row_xor<7>(acc, A);
  row_load<7>(A, a, c, 14, 0); BAR;       // k = 8
  gf_acc_shl3(acc); row_xor<7>(acc, B);
  row_load<7>(B, a, c, 11, 0); BAR;       // 7
  gf_acc_shl3(acc); row_xor<7>(acc, A);
  row_load<7>(A, a, c, 8, 0); BAR;        // 6
  gf_acc_shl3(acc); row_xor<7>(acc, B);
  row_load<7>(B, a, c, 5, 0); BAR;        // 5
  gf_acc_shl3(acc); row_xor<7>(acc, A);
  row_load<7>(A, a, c, 2, 0); BAR;        // 4
  gf_acc_shl3(acc); row_xor<7>(acc, B);
  row_load<7>(B, a, c, 0, 1); BAR;        // 3
  gf_acc_shl3(acc); row_xor<7>(acc, A);
  row_load<8>(A, a, c, 0, 4); BAR;        // 2
  gf_acc_shl3(acc); row_xor<7>(acc, B);
  row_load<8>(B, a, c, 0, 7); BAR;        // 1
  gf_acc_shl3(acc); row_xor<8>(acc, A);
  row_load<8>(A, a, c, 0, 10); BAR;       // 0
  gf_acc_shl3(acc); row_xor<8>(acc, B);
  gf_acc_shl3(acc); row_xor<8>(acc, A);
  return gf_reduce16(acc);
}
template <int V>
__global__ void __launch_bounds__(256) k_mul(Gf* out, int reps) {
  extern __shared__ char lds[];
  GfLds L = gf_lds_init(lds);
  u32 t = threadIdx.x + blockIdx.x * blockDim.x;
  Gf x, y;
  for (int i = 0; i < 8; ++i) { x.w[i] = t * 2654435761u + i; y.w[i] = t * 40503u + 7 * i; }
  x.w[7] &= 0x1ff; y.w[7] &= 0x1ff;
  for (int r = 0; r < reps; ++r) {
    gf_tab_build(L, y);
    x = V ? gf_mul_tab_pipe(x, L) : gf_mul_tab(x, L);
    y.w[0] ^= x.w[3];
  }
  out[t] = x;
}
int main() {
  Gf* d; hipMalloc(&d, (size_t)2560 * 256 * sizeof(Gf));
  Gf *h0 = new Gf[256], *h1 = new Gf[256];
  for (int v = 0; v < 2; ++v) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
      hipEventRecord(e0);
      if (v) k_mul<1><<<2560, 256, 65536>>>(d, 300); else k_mul<0><<<2560, 256, 65536>>>(d, 300);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipMemcpy(v ? h1 : h0, d, 256 * sizeof(Gf), hipMemcpyDeviceToHost);
    printf("%s: %7.2f G mul/s\n", v ? "pipelined  " : "row-at-time", 2560.0 * 256 * 300 / best / 1e6);
  }
  int bad = 0; for (int i = 0; i < 256; ++i) for (int k = 0; k < 8; ++k) bad += h0[i].w[k] != h1[i].w[k];
  printf("mismatches %d\n", bad);
}
