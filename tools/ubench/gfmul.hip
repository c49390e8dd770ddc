// gf_mul variants microbenchmark: throughput of back-to-back multiplications, many waves.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../dv-pari_amd/csrc/gf233.cuh"
using namespace dvp;

// variant B: accumulator shift by a full-rate add/addc carry chain
__device__ __forceinline__ void shl1_15(uint32_t* a) {
  asm volatile(
      "v_add_co_u32 %0, vcc, %0, %0\n"
      "v_addc_co_u32 %1, vcc, %1, %1, vcc\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n"
      "v_addc_co_u32 %4, vcc, %4, %4, vcc\n v_addc_co_u32 %5, vcc, %5, %5, vcc\n v_addc_co_u32 %6, vcc, %6, %6, vcc\n"
      "v_addc_co_u32 %7, vcc, %7, %7, vcc\n v_addc_co_u32 %8, vcc, %8, %8, vcc\n v_addc_co_u32 %9, vcc, %9, %9, vcc\n"
      "v_addc_co_u32 %10, vcc, %10, %10, vcc\n v_addc_co_u32 %11, vcc, %11, %11, vcc\n v_addc_co_u32 %12, vcc, %12, %12, vcc\n"
      "v_addc_co_u32 %13, vcc, %13, %13, vcc\n v_addc_co_u32 %14, vcc, %14, %14, vcc\n"
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
        "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14])
      :
      : "vcc");
}
__device__ __forceinline__ Gf gf_mul_B(const Gf& a, const Gf& b) {
  uint32_t acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0;
#pragma unroll 1
  for (int k = 31; k >= 9; --k) {
    shl1_15(acc);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)a.w[j], k, 1);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i + j] = gf_andxor(m, b.w[i], acc[i + j]);
    }
  }
#pragma unroll 1
  for (int k = 8; k >= 0; --k) {
    shl1_15(acc);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)a.w[j], k, 1);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i + j < 15) acc[i + j] = gf_andxor(m, b.w[i], acc[i + j]);
    }
  }
  return gf_reduce16(acc);
}
// variant C: B + masks by shifting a copy of a left (add = full rate) and taking the sign with ashr (half rate) -> same as bfe; skip.

template <int V> __global__ void __launch_bounds__(256) k(Gf* out, int reps) {
  uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  Gf x, y;
  for (int i = 0; i < 8; ++i) { x.w[i] = t * 2654435761u + i; y.w[i] = t * 40503u + 7 * i; }
  x.w[7] &= 0x1ff; y.w[7] &= 0x1ff;
  for (int r = 0; r < reps; ++r) {
    if (V == 0) x = gf_mul(x, y); else if (V == 1) x = gf_mul_B(x, y); else x = gf_sqr(x);
    y.w[0] ^= x.w[3];
  }
  out[t] = x;
}
template <int V> void run(const char* name, int w, int reps) {
  int blocks = 256 * w; Gf* d; hipMalloc(&d, blocks * 256 * sizeof(Gf));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<V><<<blocks, 256>>>(d, reps); hipDeviceSynchronize();
  hipEventRecord(e0); k<V><<<blocks, 256>>>(d, reps); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double muls = (double)blocks * 256 * reps;
  printf("%-10s w/SIMD=%d %8.3f ms  %7.2f G ops/s   cycles per op per wave-on-SIMD: %.0f\n", name, w, ms, muls / ms / 1e6, ms * 1e-3 * 2.4e9 / (reps * (double)w));
  hipFree(d);
}
__global__ void check(int* bad) {
  Gf x, y; uint32_t t = threadIdx.x;
  for (int i = 0; i < 8; ++i) { x.w[i] = t * 2654435761u + i * 97; y.w[i] = t * 40503u + 7 * i; }
  x.w[7] &= 0x1ff; y.w[7] &= 0x1ff;
  if (!gf_eq(gf_mul(x, y), gf_mul_B(x, y))) atomicAdd(bad, 1);
}
int main() {
  int* bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4); check<<<1, 256>>>(bad); int h; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost); printf("mismatches: %d\n", h);
  for (int w : {2, 4, 8}) { run<0>("mul_A", w, 200); run<1>("mul_B", w, 200); run<2>("sqr", w, 2000); }
}
