// LDS-comb gf_mul prototype: 3-bit window table of u*b (u = 0..7) per lane in LDS, lane-interleaved layout
//   addr(h, u, lane) = ((h*8 + u)*64 + lane)*16   (h = word half 0/1), 16 KB per wave.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../dv-pari_amd/csrc/gf233.cuh"
using namespace dvp;
typedef uint32_t u32;


// variant X: fully unrolled digit positions (constant shifts), XOR3 merging of adjacent lookups
__device__ __forceinline__ u32 xor3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
template <int K> __device__ __forceinline__ u32 digit_addr(u32 w, u32 lane_base) {
  constexpr int sh = 3 * K - 10;
  u32 s = sh >= 0 ? (w >> (sh >= 0 ? sh : 0)) : (w << (sh < 0 ? -sh : 0));
  return (s & 0x1C00u) | lane_base;
}
template <int K, int NW> __device__ __forceinline__ void row_x(u32* acc, const Gf& a, const GfLds& c) {
  gf_u32x4 lo[NW], hi[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    u32 addr = digit_addr<K>(a.w[j], c.lane_base);
    lo[j] = *(const gf_u32x4*)(c.lds + addr);
    hi[j] = *(const gf_u32x4*)(c.lds + addr + 8192);
  }
  // word w of lookup j lands in acc[j + w]; merge lookups (j, j+1) with one 3-input xor where they overlap
#pragma unroll
  for (int j = 0; j + 1 < NW; j += 2) {
    const gf_u32x4 &l0 = lo[j], &h0 = hi[j], &l1 = lo[j + 1], &h1 = hi[j + 1];
    acc[j + 0] ^= l0.x;
    acc[j + 1] = xor3(acc[j + 1], l0.y, l1.x);
    acc[j + 2] = xor3(acc[j + 2], l0.z, l1.y);
    acc[j + 3] = xor3(acc[j + 3], l0.w, l1.z);
    acc[j + 4] = xor3(acc[j + 4], h0.x, l1.w);
    acc[j + 5] = xor3(acc[j + 5], h0.y, h1.x);
    acc[j + 6] = xor3(acc[j + 6], h0.z, h1.y);
    if (j + 7 < 15) acc[j + 7] = xor3(acc[j + 7], h0.w, h1.z);
    if (j + 8 < 15) acc[j + 8] ^= h1.w;
  }
  if (NW & 1) {
    const int j = NW - 1;
    acc[j + 0] ^= lo[j].x; acc[j + 1] ^= lo[j].y; acc[j + 2] ^= lo[j].z; acc[j + 3] ^= lo[j].w;
    acc[j + 4] ^= hi[j].x; acc[j + 5] ^= hi[j].y; acc[j + 6] ^= hi[j].z;
    if (j + 7 < 15) acc[j + 7] ^= hi[j].w;
  }
}
__device__ __forceinline__ Gf gf_mul_tab_x(const Gf& a, const GfLds& c) {
  u32 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0;
  row_x<10, 7>(acc, a, c);
  gf_acc_shl3(acc); row_x<9, 7>(acc, a, c);
  gf_acc_shl3(acc); row_x<8, 7>(acc, a, c);
  gf_acc_shl3(acc); row_x<7, 7>(acc, a, c);
  gf_acc_shl3(acc); row_x<6, 7>(acc, a, c);
  gf_acc_shl3(acc); row_x<5, 7>(acc, a, c);
  gf_acc_shl3(acc); row_x<4, 7>(acc, a, c);
  gf_acc_shl3(acc); row_x<3, 7>(acc, a, c);
  gf_acc_shl3(acc); row_x<2, 8>(acc, a, c);
  gf_acc_shl3(acc); row_x<1, 8>(acc, a, c);
  gf_acc_shl3(acc); row_x<0, 8>(acc, a, c);
  return gf_reduce16(acc);
}

// variant Y: rolled loops, all reads of a row issued before the xors (plain xor), inlined
template <int NW> __device__ __forceinline__ void row_y(u32* acc, const Gf& a, const GfLds& c, int rsh, int lsh) {
  gf_u32x4 lo[NW], hi[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    u32 sh = (a.w[j] >> rsh) << lsh;
    u32 addr = (sh & 0x1C00u) | c.lane_base;
    lo[j] = *(const gf_u32x4*)(c.lds + addr);
    hi[j] = *(const gf_u32x4*)(c.lds + addr + 8192);
  }
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    acc[j + 0] ^= lo[j].x; acc[j + 1] ^= lo[j].y; acc[j + 2] ^= lo[j].z; acc[j + 3] ^= lo[j].w;
    acc[j + 4] ^= hi[j].x; acc[j + 5] ^= hi[j].y; acc[j + 6] ^= hi[j].z;
    if (j + 7 < 15) acc[j + 7] ^= hi[j].w;
  }
}
__device__ __forceinline__ Gf gf_mul_tab_y(const Gf& a, const GfLds& c) {
  u32 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0;
  row_y<7>(acc, a, c, 20, 0);
#pragma unroll 1
  for (int k = 9; k >= 4; --k) { gf_acc_shl3(acc); row_y<7>(acc, a, c, 3 * k - 10, 0); }
  gf_acc_shl3(acc); row_y<7>(acc, a, c, 0, 1);
#pragma unroll 1
  for (int k = 2; k >= 0; --k) { gf_acc_shl3(acc); row_y<8>(acc, a, c, 0, 10 - 3 * k); }
  return gf_reduce16(acc);
}
template <int TPB> __global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(2, 2))) ky(Gf* out, int reps) {
  extern __shared__ char lds[];
  GfLds L = gf_lds_init(lds);
  u32 t = threadIdx.x + blockIdx.x * blockDim.x;
  Gf x, y;
  for (int i = 0; i < 8; ++i) { x.w[i] = t * 2654435761u + i; y.w[i] = t * 40503u + 7 * i; }
  x.w[7] &= 0x1ff; y.w[7] &= 0x1ff;
  for (int r = 0; r < reps; ++r) { gf_tab_build(L, y); x = gf_mul_tab_y(x, L); y.w[0] ^= x.w[3]; }
  out[t] = x;
}
__global__ void __launch_bounds__(256) checky(int* bad) {
  extern __shared__ char lds[];
  GfLds L = gf_lds_init(lds);
  Gf x, y; u32 t = threadIdx.x + blockIdx.x * 256;
  for (int i = 0; i < 8; ++i) { x.w[i] = t * 2654435761u + i * 97 + (t << 20); y.w[i] = t * 40503u + 7 * i + (t << 17); }
  x.w[7] &= 0x1ff; y.w[7] &= 0x1ff;
  gf_tab_build(L, y);
  if (!gf_eq(gf_mul(x, y), gf_mul_tab_y(x, L))) atomicAdd(bad, 1);
}
template <int TPB> void runy(const char* name, int blocks_per_cu, int reps) {
  int blocks = 256 * blocks_per_cu; Gf* d; hipMalloc(&d, (size_t)blocks * TPB * sizeof(Gf));
  size_t lds_bytes = (TPB / 64) * 16384;
  hipFuncSetAttribute((const void*)ky<TPB>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  ky<TPB><<<blocks, TPB, lds_bytes>>>(d, reps); hipDeviceSynchronize();
  hipEventRecord(e0); ky<TPB><<<blocks, TPB, lds_bytes>>>(d, reps); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double muls = (double)blocks * TPB * reps;
  printf("%-10s TPB=%d blocks/CU=%d %8.3f ms  %7.2f G mul/s  (%s)\n", name, TPB, blocks_per_cu, ms, muls / ms / 1e6, hipGetErrorString(hipGetLastError()));
  hipFree(d);
}
template <int TPB> __global__ void __launch_bounds__(TPB) kx(Gf* out, int reps) {
  extern __shared__ char lds[];
  GfLds L = gf_lds_init(lds);
  u32 t = threadIdx.x + blockIdx.x * blockDim.x;
  Gf x, y;
  for (int i = 0; i < 8; ++i) { x.w[i] = t * 2654435761u + i; y.w[i] = t * 40503u + 7 * i; }
  x.w[7] &= 0x1ff; y.w[7] &= 0x1ff;
  for (int r = 0; r < reps; ++r) { gf_tab_build(L, y); x = gf_mul_tab_x(x, L); y.w[0] ^= x.w[3]; }
  out[t] = x;
}
__global__ void __launch_bounds__(256) checkx(int* bad) {
  extern __shared__ char lds[];
  GfLds L = gf_lds_init(lds);
  Gf x, y; u32 t = threadIdx.x + blockIdx.x * 256;
  for (int i = 0; i < 8; ++i) { x.w[i] = t * 2654435761u + i * 97 + (t << 20); y.w[i] = t * 40503u + 7 * i + (t << 17); }
  x.w[7] &= 0x1ff; y.w[7] &= 0x1ff;
  if (t == 0) { for (int i = 0; i < 8; ++i) x.w[i] = 0xffffffffu; x.w[7] = 0x1ff; }
  gf_tab_build(L, y);
  if (!gf_eq(gf_mul(x, y), gf_mul_tab_x(x, L))) atomicAdd(bad, 1);
}
template <int TPB> void runx(const char* name, int blocks_per_cu, int reps) {
  int blocks = 256 * blocks_per_cu; Gf* d; hipMalloc(&d, (size_t)blocks * TPB * sizeof(Gf));
  size_t lds_bytes = (TPB / 64) * 16384;
  hipFuncSetAttribute((const void*)kx<TPB>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kx<TPB><<<blocks, TPB, lds_bytes>>>(d, reps); hipDeviceSynchronize();
  hipEventRecord(e0); kx<TPB><<<blocks, TPB, lds_bytes>>>(d, reps); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double muls = (double)blocks * TPB * reps;
  printf("%-10s TPB=%d blocks/CU=%d %8.3f ms  %7.2f G mul/s  (%s)\n", name, TPB, blocks_per_cu, ms, muls / ms / 1e6, hipGetErrorString(hipGetLastError()));
  hipFree(d);
}
template <int TPB> __global__ void __launch_bounds__(TPB) kl(Gf* out, int reps) {
  extern __shared__ char lds[];
  GfLds L = gf_lds_init(lds);
  u32 t = threadIdx.x + blockIdx.x * blockDim.x;
  Gf x, y;
  for (int i = 0; i < 8; ++i) { x.w[i] = t * 2654435761u + i; y.w[i] = t * 40503u + 7 * i; }
  x.w[7] &= 0x1ff; y.w[7] &= 0x1ff;
  for (int r = 0; r < reps; ++r) { x = gf_mul(x, y, L); y.w[0] ^= x.w[3]; }
  out[t] = x;
}
template <int TPB> void runl(const char* name, int blocks_per_cu, int reps) {
  int blocks = 256 * blocks_per_cu; Gf* d; hipMalloc(&d, (size_t)blocks * TPB * sizeof(Gf));
  size_t lds_bytes = (TPB / 64) * 16384;
  hipFuncSetAttribute((const void*)kl<TPB>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kl<TPB><<<blocks, TPB, lds_bytes>>>(d, reps); hipDeviceSynchronize();
  hipEventRecord(e0); kl<TPB><<<blocks, TPB, lds_bytes>>>(d, reps); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double muls = (double)blocks * TPB * reps;
  printf("%-10s TPB=%d blocks/CU=%d %8.3f ms  %7.2f G mul/s  (%s)\n", name, TPB, blocks_per_cu, ms, muls / ms / 1e6, hipGetErrorString(hipGetLastError()));
  hipFree(d);
}
int main() {
  int* bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4); checkx<<<8, 256, 65536>>>(bad); int h; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost); printf("mismatches: %d\n", h);
  hipMemset(bad, 0, 4); checky<<<8, 256, 65536>>>(bad); hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost); printf("mismatches Y: %d\n", h);
  runy<256>("ldsY", 2, 200); runy<256>("ldsY", 4, 200);
  runx<256>("ldsX", 2, 200); runx<256>("ldsX", 4, 200); runx<512>("ldsX", 1, 200);
  runl<256>("lds", 2, 200); runl<256>("lds", 4, 200);
  runl<128>("lds", 5, 200); runl<128>("lds", 10, 200);
  runl<64>("lds", 10, 200); runl<64>("lds", 20, 200);
  runl<512>("lds", 1, 200); runl<512>("lds", 2, 200);
}
