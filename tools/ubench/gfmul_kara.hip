// One-level Karatsuba on top of the LDS comb: 3 half-products against 4-word tables (8 KB per wave instead
// of 16 KB, one ds_read_b128 per lookup instead of two).  Correctness vs the register-only gf_mul and throughput
// at 2 / 3 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../dv-pari_amd/csrc/gf233.cuh"
using namespace dvp;
typedef uint32_t u32;

constexpr unsigned KH_LDS_PER_WAVE = 8192;
// LDS addresses are kept as 32-bit integers (address space 3 offsets), so a lookup address is one shift and one
// bitop3 -- no pointer add per lookup
typedef __attribute__((address_space(3))) gf_u32x4 lds_u32x4;
__device__ __forceinline__ gf_u32x4 lds_ld(u32 addr) { return *(const lds_u32x4*)addr; }
__device__ __forceinline__ void lds_st(u32 addr, gf_u32x4 v) { *(lds_u32x4*)addr = v; }
struct GfLdsH {
  u32 lane_base;  // absolute LDS byte address of this lane's entry 0 (region 8 KB aligned)
};
__device__ __forceinline__ GfLdsH kh_init(char* base) {
  GfLdsH c;
  u32 b = (u32)(uintptr_t)(__attribute__((address_space(3))) char*)base;
  c.lane_base = b + (threadIdx.x >> 6) * KH_LDS_PER_WAVE + (threadIdx.x & 63) * 16;
  lds_st(c.lane_base, (gf_u32x4){0, 0, 0, 0});
  return c;
}
__device__ __forceinline__ void kh_shl1_4(const u32* in, u32* out) {
#pragma unroll
  for (int i = 3; i > 0; --i) out[i] = __builtin_amdgcn_alignbit(in[i], in[i - 1], 31);
  out[0] = in[0] << 1;
}
__device__ __forceinline__ void kh_store(const GfLdsH& c, int u, const u32* w) {
  lds_st(c.lane_base + (u32)u * 1024, (gf_u32x4){w[0], w[1], w[2], w[3]});
}
// table of a <= 117-bit operand (4 words)
__device__ __forceinline__ void kh_tab_build(const GfLdsH& c, const u32* b) {
  u32 t2[4], t3[4], t4[4], t6[4];
  kh_store(c, 1, b);
  kh_shl1_4(b, t2);
  kh_store(c, 2, t2);
#pragma unroll
  for (int i = 0; i < 4; ++i) t3[i] = t2[i] ^ b[i];
  kh_store(c, 3, t3);
  kh_shl1_4(t2, t4);
  kh_store(c, 4, t4);
  kh_shl1_4(t3, t6);
  kh_store(c, 6, t6);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    t4[i] ^= b[i];
    t6[i] ^= b[i];
  }
  kh_store(c, 5, t4);
  kh_store(c, 7, t6);
}
__device__ __forceinline__ u32 x3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
template <int NW>
__device__ __forceinline__ void kh_row(u32* acc, const u32* a, const GfLdsH& c, int rsh, int lsh) {
  gf_u32x4 v[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    u32 sh = (a[j] >> rsh) << lsh;
    u32 addr = __builtin_amdgcn_bitop3_b32(sh, 0x1C00u, c.lane_base, 0xEA);  // (sh & m) | base
    v[j] = lds_ld(addr);
  }
  asm volatile("" ::: "memory");
  // 4 NW values into NW + 3 accumulator words with 3-input xors
  if (NW == 4) {
    acc[0] ^= v[0].x;
    acc[1] = x3(acc[1], v[0].y, v[1].x);
    acc[2] = x3(x3(acc[2], v[0].z, v[1].y), v[2].x, 0);
    acc[3] = x3(x3(acc[3], v[0].w, v[1].z), v[2].y, v[3].x);
    acc[4] = x3(x3(acc[4], v[1].w, v[2].z), v[3].y, 0);
    acc[5] = x3(acc[5], v[2].w, v[3].z);
    acc[6] ^= v[3].w;
  } else {
    acc[0] ^= v[0].x;
    acc[1] = x3(acc[1], v[0].y, v[1].x);
    acc[2] = x3(x3(acc[2], v[0].z, v[1].y), v[2].x, 0);
    acc[3] = x3(x3(acc[3], v[0].w, v[1].z), v[2].y, 0);
    acc[4] = x3(acc[4], v[1].w, v[2].z);
    acc[5] ^= v[2].w;
  }
}
#ifdef SHL64
// 64-bit form: (pair << 3) + carry -- v_lshl_add_u64 (one half-rate op per TWO words) + a full-rate v_lshrrev for the carry
__device__ __forceinline__ void kh_shl3(u32* acc) {
#pragma unroll
  for (int i = 3; i >= 0; --i) {
    uint64_t p = ((uint64_t)acc[2 * i + 1] << 32) | acc[2 * i];
    uint64_t c = i ? (uint64_t)(acc[2 * i - 1] >> 29) : 0;
    p = (p << 3) + c;
    acc[2 * i] = (u32)p;
    acc[2 * i + 1] = (u32)(p >> 32);
  }
}
#else
__device__ __forceinline__ void kh_shl3(u32* acc) {
#pragma unroll
  for (int i = 7; i > 0; --i) acc[i] = __builtin_amdgcn_alignbit(acc[i], acc[i - 1], 29);
  acc[0] <<= 3;
}
#endif
#ifdef PIPE
// software-pipelined rows: the four lookups of row k-1 are issued before the xors of row k (addresses depend only on a)
__device__ __forceinline__ void kh_issue(gf_u32x4* v, const u32* a, const GfLdsH& c, int k) {
  const int rsh = 3 * k - 10 > 0 ? 3 * k - 10 : 0, lsh = 10 - 3 * k > 0 ? 10 - 3 * k : 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u32 sh = (a[j] >> rsh) << lsh;
    v[j] = lds_ld(__builtin_amdgcn_bitop3_b32(sh, 0x1C00u, c.lane_base, 0xEA));   // a[3] has no digits k >= 7: row 0 (zeros)
  }
}
__device__ __forceinline__ void kh_consume(u32* acc, const gf_u32x4* v) {
  acc[0] ^= v[0].x;
  acc[1] = x3(acc[1], v[0].y, v[1].x);
  acc[2] = x3(acc[2], v[0].z, v[1].y) ^ v[2].x;
  acc[3] = x3(x3(acc[3], v[0].w, v[1].z), v[2].y, v[3].x);
  acc[4] = x3(acc[4], v[1].w, v[2].z) ^ v[3].y;
  acc[5] = x3(acc[5], v[2].w, v[3].z);
  acc[6] ^= v[3].w;
}
__device__ __forceinline__ void kh_mul_tab(u32* acc, const u32* a, const GfLdsH& c) {
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0;
  gf_u32x4 p[4], q[4];
  kh_issue(p, a, c, 10);
#pragma unroll 1
  for (int k = 10; k >= 2; k -= 2) {   // rows k (in p) and k-1 (in q)
    kh_issue(q, a, c, k - 1);
    asm volatile("" ::: "memory");
    if (k != 10) kh_shl3(acc);
    kh_consume(acc, p);
    kh_issue(p, a, c, k - 2);
    asm volatile("" ::: "memory");
    kh_shl3(acc);
    kh_consume(acc, q);
  }
  kh_shl3(acc);
  kh_consume(acc, p);  // row 0
}
#else
// acc[0..7] = a (4 words, word 3 < 2^21) * table operand
__device__ __forceinline__ void kh_mul_tab(u32* acc, const u32* a, const GfLdsH& c) {
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0;
  kh_row<3>(acc, a, c, 20, 0);  // k = 10 (2-bit digit): (w >> 30) << 10
#pragma unroll 1
  for (int k = 9; k >= 7; --k) {
    kh_shl3(acc);
    kh_row<3>(acc, a, c, 3 * k - 10, 0);
  }
#pragma unroll 1
  for (int k = 6; k >= 4; --k) {
    kh_shl3(acc);
    kh_row<4>(acc, a, c, 3 * k - 10, 0);
  }
#ifdef RSH
  // rows k = 3..0 need a << (10 - 3k): take them as RIGHT shifts (full rate on gfx950) of one pre-shifted copy a << 10
  u32 ar[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) ar[j] = a[j] << 10;
#pragma unroll 1
  for (int k = 3; k >= 0; --k) {
    kh_shl3(acc);
    kh_row<4>(acc, ar, c, 3 * k, 0);
  }
#else
  kh_shl3(acc);
  kh_row<4>(acc, a, c, 0, 1);  // k = 3
#pragma unroll 1
  for (int k = 2; k >= 0; --k) {
    kh_shl3(acc);
    kh_row<4>(acc, a, c, 0, 10 - 3 * k);
  }
#endif
}
#endif
// x = lo + hi z^117
__device__ __forceinline__ void kh_split(const Gf& x, u32* lo, u32* hi) {
  lo[0] = x.w[0]; lo[1] = x.w[1]; lo[2] = x.w[2]; lo[3] = x.w[3] & 0x1FFFFFu;
#pragma unroll
  for (int i = 0; i < 4; ++i) hi[i] = __builtin_amdgcn_alignbit(x.w[i + 4], x.w[i + 3], 21);
}
__device__ __forceinline__ Gf kh_mul(const Gf& a, const Gf& b, const GfLdsH& c) {
  u32 a0[4], a1[4], b0[4], b1[4], am[4], bm[4];
  kh_split(a, a0, a1);
  kh_split(b, b0, b1);
#pragma unroll
  for (int i = 0; i < 4; ++i) { am[i] = a0[i] ^ a1[i]; bm[i] = b0[i] ^ b1[i]; }
  u32 L[8], H[8], M[8];
  kh_tab_build(c, b0);
  kh_mul_tab(L, a0, c);
  kh_tab_build(c, b1);
  kh_mul_tab(H, a1, c);
  kh_tab_build(c, bm);
  kh_mul_tab(M, am, c);
#pragma unroll
  for (int i = 0; i < 8; ++i) M[i] ^= L[i] ^ H[i];
  // c = L + M z^117 + H z^234
  u32 r[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = L[i];
#pragma unroll
  for (int i = 8; i < 16; ++i) r[i] = 0;
  // M << 117 = 3 words + 21 bits
  r[3] ^= M[0] << 21;
#pragma unroll
  for (int i = 1; i < 8; ++i) r[3 + i] ^= __builtin_amdgcn_alignbit(M[i], M[i - 1], 11);
  r[11] ^= M[7] >> 11;
  // H << 234 = 7 words + 10 bits
  r[7] ^= H[0] << 10;
#pragma unroll
  for (int i = 1; i < 8; ++i) r[7 + i] ^= __builtin_amdgcn_alignbit(H[i], H[i - 1], 22);
  r[15] ^= H[7] >> 22;
  return gf_reduce16(r);
}

template <int TPB, int WPE>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k_kara(Gf* out, int reps) {
  extern __shared__ char lds[];
  GfLdsH L = kh_init(lds);
  u32 t = threadIdx.x + blockIdx.x * blockDim.x;
  Gf x, y;
  for (int i = 0; i < 8; ++i) { x.w[i] = t * 2654435761u + i; y.w[i] = t * 40503u + 7 * i; }
  x.w[7] &= 0x1ff; y.w[7] &= 0x1ff;
  for (int r = 0; r < reps; ++r) { x = kh_mul(x, y, L); y.w[0] ^= x.w[3]; }
  out[t] = x;
}
template <int TPB>
__global__ void __launch_bounds__(TPB) k_ref(Gf* out, int reps) {
  extern __shared__ char lds[];
  GfLds L = gf_lds_init(lds);
  u32 t = threadIdx.x + blockIdx.x * blockDim.x;
  Gf x, y;
  for (int i = 0; i < 8; ++i) { x.w[i] = t * 2654435761u + i; y.w[i] = t * 40503u + 7 * i; }
  x.w[7] &= 0x1ff; y.w[7] &= 0x1ff;
  for (int r = 0; r < reps; ++r) { x = gf_mul(x, y, L); y.w[0] ^= x.w[3]; }
  out[t] = x;
}
__global__ void k_plain(Gf* out, int reps, int n) {
  u32 t = threadIdx.x + blockIdx.x * blockDim.x;
  if ((int)t >= n) return;
  Gf x, y;
  for (int i = 0; i < 8; ++i) { x.w[i] = t * 2654435761u + i; y.w[i] = t * 40503u + 7 * i; }
  x.w[7] &= 0x1ff; y.w[7] &= 0x1ff;
  for (int r = 0; r < reps; ++r) { x = gf_mul(x, y); y.w[0] ^= x.w[3]; }
  out[t] = x;
}

static const int NTHR = 256 * 48 * 64;  // threads in every throughput run

template <class K>
void timeit(K kern, const char* name, int tpb, size_t lds, Gf* d) {
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  int blocks = NTHR / tpb, reps = 300;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(tpb), lds, 0, d, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  hipError_t err = hipGetLastError();
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern, tpb, lds);
  printf("%-44s blocks/CU %2d waves/CU %2d  %7.2f G mul/s %s\n", name, occ, occ * tpb / 64, (double)blocks * tpb * reps / best / 1e6,
         err ? hipGetErrorString(err) : "");
}

int main() {
  Gf *d, *d2;
  hipMalloc(&d, (size_t)NTHR * sizeof(Gf));
  hipMalloc(&d2, (size_t)NTHR * sizeof(Gf));
  // correctness: 7 dependent products per thread, 4096 threads, vs the register-only multiplier
  {
    const int n = 4096, reps = 7;
    hipFuncSetAttribute((const void*)k_kara<256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipLaunchKernelGGL((k_kara<256, 2>), dim3(n / 256), dim3(256), 4 * KH_LDS_PER_WAVE, 0, d, reps);
    hipLaunchKernelGGL(k_plain, dim3(n / 256), dim3(256), 0, 0, d2, reps, n);
    static Gf h1[4096], h2[4096];
    hipMemcpy(h1, d, sizeof(h1), hipMemcpyDeviceToHost);
    hipMemcpy(h2, d2, sizeof(h2), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < 8; ++k) bad += h1[i].w[k] != h2[i].w[k];
    printf("karatsuba vs register multiplier: %s (%d word mismatches)\n", bad ? "MISMATCH" : "ok", bad);
    if (bad) return 1;
  }
  timeit(k_ref<256>, "comb 16 KB/wave, 256 thr (2 waves/SIMD)", 256, 65536, d);
  timeit(k_kara<256, 2>, "kara 8 KB/wave, 256 thr, wpe 2 (64 KB)", 256, 65536, d);
  timeit(k_kara<256, 3>, "kara 8 KB/wave, 256 thr, wpe 3 (48 KB)", 256, 49152, d);
  timeit(k_kara<256, 4>, "kara 8 KB/wave, 256 thr, wpe 4 (36 KB)", 256, 36864, d);
  timeit(k_kara<256, 3>, "kara 8 KB/wave, 256 thr, wpe 3 (32 KB)", 256, 32768, d);
  timeit(k_kara<512, 4>, "kara 8 KB/wave, 512 thr, wpe 4 (64 KB)", 512, 65536, d);
  return 0;
}
