// two products that share the table operand, rows interleaved in one wave (more loads in flight per wave) vs back to back
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../dv-pari_amd/csrc/gf233.cuh"
using namespace dvp;
typedef uint32_t u32;
template <int NW>
__device__ __forceinline__ void row2(u32* acc1, u32* acc2, const Gf& a1, const Gf& a2, const GfLds& c, int rsh, int lsh) {
  gf_u32x4 lo1[NW], hi1[NW], lo2[NW], hi2[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    u32 ad1 = ((((a1.w[j] >> rsh) << lsh)) & 0x1C00u) | c.lane_base;
    u32 ad2 = ((((a2.w[j] >> rsh) << lsh)) & 0x1C00u) | c.lane_base;
    lo1[j] = *(const gf_u32x4*)(c.lds + ad1); hi1[j] = *(const gf_u32x4*)(c.lds + ad1 + 8192);
    lo2[j] = *(const gf_u32x4*)(c.lds + ad2); hi2[j] = *(const gf_u32x4*)(c.lds + ad2 + 8192);
  }
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    acc1[j + 0] ^= lo1[j].x; acc1[j + 1] ^= lo1[j].y; acc1[j + 2] ^= lo1[j].z; acc1[j + 3] ^= lo1[j].w;
    acc1[j + 4] ^= hi1[j].x; acc1[j + 5] ^= hi1[j].y; acc1[j + 6] ^= hi1[j].z;
    if (j + 7 < 15) acc1[j + 7] ^= hi1[j].w;
    acc2[j + 0] ^= lo2[j].x; acc2[j + 1] ^= lo2[j].y; acc2[j + 2] ^= lo2[j].z; acc2[j + 3] ^= lo2[j].w;
    acc2[j + 4] ^= hi2[j].x; acc2[j + 5] ^= hi2[j].y; acc2[j + 6] ^= hi2[j].z;
    if (j + 7 < 15) acc2[j + 7] ^= hi2[j].w;
  }
}
__device__ __forceinline__ void gf_mul_tab2(const Gf& a1, const Gf& a2, const GfLds& c, Gf& r1, Gf& r2) {
  u32 acc1[16], acc2[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc1[i] = acc2[i] = 0;
  row2<7>(acc1, acc2, a1, a2, c, 20, 0);
#pragma unroll 1
  for (int k = 9; k >= 4; --k) { gf_acc_shl3(acc1); gf_acc_shl3(acc2); row2<7>(acc1, acc2, a1, a2, c, 3 * k - 10, 0); }
  gf_acc_shl3(acc1); gf_acc_shl3(acc2);
  row2<7>(acc1, acc2, a1, a2, c, 0, 1);
#pragma unroll 1
  for (int k = 2; k >= 0; --k) { gf_acc_shl3(acc1); gf_acc_shl3(acc2); row2<8>(acc1, acc2, a1, a2, c, 0, 10 - 3 * k); }
  r1 = gf_reduce16(acc1);
  r2 = gf_reduce16(acc2);
}
template <int V>
__global__ void __launch_bounds__(256) k_mul(Gf* out, int reps) {
  extern __shared__ char lds[];
  GfLds L = gf_lds_init(lds);
  u32 t = threadIdx.x + blockIdx.x * blockDim.x;
  Gf x, y, z;
  for (int i = 0; i < 8; ++i) { x.w[i] = t * 2654435761u + i; y.w[i] = t * 40503u + 7 * i; z.w[i] = t * 977u + 3 * i; }
  x.w[7] &= 0x1ff; y.w[7] &= 0x1ff; z.w[7] &= 0x1ff;
  for (int r = 0; r < reps; ++r) {
    gf_tab_build(L, y);
    Gf p, q;
    if (V) gf_mul_tab2(x, z, L, p, q); else { p = gf_mul_tab(x, L); q = gf_mul_tab(z, L); }
    x = p; z = q;
    y.w[0] ^= x.w[3] ^ z.w[2];
  }
  out[t] = gf_add(x, z);
}
int main() {
  Gf* d; hipMalloc(&d, (size_t)2560 * 256 * sizeof(Gf));
  Gf *h0 = new Gf[256], *h1 = new Gf[256];
  for (int v = 0; v < 2; ++v) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
      hipEventRecord(e0);
      if (v) k_mul<1><<<2560, 256, 65536>>>(d, 200); else k_mul<0><<<2560, 256, 65536>>>(d, 200);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipMemcpy(v ? h1 : h0, d, 256 * sizeof(Gf), hipMemcpyDeviceToHost);
    printf("%s: %7.2f G mul/s (one table build per two products)\n", v ? "interleaved " : "back-to-back", 2.0 * 2560 * 256 * 200 / best / 1e6);
  }
  int bad = 0; for (int i = 0; i < 256; ++i) for (int k = 0; k < 8; ++k) bad += h0[i].w[k] != h1[i].w[k];
  printf("mismatches %d\n", bad);
}
