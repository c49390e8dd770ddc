// quad-cooperative GF(2^233) product: equality with the per-lane LDS-comb product, and single-wave latency of both
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "../../dv-pari_amd/csrc/gf233.cuh"
using namespace dvp;
__global__ void __launch_bounds__(256) k_check(const Gf* a, const Gf* b, Gf* o1, Gf* o2, int n) {
  extern __shared__ char lds[];
  GfLdsQ Q = gf_ldsq_init(lds);
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int task = t >> 2;
  if (task >= n) return;
  Gf x = a[task], y = b[task];
  Gf p = gf_mul(x, y, Q.l);
  Gf q = gf_mul(x, y, Q);
  if ((t & 3) == (task & 3)) { o1[task] = p; o2[task] = q; }
}
template <bool QUAD>
__global__ void __launch_bounds__(64) k_lat(Gf* out, int reps) {
  extern __shared__ char lds[];
  GfLdsQ Q = gf_ldsq_init(lds);
  Gf x, y;
  for (int i = 0; i < 8; ++i) { x.w[i] = (threadIdx.x >> 2) * 2654435761u + i; y.w[i] = (threadIdx.x >> 2) * 40503u + 7 * i; }
  x.w[7] &= 0x1ff; y.w[7] &= 0x1ff;
  for (int r = 0; r < reps; ++r) {
    if (QUAD) x = gf_mul(x, y, Q); else x = gf_mul(x, y, Q.l);
    y.w[0] ^= x.w[3];
  }
  out[threadIdx.x] = x;
}
int main() {
  const int n = 4096;
  std::vector<Gf> a(n), b(n);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 8; ++k) { a[i].w[k] = rnd(); b[i].w[k] = rnd(); }
    a[i].w[7] &= 0x1ff; b[i].w[7] &= 0x1ff;
    if (i < 8) for (int k = 0; k < 8; ++k) a[i].w[k] = (k == 7) ? 0x1ff : 0xffffffffu;  // all-ones: every digit 7
    if (i == 8) for (int k = 0; k < 8; ++k) a[i].w[k] = 0;
  }
  Gf *da, *db, *o1, *o2;
  hipMalloc(&da, n * sizeof(Gf)); hipMalloc(&db, n * sizeof(Gf)); hipMalloc(&o1, n * sizeof(Gf)); hipMalloc(&o2, n * sizeof(Gf));
  hipMemcpy(da, a.data(), n * sizeof(Gf), hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * sizeof(Gf), hipMemcpyHostToDevice);
  k_check<<<n * 4 / 256, 256, 65536>>>(da, db, o1, o2, n);
  std::vector<Gf> r1(n), r2(n);
  hipMemcpy(r1.data(), o1, n * sizeof(Gf), hipMemcpyDeviceToHost); hipMemcpy(r2.data(), o2, n * sizeof(Gf), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) for (int k = 0; k < 8; ++k) bad += r1[i].w[k] != r2[i].w[k];
  printf("mismatching words: %d of %d\n", bad, n * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int q = 0; q < 2; ++q) {
    for (int w = 0; w < 2; ++w) {
      hipEventRecord(e0);
      if (q) k_lat<true><<<1, 64, 16384>>>(o1, 2000); else k_lat<false><<<1, 64, 16384>>>(o1, 2000);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (w) printf("%s: %.3f us per dependent product (one wave)\n", q ? "quad " : "lane ", ms * 1e3 / 2000);
    }
  }
  return bad != 0;
}
