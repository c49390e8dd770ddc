// LDS-comb multiplier throughput vs. workgroup shape / LDS footprint (how many waves per CU actually pay off)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../dv-pari_amd/csrc/gf233.cuh"
using namespace dvp;
template <int TPB>
__global__ void __launch_bounds__(TPB) k_mul(Gf* out, int reps) {
  extern __shared__ char lds[];
  GfLds L = gf_lds_init(lds);
  uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  Gf x, y;
  for (int i = 0; i < 8; ++i) { x.w[i] = t * 2654435761u + i; y.w[i] = t * 40503u + 7 * i; }
  x.w[7] &= 0x1ff; y.w[7] &= 0x1ff;
  for (int r = 0; r < reps; ++r) { x = gf_mul(x, y, L); y.w[0] ^= x.w[3]; }
  out[t] = x;
}
template <int TPB> void run(Gf* d, size_t lds, const char* name) {
  hipFuncSetAttribute((const void*)k_mul<TPB>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  int blocks = 256 * 40 * 256 / TPB / 4;  // same thread count for every shape
  int reps = 300;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    k_mul<TPB><<<blocks, TPB, lds>>>(d, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  hipError_t err = hipGetLastError();
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_mul<TPB>, TPB, lds);
  printf("%-34s blocks/CU %2d waves/CU %2d  %7.2f G mul/s %s\n", name, occ, occ * TPB / 64, (double)blocks * TPB * reps / best / 1e6, err ? hipGetErrorString(err) : "");
}
int main() {
  Gf* d; hipMalloc(&d, (size_t)256 * 40 * 256 * sizeof(Gf));
  run<256>(d, 65536, "256 thr, 64 KB (EC kernels)");
  run<256>(d, 81920, "256 thr, 80 KB (k_affine_round)");
  run<64>(d, 16384, "64 thr, 16 KB");
  run<128>(d, 32768, "128 thr, 32 KB");
  run<320>(d, 81920, "320 thr, 80 KB");
  run<192>(d, 49152, "192 thr, 48 KB");
  run<512>(d, 131072, "512 thr, 128 KB");
  run<640>(d, 163840, "640 thr, 160 KB");
  run<256>(d, 40960, "256 thr, 40 KB (invalid table overlap, rate only)");
}
