// Do the LDS-comb multiplier (LDS-limited to 8 waves/CU) and the register-only multiplier (no LDS) add up when run
// concurrently on two streams?  Measures each alone and both together.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../dv-pari_amd/csrc/gf233.cuh"
using namespace dvp;
typedef uint32_t u32;
__global__ void __launch_bounds__(256) k_lds(Gf* out, int reps) {
  extern __shared__ char lds[];
  GfLds L = gf_lds_init(lds);
  u32 t = threadIdx.x + blockIdx.x * blockDim.x;
  Gf x, y;
  for (int i = 0; i < 8; ++i) { x.w[i] = t * 2654435761u + i; y.w[i] = t * 40503u + 7 * i; }
  x.w[7] &= 0x1ff; y.w[7] &= 0x1ff;
  for (int r = 0; r < reps; ++r) { x = gf_mul(x, y, L); y.w[0] ^= x.w[3]; }
  out[t] = x;
}
__global__ void __launch_bounds__(256) k_reg(Gf* out, int reps) {
  u32 t = threadIdx.x + blockIdx.x * blockDim.x;
  Gf x, y;
  for (int i = 0; i < 8; ++i) { x.w[i] = t * 2654435761u + i; y.w[i] = t * 40503u + 7 * i; }
  x.w[7] &= 0x1ff; y.w[7] &= 0x1ff;
  for (int r = 0; r < reps; ++r) { x = gf_mul(x, y); y.w[0] ^= x.w[3]; }
  out[t] = x;
}
int main() {
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  Gf *d1, *d2; hipMalloc(&d1, 4096 * 256 * sizeof(Gf)); hipMalloc(&d2, 4096 * 256 * sizeof(Gf));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](int bl, int rl, int br, int rr, const char* name) {
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipStreamWaitEvent(s1, e0, 0); hipStreamWaitEvent(s2, e0, 0);
    if (bl) k_lds<<<bl, 256, 65536, s1>>>(d1, rl);
    if (br) k_reg<<<br, 256, 0, s2>>>(d2, rr);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, s1); hipEventRecord(b, s2);
    hipStreamWaitEvent(0, a, 0); hipStreamWaitEvent(0, b, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double muls = (double)bl * 256 * rl + (double)br * 256 * rr;
    printf("%-34s %8.3f ms  %7.2f G mul/s\n", name, ms, muls / ms / 1e6);
  };
  run(2048, 200, 0, 0, "warm"); run(0, 0, 2048, 100, "warm");
  run(2048, 400, 0, 0, "LDS only (2048 blk x 400)");
  run(0, 0, 2048, 200, "reg only (2048 blk x 200)");
  run(2048, 400, 512, 150, "LDS 2048x400 + reg 512x150");
  run(2048, 400, 1024, 100, "LDS 2048x400 + reg 1024x100");
  run(2048, 400, 2048, 60, "LDS 2048x400 + reg 2048x60");
}
