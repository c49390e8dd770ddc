// Random 64-byte gathers (one affine point per lane) out of a table of R GB: does the reach of the table (TLB, DRAM page
// locality) limit the gather rate the pair rounds need (~6.5 G points/s in round 0)?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32;
typedef uint64_t u64;
__global__ void __launch_bounds__(256) k_fill(uint4* t, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) t[i] = make_uint4((u32)i, 1, 2, 3);
}
__device__ __forceinline__ u64 mix(u64 x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
// each lane gathers PER points of 64 B; indices are independent so the loads pipeline like the kernel's prefetch
template <int PER>
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ t, u64 npts, u32* out, u32 seed) {
  u64 tid = blockIdx.x * (u64)blockDim.x + threadIdx.x;
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll 4
  for (int k = 0; k < PER; ++k) {
    u64 idx = mix(tid * PER + k + ((u64)seed << 40)) % npts;
    const uint4* p = t + idx * 4;
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc.x ^= a.x ^ b.x ^ c.x ^ d.x; acc.y ^= a.y ^ b.y ^ c.y ^ d.y; acc.z ^= a.z ^ b.z ^ c.z ^ d.z; acc.w ^= a.w ^ b.w ^ c.w ^ d.w;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
int main(int argc, char** argv) {
  double sizes[] = {1, 5, 16, 32, 64, 100, 160};
  u32* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (double gb : sizes) {
    size_t bytes = (size_t)(gb * (1ull << 30));
    uint4* t;
    if (hipMalloc(&t, bytes) != hipSuccess) { printf("%6.0f GB: alloc failed\n", gb); (void)hipGetLastError(); continue; }
    k_fill<<<4096, 256>>>(t, bytes / 16);
    hipDeviceSynchronize();
    const int PER = 32;
    const u32 blocks = 768 * 8;  // 8 chip-fulls of 3 blocks per CU
    k_gather<PER><<<blocks, 256>>>(t, bytes / 64, out, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) k_gather<PER><<<blocks, 256>>>(t, bytes / 64, out, 2 + r);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double g = 4.0 * blocks * 256 * PER;
    printf("%6.0f GB table: %7.2f G gathers/s  (%6.2f TB/s of 64-byte points)\n", gb, g / ms / 1e6, g * 64 / ms / 1e9);
    fflush(stdout);
    hipFree(t);
  }
  return 0;
}
