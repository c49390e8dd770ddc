// Which load form, if any, makes the L2 ask the fabric for 64 bytes when a lane needs a 64-byte point?
// (every read request of the first pair round is a 128-byte one, so half of every fetched byte is unused.)
// Every kernel below gathers the same number of random 64-byte points out of the same 3 GB table; they differ only in the load
// instruction / cache policy / address-to-lane mapping / memory type.  Run un-profiled for the rates, and under
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace
//   rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum --kernel-trace
// for the request sizes per kernel name (tools/pmc_digest.py prints them).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32;
typedef uint64_t u64;
typedef u32 v4u __attribute__((ext_vector_type(4)));
typedef u32 v16u __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_fill(uint4* t, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) t[i] = make_uint4((u32)i, 1, 2, 3);
}
__device__ __forceinline__ u64 mix(u64 x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
__device__ __forceinline__ void fold(uint4& acc, const uint4& a) { acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w; }
constexpr int PER = 32;   // points per lane
constexpr int U = 4;      // points in flight per lane

// ---- 1. what the pair round does: four global_load_dwordx4 per lane and point, default policy
__global__ void __launch_bounds__(256) k_g_plain(const uint4* __restrict__ t, u64 npts, u32* out, u32 seed) {
  u64 tid = blockIdx.x * (u64)blockDim.x + threadIdx.x;
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll U
  for (int k = 0; k < PER; ++k) {
    const uint4* p = t + (mix(tid * PER + k + ((u64)seed << 40)) % npts) * 4;
    fold(acc, p[0]); fold(acc, p[1]); fold(acc, p[2]); fold(acc, p[3]);
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
// the same kernel under other names, so that a table in another memory type gets its own row in the counter digest
__global__ void __launch_bounds__(256) k_g_plain_uncached(const uint4* __restrict__ t, u64 npts, u32* out, u32 seed) {
  u64 tid = blockIdx.x * (u64)blockDim.x + threadIdx.x;
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll U
  for (int k = 0; k < PER; ++k) {
    const uint4* p = t + (mix(tid * PER + k + ((u64)seed << 40)) % npts) * 4;
    fold(acc, p[0]); fold(acc, p[1]); fold(acc, p[2]); fold(acc, p[3]);
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
__global__ void __launch_bounds__(256) k_g_plain_finegrained(const uint4* __restrict__ t, u64 npts, u32* out, u32 seed) {
  u64 tid = blockIdx.x * (u64)blockDim.x + threadIdx.x;
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll U
  for (int k = 0; k < PER; ++k) {
    const uint4* p = t + (mix(tid * PER + k + ((u64)seed << 40)) % npts) * 4;
    fold(acc, p[0]); fold(acc, p[1]); fold(acc, p[2]); fold(acc, p[3]);
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
// ---- 2. non-temporal (compiler builtin -> "nt")
__global__ void __launch_bounds__(256) k_g_nt(const uint4* __restrict__ t, u64 npts, u32* out, u32 seed) {
  u64 tid = blockIdx.x * (u64)blockDim.x + threadIdx.x;
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll U
  for (int k = 0; k < PER; ++k) {
    const uint4* p = t + (mix(tid * PER + k + ((u64)seed << 40)) % npts) * 4;
    const v4u* pv = (const v4u*)p;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v4u a = __builtin_nontemporal_load(pv + j); acc.x ^= a[0]; acc.y ^= a[1]; acc.z ^= a[2]; acc.w ^= a[3]; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
// ---- 3. explicit cache-policy bits by inline asm: U points (4 loads each) in flight, then one wait
// (the loads AND their s_waitcnt sit in ONE asm statement: the compiler treats an asm's outputs as valid when the statement ends, so
// with a separate wait statement it folded -- and re-allocated -- the destination registers while the loads were still in flight: a
// memory fault on the first run of this file)
#define LD16(MOD)                                                                         \
  asm volatile("global_load_dwordx4 %0, %16, off " MOD "\n\t" \
               "global_load_dwordx4 %1, %16, off offset:16 " MOD "\n\t" \
               "global_load_dwordx4 %2, %16, off offset:32 " MOD "\n\t" \
               "global_load_dwordx4 %3, %16, off offset:48 " MOD "\n\t" \
               "global_load_dwordx4 %4, %17, off " MOD "\n\t" \
               "global_load_dwordx4 %5, %17, off offset:16 " MOD "\n\t" \
               "global_load_dwordx4 %6, %17, off offset:32 " MOD "\n\t" \
               "global_load_dwordx4 %7, %17, off offset:48 " MOD "\n\t" \
               "global_load_dwordx4 %8, %18, off " MOD "\n\t" \
               "global_load_dwordx4 %9, %18, off offset:16 " MOD "\n\t" \
               "global_load_dwordx4 %10, %18, off offset:32 " MOD "\n\t" \
               "global_load_dwordx4 %11, %18, off offset:48 " MOD "\n\t" \
               "global_load_dwordx4 %12, %19, off " MOD "\n\t" \
               "global_load_dwordx4 %13, %19, off offset:16 " MOD "\n\t" \
               "global_load_dwordx4 %14, %19, off offset:32 " MOD "\n\t" \
               "global_load_dwordx4 %15, %19, off offset:48 " MOD "\n\t" \
               "s_waitcnt vmcnt(0)" \
               : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]), "=&v"(v[0][3]), "=&v"(v[1][0]), "=&v"(v[1][1]), "=&v"(v[1][2]), "=&v"(v[1][3]), "=&v"(v[2][0]), "=&v"(v[2][1]), "=&v"(v[2][2]), "=&v"(v[2][3]), "=&v"(v[3][0]), "=&v"(v[3][1]), "=&v"(v[3][2]), "=&v"(v[3][3]) \
               : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory")
#define ASM_KERNEL(NAME, MOD)                                                             \
  __global__ void __launch_bounds__(256) NAME(const uint4* __restrict__ t, u64 npts, u32* out, u32 seed) { \
    static_assert(U == 4, "LD16 is written for four points in flight");                  \
    u64 tid = blockIdx.x * (u64)blockDim.x + threadIdx.x;                                 \
    uint4 acc = make_uint4(0, 0, 0, 0);                                                   \
    for (int k = 0; k < PER; k += U) {                                                    \
      v4u v[U][4];                                                                        \
      const uint4* p0 = t + (mix(tid * PER + k + 0 + ((u64)seed << 40)) % npts) * 4;      \
      const uint4* p1 = t + (mix(tid * PER + k + 1 + ((u64)seed << 40)) % npts) * 4;      \
      const uint4* p2 = t + (mix(tid * PER + k + 2 + ((u64)seed << 40)) % npts) * 4;      \
      const uint4* p3 = t + (mix(tid * PER + k + 3 + ((u64)seed << 40)) % npts) * 4;      \
      LD16(MOD);                                                                          \
      _Pragma("unroll") for (int u = 0; u < U; ++u)                                       \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { acc.x ^= v[u][j][0]; acc.y ^= v[u][j][1]; acc.z ^= v[u][j][2]; acc.w ^= v[u][j][3]; } \
    }                                                                                     \
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;                       \
  }
ASM_KERNEL(k_g_asm_default, "")
ASM_KERNEL(k_g_asm_sc0, "sc0")
ASM_KERNEL(k_g_asm_sc1, "sc1")
ASM_KERNEL(k_g_asm_sc0sc1, "sc0 sc1")
ASM_KERNEL(k_g_asm_nt, "nt")
ASM_KERNEL(k_g_asm_sc1nt, "sc1 nt")
ASM_KERNEL(k_g_asm_sc0sc1nt, "sc0 sc1 nt")
// ---- 4. only the x-coordinate (the first 32 bytes of the point): what pass 1 of the pair round needs
__global__ void __launch_bounds__(256) k_g_x32(const uint4* __restrict__ t, u64 npts, u32* out, u32 seed) {
  u64 tid = blockIdx.x * (u64)blockDim.x + threadIdx.x;
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll U
  for (int k = 0; k < PER; ++k) {
    const uint4* p = t + (mix(tid * PER + k + ((u64)seed << 40)) % npts) * 4;
    fold(acc, p[0]); fold(acc, p[1]);
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
// ---- 5. a quad of lanes per point: one global_load_dwordx4 per lane, the four lanes of a quad cover the 64 bytes (one instruction
// touches 16 points instead of 64 quarter-points); same number of points per wave
__global__ void __launch_bounds__(256) k_g_quad(const uint4* __restrict__ t, u64 npts, u32* out, u32 seed) {
  u64 tid = blockIdx.x * (u64)blockDim.x + threadIdx.x;
  const u32 q = threadIdx.x & 3;
  const u64 grp = tid >> 2;
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll 8
  for (int k = 0; k < 4 * PER; ++k) {  // a quad fetches 4 x PER points, 16 bytes per lane each: the same bytes per lane as above
    const uint4* p = t + (mix(grp * (4 * PER) + k + ((u64)seed << 40)) % npts) * 4;
    fold(acc, p[q]);
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
// ---- 6. scalar loads: one wave-uniform point per s_load_dwordx16 (the scalar cache has 64-byte lines).  Not usable for per-lane
// gathers as it stands -- it only answers whether that path produces 64-byte fabric requests at all.
__global__ void __launch_bounds__(256) k_g_sload(const uint4* __restrict__ t, u64 npts, u32* out, u32 seed) {
  const u64 wave = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 6;
  u32 acc = 0;
  for (int k = 0; k < 64 * PER / 16; ++k) {  // 1/16 of the points of the vector kernels (a wave fetches one point per instruction)
    u64 idx = __builtin_amdgcn_readfirstlane((u32)(mix(wave * 4096 + k + ((u64)seed << 40)) % npts));
    const u32* p = (const u32*)(t + idx * 4);
    v16u r;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
#pragma unroll
    for (int j = 0; j < 16; ++j) acc ^= r[j];
  }
  if (acc == 0x12345678u) out[0] = 1;
}

typedef void (*kern_t)(const uint4*, u64, u32*, u32);
struct Var { const char* name; kern_t k; int mem; double pts_scale; };

int main(int argc, char** argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 3.0;
  const size_t bytes = (size_t)(gb * (1ull << 30));
  u32* out; CK(hipMalloc(&out, 4));
  uint4* tab[3] = {nullptr, nullptr, nullptr};
  CK(hipMalloc(&tab[0], bytes));
  if (hipExtMallocWithFlags((void**)&tab[1], bytes, hipDeviceMallocUncached) != hipSuccess) { tab[1] = nullptr; (void)hipGetLastError(); printf("uncached alloc failed\n"); }
  if (hipExtMallocWithFlags((void**)&tab[2], bytes, hipDeviceMallocFinegrained) != hipSuccess) { tab[2] = nullptr; (void)hipGetLastError(); printf("fine-grained alloc failed\n"); }
  for (int m = 0; m < 3; ++m) if (tab[m]) k_fill<<<4096, 256>>>(tab[m], bytes / 16);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const Var vars[] = {
    {"plain (4 x dwordx4 per lane, default policy)", k_g_plain, 0, 1.0},
    {"nt (__builtin_nontemporal_load)", k_g_nt, 0, 1.0},
    {"asm default", k_g_asm_default, 0, 1.0},
    {"asm sc0", k_g_asm_sc0, 0, 1.0},
    {"asm sc1", k_g_asm_sc1, 0, 1.0},
    {"asm sc0 sc1", k_g_asm_sc0sc1, 0, 1.0},
    {"asm nt", k_g_asm_nt, 0, 1.0},
    {"asm sc1 nt", k_g_asm_sc1nt, 0, 1.0},
    {"asm sc0 sc1 nt", k_g_asm_sc0sc1nt, 0, 1.0},
    {"x only (first 32 bytes of the point)", k_g_x32, 0, 1.0},
    {"quad of lanes per point (1 x dwordx4 per lane)", k_g_quad, 0, 1.0},
    {"scalar loads (s_load_dwordx16, one point per wave-instruction)", k_g_sload, 0, 1.0 / 16},
    {"plain, table in hipDeviceMallocUncached memory", k_g_plain_uncached, 1, 1.0},
    {"plain, table in hipDeviceMallocFinegrained memory", k_g_plain_finegrained, 2, 1.0},
  };
  const u32 blocks = 768 * 4;
  printf("table %.1f GB, %u blocks x 256 lanes x %d points per lane = %.1f M points per launch\n", gb, blocks, PER, blocks * 256.0 * PER / 1e6);
  for (const Var& v : vars) {
    if (!tab[v.mem]) { printf("%-64s skipped (no table)\n", v.name); continue; }
    v.k<<<blocks, 256>>>(tab[v.mem], bytes / 64, out, 1);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) v.k<<<blocks, 256>>>(tab[v.mem], bytes / 64, out, 2 + r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double g = 3.0 * blocks * 256.0 * PER * v.pts_scale;
    printf("%-64s %7.2f G points/s  %6.3f ms per launch\n", v.name, g / ms / 1e6, ms / 3);
    fflush(stdout);
  }
  return 0;
}
