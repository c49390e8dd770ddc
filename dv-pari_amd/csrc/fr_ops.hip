// Fr vector kernels of the prover (HBM-bound scans over m-length vectors):
//   ark_ff::batch_inversion                              src/proving.rs:604,614; src/ec_fft.rs:482
//   evaluate_poly_at_alpha_using_barycentric_weights     src/ec_fft.rs:455-491 (called 3x, src/proving.rs:571-591)
// Algorithmic bytes: batch inverse 64 B/element (read + write); barycentric 96 B/element.
#include <cstring>

#include "common.h"
#include "fr.cuh"

namespace dvp {

constexpr int INV_CHUNK = 16;  // elements per thread sharing one Fermat inversion

// in-place element-wise inverse of canonical values; zeros stay zero (ark semantics).
// Montgomery's trick on two levels.  A thread owns INV_CHUNK strided elements (coalesced across the block) and keeps
// their prefix products; the 512 thread products of a block then share ONE Fermat inversion: wave 0 multiplies them up
// (8 per lane, prefixes parked in LDS, a shuffle scan across the 64 lanes), inverts the block product and hands every
// thread the inverse of its own product.  A wave pays for an inversion (232 squarings + ~116 products) whether one lane
// needs it or all 64, so the first version -- every thread its own inversion -- spent 22 of its 27 products per element
// there; this one spends ~3 (8192 elements per inversion): 3 + 2 conversions + ~3 per element.
// Shape (round 5): the kernel is a chain -- CHUNK loads and products per thread, the wave-0 stretch with the inversion, CHUNK more
// loads and three products each -- on ONE workgroup per CU at 512 x 16 (2^21 elements: 256 workgroups), 223 us in the prover's
// challenge phase for 30 us' worth of products.  Smaller workgroups with fewer elements per thread run more, shorter chains side by
// side (Tune::fr_bi_shape; the inversions are per workgroup and run in parallel).
__device__ __forceinline__ Fr fr_shfl(const Fr& a, int src_lane) {
  Fr r;
#pragma unroll
  for (int k = 0; k < 8; ++k) r.v[k] = (uint32_t)__shfl((int)a.v[k], src_lane);
  return r;
}
template <int BI_TPB, int BI_CHUNK>
__global__ void __launch_bounds__(BI_TPB) k_batch_inverse(Fr* __restrict__ v, size_t n) {
  constexpr int BI_PER_LANE = BI_TPB / 64;
  constexpr int INV_CHUNK = BI_CHUNK;
  __shared__ Fr s_acc[BI_TPB], s_pre2[BI_TPB];
  const int t = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * (BI_TPB * INV_CHUNK);
  // Round 5: no conversions.  The values stay canonical and fr_mul (= a b / R) is applied to them as they are: with acc_0 = 1 the
  // running product after k factors is P_k / R^k, and with J = the TRUE inverse of the thread's product the way back is
  // 1 / e_k = P_(k-1) / P_k = fr_mul(J, pre_k), J <- fr_mul(J, e_k) -- every power of R cancels.  Three products per element where
  // the Montgomery-form version paid six (to_mont on the way in, twice, from_mont on the way out): the kernel is bound by them.
  Fr pre[INV_CHUNK];
  Fr acc = fr_zero();
  acc.v[0] = 1;
#pragma unroll
  for (int k = 0; k < INV_CHUNK; ++k) {
    const size_t i = base + (size_t)k * BI_TPB + t;
    Fr e = (i < n) ? v[i] : fr_zero();
    pre[k] = acc;
    if (!fr_is_zero(e)) acc = fr_mul(acc, e);  // a zero stays out of the product (its output is zero)
  }
  s_acc[t] = acc;
  __syncthreads();
  // the wave that inverts: not always wave 0 -- the workgroups sharing a CU would all run their inversions on its SIMD 0
  const int inv_wave = (int)(blockIdx.x % (unsigned)BI_PER_LANE);
  if ((threadIdx.x >> 6) == inv_wave) {
    const int t = threadIdx.x & 63;
    // lane t owns the thread products t, 64 + t, ... (conflict-free LDS columns)
    Fr a2 = fr_one_mont();
#pragma unroll
    for (int j = 0; j < BI_PER_LANE; ++j) {
      s_pre2[j * 64 + t] = a2;
      a2 = fr_mul(a2, s_acc[j * 64 + t]);
    }
    Fr pf = a2, sf = a2;  // inclusive prefix / suffix products across the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      Fr y = fr_shfl(pf, t - o >= 0 ? t - o : t);
      if (t >= o) pf = fr_mul(pf, y);
      Fr z = fr_shfl(sf, t + o < 64 ? t + o : t);
      if (t + o < 64) sf = fr_mul(sf, z);
    }
    Fr inv = fr_inv(fr_shfl(pf, 63));
    Fr left = fr_shfl(pf, t > 0 ? t - 1 : 0), right = fr_shfl(sf, t < 63 ? t + 1 : 63);
    if (t > 0) inv = fr_mul(inv, left);
    if (t < 63) inv = fr_mul(inv, right);  // = 1 / a2 of this lane
#pragma unroll
    for (int j = BI_PER_LANE - 1; j >= 0; --j) {
      Fr r = fr_mul(inv, s_pre2[j * 64 + t]);
      inv = fr_mul(inv, s_acc[j * 64 + t]);
      s_acc[j * 64 + t] = r;  // 1 / (thread product)
    }
  }
  __syncthreads();
  // wave-level stretch above: s_acc[t] = R^2 / (thread product) (the Montgomery-form inverse of a value read as x R); two products
  // with the plain 1 strip the two factors of R: the true inverse
  Fr one = fr_zero();
  one.v[0] = 1;
  Fr inv = fr_mul(fr_mul(s_acc[t], one), one);
#pragma unroll
  for (int k = INV_CHUNK - 1; k >= 0; --k) {
    const size_t i = base + (size_t)k * BI_TPB + t;
    Fr e = (i < n) ? v[i] : fr_zero();  // re-read instead of kept: 16 more live elements would not fit the register file
    if (!fr_is_zero(e)) {
      v[i] = fr_mul(inv, pre[k]);
      inv = fr_mul(inv, e);
    }
  }
}

// partial[b] = sum over the block's elements of y_i * w_i / (alpha - s_i)   (Montgomery partials)
// domain / weights canonical or Montgomery per flags; evals canonical.
__global__ void __launch_bounds__(256)
k_bary_partial(const Fr* __restrict__ dom, const Fr* __restrict__ wts, const Fr* __restrict__ ev, size_t n, Fr alpha_m,
               int tables_mont, Fr* __restrict__ partial) {
  __shared__ Fr sh[256];
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  Fr d[INV_CHUNK], pre[INV_CHUNK];
  Fr acc = fr_one_mont();
#pragma unroll
  for (int k = 0; k < INV_CHUNK; ++k) {
    size_t i = t + (size_t)k * stride;
    Fr s = fr_zero();
    if (i < n) s = tables_mont ? dom[i] : fr_to_mont(dom[i]);
    d[k] = (i < n) ? fr_sub(alpha_m, s) : fr_one_mont();
    pre[k] = acc;
    acc = fr_mul(acc, d[k]);
  }
  Fr inv = fr_inv(acc);
  Fr sum = fr_zero();
#pragma unroll
  for (int k = INV_CHUNK - 1; k >= 0; --k) {
    size_t i = t + (size_t)k * stride;
    Fr di = fr_mul(inv, pre[k]);  // 1/(alpha - s_i), Montgomery
    inv = fr_mul(inv, d[k]);
    if (i < n) {
      Fr w = tables_mont ? wts[i] : fr_to_mont(wts[i]);
      Fr term = fr_mul(fr_mul(w, di), fr_to_mont(ev[i]));
      sum = fr_add(sum, term);
    }
  }
  sh[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fr_add(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// out = z_alpha * sum(partials), canonical
__global__ void __launch_bounds__(256) k_bary_final(const Fr* __restrict__ partial, uint32_t nb, Fr z_alpha_m, Fr* __restrict__ out) {
  __shared__ Fr sh[256];
  Fr s = fr_zero();
  for (uint32_t i = threadIdx.x; i < nb; i += 256) s = fr_add(s, partial[i]);
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fr_add(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = fr_from_mont(fr_mul(sh[0], z_alpha_m));
}

// microbenchmark of the ECFFT's multiplier alone (bench.py's work model for configs #3 / #4's extends): pairs of independent
// multiply-adds r = a b / R' + c on lazy 30-bit limbs (fr30_muladd_x2: what a twisted butterfly is made of), a dependent chain per
// thread at the unfused passes' shape (256-thread workgroups, the compiler's own occupancy)
__global__ void __launch_bounds__(256) k_ubench_fr30(Fr30* __restrict__ out, int reps) {
  const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  Fr30 a0, a1, x0, x1, c;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a0.l[i] = (t * 2654435761u + 977u * i) & FR_M30;
    a1.l[i] = (t * 40503u + 7919u * i) & FR_M30;
    x0.l[i] = (t + 13u * i) & FR_M30;
    x1.l[i] = (3u * t + 17u * i) & FR_M30;
    c.l[i] = (5u * t + i) & FR_M30;
  }
  a0.l[7] &= 0x1fffffu; a1.l[7] &= 0x1fffffu; x0.l[7] &= 0x1fffffu; x1.l[7] &= 0x1fffffu; c.l[7] &= 0x1fffffu;  // constants below p, values of a few p
#pragma unroll 1
  for (int r = 0; r < reps; ++r) fr30_muladd_x2(a0, x0, c, a1, x1, c, x0, x1);
  Fr30 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o.l[i] = x0.l[i] ^ x1.l[i];
  out[t] = o;
}

int batch_inverse_dev(Fr* d, size_t n, hipStream_t st) {
  if (!n) return DVP_OK;
#define DVP_BI(TPB, CH) hipLaunchKernelGGL((k_batch_inverse<TPB, CH>), dim3(cdiv(n, (size_t)TPB * CH)), dim3(TPB), 0, st, d, n)
  switch ((int)tune().fr_bi_shape) {
    case 0: DVP_BI(512, 16); break;  // rounds 3-4
    case 1: DVP_BI(256, 8); break;
    case 2: DVP_BI(256, 4); break;
    case 3: DVP_BI(512, 8); break;
    case 4: DVP_BI(128, 8); break;
    case 5: DVP_BI(128, 4); break;
    case 6: DVP_BI(256, 16); break;
    case 7: DVP_BI(1024, 8); break;
    default: DVP_BI(256, 16); break;
  }
#undef DVP_BI
  DVP_HIP(hipGetLastError());
  return DVP_OK;
}

// d_out: one Fr (canonical).  d_partial: >= cdiv(n,256*INV_CHUNK) Fr of scratch.
int barycentric_dev(const Fr* dom, const Fr* wts, const Fr* ev, size_t n, Fr alpha_m, Fr z_alpha_m, int tables_mont,
                    Fr* d_partial, Fr* d_out, hipStream_t st) {
  size_t threads = (n + INV_CHUNK - 1) / INV_CHUNK;
  uint32_t nb = cdiv(threads, 256);
  hipLaunchKernelGGL(k_bary_partial, dim3(nb), dim3(256), 0, st, dom, wts, ev, n, alpha_m, tables_mont, d_partial);
  hipLaunchKernelGGL(k_bary_final, dim3(1), dim3(256), 0, st, d_partial, nb, z_alpha_m, d_out);
  DVP_HIP(hipGetLastError());
  return DVP_OK;
}

}  // namespace dvp

using namespace dvp;

// multiply-adds per second of the lazy 30-bit Fr multiplier, whole chip (the ceiling of bench.py's ECFFT work model)
extern "C" int dvp_ubench_fr_mul(int reps, double* muladds_per_s) {
  if (reps < 1 || !muladds_per_s) return DVP_EINVAL;
  int n_cu = 256;
  int dev = 0;
  DVP_HIP(hipGetDevice(&dev));
  DVP_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  const uint32_t blocks = (uint32_t)n_cu * 8;  // 8 workgroups of 256 per CU
  DevBuf out;
  DVP_TRY(out.alloc((size_t)blocks * 256 * sizeof(Fr30)));
  hipEvent_t e0, e1;
  DVP_HIP(hipEventCreate(&e0));
  DVP_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_ubench_fr30, dim3(blocks), dim3(256), 0, 0, out.as<Fr30>(), 8);
  DVP_HIP(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_ubench_fr30, dim3(blocks), dim3(256), 0, 0, out.as<Fr30>(), reps);
  DVP_HIP(hipEventRecord(e1, 0));
  DVP_HIP(hipEventSynchronize(e1));
  float ms = 0;
  DVP_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *muladds_per_s = 2.0 * (double)blocks * 256.0 * (double)reps / ((double)ms * 1e-3);
  return DVP_OK;
}

extern "C" int dvp_fr_batch_inverse_dev(void* d_vals, size_t n, void* stream) {
  if (n && !d_vals) return DVP_EINVAL;
  return batch_inverse_dev((Fr*)d_vals, n, (hipStream_t)stream);
}

extern "C" int dvp_fr_batch_inverse(uint64_t* vals, size_t n) {
  if (!n) return DVP_OK;
  if (!vals) return DVP_EINVAL;
  DevBuf b;
  DVP_TRY(b.alloc(n * sizeof(Fr)));
  DVP_HIP(hipMemcpy(b.p, vals, n * sizeof(Fr), hipMemcpyHostToDevice));
  DVP_TRY(batch_inverse_dev(b.as<Fr>(), n, 0));
  DVP_HIP(hipMemcpy(vals, b.p, n * sizeof(Fr), hipMemcpyDeviceToHost));
  return DVP_OK;
}

extern "C" int dvp_barycentric_eval(const uint64_t* domain, const uint64_t* bar_weights, const uint64_t z_at_alpha[4],
                                    const uint64_t* evals, size_t n, const uint64_t alpha[4], uint64_t out[4]) {
  if (!domain || !bar_weights || !z_at_alpha || !evals || !alpha || !out || !n) return DVP_EINVAL;
  DevBuf d, w, e, part, o;
  DVP_TRY(d.alloc(n * sizeof(Fr)));
  DVP_TRY(w.alloc(n * sizeof(Fr)));
  DVP_TRY(e.alloc(n * sizeof(Fr)));
  DVP_TRY(part.alloc((n / (256 * INV_CHUNK) + 2) * sizeof(Fr)));
  DVP_TRY(o.alloc(sizeof(Fr)));
  DVP_HIP(hipMemcpy(d.p, domain, n * sizeof(Fr), hipMemcpyHostToDevice));
  DVP_HIP(hipMemcpy(w.p, bar_weights, n * sizeof(Fr), hipMemcpyHostToDevice));
  DVP_HIP(hipMemcpy(e.p, evals, n * sizeof(Fr), hipMemcpyHostToDevice));
  Fr a, z;
  memcpy(a.v, alpha, 32);
  memcpy(z.v, z_at_alpha, 32);
  if (!fr_is_canonical(a) || !fr_is_canonical(z)) return DVP_EINVAL;
  DVP_TRY(barycentric_dev(d.as<Fr>(), w.as<Fr>(), e.as<Fr>(), n, fr_to_mont(a), fr_to_mont(z), 0, part.as<Fr>(), o.as<Fr>(), 0));
  DVP_HIP(hipMemcpy(out, o.p, sizeof(Fr), hipMemcpyDeviceToHost));
  return DVP_OK;
}

// ---- generic pointwise / reduction helpers (the rayon maps of src/proving.rs:492-654 and the setup
// loops of src/srs.rs:53-84,138-160 expressed as reusable vector ops; host-pointer flavours) ---------
namespace dvp {
__global__ void __launch_bounds__(256) k_vec_mul(const Fr* __restrict__ a, const Fr* __restrict__ b, Fr* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = fr_mul(fr_to_mont(a[i]), b[i]);
}
__global__ void __launch_bounds__(256) k_vec_scale(const Fr* __restrict__ a, Fr s_m, Fr* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = fr_mul(s_m, a[i]);
}
// o = a + s * b (canonical in, canonical out; s in Montgomery form)
__global__ void __launch_bounds__(256) k_vec_axpy(const Fr* __restrict__ a, Fr s_m, const Fr* __restrict__ b, Fr* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = fr_add(a[i], fr_mul(s_m, b[i]));
}
__global__ void __launch_bounds__(256) k_vec_scalar_sub(Fr s, const Fr* __restrict__ a, Fr* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = fr_sub(s, a[i]);
}
__global__ void __launch_bounds__(256) k_vec_dot_partial(const Fr* __restrict__ a, const Fr* __restrict__ b, size_t n, Fr* __restrict__ partial) {
  __shared__ Fr sh[256];
  Fr s = fr_zero();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    s = fr_add(s, fr_mul(fr_to_mont(a[i]), b[i]));
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fr_add(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
__global__ void __launch_bounds__(256) k_sum_final(const Fr* __restrict__ partial, uint32_t nb, Fr* __restrict__ out) {
  __shared__ Fr sh[256];
  Fr s = fr_zero();
  for (uint32_t i = threadIdx.x; i < nb; i += 256) s = fr_add(s, partial[i]);
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fr_add(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sh[0];
}
// out[r] = sum_k coeffs[cid[k]] * x[col[k]] over row r (eval_row, src/gnark_r1cs.rs:273-280)
__global__ void __launch_bounds__(256)
k_spmv(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, const uint32_t* __restrict__ cid,
       const Fr* __restrict__ coeffs, const Fr* __restrict__ x, uint32_t n_rows, Fr* __restrict__ out) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  Fr acc = fr_zero();
  for (uint32_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) acc = fr_add(acc, fr_mul(fr_to_mont(coeffs[cid[k]]), x[col[k]]));
  out[r] = acc;
}
}  // namespace dvp

static int up(DevBuf& b, const void* h, size_t bytes) {
  DVP_TRY(b.alloc(bytes));
  DVP_HIP(hipMemcpy(b.p, h, bytes, hipMemcpyHostToDevice));
  return DVP_OK;
}

extern "C" int dvp_fr_vec_mul(const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
  if (!n) return DVP_OK;
  if (!a || !b || !out) return DVP_EINVAL;
  DevBuf da, db;
  DVP_TRY(up(da, a, n * 32));
  DVP_TRY(up(db, b, n * 32));
  hipLaunchKernelGGL(k_vec_mul, dim3(cdiv(n, 256)), dim3(256), 0, 0, da.as<Fr>(), db.as<Fr>(), da.as<Fr>(), n);
  DVP_HIP(hipGetLastError());
  DVP_HIP(hipMemcpy(out, da.p, n * 32, hipMemcpyDeviceToHost));
  return DVP_OK;
}
extern "C" int dvp_fr_vec_scale(const uint64_t* a, const uint64_t s[4], size_t n, uint64_t* out) {
  if (!n) return DVP_OK;
  if (!a || !s || !out) return DVP_EINVAL;
  Fr sc;
  memcpy(sc.v, s, 32);
  if (!fr_is_canonical(sc)) return DVP_EINVAL;
  DevBuf da;
  DVP_TRY(up(da, a, n * 32));
  hipLaunchKernelGGL(k_vec_scale, dim3(cdiv(n, 256)), dim3(256), 0, 0, da.as<Fr>(), fr_to_mont(sc), da.as<Fr>(), n);
  DVP_HIP(hipGetLastError());
  DVP_HIP(hipMemcpy(out, da.p, n * 32, hipMemcpyDeviceToHost));
  return DVP_OK;
}
extern "C" int dvp_fr_vec_axpy(const uint64_t* a, const uint64_t s[4], const uint64_t* b, size_t n, uint64_t* out) {
  if (!n) return DVP_OK;
  if (!a || !b || !s || !out) return DVP_EINVAL;
  Fr sc;
  memcpy(sc.v, s, 32);
  if (!fr_is_canonical(sc)) return DVP_EINVAL;
  DevBuf da, db;
  DVP_TRY(up(da, a, n * 32));
  DVP_TRY(up(db, b, n * 32));
  hipLaunchKernelGGL(k_vec_axpy, dim3(cdiv(n, 256)), dim3(256), 0, 0, da.as<Fr>(), fr_to_mont(sc), db.as<Fr>(), da.as<Fr>(), n);
  DVP_HIP(hipGetLastError());
  DVP_HIP(hipMemcpy(out, da.p, n * 32, hipMemcpyDeviceToHost));
  return DVP_OK;
}
extern "C" int dvp_fr_vec_scalar_sub(const uint64_t s[4], const uint64_t* a, size_t n, uint64_t* out) {
  if (!n) return DVP_OK;
  if (!a || !s || !out) return DVP_EINVAL;
  Fr sc;
  memcpy(sc.v, s, 32);
  if (!fr_is_canonical(sc)) return DVP_EINVAL;
  DevBuf da;
  DVP_TRY(up(da, a, n * 32));
  hipLaunchKernelGGL(k_vec_scalar_sub, dim3(cdiv(n, 256)), dim3(256), 0, 0, sc, da.as<Fr>(), da.as<Fr>(), n);
  DVP_HIP(hipGetLastError());
  DVP_HIP(hipMemcpy(out, da.p, n * 32, hipMemcpyDeviceToHost));
  return DVP_OK;
}
extern "C" int dvp_fr_vec_dot(const uint64_t* a, const uint64_t* b, size_t n, uint64_t out[4]) {
  if (!a || !b || !out) return DVP_EINVAL;
  DevBuf da, db, part, o;
  DVP_TRY(up(da, a, (n ? n : 1) * 32));
  DVP_TRY(up(db, b, (n ? n : 1) * 32));
  uint32_t nb = cdiv(n ? n : 1, 256);
  if (nb > 1024) nb = 1024;
  DVP_TRY(part.alloc(nb * sizeof(Fr)));
  DVP_TRY(o.alloc(sizeof(Fr)));
  hipLaunchKernelGGL(k_vec_dot_partial, dim3(nb), dim3(256), 0, 0, da.as<Fr>(), db.as<Fr>(), n, part.as<Fr>());
  hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, 0, part.as<Fr>(), nb, o.as<Fr>());
  DVP_HIP(hipGetLastError());
  DVP_HIP(hipMemcpy(out, o.p, 32, hipMemcpyDeviceToHost));
  return DVP_OK;
}
extern "C" int dvp_fr_spmv(const uint32_t* row_ptr, const uint32_t* col, const uint32_t* coeff_ids, uint32_t n_rows,
                           const uint64_t* coeffs, uint32_t n_coeffs, const uint64_t* x, uint32_t n_cols, uint64_t* out) {
  if (!n_rows) return DVP_OK;
  if (!row_ptr || !coeffs || !x || !out) return DVP_EINVAL;
  size_t nnz = row_ptr[n_rows];
  for (size_t k = 0; k < nnz; ++k)
    if (col[k] >= n_cols || coeff_ids[k] >= n_coeffs) { g_last_error_index = (int64_t)k; return DVP_EINVAL; }
  DevBuf rp, c, id, cf, dx, o;
  DVP_TRY(up(rp, row_ptr, ((size_t)n_rows + 1) * 4));
  DVP_TRY(up(c, col, (nnz ? nnz : 1) * 4));
  DVP_TRY(up(id, coeff_ids, (nnz ? nnz : 1) * 4));
  DVP_TRY(up(cf, coeffs, (size_t)n_coeffs * 32));
  DVP_TRY(up(dx, x, (size_t)n_cols * 32));
  DVP_TRY(o.alloc((size_t)n_rows * 32));
  hipLaunchKernelGGL(k_spmv, dim3(cdiv(n_rows, 256)), dim3(256), 0, 0, rp.as<uint32_t>(), c.as<uint32_t>(), id.as<uint32_t>(),
                     cf.as<Fr>(), dx.as<Fr>(), n_rows, o.as<Fr>());
  DVP_HIP(hipGetLastError());
  DVP_HIP(hipMemcpy(out, o.p, (size_t)n_rows * 32, hipMemcpyDeviceToHost));
  return DVP_OK;
}
