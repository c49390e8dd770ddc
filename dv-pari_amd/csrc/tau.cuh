// tau-adic recoding of sect233k1 scalars (shared by msm.hip and codec.hip).
//
// On K-233 (a = 0, mu = -1) the Frobenius tau(x,y) = (x^2,y^2) satisfies tau^2 + tau + 2 = 0 and acts
// on E[r] as multiplication by lambda (the root of x^2+x+2 mod r with tau(G) = lambda*G).  A scalar
// s is first reduced modulo delta = (tau^233 - 1)/(tau - 1) (Solinas' partial reduction, with a
// 256-bit fixed-point reciprocal instead of an exact rounding: any rho = s (mod delta) is valid, the
// rounding only affects the length) and then expanded in base tau with digits {0,1}:
// x^2 + x + 2 is a canonical-number-system polynomial, so the expansion is finite and unique.
// The reference gets the same effect inside xs233's xsk233_mul_frob (src/curve.rs:118-123).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dvp {

// constants: oracle/pyref.py tau_constants()
// delta = D0 + D1*tau, N(delta) = r;  conj(delta) = (D0 - D1) - D1*tau;  A_i = floor(|conj_i| 2^256 / r)
static __constant__ uint32_t TAU_D0[4] = {0xba75bb3bu, 0xda32c0f4u, 0x2dcb0ed1u, 0x00032540u};
static __constant__ uint32_t TAU_D1[4] = {0xcb36bee6u, 0x16aa143cu, 0x2d7ae36eu, 0x000882d7u};
static __constant__ uint32_t TAU_C0M[4] = {0x10c103abu, 0x3c775348u, 0xffafd49cu, 0x00055d96u};  // D1 - D0
static __constant__ uint32_t TAU_A0[5] = {0x55720891u, 0x90218207u, 0x3878eea6u, 0x2dff5fa9u, 0x00000abbu};
static __constant__ uint32_t TAU_A1[5] = {0xcb1ecea9u, 0x79966d7du, 0xdc2d5428u, 0xae5af5c6u, 0x00001105u};
// TAU_DIGITS = 240 is a PROVEN bound on the length of the expansion, not an observed one (observed maximum: 236):
//   rho = s - kappa*delta with kappa_i = round(s c_i / r) computed through a 256-bit fixed-point reciprocal, so
//   rho/delta = e0 + e1 tau with |e_i| <= 1/2 + 2^-24, hence N(rho) = N(delta) (e0^2 - e0 e1 + 2 e1^2) <= r (1 + 2^-21)
//   and |rho| = sqrt(N(rho)) < 2^115.51 (r < 2^231.01).  One expansion step maps rho to (rho - u)/tau, u in {0,1}, and
//   |tau| = sqrt 2, so |rho_(k+1)| <= (|rho_k| + 1)/sqrt 2, i.e. |rho_k| < |rho_0| 2^(-k/2) + 1/(sqrt 2 - 1).  After
//   k = 233 steps |rho_233| < 2^(115.51-116.5) + 2.4143 < 2.92, so N(rho_233) <= 8; every element of Z[tau] of norm
//   <= 9 has an expansion of at most 7 digits (exhaustive enumeration, tests/test_oracle.py::test_tau_length_bound).
//   Total <= 233 + 7 = 240.  The windows of both MSM modes cover >= 240 digits (msm.hip: MsmFixedCtx::set_c keeps an
//   overflow window of c >= 8 digits above digit 234; msm_plan adds ceil(6/c) overflow windows), so the "expansion
//   longer than the windows" flag of k_recode cannot fire for a canonical scalar; it stays as an internal check.
constexpr int TAU_DIGITS = 240;

// out[0..no) = low `no` limbs of a[0..na) * b[0..nb)
template <int NA, int NB, int NO>
__device__ __forceinline__ void mp_mul_lo(const uint32_t* a, const uint32_t* b, uint32_t* out) {
#pragma unroll
  for (int i = 0; i < NO; ++i) out[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (i + j < NO) {
        c += (uint64_t)a[i] * b[j] + out[i + j];
        out[i + j] = (uint32_t)c;
        c >>= 32;
      }
    }
    if (i + NB < NO) out[i + NB] = (uint32_t)c;
  }
}

// Q = round(s * A / 2^256), s: 8 limbs, A: 5 limbs -> 4 limbs (value < 2^117)
__device__ __forceinline__ void tau_round_mul(const uint32_t* s, const uint32_t* A, uint32_t* Q) {
  uint32_t t[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      c += (uint64_t)s[i] * A[j] + t[i + j];
      t[i + j] = (uint32_t)c;
      c >>= 32;
    }
    t[i + 5] = (uint32_t)c;
  }
  // + 2^255, then >> 256
  uint64_t c = (uint64_t)t[7] + 0x80000000u;
  c >>= 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    c += t[8 + i];
    Q[i] = (uint32_t)c;
    c >>= 32;
  }
}


// s < r ?
__device__ __forceinline__ bool tau_scalar_is_canonical(const uint32_t* s) {
  const uint32_t p[8] = {0xf173abdfu, 0x6efb1ad5u, 0xb915bcd4u, 0x00069d5bu, 0, 0, 0, 0x00000080u};
  uint64_t borrow = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    uint64_t t = (uint64_t)s[k] - p[k] - borrow;
    borrow = (t >> 63) & 1;
  }
  return borrow != 0;
}

// rho = r0 + r1*tau = s (mod delta); components are 160-bit two's complement, |r_i| < 2^118
__device__ __forceinline__ void tau_partial_reduce(const uint32_t* s, uint32_t* r0, uint32_t* r1) {
  uint32_t Q0[4], Q1[4];
  tau_round_mul(s, TAU_A0, Q0);
  tau_round_mul(s, TAU_A1, Q1);
  // rho0 = s + Q0*D0 - 2*Q1*D1 ; rho1 = Q0*D1 - Q1*(D1-D0)      (mod 2^160)
  uint32_t t0[5], t1[5];
  mp_mul_lo<4, 4, 5>(Q0, TAU_D0, t0);
  mp_mul_lo<4, 4, 5>(Q1, TAU_D1, t1);
  {
    uint64_t cy = 0;
    int64_t bw = 0;
    uint32_t acc[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {  // s + t0
      cy += (uint64_t)s[k] + t0[k];
      acc[k] = (uint32_t)cy;
      cy >>= 32;
    }
    uint32_t prev = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {  // - 2*t1
      uint32_t d = (t1[k] << 1) | prev;
      prev = t1[k] >> 31;
      int64_t v = (int64_t)acc[k] - d + bw;
      r0[k] = (uint32_t)v;
      bw = v >> 32;
    }
  }
  mp_mul_lo<4, 4, 5>(Q0, TAU_D1, t0);
  mp_mul_lo<4, 4, 5>(Q1, TAU_C0M, t1);
  {
    int64_t bw = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      int64_t v = (int64_t)t0[k] - t1[k] + bw;
      r1[k] = (uint32_t)v;
      bw = v >> 32;
    }
  }
}

// one digit: u = rho mod tau in {0,1}; rho <- (rho - u)/tau, i.e. (r0,r1) <- (r1 - h, -h), h = (r0-u)/2
__device__ __forceinline__ uint32_t tau_step(uint32_t* r0, uint32_t* r1) {
  uint32_t u = r0[0] & 1u;
  uint32_t h[5];
#pragma unroll
  for (int k = 0; k < 4; ++k) h[k] = (r0[k] >> 1) | (r0[k + 1] << 31);
  h[4] = (uint32_t)((int32_t)r0[4] >> 1);
  int64_t b1 = 0, b2 = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    int64_t v = (int64_t)r1[k] - h[k] + b1;
    r0[k] = (uint32_t)v;
    b1 = v >> 32;
    int64_t z = (int64_t)0 - h[k] + b2;
    r1[k] = (uint32_t)z;
    b2 = z >> 32;
  }
  return u;
}

// sixteen digits at once.  The digits depend only on the low bits of (r0, r1), so they come from a 32-bit copy of the
// one-digit recurrence (after j steps the low 32-j bits are still exact); then rho <- (rho - D) / tau^16 in one
// multi-limb pass: D = sum d_i tau^i = p + q tau by Horner, tau^16 = 178 - 93 tau, N(tau^16) = 2^16 and
// conj(tau^16) = 271 + 93 tau, so with x = r0 - p, y = r1 - q:
//     r0' = (271 x - 186 y) / 2^16,   r1' = (93 x + 178 y) / 2^16          (exact divisions).
// ~3x fewer instructions than sixteen tau_step calls.  Returns the digits, digit i in bit i.
__device__ __forceinline__ uint32_t tau_step16(uint32_t* r0, uint32_t* r1) {
  uint32_t lo0 = r0[0], lo1 = r1[0], dw = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    uint32_t u = lo0 & 1u;
    uint32_t h = (lo0 - u) >> 1;
    lo0 = lo1 - h;
    lo1 = 0u - h;
    dw |= u << i;
  }
  int32_t p = 0, q = 0;
#pragma unroll
  for (int i = 15; i >= 0; --i) {
    int32_t d = (int32_t)((dw >> i) & 1u);
    int32_t np = d - 2 * q, nq = p - q;
    p = np;
    q = nq;
  }
  // x = r0 - p, y = r1 - q (5-limb two's complement), then the two linear combinations with carries in int64
  uint32_t x[5], y[5];
  {
    int64_t bx = -(int64_t)p, by = -(int64_t)q;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      bx += (int64_t)(uint64_t)r0[k];
      x[k] = (uint32_t)bx;
      bx >>= 32;
      by += (int64_t)(uint64_t)r1[k];
      y[k] = (uint32_t)by;
      by >>= 32;
    }
  }
  uint32_t n0[5], n1[5];
  int64_t c0 = 0, c1 = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    // limbs 0..3 are unsigned digits, limb 4 carries the sign
    int64_t xv = k < 4 ? (int64_t)(uint64_t)x[k] : (int64_t)(int32_t)x[k];
    int64_t yv = k < 4 ? (int64_t)(uint64_t)y[k] : (int64_t)(int32_t)y[k];
    c0 += 271 * xv - 186 * yv;
    c1 += 93 * xv + 178 * yv;
    n0[k] = (uint32_t)c0;
    n1[k] = (uint32_t)c1;
    c0 >>= 32;
    c1 >>= 32;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r0[k] = (n0[k] >> 16) | (n0[k + 1] << 16);
    r1[k] = (n1[k] >> 16) | (n1[k + 1] << 16);
  }
  r0[4] = (uint32_t)((int32_t)n0[4] >> 16);
  r1[4] = (uint32_t)((int32_t)n1[4] >> 16);
  return dw;
}

}  // namespace dvp
