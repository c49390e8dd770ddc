// Fr: the 232-bit prime scalar field of sect233k1 (reference: src/curve.rs:16-22), as used by the
// ECFFT and all pointwise prover math.  8 x 32-bit little-endian limbs, Montgomery radix R = 2^256.
//
// The C ABI carries canonical values; kernels keep *data* canonical and *constants* (twiddle
// matrices, per-layer isogeny constants) in Montgomery form, because
//     mont_mul(c*R, x) = c*x            (canonical result)
// so a linear map applied with Montgomery-form constants needs no conversions on the data.
//
// p = 2^231 + 0x69d5bb915bcd46efb1ad5f173abdf: limbs 4..6 are zero and limb 7 is 0x80, which
// the reduction half of the CIOS loop exploits (3 zero products folded, top limb is a shift).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DVP_HD __host__ __device__ __forceinline__
#else
#define DVP_HD inline
#endif

namespace dvp {

struct Fr {
  uint32_t v[8];
};

#define DVP_FR_P_LIMBS \
  { 0xf173abdfu, 0x6efb1ad5u, 0xb915bcd4u, 0x00069d5bu, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000080u }
#define DVP_FR_R1_LIMBS \
  { 0x3373abdfu, 0xc318337eu, 0x1037c69eu, 0x489471e2u, 0xfffff2c5u, 0xffffffffu, 0xffffffffu, 0x0000007fu }
#define DVP_FR_R2_LIMBS \
  { 0x09468bb6u, 0x1710ac10u, 0xdb9a5b86u, 0xf7e3eb91u, 0xb5b58a0au, 0x93c813eeu, 0xbebed802u, 0x00000059u }
#define DVP_FR_PM2_LIMBS \
  { 0xf173abddu, 0x6efb1ad5u, 0xb915bcd4u, 0x00069d5bu, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000080u }
constexpr uint32_t FR_N0 = 0x8c382fe1u;  // -p^{-1} mod 2^32

DVP_HD uint32_t fr_p_limb(int i) {
  constexpr uint32_t p[8] = DVP_FR_P_LIMBS;
  return p[i];
}

DVP_HD Fr fr_zero() {
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = 0;
  return r;
}
DVP_HD Fr fr_one_mont() {
  constexpr uint32_t c[8] = DVP_FR_R1_LIMBS;
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = c[i];
  return r;
}
DVP_HD Fr fr_r2() {
  constexpr uint32_t c[8] = DVP_FR_R2_LIMBS;
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = c[i];
  return r;
}
DVP_HD Fr fr_one_canon() {
  Fr r = fr_zero();
  r.v[0] = 1;
  return r;
}

DVP_HD bool fr_is_zero(const Fr& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.v[i];
  return o == 0;
}
DVP_HD bool fr_eq(const Fr& a, const Fr& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.v[i] ^ b.v[i];
  return o == 0;
}

// r = a - p if a >= p else a   (a < 2p)
DVP_HD Fr fr_cond_sub_p(const Fr& a) {
  constexpr uint32_t p[8] = DVP_FR_P_LIMBS;
  Fr d;
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t t = (uint64_t)a.v[i] - p[i] - borrow;
    d.v[i] = (uint32_t)t;
    borrow = (t >> 63) & 1;
  }
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = borrow ? a.v[i] : d.v[i];
  return r;
}

DVP_HD Fr fr_add(const Fr& a, const Fr& b) {
  Fr s;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)a.v[i] + b.v[i];
    s.v[i] = (uint32_t)c;
    c >>= 32;
  }
  return fr_cond_sub_p(s);  // a,b < p < 2^232 so no carry out of limb 7
}

DVP_HD Fr fr_sub(const Fr& a, const Fr& b) {
  constexpr uint32_t p[8] = DVP_FR_P_LIMBS;
  Fr d;
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t t = (uint64_t)a.v[i] - b.v[i] - borrow;
    d.v[i] = (uint32_t)t;
    borrow = (t >> 63) & 1;
  }
  uint32_t mask = (uint32_t)0 - (uint32_t)borrow;
  uint64_t c = 0;
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)d.v[i] + (p[i] & mask);
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  return r;
}

DVP_HD Fr fr_neg(const Fr& a) { return fr_sub(fr_zero(), a); }

// Montgomery product a*b/R mod p, fully reduced.  CIOS over 32-bit limbs; v_mad_u64_u32 on gfx950.
DVP_HD Fr fr_mul(const Fr& a, const Fr& b) {
  constexpr uint32_t p[8] = DVP_FR_P_LIMBS;
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t c = 0;
    const uint32_t bi = b.v[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      c += (uint64_t)a.v[j] * bi + t[j];
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[8] = (uint32_t)c;
    t[9] = (uint32_t)(c >> 32);
    const uint32_t m = t[0] * FR_N0;
    c = ((uint64_t)m * p[0] + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      if (p[j] != 0) c += (uint64_t)m * p[j];
      c += t[j];
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[7] = (uint32_t)c;
    t[8] = t[9] + (uint32_t)(c >> 32);
  }
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = t[i];
  // a,b < p  =>  t < 2p and t[8] == 0
  return fr_cond_sub_p(r);
}

DVP_HD Fr fr_sqr(const Fr& a) { return fr_mul(a, a); }
DVP_HD Fr fr_to_mont(const Fr& a) { return fr_mul(a, fr_r2()); }
DVP_HD Fr fr_from_mont(const Fr& a) { return fr_mul(a, fr_one_canon()); }
DVP_HD Fr fr_dbl(const Fr& a) { return fr_add(a, a); }

// a^e for a 64-bit exponent (Montgomery in, Montgomery out)
DVP_HD Fr fr_pow_u64(const Fr& a, uint64_t e) {
  Fr r = fr_one_mont();
  Fr b = a;
  while (e) {
    if (e & 1) r = fr_mul(r, b);
    b = fr_sqr(b);
    e >>= 1;
  }
  return r;
}

// a^(p-2) (Montgomery in/out); a == 0 -> 0
DVP_HD Fr fr_inv(const Fr& a) {
  constexpr uint32_t e[8] = DVP_FR_PM2_LIMBS;
  Fr r = fr_one_mont();
  for (int i = 231; i >= 0; --i) {
    r = fr_sqr(r);
    if ((e[i >> 5] >> (i & 31)) & 1) r = fr_mul(r, a);
  }
  return r;
}

// canonical value < p ?
DVP_HD bool fr_is_canonical(const Fr& a) {
  constexpr uint32_t p[8] = DVP_FR_P_LIMBS;
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t t = (uint64_t)a.v[i] - p[i] - borrow;
    borrow = (t >> 63) & 1;
  }
  return borrow != 0;
}

}  // namespace dvp
