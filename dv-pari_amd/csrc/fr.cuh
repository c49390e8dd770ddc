// Fr: the 232-bit prime scalar field of sect233k1 (reference: src/curve.rs:16-22), as used by the
// ECFFT and all pointwise prover math.  8 x 32-bit little-endian limbs in memory, Montgomery radix
// R = 2^232 (the multiplier re-slices into 8 x 29-bit limbs, see fr_mul).
//
// The C ABI carries canonical values; kernels keep *data* canonical and *constants* (twiddle
// matrices, per-layer isogeny constants) in Montgomery form, because
//     mont_mul(c*R, x) = c*x            (canonical result)
// so a linear map applied with Montgomery-form constants needs no conversions on the data.
//
// p = 2^231 + 0x69d5bb915bcd46efb1ad5f173abdf: its 29-bit limbs 4..6 are zero, which the reduction half of
// the multiplier exploits.
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
// Montgomery radix R = 2^232 (8 limbs of 29 bits inside the multiplier, see fr_mul)
#define DVP_FR_R1_LIMBS \
  { 0x0e8c5421u, 0x9104e52au, 0x46ea432bu, 0xfff962a4u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x0000007fu }
#define DVP_FR_R2_LIMBS \
  { 0xfd830525u, 0xf2ae10f8u, 0x70af2a02u, 0x13f123b3u, 0xd80293c8u, 0x7b59bebeu, 0x04193b9au, 0x0000002fu }
#define DVP_FR_P29_LIMBS \
  { 0x1173abdfu, 0x17d8d6afu, 0x056f351bu, 0x0d3ab772u, 0x00000000u, 0x00000000u, 0x00000000u, 0x10000000u }
#define DVP_FR_PM2_LIMBS \
  { 0xf173abddu, 0x6efb1ad5u, 0xb915bcd4u, 0x00069d5bu, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000080u }
constexpr uint32_t FR_N0_29 = 0x0c382fe1u;  // -p^{-1} mod 2^29
constexpr uint32_t FR_M29 = 0x1fffffffu;

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

// Montgomery product a*b/R mod p (R = 2^232), fully reduced.
// p has exactly 232 bits, so the operands are re-sliced into 8 limbs of 29 bits: a column of the schoolbook
// product (<= 8 terms of 58 bits) plus the reduction terms fits one 64-bit accumulator with NO carry
// handling, i.e. one v_mad_u64_u32 (half rate on gfx950, measured) per limb product and nothing else --
// the 32-bit-limb CIOS spent 4x more instructions shuffling carries than multiplying.  p's 29-bit limbs
// 4..6 are zero, so the reduction half needs 5 products per column instead of 8.
DVP_HD void fr_to29(const Fr& a, uint32_t* l) {
  l[0] = a.v[0] & FR_M29;
  l[1] = ((a.v[0] >> 29) | (a.v[1] << 3)) & FR_M29;
  l[2] = ((a.v[1] >> 26) | (a.v[2] << 6)) & FR_M29;
  l[3] = ((a.v[2] >> 23) | (a.v[3] << 9)) & FR_M29;
  l[4] = ((a.v[3] >> 20) | (a.v[4] << 12)) & FR_M29;
  l[5] = ((a.v[4] >> 17) | (a.v[5] << 15)) & FR_M29;
  l[6] = ((a.v[5] >> 14) | (a.v[6] << 18)) & FR_M29;
  l[7] = ((a.v[6] >> 11) | (a.v[7] << 21)) & FR_M29;  // a < 2^232: nothing above bit 231
}
DVP_HD Fr fr_from29(const uint32_t* l) {
  Fr r;
  r.v[0] = l[0] | (l[1] << 29);
  r.v[1] = (l[1] >> 3) | (l[2] << 26);
  r.v[2] = (l[2] >> 6) | (l[3] << 23);
  r.v[3] = (l[3] >> 9) | (l[4] << 20);
  r.v[4] = (l[4] >> 12) | (l[5] << 17);
  r.v[5] = (l[5] >> 15) | (l[6] << 14);
  r.v[6] = (l[6] >> 18) | (l[7] << 11);
  r.v[7] = l[7] >> 21;
  return r;
}
DVP_HD Fr fr_mul(const Fr& a, const Fr& b) {
  constexpr uint32_t p[8] = DVP_FR_P29_LIMBS;
  uint32_t x[8], y[8], m[8], r[8];
  fr_to29(a, x);
  fr_to29(b, y);
  uint64_t t = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) t += (uint64_t)x[j] * y[i - j];
#pragma unroll
    for (int j = 0; j < i; ++j)
      if (p[i - j] != 0) t += (uint64_t)m[j] * p[i - j];
    m[i] = ((uint32_t)t * FR_N0_29) & FR_M29;
    t += (uint64_t)m[i] * p[0];
    t >>= 29;
  }
#pragma unroll
  for (int i = 8; i < 16; ++i) {
#pragma unroll
    for (int j = i - 7; j < 8; ++j) t += (uint64_t)x[j] * y[i - j];
#pragma unroll
    for (int j = i - 7; j < 8; ++j)
      if (p[i - j] != 0) t += (uint64_t)m[j] * p[i - j];
    r[i - 8] = (uint32_t)t & FR_M29;
    t >>= 29;
  }
  // a,b < p  =>  result < 2p < 2^233: the last carry holds bit 232 and belongs to limb 7
  r[7] |= (uint32_t)t << 29;
  Fr out;
  out.v[0] = r[0] | (r[1] << 29);
  out.v[1] = (r[1] >> 3) | (r[2] << 26);
  out.v[2] = (r[2] >> 6) | (r[3] << 23);
  out.v[3] = (r[3] >> 9) | (r[4] << 20);
  out.v[4] = (r[4] >> 12) | (r[5] << 17);
  out.v[5] = (r[5] >> 15) | (r[6] << 14);
  out.v[6] = (r[6] >> 18) | (r[7] << 11);
  out.v[7] = r[7] >> 21;
  return fr_cond_sub_p(out);
}

// ---- fused 2-term dot product (the ECFFT butterfly: out = m0*e0 + m1*e1) -------------------------------------------
// Both schoolbook products share the column accumulators (16 terms of 58 bits + the reduction terms stay below
// 2^63) and ONE Montgomery reduction serves the sum: 160 limb products instead of 192, no modular addition, and the
// matrix operand arrives pre-sliced (Fr29) so only the data is re-sliced, once per butterfly.
struct Fr29 {
  uint32_t l[8];  // 29-bit limbs of a value < p
};
DVP_HD Fr29 fr29_from(const Fr& a) {
  Fr29 r;
  fr_to29(a, r.l);
  return r;
}
// (a0*b0 + a1*b1) / R mod p, fully reduced.  T < 2p^2 gives (T + m p)/R < 3p (p/R is a hair above 1/2), hence the two
// conditional subtractions.
DVP_HD Fr fr_dot2(const Fr29& a0, const Fr29& b0, const Fr29& a1, const Fr29& b1) {
  constexpr uint32_t p[8] = DVP_FR_P29_LIMBS;
  uint32_t m[8], r[8];
  uint64_t t = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      t += (uint64_t)a0.l[j] * b0.l[i - j];
      t += (uint64_t)a1.l[j] * b1.l[i - j];
    }
#pragma unroll
    for (int j = 0; j < i; ++j)
      if (p[i - j] != 0) t += (uint64_t)m[j] * p[i - j];
    m[i] = ((uint32_t)t * FR_N0_29) & FR_M29;
    t += (uint64_t)m[i] * p[0];
    t >>= 29;
  }
#pragma unroll
  for (int i = 8; i < 16; ++i) {
#pragma unroll
    for (int j = i - 7; j < 8; ++j) {
      t += (uint64_t)a0.l[j] * b0.l[i - j];
      t += (uint64_t)a1.l[j] * b1.l[i - j];
    }
#pragma unroll
    for (int j = i - 7; j < 8; ++j)
      if (p[i - j] != 0) t += (uint64_t)m[j] * p[i - j];
    r[i - 8] = (uint32_t)t & FR_M29;
    t >>= 29;
  }
  r[7] |= (uint32_t)t << 29;  // result < 3p < 2^234: the carry belongs to limb 7
  return fr_cond_sub_p(fr_cond_sub_p(fr_from29(r)));
}

// (a*b)/R + c mod p, fully reduced, for pre-sliced operands: c*R is c's limbs eight columns up (R = 2^232 = eight 29-bit limbs), so the
// addition rides in the column accumulators of the Montgomery product: 64 + 32 limb products, no modular addition.  The twisted ECFFT
// butterflies (ecfft.hip) are two of these per pair, against two fr_dot2 (2 x 160 limb products) for the untwisted 2x2 matrices.
// T = a b + c R + m p < p^2 + 2 p R gives T / R < 3p: two conditional subtractions.  c = nullptr-like zero limbs -> plain product.
// One limb product accumulated: on the device an explicit v_mad_u64_u32 -- left to itself the compiler re-associates the column sums
// into several partial chains and then pays a half-rate 64-bit add (v_lshl_add_u64) per joint, ~47 per product against 96 multiply-adds.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DVP_FR_NO_ASM_MAD)
__device__ __forceinline__ uint64_t fr_mad64(uint32_t a, uint32_t b, uint64_t c) {
  uint64_t d;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
  return d;
}
// one multiply-add of each of two independent chains in ONE asm statement (the compiler puts a wait state after every asm
// statement that writes a VGPR, whatever follows it)
__device__ __forceinline__ void fr_mad64_2(uint32_t a0, uint32_t b0, uint64_t& t0, uint32_t a1, uint32_t b1, uint64_t& t1) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1" : "+v"(t0), "+v"(t1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "vcc");
}
#else
DVP_HD uint64_t fr_mad64(uint32_t a, uint32_t b, uint64_t c) { return c + (uint64_t)a * b; }
DVP_HD void fr_mad64_2(uint32_t a0, uint32_t b0, uint64_t& t0, uint32_t a1, uint32_t b1, uint64_t& t1) {
  t0 += (uint64_t)a0 * b0;
  t1 += (uint64_t)a1 * b1;
}
#endif

DVP_HD Fr fr_muladd29(const Fr29& a, const Fr29& b, const Fr29& c) {
  constexpr uint32_t p[8] = DVP_FR_P29_LIMBS;
  uint32_t m[8], r[8];
  uint64_t t = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) t = fr_mad64(a.l[j], b.l[i - j], t);
#pragma unroll
    for (int j = 0; j < i; ++j)
      if (p[i - j] != 0) t = fr_mad64(m[j], p[i - j], t);
    m[i] = ((uint32_t)t * FR_N0_29) & FR_M29;
    t = fr_mad64(m[i], p[0], t);
    t >>= 29;
  }
#pragma unroll
  for (int i = 8; i < 16; ++i) {
#pragma unroll
    for (int j = i - 7; j < 8; ++j) t = fr_mad64(a.l[j], b.l[i - j], t);
#pragma unroll
    for (int j = i - 7; j < 8; ++j)
      if (p[i - j] != 0) t = fr_mad64(m[j], p[i - j], t);
    t += c.l[i - 8];
    r[i - 8] = (uint32_t)t & FR_M29;
    t >>= 29;
  }
  r[7] |= (uint32_t)t << 29;  // result < 3p < 2^234: the carry belongs to limb 7
  return fr_cond_sub_p(fr_cond_sub_p(fr_from29(r)));
}
DVP_HD Fr fr_mul29(const Fr29& a, const Fr29& b) {
  Fr29 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z.l[i] = 0;
  return fr_muladd29(a, b, z);
}

// ---- lazy 30-bit-limb arithmetic: the ECFFT butterflies (round 4) ------------------------------------------------------------
// Inside an extend a value lives as 8 limbs of 30 bits (limb 7 keeps whatever is left), only CONGRUENT to the field element and
// bounded by a few dozen p instead of reduced below p.  Montgomery radix R' = 2^240 = eight 30-bit limbs, so p / R' = 2^-9: a product
// constant * x / R' + p is below p (1 + B / 512) for x < B p -- the multiplication itself pulls a value back to ~p, and an
// accumulated addend only grows the bound by one p per layer.  Over the <= 2 x 27 layers of an extend every value stays below 64 p
// < 2^238 < 2^240, so a butterfly needs NO conditional subtraction and no re-slicing between 32-bit and 29/30-bit limbs (the two
// fully reduced 29-bit products they replace spent ~45 % of their instructions on exactly that).  A column of the schoolbook
// product is <= 8 + 5 terms of 60 bits: < 0.54 * 2^64 in the worst case (tools checked with all-ones limbs), one 64-bit accumulator,
// one v_mad_u64_u32 per limb product.  Constants (twiddles, twists) are canonical values times R' mod p, pre-sliced; the data keeps
// whatever domain the caller uses.  The first pass of an extend slices its canonical input, the last one reduces (one conditional
// subtraction: the output twist product is below 2p) and re-slices to Fr.
struct Fr30 {
  uint32_t l[8];
};
constexpr uint32_t FR_M30 = 0x3fffffffu;
constexpr uint32_t FR_N0_30 = 0x0c382fe1u;  // -p^{-1} mod 2^30
#define DVP_FR_P30_LIMBS \
  { 0x3173abdfu, 0x3bec6b57u, 0x115bcd46u, 0x01a756eeu, 0x00000000u, 0x00000000u, 0x00000000u, 0x00200000u }
// 128 p with 2^30 lent from every limb to the one below (limb i < 7: + 2^30, limb i > 0: - 1): e0 + this - e1 never borrows
#define DVP_FR_128P30_LIMBS \
  { 0x79d5ef80u, 0x7635abe1u, 0x6de6a376u, 0x53ab7721u, 0x40000002u, 0x3fffffffu, 0x3fffffffu, 0x0fffffffu }
// 2^8 in Montgomery form (R = 2^232): fr_mul(x R, this) = x 2^240, the constant form of the 30-bit multiplier
#define DVP_FR_2P8_MONT_LIMBS \
  { 0x0a1beddfu, 0x78c56ef3u, 0x8d9c13f6u, 0xf2cbe5e9u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x0000007fu }

DVP_HD Fr30 fr30_from(const Fr& a) {  // a < 2^240
  Fr30 r;
  r.l[0] = a.v[0] & FR_M30;
  r.l[1] = ((a.v[0] >> 30) | (a.v[1] << 2)) & FR_M30;
  r.l[2] = ((a.v[1] >> 28) | (a.v[2] << 4)) & FR_M30;
  r.l[3] = ((a.v[2] >> 26) | (a.v[3] << 6)) & FR_M30;
  r.l[4] = ((a.v[3] >> 24) | (a.v[4] << 8)) & FR_M30;
  r.l[5] = ((a.v[4] >> 22) | (a.v[5] << 10)) & FR_M30;
  r.l[6] = ((a.v[5] >> 20) | (a.v[6] << 12)) & FR_M30;
  r.l[7] = (a.v[6] >> 18) | (a.v[7] << 14);
  return r;
}
DVP_HD Fr fr30_to_fr(const Fr30& a) {  // plain re-slicing of a normalized value < 2^240
  Fr r;
  r.v[0] = a.l[0] | (a.l[1] << 30);
  r.v[1] = (a.l[1] >> 2) | (a.l[2] << 28);
  r.v[2] = (a.l[2] >> 4) | (a.l[3] << 26);
  r.v[3] = (a.l[3] >> 6) | (a.l[4] << 24);
  r.v[4] = (a.l[4] >> 8) | (a.l[5] << 22);
  r.v[5] = (a.l[5] >> 10) | (a.l[6] << 20);
  r.v[6] = (a.l[6] >> 12) | (a.l[7] << 18);
  r.v[7] = a.l[7] >> 14;
  return r;
}
// a Montgomery-form constant (x R, R = 2^232, fully reduced) -> the multiplier's constant form x R' mod p, sliced
DVP_HD Fr30 fr30_const(const Fr& x_mont) {
  constexpr uint32_t c[8] = DVP_FR_2P8_MONT_LIMBS;
  Fr k;
#pragma unroll
  for (int i = 0; i < 8; ++i) k.v[i] = c[i];
  return fr30_from(fr_mul(x_mont, k));
}
// e0 - e1 + 128 p, normalized (e1 < 128 p)
DVP_HD Fr30 fr30_sub_lazy(const Fr30& e0, const Fr30& e1) {
  constexpr uint32_t kp[8] = DVP_FR_128P30_LIMBS;
  Fr30 d;
  uint32_t cy = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t t = e0.l[i] + kp[i] + cy - e1.l[i];  // in [0, 2^32): the lent 2^30 covers e1's limb
    d.l[i] = i < 7 ? (t & FR_M30) : t;
    cy = t >> 30;
  }
  return d;
}
// Two INDEPENDENT products column by column: r_k = a_k b_k / R' + c_k (+ at most p), a_k a constant below p.  Interleaving the two
// accumulator chains is what keeps the explicit v_mad_u64_u32 (fr_mad64) free: the instruction after one never reads its result.
DVP_HD void fr30_muladd_x2(const Fr30& a0, const Fr30& b0, const Fr30& c0, const Fr30& a1, const Fr30& b1, const Fr30& c1, Fr30& r0, Fr30& r1) {
  constexpr uint32_t p[8] = DVP_FR_P30_LIMBS;
  uint32_t m0[8], m1[8];
  Fr30 o0, o1;
  uint64_t t0 = 0, t1 = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) fr_mad64_2(a0.l[j], b0.l[i - j], t0, a1.l[j], b1.l[i - j], t1);
#pragma unroll
    for (int j = 0; j < i; ++j)
      if (p[i - j] != 0) fr_mad64_2(m0[j], p[i - j], t0, m1[j], p[i - j], t1);
    m0[i] = ((uint32_t)t0 * FR_N0_30) & FR_M30;
    m1[i] = ((uint32_t)t1 * FR_N0_30) & FR_M30;
    fr_mad64_2(m0[i], p[0], t0, m1[i], p[0], t1);
    t0 >>= 30;
    t1 >>= 30;
  }
#pragma unroll
  for (int i = 8; i < 16; ++i) {
#pragma unroll
    for (int j = i - 7; j < 8; ++j) fr_mad64_2(a0.l[j], b0.l[i - j], t0, a1.l[j], b1.l[i - j], t1);
#pragma unroll
    for (int j = i - 7; j < 8; ++j)
      if (p[i - j] != 0) fr_mad64_2(m0[j], p[i - j], t0, m1[j], p[i - j], t1);
    t0 += c0.l[i - 8];
    t1 += c1.l[i - 8];
    o0.l[i - 8] = i < 15 ? ((uint32_t)t0 & FR_M30) : (uint32_t)t0;  // limb 7 keeps the rest (the value is below 2^240)
    o1.l[i - 8] = i < 15 ? ((uint32_t)t1 & FR_M30) : (uint32_t)t1;
    t0 >>= 30;
    t1 >>= 30;
  }
  r0 = o0;
  r1 = o1;
}
DVP_HD Fr30 fr30_zero() {
  Fr30 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z.l[i] = 0;
  return z;
}
// one product (the chain is serial: on the device every explicit multiply-add is then followed by a wait state; the kernels pair
// their products through fr30_muladd_x2 wherever two are independent)
DVP_HD Fr30 fr30_muladd(const Fr30& a, const Fr30& b, const Fr30& c) {
  constexpr uint32_t p[8] = DVP_FR_P30_LIMBS;
  uint32_t m[8];
  Fr30 o;
  uint64_t t = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) t = fr_mad64(a.l[j], b.l[i - j], t);
#pragma unroll
    for (int j = 0; j < i; ++j)
      if (p[i - j] != 0) t = fr_mad64(m[j], p[i - j], t);
    m[i] = ((uint32_t)t * FR_N0_30) & FR_M30;
    t = fr_mad64(m[i], p[0], t);
    t >>= 30;
  }
#pragma unroll
  for (int i = 8; i < 16; ++i) {
#pragma unroll
    for (int j = i - 7; j < 8; ++j) t = fr_mad64(a.l[j], b.l[i - j], t);
#pragma unroll
    for (int j = i - 7; j < 8; ++j)
      if (p[i - j] != 0) t = fr_mad64(m[j], p[i - j], t);
    t += c.l[i - 8];
    o.l[i - 8] = i < 15 ? ((uint32_t)t & FR_M30) : (uint32_t)t;
    t >>= 30;
  }
  return o;
}
// a lazy value below 2p (what the output twist leaves) -> canonical Fr
DVP_HD Fr fr30_canon(const Fr30& a) { return fr_cond_sub_p(fr30_to_fr(a)); }

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

// a^(p-2) (Montgomery in/out); a == 0 -> 0.  232 squarings + 116 products: the reference route (ark-ff inverts by an extended
// Euclid; the value is unique).  Kept as the cross-check of fr_inv below.
DVP_HD Fr fr_inv_fermat(const Fr& a) {
  constexpr uint32_t e[8] = DVP_FR_PM2_LIMBS;
  Fr r = fr_one_mont();
  for (int i = 231; i >= 0; --i) {
    r = fr_sqr(r);
    if ((e[i >> 5] >> (i & 31)) & 1) r = fr_mul(r, a);
  }
  return r;
}

// ---- inversion by a binary GCD on 62-bit approximations (round 5) --------------------------------------------------------
// T. Pornin, "Optimized Binary GCD for Modular Inversion" (ePrint 2020/972), Algorithm 2 with k = 31: the classic binary extended
// GCD keeps a = u y, b = v y (mod p) and spends a full-width subtraction and two full-width halvings per step; here 30 steps at a
// time run on 62-bit APPROXIMATIONS of a and b (their low 30 bits, which decide every parity, and their top 32 bits, which decide
// the comparisons) while the steps are recorded as a 2 x 2 matrix of small integers (|f| + |g| <= 2^30), which is then applied ONCE to
// the full-width (a, b) -- exactly divisible by 2^30 -- and to (u, v) modulo p (one Montgomery-style step makes that division exact as
// well).  2 x 232 - 1 = 463 steps reach a = 0, b = 1 for every invertible input (the paper's bound for this variant): 16 rounds of 30.
// A wrong comparison (the approximation can misjudge a and b when they are close) only makes a value negative, which the sign fix
// after the update absorbs.  ~16 x 0.9 k instructions against ~348 x 250 for the Fermat chain: what the one serial inversion of a
// batch inversion (fr_ops.hip: a lone wave per workgroup) and the per-point inversions of the domain tables are made of.
#define DVP_FR_R3_LIMBS \
  { 0xa736e62au, 0xd5434168u, 0x8fe1c0d5u, 0x4ddf2901u, 0x8acce292u, 0xc70972a5u, 0xbf5ec775u, 0x00000072u }
// t = f x + g y as a 9-limb two's complement number; |f| + |g| <= 2^30
DVP_HD void fr_gcd_lin(int32_t f, const uint32_t* x, int32_t g, const uint32_t* y, uint32_t* t) {
  int64_t acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc += (int64_t)f * (int64_t)(uint64_t)x[i] + (int64_t)g * (int64_t)(uint64_t)y[i];
    t[i] = (uint32_t)acc;
    acc >>= 32;  // arithmetic: the carry keeps its sign
  }
  t[8] = (uint32_t)acc;
}
// bits [s, s + 32) of the 256-bit x (zero beyond bit 255), 30 <= s < 256
DVP_HD uint32_t fr_gcd_window(const uint32_t* x, uint32_t s) {
  const uint32_t q = s >> 5, r = s & 31u;
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (uint32_t i = 0; i < 8; ++i) {
    lo = (i == q) ? x[i] : lo;
    hi = (i == q + 1) ? x[i] : hi;
  }
  return r ? (lo >> r) | (hi << (32 - r)) : lo;
}
// y^{-1} mod p for a canonical integer y < p (no Montgomery factors); 0 -> 0
DVP_HD Fr fr_inv_gcd_raw(const Fr& y) {
  constexpr uint32_t P[8] = DVP_FR_P_LIMBS;
  uint32_t a[8], b[8], u[8], v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = y.v[i]; b[i] = P[i]; u[i] = 0; v[i] = 0; }
  u[0] = 1;
#pragma unroll 1
  for (int it = 0; it < 16; ++it) {
    // n = max(len(a), len(b), 62); the approximations keep bits [0, 30) and [n - 32, n)
    uint32_t top = 0, w = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
      const uint32_t o = a[i] | b[i];
      top = o ? o : top;
      w = o ? i : w;
    }
    uint32_t n = top ? 32u * (w + 1) - (uint32_t)__builtin_clz(top) : 0u;
    n = n < 62u ? 62u : n;
    uint64_t xa = (uint64_t)(a[0] & 0x3fffffffu) | ((uint64_t)fr_gcd_window(a, n - 32) << 30);
    uint64_t xb = (uint64_t)(b[0] & 0x3fffffffu) | ((uint64_t)fr_gcd_window(b, n - 32) << 30);
    int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
#pragma unroll 2
    for (int j = 0; j < 30; ++j) {
      const bool odd = xa & 1u;
      const bool sw = odd && xa < xb;
      const uint64_t ta = sw ? xb : xa, tb = sw ? xa : xb;
      const int32_t tf0 = sw ? f1 : f0, tf1 = sw ? f0 : f1, tg0 = sw ? g1 : g0, tg1 = sw ? g0 : g1;
      xa = (odd ? ta - tb : ta) >> 1;
      xb = tb;
      f0 = odd ? tf0 - tf1 : tf0;
      g0 = odd ? tg0 - tg1 : tg0;
      f1 = tf1 << 1;
      g1 = tg1 << 1;
    }
    // (a, b) <- (f0 a + g0 b, f1 a + g1 b) / 2^30, signs folded into the matrix rows
    uint32_t ta[9], tb[9];
    fr_gcd_lin(f0, a, g0, b, ta);
    fr_gcd_lin(f1, a, g1, b, tb);
    const bool na = (int32_t)ta[8] < 0, nb = (int32_t)tb[8] < 0;
    uint64_t ca = na ? 1 : 0, cb = nb ? 1 : 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint32_t ra = (ta[i] >> 30) | (ta[i + 1] << 2), rb = (tb[i] >> 30) | (tb[i + 1] << 2);
      ca += na ? (uint32_t)~ra : ra;  // two's complement negation when the value came out negative
      cb += nb ? (uint32_t)~rb : rb;
      a[i] = (uint32_t)ca;
      b[i] = (uint32_t)cb;
      ca >>= 32;
      cb >>= 32;
    }
    if (na) { f0 = -f0; g0 = -g0; }
    if (nb) { f1 = -f1; g1 = -g1; }
    // (u, v) <- the same rows / 2^30 modulo p: make the low 30 bits vanish with a multiple of p, shift, fold into [0, p)
    fr_gcd_lin(f0, u, g0, v, ta);
    fr_gcd_lin(f1, u, g1, v, tb);
    const uint32_t qa = (ta[0] * FR_N0_30) & 0x3fffffffu, qb = (tb[0] * FR_N0_30) & 0x3fffffffu;
    uint64_t da = 0, db = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      da += (uint64_t)ta[i] + (uint64_t)qa * P[i];
      db += (uint64_t)tb[i] + (uint64_t)qb * P[i];
      ta[i] = (uint32_t)da;
      tb[i] = (uint32_t)db;
      da >>= 32;
      db >>= 32;
    }
    ta[8] += (uint32_t)da;
    tb[8] += (uint32_t)db;
    Fr ru, rv;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      ru.v[i] = (ta[i] >> 30) | (ta[i + 1] << 2);
      rv.v[i] = (tb[i] >> 30) | (tb[i + 1] << 2);
    }
    // a value in (-p, 2p) as 256-bit two's complement
    const bool nu = (int32_t)ru.v[7] < 0, nv = (int32_t)rv.v[7] < 0;
    uint64_t cu = 0, cv = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      cu += (uint64_t)ru.v[i] + (nu ? P[i] : 0u);
      cv += (uint64_t)rv.v[i] + (nv ? P[i] : 0u);
      ru.v[i] = (uint32_t)cu;
      rv.v[i] = (uint32_t)cv;
      cu >>= 32;
      cv >>= 32;
    }
    ru = fr_cond_sub_p(ru);
    rv = fr_cond_sub_p(rv);
#pragma unroll
    for (int i = 0; i < 8; ++i) { u[i] = ru.v[i]; v[i] = rv.v[i]; }
  }
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = v[i];
  return r;
}
// a^{-1} (Montgomery in / out): (a R)^{-1} = a^{-1} R^{-1}, times R^3 / R; a == 0 -> 0
DVP_HD Fr fr_inv(const Fr& a) {
  constexpr uint32_t c[8] = DVP_FR_R3_LIMBS;
  Fr r3;
#pragma unroll
  for (int i = 0; i < 8; ++i) r3.v[i] = c[i];
  return fr_mul(fr_inv_gcd_raw(a), r3);
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
