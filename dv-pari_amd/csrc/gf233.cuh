// GF(2^233) = GF(2)[z]/(z^233 + z^74 + 1): the base field of sect233k1 (the arithmetic the
// reference reaches through xs233-sys, src/curve.rs:13).  8 x 32-bit words, polynomial basis,
// bit i of the 256-bit little-endian value = coefficient of z^i; bits 233..255 are zero.
//
// gfx950 has no carry-less multiply (v_clmul_* does not assemble for this target), so the
// product is built on the integer VALU: a is scanned bit-serially (v_bfe_i32 turns a bit into a
// 0/-1 mask), and each mask gates one word of b into the accumulator with a single
// v_bitop3_b32 ((m & b) ^ acc, truth table 0x6a); the accumulator is shifted by one bit per
// scanned bit position with v_alignbit_b32.  One 233x233 product = 32 x (15 shifts + 8 masks +
// 64 and-xor) ~ 2.8k VALU lane-ops; a squaring is a bit spread (~0.2k).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dvp {

struct Gf {
  uint32_t w[8];
};

#define GF_DEV __device__ __forceinline__

GF_DEV Gf gf_zero() {
  Gf r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.w[i] = 0;
  return r;
}
GF_DEV Gf gf_one() {
  Gf r = gf_zero();
  r.w[0] = 1;
  return r;
}
GF_DEV Gf gf_add(const Gf& a, const Gf& b) {
  Gf r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.w[i] = a.w[i] ^ b.w[i];
  return r;
}
GF_DEV bool gf_is_zero(const Gf& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.w[i];
  return o == 0;
}
GF_DEV bool gf_eq(const Gf& a, const Gf& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.w[i] ^ b.w[i];
  return o == 0;
}
// c ? a : b
GF_DEV Gf gf_select(bool c, const Gf& a, const Gf& b) {
  Gf r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.w[i] = c ? a.w[i] : b.w[i];
  return r;
}

GF_DEV uint32_t gf_andxor(uint32_t m, uint32_t b, uint32_t acc) {
  return __builtin_amdgcn_bitop3_b32(m, b, acc, 0x6a);  // (m & b) ^ acc
}

// c[0..15] (466 significant bits) -> reduced element.  z^233 = z^74 + 1:
// word j >= 8 sits at bit 32j = 233 + (32(j-8)+23), so it folds into bit offsets 32(j-8)+23 and
// 32(j-8)+97 = 32(j-5)+1.
// Round 5: word by word.  The words 12 .. 15 fold into 7 .. 11 first (T8 .. T11 below are the unreduced words 8 .. 11 with those
// folds applied); after that every output word takes the fold of TWO neighbouring high words, whose shifted images do not overlap --
// (T[i+8] << 23) | (T[i+7] >> 9) and (T[i+5] << 1) | (T[i+4] >> 31) -- so each fold is ONE v_alignbit_b32 and the sums are 3-input
// xors: ~36 instructions where the word-serial loop (four shifts and four xors per high word) took ~72.
// TOP: the highest word that can be nonzero (a square of a reduced element, or a Karatsuba product, ends in word 14).
GF_DEV uint32_t gf_xor3_(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
template <int TOP = 15>
GF_DEV Gf gf_reduce16(const uint32_t* c) {
  const uint32_t T15 = TOP >= 15 ? c[15] : 0u;
  const uint32_t T14 = c[14], T13 = c[13], T12 = c[12];
  const uint32_t T11 = TOP >= 15 ? c[11] ^ (T15 >> 31) : c[11];
  const uint32_t T10 = c[10] ^ (TOP >= 15 ? __builtin_amdgcn_alignbit(T15, T14, 31) : T14 >> 31);
  const uint32_t T9 = c[9] ^ __builtin_amdgcn_alignbit(T14, T13, 31);
  const uint32_t T8 = TOP >= 15 ? gf_xor3_(c[8], __builtin_amdgcn_alignbit(T13, T12, 31), T15 >> 9) : c[8] ^ __builtin_amdgcn_alignbit(T13, T12, 31);
  Gf r;
  r.w[0] = c[0] ^ (T8 << 23);
  r.w[1] = c[1] ^ __builtin_amdgcn_alignbit(T9, T8, 9);
  r.w[2] = c[2] ^ __builtin_amdgcn_alignbit(T10, T9, 9);
  r.w[3] = gf_xor3_(c[3], __builtin_amdgcn_alignbit(T11, T10, 9), T8 << 1);
  r.w[4] = gf_xor3_(c[4], __builtin_amdgcn_alignbit(T12, T11, 9), __builtin_amdgcn_alignbit(T9, T8, 31));
  r.w[5] = gf_xor3_(c[5], __builtin_amdgcn_alignbit(T13, T12, 9), __builtin_amdgcn_alignbit(T10, T9, 31));
  r.w[6] = gf_xor3_(c[6], __builtin_amdgcn_alignbit(T14, T13, 9), __builtin_amdgcn_alignbit(T11, T10, 31));
  r.w[7] = gf_xor3_(c[7], TOP >= 15 ? __builtin_amdgcn_alignbit(T15, T14, 9) : T14 >> 9, __builtin_amdgcn_alignbit(T12, T11, 31));
  const uint32_t t = r.w[7] >> 9;  // bits 233 .. 255
  r.w[0] ^= t;
  r.w[2] ^= t << 10;
  r.w[3] ^= t >> 22;
  r.w[7] &= 0x1FFu;
  return r;
}

GF_DEV Gf gf_mul(const Gf& a, const Gf& b) {
  uint32_t acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0;
  // bits 9..31: a.w[7] has only 9 significant bits, so its rows are skipped here
#pragma unroll 1
  for (int k = 31; k >= 9; --k) {
#pragma unroll
    for (int i = 14; i > 0; --i) acc[i] = __builtin_amdgcn_alignbit(acc[i], acc[i - 1], 31);
    acc[0] <<= 1;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)a.w[j], k, 1);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i + j] = gf_andxor(m, b.w[i], acc[i + j]);
    }
  }
#pragma unroll 1
  for (int k = 8; k >= 0; --k) {
#pragma unroll
    for (int i = 14; i > 0; --i) acc[i] = __builtin_amdgcn_alignbit(acc[i], acc[i - 1], 31);
    acc[0] <<= 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)a.w[j], k, 1);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i + j < 15) acc[i + j] = gf_andxor(m, b.w[i], acc[i + j]);
      }
    }
  }
  return gf_reduce16(acc);
}


// ------------------------------------------------------------------------------------------------------
// LDS-comb multiplication (the hot kernels' multiplier; 1.8x the register-only form on MI355X).
// One operand is expanded into a 3-bit window table T[u] = u(z)*b(z), u = 0..7, stored in LDS in a
// lane-interleaved layout so that 64 lanes reading 64 different entries are bank-conflict-free:
//     byte address = region + (half*8 + u)*1024 + lane*16          (half = words 0-3 / 4-7)
// i.e. 16 KB per wave.  The other operand is scanned as 11 three-bit digits per 32-bit word (comb): per
// digit position one accumulator shift (15 x v_alignbit) serves 8 table lookups (2 x ds_read_b128 + 8 xor).
// Per product: ~14 ds_write_b128 + 176 ds_read_b128 + ~1.1k VALU ops, against ~2.8k VALU ops.
// Entry 0 (all zero) is written once per kernel by gf_lds_init.  A kernel using it launches with
// GF_LDS_BYTES_PER_WAVE * waves of dynamic LDS, which caps residency at 8-10 waves per CU.
typedef uint32_t gf_u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned GF_LDS_BYTES_PER_WAVE = 16384;

struct GfLds {
  char* lds;           // base of the block's dynamic LDS
  uint32_t lane_base;  // wave_region + lane*16
};

GF_DEV GfLds gf_lds_init(char* lds_base) {
  GfLds c;
  c.lds = lds_base;
  c.lane_base = (threadIdx.x >> 6) * GF_LDS_BYTES_PER_WAVE + (threadIdx.x & 63) * 16;
  *(gf_u32x4*)(c.lds + c.lane_base) = (gf_u32x4){0, 0, 0, 0};
  *(gf_u32x4*)(c.lds + c.lane_base + 8192) = (gf_u32x4){0, 0, 0, 0};
  return c;
}
GF_DEV void gf_tab_store(const GfLds& c, int u, const uint32_t* w) {
  *(gf_u32x4*)(c.lds + c.lane_base + (uint32_t)u * 1024) = (gf_u32x4){w[0], w[1], w[2], w[3]};
  *(gf_u32x4*)(c.lds + c.lane_base + 8192 + (uint32_t)u * 1024) = (gf_u32x4){w[4], w[5], w[6], w[7]};
}
GF_DEV void gf_shl1_8(const uint32_t* in, uint32_t* out) {
#pragma unroll
  for (int i = 7; i > 0; --i) out[i] = __builtin_amdgcn_alignbit(in[i], in[i - 1], 31);
  out[0] = in[0] << 1;
}
// table of b
GF_DEV void gf_tab_build(const GfLds& c, const Gf& b) {
  uint32_t t1[8], t2[8], t3[8], t4[8], t6[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) t1[i] = b.w[i];
  gf_tab_store(c, 1, t1);
  gf_shl1_8(t1, t2);
  gf_tab_store(c, 2, t2);
#pragma unroll
  for (int i = 0; i < 8; ++i) t3[i] = t2[i] ^ t1[i];
  gf_tab_store(c, 3, t3);
  gf_shl1_8(t2, t4);
  gf_tab_store(c, 4, t4);
  gf_shl1_8(t3, t6);
  gf_tab_store(c, 6, t6);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    t4[i] ^= t1[i];
    t6[i] ^= t1[i];
  }
  gf_tab_store(c, 5, t4);
  gf_tab_store(c, 7, t6);
}
// one digit position: all table reads of the row are issued before the first xor consumes one (the
// compiler otherwise waits on each lookup right after issuing it; +5% on the multiplier microbenchmark)
template <int NW>
GF_DEV void gf_tab_row(uint32_t* acc, const Gf& a, const GfLds& c, int rsh, int lsh) {
  gf_u32x4 lo[NW], hi[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    uint32_t sh = (a.w[j] >> rsh) << lsh;
    uint32_t addr = (sh & 0x1C00u) | c.lane_base;
    lo[j] = *(const gf_u32x4*)(c.lds + addr);
    hi[j] = *(const gf_u32x4*)(c.lds + addr + 8192);
  }
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    acc[j + 0] ^= lo[j].x; acc[j + 1] ^= lo[j].y; acc[j + 2] ^= lo[j].z; acc[j + 3] ^= lo[j].w;
    acc[j + 4] ^= hi[j].x; acc[j + 5] ^= hi[j].y; acc[j + 6] ^= hi[j].z;
    if (j + 7 < 15) acc[j + 7] ^= hi[j].w;
  }
}
GF_DEV void gf_acc_shl3(uint32_t* acc) {
#pragma unroll
  for (int i = 14; i > 0; --i) acc[i] = __builtin_amdgcn_alignbit(acc[i], acc[i - 1], 29);
  acc[0] <<= 3;
}
// a * (operand whose table is currently in LDS)
GF_DEV Gf gf_mul_tab(const Gf& a, const GfLds& c) {
  uint32_t acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0;
  // digit k of every word = bits [3k, 3k+3); k = 10 is the 2-bit top digit; a.w[7] has digits 0..2 only
  gf_tab_row<7>(acc, a, c, 20, 0);  // k = 10: (w >> 30) << 10 == (w >> 20) & 0xC00
#pragma unroll 1
  for (int k = 9; k >= 4; --k) {
    gf_acc_shl3(acc);
    gf_tab_row<7>(acc, a, c, 3 * k - 10, 0);
  }
  gf_acc_shl3(acc);
  gf_tab_row<7>(acc, a, c, 0, 1);   // k = 3: bits 9..11 -> << 1
#pragma unroll 1
  for (int k = 2; k >= 0; --k) {
    gf_acc_shl3(acc);
    gf_tab_row<8>(acc, a, c, 0, 10 - 3 * k);
  }
  return gf_reduce16(acc);
}
GF_DEV Gf gf_mul(const Gf& a, const Gf& b, const GfLds& c) {
  gf_tab_build(c, b);
  return gf_mul_tab(a, c);
}

// ---- quad-cooperative product (latency-bound stages: deep merge levels, the Frobenius tail) ------------------
// The 4 lanes of a DPP quad hold the SAME operands and compute ONE product together: lane r scans only the comb
// digits k = 3r..3r+2 of every word of a (24 table lookups instead of 88), shifts its partial sum by 9r bits, and the
// quad XOR-reduces the four partial sums with two v_xor_dpp quad_perm steps per word.  ~0.4x the instructions of
// gf_mul_tab on the critical path; all four lanes end up with the full product.  Every lane of the quad must be
// active (callers retire whole quads).  Each lane still keeps its own table copy, so the LDS layout is unchanged.
struct GfLdsQ {
  GfLds l;
  uint32_t r;  // lane within the quad
};
GF_DEV GfLdsQ gf_ldsq_init(char* lds_base) {
  GfLdsQ q;
  q.l = gf_lds_init(lds_base);
  q.r = threadIdx.x & 3u;
  return q;
}
GF_DEV uint32_t gf_quad_xor(uint32_t x) {
  x ^= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  x ^= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  return x;
}
GF_DEV void gf_tab_build(const GfLdsQ& c, const Gf& b) { gf_tab_build(c.l, b); }
GF_DEV Gf gf_mul_tab(const Gf& a, const GfLdsQ& c) {
  uint32_t acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0;
#pragma unroll
  for (int t = 2; t >= 0; --t) {
    if (t != 2) gf_acc_shl3(acc);
    const uint32_t bit = 9u * c.r + 3u * (uint32_t)t;          // 3k; k = 11 (lane 3, t = 2) does not exist
    const uint32_t mask = (t == 2 && c.r == 3u) ? 0u : 0x1C00u;
    gf_u32x4 lo[8], hi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t addr = (((a.w[j] >> (bit & 31u)) << 10) & mask) | c.l.lane_base;
      lo[j] = *(const gf_u32x4*)(c.l.lds + addr);
      hi[j] = *(const gf_u32x4*)(c.l.lds + addr + 8192);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[j + 0] ^= lo[j].x; acc[j + 1] ^= lo[j].y; acc[j + 2] ^= lo[j].z; acc[j + 3] ^= lo[j].w;
      acc[j + 4] ^= hi[j].x; acc[j + 5] ^= hi[j].y; acc[j + 6] ^= hi[j].z;
      if (j + 7 < 15) acc[j + 7] ^= hi[j].w;
    }
  }
  // partial sum of lane r sits 9r bits up
  const uint32_t sh = 32u - 9u * c.r;
  const bool r0 = c.r == 0u;
#pragma unroll
  for (int i = 14; i > 0; --i) {
    uint32_t v = __builtin_amdgcn_alignbit(acc[i], acc[i - 1], sh);
    acc[i] = r0 ? acc[i] : v;
  }
  acc[0] = r0 ? acc[0] : (acc[0] << (9u * c.r));
#pragma unroll
  for (int i = 0; i < 15; ++i) acc[i] = gf_quad_xor(acc[i]);
  return gf_reduce16(acc);
}
GF_DEV Gf gf_mul(const Gf& a, const Gf& b, const GfLdsQ& c) {
  gf_tab_build(c.l, b);
  return gf_mul_tab(a, c);
}

// ---- sixteen lanes per product (round 4: the deepest merge levels and the doubling tail) ----------------------------------------
// A lone wave issues one VALU instruction every ~4.5 cycles whatever its lanes do, so a dependent product costs its INSTRUCTION
// COUNT: ~450 for the quad form.  Here the 16 lanes of a DPP row hold the same operands and lane h = 4 kg + jg scans the comb
// digits 3 kg .. 3 kg + 2 of the two words 2 jg, 2 jg + 1 of a only -- 6 lookups instead of 24 -- into a 10-word partial sum that sits
// 9 kg bits and 2 jg words up.  The partial sums of equal jg are XOR-reduced over kg with two row rotations (by 4 and by 8 lanes) per
// word; the four jg partials, which differ by whole words, are gathered with quad broadcasts at compile-time word offsets.  ~290
// instructions; every lane of the row ends up with the product.  Each lane still keeps its own table copy (the LDS layout of the
// quad form), and every lane of a row must be active.
struct GfLdsH {
  GfLds l;
  uint32_t r;  // lane within its row of 16
};
GF_DEV GfLdsH gf_ldsh_init(char* lds_base) {
  GfLdsH q;
  q.l = gf_lds_init(lds_base);
  q.r = threadIdx.x & 15u;
  return q;
}
template <int CTRL>
GF_DEV uint32_t gf_dpp(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, CTRL, 0xF, 0xF, true);
}
GF_DEV void gf_tab_build(const GfLdsH& c, const Gf& b) { gf_tab_build(c.l, b); }
GF_DEV Gf gf_mul_tab(const Gf& a, const GfLdsH& c) {
  const uint32_t kg = c.r >> 2, jg = c.r & 3u;
  // the lane's two words of a, selected with masks (a ternary chain becomes an indexed load from a scratch copy of a)
  const uint32_t m0 = 0u - (uint32_t)(jg == 0u), m1 = 0u - (uint32_t)(jg == 1u), m2 = 0u - (uint32_t)(jg == 2u), m3 = 0u - (uint32_t)(jg == 3u);
  const uint32_t aw0 = (a.w[0] & m0) | (a.w[2] & m1) | (a.w[4] & m2) | (a.w[6] & m3);
  const uint32_t aw1 = (a.w[1] & m0) | (a.w[3] & m1) | (a.w[5] & m2) | (a.w[7] & m3);
  uint32_t acc[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = 0;
#pragma unroll
  for (int t = 2; t >= 0; --t) {
    if (t != 2) {
#pragma unroll
      for (int i = 9; i > 0; --i) acc[i] = __builtin_amdgcn_alignbit(acc[i], acc[i - 1], 29);
      acc[0] <<= 3;
    }
    const uint32_t bit = 9u * kg + 3u * (uint32_t)t;                    // 3k; k = 11 (kg = 3, t = 2) does not exist
    const uint32_t mask = (t == 2 && kg == 3u) ? 0u : 0x1C00u;
    const uint32_t ad0 = (((aw0 >> (bit & 31u)) << 10) & mask) | c.l.lane_base;
    const uint32_t ad1 = (((aw1 >> (bit & 31u)) << 10) & mask) | c.l.lane_base;
    const gf_u32x4 lo0 = *(const gf_u32x4*)(c.l.lds + ad0), hi0 = *(const gf_u32x4*)(c.l.lds + ad0 + 8192);
    const gf_u32x4 lo1 = *(const gf_u32x4*)(c.l.lds + ad1), hi1 = *(const gf_u32x4*)(c.l.lds + ad1 + 8192);
    asm volatile("" ::: "memory");
    acc[0] ^= lo0.x; acc[1] ^= lo0.y ^ lo1.x; acc[2] ^= lo0.z ^ lo1.y; acc[3] ^= lo0.w ^ lo1.z;
    acc[4] ^= hi0.x ^ lo1.w; acc[5] ^= hi0.y ^ hi1.x; acc[6] ^= hi0.z ^ hi1.y; acc[7] ^= hi0.w ^ hi1.z;
    acc[8] ^= hi1.w;
  }
  // the partial sum of group kg sits 9 kg bits up
  const uint32_t sh = 32u - 9u * kg;
  const bool k0 = kg == 0u;
#pragma unroll
  for (int i = 9; i > 0; --i) {
    const uint32_t v = __builtin_amdgcn_alignbit(acc[i], acc[i - 1], sh);
    acc[i] = k0 ? acc[i] : v;
  }
  acc[0] = k0 ? acc[0] : (acc[0] << (9u * kg));
  // over kg: lanes h, h + 4, h + 8, h + 12 of the row
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    acc[i] ^= gf_dpp<0x124>(acc[i]);  // row_ror:4
    acc[i] ^= gf_dpp<0x128>(acc[i]);  // row_ror:8
  }
  // over jg: the partial of lane q of each quad sits 2 q words up
  uint32_t f[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) f[i] = 0;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    f[i] ^= gf_dpp<0x00>(acc[i]);      // quad_perm [0,0,0,0]
    f[2 + i] ^= gf_dpp<0x55>(acc[i]);  // [1,1,1,1]
    f[4 + i] ^= gf_dpp<0xAA>(acc[i]);  // [2,2,2,2]
    f[6 + i] ^= gf_dpp<0xFF>(acc[i]);  // [3,3,3,3]
  }
  return gf_reduce16(f);
}
GF_DEV Gf gf_mul(const Gf& a, const Gf& b, const GfLdsH& c) {
  gf_tab_build(c.l, b);
  return gf_mul_tab(a, c);
}
GF_DEV void gf_mul2(const Gf& a1, const Gf& a2, const Gf& b, const GfLdsH& c, Gf& r1, Gf& r2) {
  gf_tab_build(c.l, b);
  r1 = gf_mul_tab(a1, c);
  r2 = gf_mul_tab(a2, c);
}

// two products against the same operand b: the table of b is built once
GF_DEV void gf_mul2(const Gf& a1, const Gf& a2, const Gf& b, const GfLds& c, Gf& r1, Gf& r2) {
  gf_tab_build(c, b);
  r1 = gf_mul_tab(a1, c);
  r2 = gf_mul_tab(a2, c);
}
GF_DEV void gf_mul2(const Gf& a1, const Gf& a2, const Gf& b, const GfLdsQ& c, Gf& r1, Gf& r2) {
  gf_tab_build(c.l, b);
  r1 = gf_mul_tab(a1, c);
  r2 = gf_mul_tab(a2, c);
}

// ------------------------------------------------------------------------------------------------------
// Karatsuba over the LDS comb (the throughput kernels' multiplier).  a = a0 + a1 z^117, b = b0 + b1 z^117:
//     a b = L + (M + L + H) z^117 + H z^234,   L = a0 b0,  H = a1 b1,  M = (a0 + a1)(b0 + b1).
// The three half-products run against 3-bit window tables of 117-bit operands, whose entries u(z) * bh(z) have 119
// bits = FOUR words: one ds_read_b128 per lookup instead of two, 120 lookups per product instead of 176, and 8 KB of
// LDS per wave instead of 16 KB -- which is what lets the EC kernels run 3-4 waves per SIMD instead of 2 (the comb is
// LDS-pipe-bound at 2 waves: 176 x 4 + 14 x 13 LDS cycles per wave-product against 120 x 4 + 21 x 13 here).
// Measured on MI355X (tools/ubench/gfmul_kara.hip): 31.0 G products/s for the 16 KB comb, 38.0 / 39.7 G/s for this
// form at 3 / 4 waves per SIMD.  Details that matter on gfx950: lookups of a row are combined with 3-input xors
// (v_bitop3 0x96: 8 ops for 16 loaded words), and LDS addresses are kept as 32-bit integers so that a lookup address
// is one shift + one bitop3 ((x & 0x1C00) | lane_base) with no pointer add.
// Layout: byte address = wave region + u * 1024 + lane * 16, u = 0..7 (row 0 stays zero).
constexpr unsigned GF_LDSK_BYTES_PER_WAVE = 8192;
typedef __attribute__((address_space(3))) gf_u32x4 gf_lds_u32x4;
GF_DEV gf_u32x4 gf_lds_ld(uint32_t addr) { return *(const gf_lds_u32x4*)addr; }
GF_DEV void gf_lds_st(uint32_t addr, gf_u32x4 v) { *(gf_lds_u32x4*)addr = v; }
GF_DEV uint32_t gf_xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

struct GfLdsK {
  uint32_t lane_base;  // LDS byte address of this lane's row-0 entry (wave regions are 8 KB aligned)
};
GF_DEV GfLdsK gf_ldsk_init(char* lds_base) {
  GfLdsK c;
  uint32_t b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_base;
  c.lane_base = b + (threadIdx.x >> 6) * GF_LDSK_BYTES_PER_WAVE + (threadIdx.x & 63) * 16;
  gf_lds_st(c.lane_base, (gf_u32x4){0, 0, 0, 0});
  return c;
}
GF_DEV void gf_k_shl1(const uint32_t* in, uint32_t* out) {
#pragma unroll
  for (int i = 3; i > 0; --i) out[i] = __builtin_amdgcn_alignbit(in[i], in[i - 1], 31);
  out[0] = in[0] << 1;
}
GF_DEV void gf_k_store(const GfLdsK& c, int u, const uint32_t* w) {
  gf_lds_st(c.lane_base + (uint32_t)u * 1024, (gf_u32x4){w[0], w[1], w[2], w[3]});
}
// table of a half operand (<= 117 bits in 4 words)
GF_DEV void gf_k_tab_build(const GfLdsK& c, const uint32_t* b) {
  uint32_t t2[4], t3[4], t4[4], t6[4];
  gf_k_store(c, 1, b);
  gf_k_shl1(b, t2);
  gf_k_store(c, 2, t2);
#pragma unroll
  for (int i = 0; i < 4; ++i) t3[i] = t2[i] ^ b[i];
  gf_k_store(c, 3, t3);
  gf_k_shl1(t2, t4);
  gf_k_store(c, 4, t4);
  gf_k_shl1(t3, t6);
  gf_k_store(c, 6, t6);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    t4[i] ^= b[i];
    t6[i] ^= b[i];
  }
  gf_k_store(c, 5, t4);
  gf_k_store(c, 7, t6);
}
// one digit position of a half operand: NW lookups (all issued before the first xor), 4 NW words into NW + 3.
// Round 5: in two halves, so that a row is  lookups issued -> accumulator shifted -> lookups xored in: the shift (7 funnel shifts,
// ~35 cycles of the SIMD) does not depend on the lookups and runs while the LDS serves them; hipcc's own order was shift, then
// lookups, then wait (gf_k_pin keeps it from moving the shift back up).
template <int NW>
GF_DEV void gf_k_row_ld(gf_u32x4* v, const uint32_t* a, const GfLdsK& c, int rsh, int lsh) {
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    uint32_t sh = (a[j] >> rsh) << lsh;
    v[j] = gf_lds_ld(__builtin_amdgcn_bitop3_b32(sh, 0x1C00u, c.lane_base, 0xEA));  // (sh & 0x1C00) | lane_base
  }
  asm volatile("" ::: "memory");
}
template <int NW>
GF_DEV void gf_k_row_acc(uint32_t* acc, const gf_u32x4* v) {
  acc[0] ^= v[0].x;
  acc[1] = gf_xor3(acc[1], v[0].y, v[1].x);
  if (NW == 4) {
    acc[2] = gf_xor3(acc[2], v[0].z, v[1].y) ^ v[2].x;
    acc[3] = gf_xor3(gf_xor3(acc[3], v[0].w, v[1].z), v[2].y, v[3].x);
    acc[4] = gf_xor3(acc[4], v[1].w, v[2].z) ^ v[3].y;
    acc[5] = gf_xor3(acc[5], v[2].w, v[3].z);
    acc[6] ^= v[3].w;
  } else {
    acc[2] = gf_xor3(acc[2], v[0].z, v[1].y) ^ v[2].x;
    acc[3] = gf_xor3(acc[3], v[0].w, v[1].z) ^ v[2].y;
    acc[4] = gf_xor3(acc[4], v[1].w, v[2].z);
    acc[5] ^= v[2].w;
  }
}
template <int NW>
GF_DEV void gf_k_row(uint32_t* acc, const uint32_t* a, const GfLdsK& c, int rsh, int lsh) {
  gf_u32x4 v[NW];
  gf_k_row_ld<NW>(v, a, c, rsh, lsh);
  gf_k_row_acc<NW>(acc, v);
}
// the accumulator words as they are, as far as the optimiser is concerned, only from here on
template <int N>
GF_DEV void gf_k_pin(uint32_t* acc) {
  if (N == 6) asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]));
  else if (N == 7) asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]));
  else asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
}
// the low N words of the accumulator, three bits up (the words above them are still zero, see gf_k_mul_tab)
template <int N = 8>
GF_DEV void gf_k_shl3(uint32_t* acc) {
#pragma unroll
  for (int i = N - 1; i > 0; --i) acc[i] = __builtin_amdgcn_alignbit(acc[i], acc[i - 1], 29);
  acc[0] <<= 3;
}
// lookups of a row, then the shift of the N filled words, then the xors
template <int NW, int N>
GF_DEV void gf_k_shl3_row(uint32_t* acc, const uint32_t* a, const GfLdsK& c, int rsh, int lsh) {
  gf_u32x4 v[NW];
  gf_k_row_ld<NW>(v, a, c, rsh, lsh);
  gf_k_pin<N>(acc);
  gf_k_shl3<N>(acc);
  gf_k_row_acc<NW>(acc, v);
}
// acc[0..7] = a (4 words, a[3] < 2^21) * (half operand whose table is in LDS); digit k of a word = bits [3k, 3k+3),
// k = 10 is the 2-bit top digit, a[3] has digits 0..6 only.
// Round 5: the accumulator is shifted only as far up as it is filled.  An entry u(z) bh(z) has 119 bits, so the three-word rows
// k = 10 .. 7 reach bit 183 and each shift adds three: 186, 189, 192 bits after the shifts in front of rows 9, 8, 7 -- words 6 and 7
// are still zero and six words are shifted; the four-word rows reach bit 215, the shifts in front of rows 6 .. 3 leave at most
// 195, 218, 221, 224 bits (seven words), and only the last three fill word 7 (227, 230, 233 bits).  60 funnel shifts per half
// product instead of 70 -- half-rate instructions (v_alignbit_b32), 30 fewer of ~460 per product.
// (GF_K_UNROLL = 1 -- the three-trip row loops unrolled, four scalar loop instructions per row fewer -- was measured in round 5:
// the pair rounds ran EIGHT times slower (162 against 20 ms per proof; msm.o 2.1 -> 2.8 MB): the rolled loops are what keeps the
// inlined products inside the register budget and the instruction cache)
#ifndef GF_K_UNROLL
#define GF_K_UNROLL 0
#endif
#if GF_K_UNROLL
#define GF_K_LOOP _Pragma("unroll")
#else
#define GF_K_LOOP _Pragma("unroll 1")
#endif
GF_DEV void gf_k_rows_9_0(uint32_t* acc, const uint32_t* a, const GfLdsK& c);
GF_DEV void gf_k_mul_tab(uint32_t* acc, const uint32_t* a, const GfLdsK& c) {
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0;
  gf_k_row<3>(acc, a, c, 20, 0);  // k = 10: (w >> 30) << 10 == (w >> 20) & 0xC00
  gf_k_rows_9_0(acc, a, c);
}
// Table build and the first row of the product in one (round 5).  Row k = 10 looks up the 2-bit top digits of the words 0 .. 2:
// entries 0 .. 3 only.  A wave's LDS operations execute in order, so with the row's three lookups placed right behind the stores of
// the entries 1, 2, 3 their data is back four stores (4 x 13 LDS cycles) sooner than behind all seven, and the other four stores
// run under the row's xors -- every half product used to start with the whole table's stores in front of its first lookups.
GF_DEV void gf_k_tab_build_mul(uint32_t* acc, const uint32_t* a, const uint32_t* b, const GfLdsK& c) {
  uint32_t t2[4], t3[4], t4[4], t6[4];
  gf_k_store(c, 1, b);
  gf_k_shl1(b, t2);
  gf_k_store(c, 2, t2);
#pragma unroll
  for (int i = 0; i < 4; ++i) t3[i] = t2[i] ^ b[i];
  gf_k_store(c, 3, t3);
  asm volatile("" ::: "memory");
  gf_u32x4 v[3];
  gf_k_row_ld<3>(v, a, c, 20, 0);  // k = 10: (w >> 30) << 10 == (w >> 20) & 0xC00
  gf_k_shl1(t2, t4);
  gf_k_store(c, 4, t4);
  gf_k_shl1(t3, t6);
  gf_k_store(c, 6, t6);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    t4[i] ^= b[i];
    t6[i] ^= b[i];
  }
  gf_k_store(c, 5, t4);
  gf_k_store(c, 7, t6);
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0;
  gf_k_row_acc<3>(acc, v);
  gf_k_rows_9_0(acc, a, c);
}
GF_DEV void gf_k_rows_9_0(uint32_t* acc, const uint32_t* a, const GfLdsK& c) {
  // (the loops count the shift amounts themselves: one scalar addition per row instead of two)
GF_K_LOOP
  for (int rsh = 17; rsh >= 11; rsh -= 3) gf_k_shl3_row<3, 6>(acc, a, c, rsh, 0);  // k = 9, 8, 7: 3 k - 10
GF_K_LOOP
  for (int rsh = 8; rsh >= 2; rsh -= 3) gf_k_shl3_row<4, 7>(acc, a, c, rsh, 0);  // k = 6, 5, 4
  gf_k_shl3_row<4, 7>(acc, a, c, 0, 1);  // k = 3: bits 9..11 -> << 1
GF_K_LOOP
  for (int lsh = 4; lsh <= 10; lsh += 3) gf_k_shl3_row<4, 8>(acc, a, c, 0, lsh);  // k = 2, 1, 0: 10 - 3 k
}
// x = lo + hi z^117
GF_DEV void gf_k_split(const Gf& x, uint32_t* lo, uint32_t* hi) {
  lo[0] = x.w[0]; lo[1] = x.w[1]; lo[2] = x.w[2]; lo[3] = x.w[3] & 0x1FFFFFu;
#pragma unroll
  for (int i = 0; i < 4; ++i) hi[i] = __builtin_amdgcn_alignbit(i + 4 < 8 ? x.w[i + 4] : 0u, x.w[i + 3], 21);
}
// L + (M + L + H) z^117 + H z^234, reduced.
// Round 5: assembled and reduced word by word instead of through a 16-word buffer and the general gf_reduce16.  With reduced
// operands L and M + L + H have at most 233 bits and H (the product of two 116-bit halves) 231, so the words 11 and 15 the general
// route computed and folded are zero; the unreduced words 8 .. 14 are T8 .. T14 below (T8 .. T10 take the z^74 folds of T12 .. T14
// first), and a word's two shifted neighbours never overlap, so each fold is ONE funnel shift:
//   z^233 = 1   : word j -> bit 32 (j - 8) + 23:  o[i] ^= alignbit(T[i+8], T[i+7], 9)
//   z^233 = z^74: word j -> bit 32 (j - 5) + 1 :  o[i] ^= alignbit(T[i+5], T[i+4], 31)
// ~65 instructions where the buffer route took ~115 (the same 0.7 % of a proof again as the accumulator shifts above).
GF_DEV Gf gf_k_combine(const uint32_t* L, const uint32_t* H, uint32_t* M) {
#pragma unroll
  for (int i = 0; i < 8; ++i) M[i] = gf_xor3(M[i], L[i], H[i]);
  uint32_t ms[8], hs[8];  // ms[i] = word 3 + i of (M + L + H) z^117, hs[i] = word 7 + i of H z^234
  ms[0] = M[0] << 21;
  hs[0] = H[0] << 10;
#pragma unroll
  for (int i = 1; i < 8; ++i) {
    ms[i] = __builtin_amdgcn_alignbit(M[i], M[i - 1], 11);
    hs[i] = __builtin_amdgcn_alignbit(H[i], H[i - 1], 22);
  }
  const uint32_t T14 = hs[7], T13 = hs[6], T12 = hs[5], T11 = hs[4];
  const uint32_t T10 = gf_xor3(ms[7], hs[3], T14 >> 31);
  const uint32_t T9 = gf_xor3(ms[6], hs[2], __builtin_amdgcn_alignbit(T14, T13, 31));
  const uint32_t T8 = gf_xor3(ms[5], hs[1], __builtin_amdgcn_alignbit(T13, T12, 31));
  Gf r;
  r.w[0] = L[0] ^ (T8 << 23);
  r.w[1] = L[1] ^ __builtin_amdgcn_alignbit(T9, T8, 9);
  r.w[2] = L[2] ^ __builtin_amdgcn_alignbit(T10, T9, 9);
  r.w[3] = gf_xor3(L[3], ms[0], __builtin_amdgcn_alignbit(T11, T10, 9)) ^ (T8 << 1);
  r.w[4] = gf_xor3(L[4], ms[1], __builtin_amdgcn_alignbit(T12, T11, 9)) ^ __builtin_amdgcn_alignbit(T9, T8, 31);
  r.w[5] = gf_xor3(L[5], ms[2], __builtin_amdgcn_alignbit(T13, T12, 9)) ^ __builtin_amdgcn_alignbit(T10, T9, 31);
  r.w[6] = gf_xor3(L[6], ms[3], __builtin_amdgcn_alignbit(T14, T13, 9)) ^ __builtin_amdgcn_alignbit(T11, T10, 31);
  r.w[7] = gf_xor3(gf_xor3(L[7], ms[4], hs[0]), T14 >> 9, __builtin_amdgcn_alignbit(T12, T11, 31));
  const uint32_t t = r.w[7] >> 9;  // bits 233 .. 255
  r.w[0] ^= t;
  r.w[2] ^= t << 10;
  r.w[3] ^= t >> 22;
  r.w[7] &= 0x1FFu;
  return r;
}
GF_DEV Gf gf_mul(const Gf& a, const Gf& b, const GfLdsK& c) {
  uint32_t a0[4], a1[4], b0[4], b1[4];
  gf_k_split(a, a0, a1);
  gf_k_split(b, b0, b1);
  uint32_t L[8], H[8], M[8];
  gf_k_tab_build_mul(L, a0, b0, c);
  gf_k_tab_build_mul(H, a1, b1, c);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a0[i] ^= a1[i];
    b0[i] ^= b1[i];
  }
  gf_k_tab_build_mul(M, a0, b0, c);
  return gf_k_combine(L, H, M);
}
// a1 * b and a2 * b: each half table of b is built once and serves both products
GF_DEV void gf_mul2(const Gf& a1, const Gf& a2, const Gf& b, const GfLdsK& c, Gf& r1, Gf& r2) {
  uint32_t p0[4], p1[4], q0[4], q1[4], b0[4], b1[4];
  gf_k_split(a1, p0, p1);
  gf_k_split(a2, q0, q1);
  gf_k_split(b, b0, b1);
  uint32_t L1[8], H1[8], M1[8], L2[8], H2[8], M2[8];
  gf_k_tab_build_mul(L1, p0, b0, c);
  gf_k_mul_tab(L2, q0, c);
  gf_k_tab_build_mul(H1, p1, b1, c);
  gf_k_mul_tab(H2, q1, c);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    p0[i] ^= p1[i];
    q0[i] ^= q1[i];
    b0[i] ^= b1[i];
  }
  gf_k_tab_build_mul(M1, p0, b0, c);
  gf_k_mul_tab(M2, q0, c);
  r1 = gf_k_combine(L1, H1, M1);
  r2 = gf_k_combine(L2, H2, M2);
}

// 16 bits -> 32 bits with zeros interleaved
GF_DEV uint32_t gf_spread16(uint32_t x) {
  x = (x | (x << 8)) & 0x00FF00FFu;
  x = (x | (x << 4)) & 0x0F0F0F0Fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
}

GF_DEV Gf gf_sqr(const Gf& a) {
  uint32_t c[16];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    c[2 * i] = gf_spread16(a.w[i] & 0xFFFFu);
    c[2 * i + 1] = gf_spread16(a.w[i] >> 16);
  }
  c[14] = gf_spread16(a.w[7] & 0xFFFFu);
  c[15] = 0;
  return gf_reduce16<14>(c);
}

GF_DEV Gf gf_sqr_n(Gf a, int n) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) a = gf_sqr(a);
  return a;
}


// ------------------------------------------------------------------------------------------------------
// Table-driven multi-squaring: x -> x^(2^k) is GF(2)-linear, so for the three long squaring runs of the
// Itoh-Tsujii chain (k = 29, 58, 116: 203 of its 232 squarings) the map is applied as 30 byte-indexed
// lookups into T_k[pos][byte] = (byte * z^(8 pos))^(2^k)  (30 x 256 x 32 B = 240 KB per k, L2-resident).
// Inversion drops from 10 M + 232 S to 10 M + 29 S + 3 table passes.
struct GfSqrTables {
  const Gf* t29;
  const Gf* t58;
  const Gf* t116;
  const Gf* th;  // the half-trace (also GF(2)-linear): th[pos][byte] = H(byte * z^(8 pos))
  // round 6: the runs of 14 and 7 squarings of the chain as table passes too (nullptr = plain squarings; Tune::gf_inv_tabs).  A plain
  // squaring is ~215 VALU instructions, a table pass ~430 + eight round trips to the L2: 14 squarings -> one pass takes 2.6 k of the
  // ~16 k instructions of an inversion, which every pair-round thread pays once per round whatever its slot count
  const Gf* t14;
  const Gf* t7;
};
// Round 5: the lookups of one word of a (four bytes: eight 16-byte loads) are issued together and xored in afterwards, with 3-input
// xors.  The first version walked the bytes in a rolled loop -- load, wait, xor, thirty times: thirty dependent round trips to the L2 per
// pass, three passes per inversion, and every wave of a pair round reaches its inversion at the same time, so nothing hid them.
template <int NB, bool FENCE = true>
GF_DEV void gf_sqr_tab_word(Gf& r, uint32_t word, const Gf* __restrict__ Tw) {  // NB byte lookups of one word, issued together
  gf_u32x4 lo[4], hi[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (b < NB) {
      const uint32_t byte = (word >> (8 * b)) & 0xFFu;
      const gf_u32x4* e = (const gf_u32x4*)(Tw + b * 256 + byte);
      lo[b] = e[0];
      hi[b] = e[1];
    } else {
      lo[b] = (gf_u32x4){0, 0, 0, 0};
      hi[b] = (gf_u32x4){0, 0, 0, 0};
    }
  }
  if (FENCE) asm volatile("" ::: "memory");  // (keeps the next word's loads behind this word's: the register budget of a throughput kernel)
#pragma unroll
  for (int b = 0; b < 4; b += 2) {
    r.w[0] = gf_xor3_(r.w[0], lo[b].x, lo[b + 1].x); r.w[1] = gf_xor3_(r.w[1], lo[b].y, lo[b + 1].y);
    r.w[2] = gf_xor3_(r.w[2], lo[b].z, lo[b + 1].z); r.w[3] = gf_xor3_(r.w[3], lo[b].w, lo[b + 1].w);
    r.w[4] = gf_xor3_(r.w[4], hi[b].x, hi[b + 1].x); r.w[5] = gf_xor3_(r.w[5], hi[b].y, hi[b + 1].y);
    r.w[6] = gf_xor3_(r.w[6], hi[b].z, hi[b + 1].z); r.w[7] = gf_xor3_(r.w[7], hi[b].w, hi[b + 1].w);
  }
}
GF_DEV Gf gf_sqr_tab(const Gf& a, const Gf* __restrict__ T) {
  Gf r = gf_zero();
  Gf t = a;  // walked by ROTATING the words through t.w[0] (a runtime word index would push a to scratch; unrolling the words lets the
             // scheduler pull sixty loads and their 64-bit addresses in front of the products around them: 390 B of scratch per lane)
#pragma unroll 1
  for (int w = 0; w < 7; ++w) {
    gf_sqr_tab_word<4>(r, t.w[0], T + (size_t)w * 1024);
#pragma unroll
    for (int i = 0; i < 7; ++i) t.w[i] = t.w[i + 1];
  }
  gf_sqr_tab_word<2>(r, t.w[0], T + 7 * 1024);  // bits 224 .. 232: two bytes
  return r;
}
// the same for a latency CHAIN (one point per quad / row of lanes on a lone wave: k_tail, the deep merge levels): the words unrolled, so
// that the scheduler issues the sixty loads of a pass as early as its registers allow -- two or three round trips to the L2 per pass
// instead of eight (a kernel at 3 waves per SIMD has no registers for that: the note in gf_sqr_tab)
GF_DEV Gf gf_sqr_tab_wide(const Gf& a, const Gf* __restrict__ T) {
  Gf r = gf_zero();
#pragma unroll
  for (int w = 0; w < 7; ++w) gf_sqr_tab_word<4, false>(r, a.w[w], T + (size_t)w * 1024);
  gf_sqr_tab_word<2, false>(r, a.w[7], T + 7 * 1024);
  return r;
}
template <class LT> struct GfIsChain { static constexpr bool value = false; };
template <> struct GfIsChain<GfLdsQ> { static constexpr bool value = true; };
template <> struct GfIsChain<GfLdsH> { static constexpr bool value = true; };
template <class LT>
GF_DEV Gf gf_sqr_tab(const Gf& a, const Gf* __restrict__ T, const LT&) {
  return GfIsChain<LT>::value ? gf_sqr_tab_wide(a, T) : gf_sqr_tab(a, T);
}
// a^(2^k) for any k: table passes for the 116 / 58 / 29 parts, plain squarings for the rest (< 29)
GF_DEV Gf gf_sqr_n_fast(Gf a, int k, const GfSqrTables& T) {
  while (k >= 116) { a = gf_sqr_tab(a, T.t116); k -= 116; }
  if (k >= 58) { a = gf_sqr_tab(a, T.t58); k -= 58; }
  if (k >= 29) { a = gf_sqr_tab(a, T.t29); k -= 29; }
  return gf_sqr_n(a, k);
}
// a^(2^233-2) with table-driven runs; products through the LDS multiplier.  a == 0 -> 0.
template <class LT>
GF_DEV Gf gf_inv_fast(const Gf& a, const GfSqrTables& T, const LT& L) {
  Gf b1 = a;
  Gf b2 = gf_mul(gf_sqr(b1), b1, L);
  Gf b3 = gf_mul(gf_sqr(b2), b1, L);
  Gf b6 = gf_mul(gf_sqr_n(b3, 3), b3, L);
  Gf b7 = gf_mul(gf_sqr(b6), b1, L);
  Gf b14 = gf_mul(T.t7 ? gf_sqr_tab(b7, T.t7, L) : gf_sqr_n(b7, 7), b7, L);
  Gf b28 = gf_mul(T.t14 ? gf_sqr_tab(b14, T.t14, L) : gf_sqr_n(b14, 14), b14, L);
  Gf b29 = gf_mul(gf_sqr(b28), b1, L);
  Gf b58 = gf_mul(gf_sqr_tab(b29, T.t29, L), b29, L);
  Gf b116 = gf_mul(gf_sqr_tab(b58, T.t58, L), b58, L);
  Gf b232 = gf_mul(gf_sqr_tab(b116, T.t116, L), b116, L);
  return gf_sqr(b232);
}

// Itoh-Tsujii inversion a^(2^233 - 2): 10 multiplications + 232 squarings.  a == 0 -> 0.
GF_DEV Gf gf_inv(const Gf& a) {
  Gf b1 = a;
  Gf b2 = gf_mul(gf_sqr(b1), b1);
  Gf b3 = gf_mul(gf_sqr(b2), b1);
  Gf b6 = gf_mul(gf_sqr_n(b3, 3), b3);
  Gf b7 = gf_mul(gf_sqr(b6), b1);
  Gf b14 = gf_mul(gf_sqr_n(b7, 7), b7);
  Gf b28 = gf_mul(gf_sqr_n(b14, 14), b14);
  Gf b29 = gf_mul(gf_sqr(b28), b1);
  Gf b58 = gf_mul(gf_sqr_n(b29, 29), b29);
  Gf b116 = gf_mul(gf_sqr_n(b58, 58), b58);
  Gf b232 = gf_mul(gf_sqr_n(b116, 116), b116);
  return gf_sqr(b232);
}

GF_DEV Gf gf_sqrt(const Gf& a) { return gf_sqr_n(a, 232); }

// Tr(a): for z^233+z^74+1, Tr(z^i) = 1 exactly for i in {0, 159}
GF_DEV uint32_t gf_trace(const Gf& a) { return (a.w[0] ^ (a.w[4] >> 31)) & 1u; }

// half-trace H(c) = sum_{i=0}^{116} c^(4^i): a root of z^2 + z = c when Tr(c) = 0
GF_DEV Gf gf_halftrace(const Gf& c) {
  Gf h = c, x = c;
#pragma unroll 1
  for (int i = 0; i < 116; ++i) {
    x = gf_sqr(gf_sqr(x));
    h = gf_add(h, x);
  }
  return h;
}

}  // namespace dvp
