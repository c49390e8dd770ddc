// The 30-byte presentation of an encoded point (shared by codec.hip's kernels and the MSM tail, which encodes its own result).
#pragma once
#include "k233.cuh"

namespace dvp {

// ---- the encoding RULE as a run-time constant (dvp_codec_set_rule) -------------------------------------------------------
// What is settled (tools/pin_xsk233.py): a point travels as ONE field element of 30 bytes from which the
// decode equation x^2 + (w^2 + w + a) x + b = 0 recovers it, and w = y'/x on the N = (0,0) model satisfies it.  What one vector
// from xs233 would settle is which equivalent presentation of that element the bytes hold.  Every candidate tools/pin_xsk233.py
// enumerates (8 formulas x 2 views x 2 generator signs x "+1" x 2 byte orders) collapses -- on the curve sqrt(s/x) = y'/x =
// sqrt(lambda), s/x = lambda = (y'/x)^2, and w(Q + N) = w(-Q) = w(Q) + 1 -- into the classes below (the formulas y/x and
// "x alone" are not encodings of this family: their decode is a cubic / needs a sign bit).  rule =
//   bit 0      the element is w + 1  (= the E[r] representative's own value, = the negated generator's convention)
//   bit 1      bytes are big-endian
//   bits 2..3  0: w = sqrt(s/x) itself   1: w^2 (= s/x = lambda)   2: sqrt(w) (= sqrt(y'/x))
// Rule 0 is the candidate followed since round 1 (Pornin, ePrint 2022/1325, as recalled).  The neutral element is 30 zero bytes
// under every rule.  A vector from a machine with cargo turns into `dvp_codec_set_rule(k)` (or DVP_CODEC_RULE=k), not a port.
constexpr int CODEC_RULES = 12;
__device__ __forceinline__ Gf codec_present(Gf w, int rule, const GfSqrTables& T) {
  const int tr = (rule >> 2) & 3;
  if (tr == 1) w = gf_sqr(w);
  else if (tr == 2) w = gf_sqr_tab(gf_sqr_tab(w, T.t116), T.t116);  // sqrt = 232 squarings = two 116-step table passes
  if (rule & 1) w.w[0] ^= 1u;
  return w;
}
__device__ __forceinline__ Gf codec_absorb(Gf t, int rule, const GfSqrTables& T) {
  if (rule & 1) t.w[0] ^= 1u;
  const int tr = (rule >> 2) & 3;
  if (tr == 1) t = gf_sqr_tab(gf_sqr_tab(t, T.t116), T.t116);
  else if (tr == 2) t = gf_sqr(t);
  return t;
}
__device__ __forceinline__ void store30(uint8_t* dst, const Gf& w, int rule) {
  const bool be = (rule & 2) != 0;
#pragma unroll
  for (int b = 0; b < 30; ++b) dst[be ? 29 - b : b] = (uint8_t)(w.w[b >> 2] >> (8 * (b & 3)));
}
__device__ __forceinline__ Gf load30(const uint8_t* src, uint32_t* top_bits, int rule) {
  const bool be = (rule & 2) != 0;
  Gf w = gf_zero();
#pragma unroll
  for (int b = 0; b < 30; ++b) w.w[b >> 2] |= (uint32_t)src[be ? 29 - b : b] << (8 * (b & 3));
  *top_bits = w.w[7] >> 9;
  return w;
}

}  // namespace dvp
