// sect233k1 (K-233): y^2 + xy = x^3 + 1 over GF(2^233), prime-order subgroup E[r].
//
// The reference reaches this curve through xs233's prime-order group xsk233 = { P + N : P in E[r] }
// with N = (0,1) the point of order 2 (src/curve.rs:13,72-158).  P -> P + N is a group isomorphism
// E[r] -> xsk233, so every linear operation of the reference (multi_scalar_mul, point_scalar_mul,
// add) is carried out here on the E[r] representative and only the 30-byte codec moves between the
// two views (codec.hip).
//
// Coordinates: affine (x,y) in HBM for bases; Lopez-Dahab projective (X,Y,Z), x = X/Z, y = Y/Z^2,
// for accumulators.  Z == 0 encodes the point at infinity.  Formula costs (a = 0, b = 1):
//   doubling 3M+5S, mixed addition 8M+5S, full addition 13M+5S, Frobenius 3S.
#pragma once
#include "gf233.cuh"

namespace dvp {

struct Aff {
  Gf x, y;
};
struct Ld {
  Gf X, Y, Z;
};

GF_DEV Ld ld_infinity() {
  Ld r;
  r.X = gf_one();
  r.Y = gf_zero();
  r.Z = gf_zero();
  return r;
}
GF_DEV bool ld_is_inf(const Ld& p) { return gf_is_zero(p.Z); }
GF_DEV Ld ld_from_aff(const Aff& a) {
  Ld r;
  r.X = a.x;
  r.Y = a.y;
  r.Z = gf_one();
  return r;
}
GF_DEV Aff aff_neg(const Aff& a) {
  Aff r;
  r.x = a.x;
  r.y = gf_add(a.x, a.y);
  return r;
}

GF_DEV Ld ld_dbl(const Ld& p) {
  Gf z1s = gf_sqr(p.Z), x1s = gf_sqr(p.X);
  Ld r;
  r.Z = gf_mul(x1s, z1s);
  Gf z1q = gf_sqr(z1s);
  r.X = gf_add(gf_sqr(x1s), z1q);
  r.Y = gf_add(gf_mul(z1q, r.Z), gf_mul(r.X, gf_add(gf_sqr(p.Y), z1q)));
  return r;  // Z1 == 0 -> Z3 == 0; X1 == 0 (the point N, never in E[r]) -> Z3 == 0
}

// p + q, q affine and not infinity.
GF_DEV Ld ld_madd(const Ld& p, const Aff& q) {
  if (ld_is_inf(p)) return ld_from_aff(q);
  Gf z1s = gf_sqr(p.Z);
  Gf A = gf_add(p.Y, gf_mul(q.y, z1s));
  Gf B = gf_add(p.X, gf_mul(q.x, p.Z));
  if (gf_is_zero(B)) {
    if (gf_is_zero(A)) return ld_dbl(ld_from_aff(q));  // p == q
    return ld_infinity();                              // p == -q
  }
  Gf C = gf_mul(p.Z, B);
  Gf D = gf_mul(gf_sqr(B), C);
  Ld r;
  r.Z = gf_sqr(C);
  Gf E = gf_mul(A, C);
  r.X = gf_add(gf_add(gf_sqr(A), D), E);
  Gf F = gf_add(r.X, gf_mul(q.x, r.Z));
  Gf G = gf_mul(gf_add(q.x, q.y), gf_sqr(r.Z));
  r.Y = gf_add(gf_mul(gf_add(E, r.Z), F), G);
  return r;
}

// p + q, both projective
GF_DEV Ld ld_add(const Ld& p, const Ld& q) {
  if (ld_is_inf(p)) return q;
  if (ld_is_inf(q)) return p;
  Gf A1 = gf_mul(q.Y, gf_sqr(p.Z));
  Gf A2 = gf_mul(p.Y, gf_sqr(q.Z));
  Gf B1 = gf_mul(q.X, p.Z);
  Gf B2 = gf_mul(p.X, q.Z);
  Gf C = gf_add(A1, A2);
  Gf D = gf_add(B1, B2);
  if (gf_is_zero(D)) {
    if (gf_is_zero(C)) return ld_dbl(p);
    return ld_infinity();
  }
  Gf E = gf_mul(p.Z, q.Z);
  Gf F = gf_mul(D, E);
  Ld r;
  r.Z = gf_sqr(F);
  Gf Ds = gf_sqr(D);
  Gf G = gf_mul(Ds, F);
  Gf H = gf_mul(C, F);
  r.X = gf_add(gf_add(gf_sqr(C), H), G);
  Gf I = gf_add(gf_mul(gf_mul(Ds, B1), E), r.X);
  Gf J = gf_add(gf_mul(Ds, A1), r.X);
  r.Y = gf_add(gf_mul(H, I), gf_mul(r.Z, J));
  return r;
}


// ---- LDS-multiplier flavours (hot kernels): same formulas, products through one of the LDS multipliers of
// gf233.cuh.  LT = GfLdsK (Karatsuba half tables, the throughput kernels), GfLds (16 KB comb) or GfLdsQ (the four
// lanes of a quad compute one addition together: latency-bound stages).  Products that share an operand go through
// gf_mul2, which builds the shared operand's table(s) once: 6 table sets for the 8 products of the mixed addition,
// 9 for the 13 of the full addition. ----------------
template <class LT>
GF_DEV Ld ld_dbl(const Ld& p, const LT& L) {
  Gf z1s = gf_sqr(p.Z), x1s = gf_sqr(p.X);
  Ld r;
  r.Z = gf_mul(x1s, z1s, L);
  Gf z1q = gf_sqr(z1s);
  r.X = gf_add(gf_sqr(x1s), z1q);
  Gf t = gf_mul(z1q, r.Z, L);
  r.Y = gf_add(t, gf_mul(gf_add(gf_sqr(p.Y), z1q), r.X, L));
  return r;
}

// in-place forms (no aggregate returns through divergent paths: those made hipcc keep the accumulator
// in scratch, 68 B/lane of write+read traffic per addition)
template <class LT>
GF_DEV void ld_madd_ip(Ld& p, const Aff& q, const LT& L) {
  if (ld_is_inf(p)) {
    p.X = q.x;
    p.Y = q.y;
    p.Z = gf_one();
    return;
  }
  Gf A = gf_add(p.Y, gf_mul(q.y, gf_sqr(p.Z), L));
  Gf B = gf_add(p.X, gf_mul(q.x, p.Z, L));
  if (gf_is_zero(B)) {
    if (gf_is_zero(A)) {  // p == q: double the affine point
      Ld t;
      t.X = q.x; t.Y = q.y; t.Z = gf_one();
      t = ld_dbl(t, L);
      p.X = t.X; p.Y = t.Y; p.Z = t.Z;
    } else {              // p == -q
      p.X = gf_one(); p.Y = gf_zero(); p.Z = gf_zero();
    }
    return;
  }
  Gf C = gf_mul(B, p.Z, L);  // Z1 * B
  Gf D, E;
  gf_mul2(gf_sqr(B), A, C, L, D, E);
  Gf Z3 = gf_sqr(C);
  Gf X3 = gf_add(gf_add(gf_sqr(A), D), E);
  Gf F = gf_add(X3, gf_mul(q.x, Z3, L));
  Gf G = gf_mul(gf_add(q.x, q.y), gf_sqr(Z3), L);
  p.Y = gf_add(gf_mul(gf_add(E, Z3), F, L), G);
  p.X = X3;
  p.Z = Z3;
}
// p + q from two AFFINE points, both finite and p.x != q.x (the caller sends everything else down ld_madd_ip): the mixed
// addition with Z1 = 1, where three of its eight products are products by one -- 5M + 3S, and the same (X, Y, Z) as
// ld_madd_ip(ld_from_aff(p), q) word for word.
template <class LT>
GF_DEV void ld_add_aff_aff(const Aff& p, const Aff& q, Ld& r, const LT& L) {
  const Gf A = gf_add(p.y, q.y), B = gf_add(p.x, q.x);
  const Gf Z3 = gf_sqr(B);
  Gf D, E;
  gf_mul2(Z3, A, B, L, D, E);  // B^3, A B
  const Gf X3 = gf_add(gf_add(gf_sqr(A), D), E);
  const Gf F = gf_add(X3, gf_mul(q.x, Z3, L));
  const Gf G = gf_mul(gf_add(q.x, q.y), gf_sqr(Z3), L);
  r.Y = gf_add(gf_mul(gf_add(E, Z3), F, L), G);
  r.X = X3;
  r.Z = Z3;
}
// p + q for a FINITE p and a finite affine q with p != +-q: the body of ld_madd_ip without its exceptional branches (a kernel
// that keeps them pays their registers on every lane).  Returns false -- p untouched -- when p == +-q.
template <class LT>
GF_DEV bool ld_madd_fast(Ld& p, const Aff& q, const LT& L) {
  const Gf A = gf_add(p.Y, gf_mul(q.y, gf_sqr(p.Z), L));
  const Gf B = gf_add(p.X, gf_mul(q.x, p.Z, L));
  if (gf_is_zero(B)) return false;
  const Gf C = gf_mul(B, p.Z, L);
  Gf D, E;
  gf_mul2(gf_sqr(B), A, C, L, D, E);
  const Gf Z3 = gf_sqr(C);
  const Gf X3 = gf_add(gf_add(gf_sqr(A), D), E);
  const Gf F = gf_add(X3, gf_mul(q.x, Z3, L));
  const Gf G = gf_mul(gf_add(q.x, q.y), gf_sqr(Z3), L);
  p.Y = gf_add(gf_mul(gf_add(E, Z3), F, L), G);
  p.X = X3;
  p.Z = Z3;
  return true;
}
template <class LT>
GF_DEV Ld ld_madd(const Ld& p, const Aff& q, const LT& L) {
  Ld r = p;
  ld_madd_ip(r, q, L);
  return r;
}

template <class LT>
GF_DEV void ld_add_ip(Ld& p, const Ld& q, const LT& L) {
  if (ld_is_inf(q)) return;
  if (ld_is_inf(p)) {
    p.X = q.X; p.Y = q.Y; p.Z = q.Z;
    return;
  }
  Gf A1 = gf_mul(q.Y, gf_sqr(p.Z), L);
  Gf A2 = gf_mul(p.Y, gf_sqr(q.Z), L);
  Gf B1, E;
  gf_mul2(q.X, q.Z, p.Z, L, B1, E);
  Gf B2 = gf_mul(p.X, q.Z, L);
  Gf C = gf_add(A1, A2);
  Gf D = gf_add(B1, B2);
  if (gf_is_zero(D)) {
    if (gf_is_zero(C)) {
      Ld t = ld_dbl(p, L);
      p.X = t.X; p.Y = t.Y; p.Z = t.Z;
    } else {
      p.X = gf_one(); p.Y = gf_zero(); p.Z = gf_zero();
    }
    return;
  }
  Gf Ds = gf_sqr(D);
  Gf DB, DA;
  gf_mul2(B1, A1, Ds, L, DB, DA);
  Gf F, I;
  gf_mul2(D, DB, E, L, F, I);
  Gf G, H;
  gf_mul2(Ds, C, F, L, G, H);
  Gf Z3 = gf_sqr(F);
  Gf X3 = gf_add(gf_add(gf_sqr(C), H), G);
  I = gf_add(I, X3);
  Gf J = gf_add(DA, X3);
  p.Y = gf_add(gf_mul(I, H, L), gf_mul(J, Z3, L));
  p.X = X3;
  p.Z = Z3;
}
template <class LT>
GF_DEV Ld ld_add(const Ld& p, const Ld& q, const LT& L) {
  Ld r = p;
  ld_add_ip(r, q, L);
  return r;
}
// ld_add_ip without the inlined doubling: returns false -- p untouched -- when p == q (both finite); the CALLER doubles, from a
// copy it re-reads (as lam_add_ip's callers do): the doubling inside the addition kept all of p alive across it and cost the fan-in
// reducer's quad flavour 364 B of scratch per lane
template <class LT>
GF_DEV bool ld_add_nodbl(Ld& p, const Ld& q, const LT& L) {
  if (ld_is_inf(q)) return true;
  if (ld_is_inf(p)) {
    p.X = q.X; p.Y = q.Y; p.Z = q.Z;
    return true;
  }
  Gf A1 = gf_mul(q.Y, gf_sqr(p.Z), L);
  Gf A2 = gf_mul(p.Y, gf_sqr(q.Z), L);
  Gf B1, E;
  gf_mul2(q.X, q.Z, p.Z, L, B1, E);
  Gf B2 = gf_mul(p.X, q.Z, L);
  Gf C = gf_add(A1, A2);
  Gf D = gf_add(B1, B2);
  if (gf_is_zero(D)) {
    if (gf_is_zero(C)) return false;
    p.X = gf_one(); p.Y = gf_zero(); p.Z = gf_zero();
    return true;
  }
  Gf Ds = gf_sqr(D);
  Gf DB, DA;
  gf_mul2(B1, A1, Ds, L, DB, DA);
  Gf F, I;
  gf_mul2(D, DB, E, L, F, I);
  Gf G, H;
  gf_mul2(Ds, C, F, L, G, H);
  Gf Z3 = gf_sqr(F);
  Gf X3 = gf_add(gf_add(gf_sqr(C), H), G);
  I = gf_add(I, X3);
  Gf J = gf_add(DA, X3);
  p.Y = gf_add(gf_mul(I, H, L), gf_mul(J, Z3, L));
  p.X = X3;
  p.Z = Z3;
  return true;
}

// ---- lambda-projective coordinates (Oliveira, Lopez, Aranha, Rodriguez-Henriquez 2013): (X, L, Z) with x = X / Z and
// lambda = x + y / x = L / Z, kept in an Ld whose Y field holds L; Z == 0 is the point at infinity.  Full addition 11M + 2S against
// Lopez-Dahab's 13M + 5S, doubling 4M + 4S against 3M + 5S: what the merge tree runs on (2^(c+1) full additions per MSM, no
// doublings); the tail goes back to Lopez-Dahab for its doublings.  Formulas for a = 0, checked against the big-int group law
// (tools/lambda_check.py: addition, doubling and both conversions on the big-int oracle; every MSM test runs through the tree).
// With x_i = X_i / Z_i, lambda_i = L_i / Z_i:  A = L1 Z2 + L2 Z1, U = X1 Z2, V = X2 Z1, B = (U + V)^2,
//   X3 = (A U)(A V),  Z3 = (A B Z2) Z1,  L3 = (A V + B)^2 + (A B Z2)(L1 + Z1);
// doubling: T = L^2 + L Z,  X3 = T^2,  Z3 = T Z^2,  L3 = (X Z)^2 + X3 + T (L Z) + Z3.
template <class LT>
GF_DEV void lam_from_ld(Ld& p, const LT& L) {  // (X, Y, Z) -> (X^2, X^2 + Y, X Z); infinity (Z = 0) stays infinity
  const Gf xs = gf_sqr(p.X);
  p.Z = gf_mul(p.X, p.Z, L);
  p.Y = gf_add(xs, p.Y);
  p.X = xs;
}
template <class LT>
GF_DEV void lam_to_ld(Ld& p, const LT& L) {  // y = x (lambda + x): (X, L, Z) -> (X, X (L + X), Z)
  p.Y = gf_mul(p.X, gf_add(p.Y, p.X), L);
}
template <class LT>
GF_DEV void lam_dbl_ip(Ld& p, const LT& L) {
  Gf lz, xz;
  gf_mul2(p.Y, p.X, p.Z, L, lz, xz);
  const Gf T = gf_add(gf_sqr(p.Y), lz);  // L^2 + L Z (+ a Z^2, a = 0)
  const Gf X3 = gf_sqr(T);
  Gf Z3, tlz;
  gf_mul2(gf_sqr(p.Z), lz, T, L, Z3, tlz);
  p.Y = gf_add(gf_add(gf_sqr(xz), X3), gf_add(tlz, Z3));
  p.X = X3;
  p.Z = Z3;
}
// p += q.  Returns false -- p untouched -- when p == q (both finite): the CALLER doubles, from a copy it re-reads: an inlined
// doubling here keeps all of p alive across the whole addition and costs 460 B of scratch per lane in k_merge.
template <class LT>
GF_DEV bool lam_add_ip(Ld& p, const Ld& q, const LT& L) {
  if (gf_is_zero(q.Z)) return true;
  if (gf_is_zero(p.Z)) {
    p.X = q.X; p.Y = q.Y; p.Z = q.Z;
    return true;
  }
  Gf l1, U, l2, V;
  gf_mul2(p.Y, p.X, q.Z, L, l1, U);  // L1 Z2, X1 Z2
  gf_mul2(q.Y, q.X, p.Z, L, l2, V);  // L2 Z1, X2 Z1
  const Gf A = gf_add(l1, l2), D = gf_add(U, V);
  if (gf_is_zero(D)) {  // same x
    if (gf_is_zero(A)) return false;  // same lambda: p == q
    p.X = gf_one(); p.Y = gf_one(); p.Z = gf_zero();  // p == -q
    return true;
  }
  // order: what needs q.Z, p.Y and p.Z first, so that only U, V, A, B and two results are live across the last products
  const Gf B = gf_sqr(D);
  const Gf ABZ2 = gf_mul(gf_mul(B, A, L), q.Z, L);
  Gf T1, Z3;
  gf_mul2(gf_add(p.Y, p.Z), p.Z, ABZ2, L, T1, Z3);
  Gf AV, AU;
  gf_mul2(V, U, A, L, AV, AU);
  p.X = gf_mul(AU, AV, L);
  p.Y = gf_add(gf_sqr(gf_add(AV, B)), T1);
  p.Z = Z3;
  return true;
}

// Frobenius tau(x,y) = (x^2,y^2), applied k times
GF_DEV Ld ld_frob_n(Ld p, int k) {
#pragma unroll 1
  for (int i = 0; i < k; ++i) {
    p.X = gf_sqr(p.X);
    p.Y = gf_sqr(p.Y);
    p.Z = gf_sqr(p.Z);
  }
  return p;
}

// projective -> affine; returns false for infinity
GF_DEV bool ld_to_aff(const Ld& p, Aff* out) {
  if (ld_is_inf(p)) {
    out->x = gf_zero();
    out->y = gf_zero();
    return false;
  }
  Gf zi = gf_inv(p.Z);
  out->x = gf_mul(p.X, zi);
  out->y = gf_mul(p.Y, gf_sqr(zi));
  return true;
}

}  // namespace dvp
