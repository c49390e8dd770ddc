// ECFFT over Fr on gfx950: domain/twiddle generation and the extend / enter / exit butterflies.
//
// Replaces the third-party `ecfft` crate as the reference uses it:
//   build_ec_fftrees / build_sect_ecfft_tree   src/ec_fft.rs:93-170,197-239   -> dvp_ecfft_create
//   FFTree::extend(evals, Moiety::S1)           src/proving.rs:410-422         -> dvp_ecfft_extend
//   FFTree::enter / FFTree::exit                src/ec_fft.rs:266,317,411      -> dvp_ecfft_enter/exit
//
// Math (Ben-Sasson, Carmon, Kopparty, Levit, "ECFFT part I"):
//   layer d has N_d = N >> d leaves L_d; psi_d(x) = x + t_d/(x - x0_d) is the x-map of the
//   2-isogeny with kernel <(x0_d,0)> (the one lowering the 2-adicity of the generator,
//   src/ec_fft.rs:131-148); L_{d+1}[i] = psi_d(L_d[i]) = psi_d(L_d[i + N_d/2]).
//   A polynomial P of degree < n on n points decomposes as
//        P(x) = (P0(psi(x)) + x P1(psi(x))) * (x - x0)^(n/2-1)
//   so moving n evaluations from the even leaves to the odd leaves of a 2n-leaf layer is
//   log2(n) "decompose" butterfly passes (2x2 inverse matrices built from even-leaf pairs) followed
//   by log2(n) "recombine" passes (2x2 matrices built from odd-leaf pairs).  At pass d the vector
//   is 2^d independent blocks that all share the layer-d matrices.
//
// Data stays CANONICAL in HBM; matrices/tables are stored in Montgomery form so that
// mont_mul(const, data) is already the canonical product (see fr.cuh).
#include <vector>
#include <map>
#include <mutex>
#include <cstring>

#include "common.h"
#include "fr.cuh"

namespace dvp {

// ---- constants from src/ec_fft.rs:205-229 (canonical limbs) -----------------------------------
// a = 2125753088427212854352924174339172498722499297750753614229533284661082
constexpr uint32_t ECFFT_A_CANON[8] = {0xace0775au, 0x44fd5f67u, 0x76ead179u, 0x030cf18fu,
                                       0xaff1f871u, 0xd2776c4bu, 0xd93b829fu, 0x0000004eu};
// subgroup generator (order 2^28)
constexpr uint32_t ECFFT_GX_CANON[8] = {0xd0a7e150u, 0xb8963584u, 0x9981c323u, 0x88c5a0ceu,
                                        0xe7ba27e5u, 0x87f53e0eu, 0x0c8ea06bu, 0x00000049u};
constexpr uint32_t ECFFT_GY_CANON[8] = {0xec5cc36eu, 0x779340c8u, 0x3c4ec51bu, 0x92c4b0adu,
                                        0x94bbcf10u, 0xe10365e3u, 0x0a0d1e5fu, 0x00000022u};
// coset offset
constexpr uint32_t ECFFT_CX_CANON[8] = {0x7608b3eau, 0x786f6278u, 0xd29e01e1u, 0x4da67d47u,
                                        0x62d59c79u, 0xac90a9e4u, 0xc2a60c19u, 0x00000039u};
constexpr uint32_t ECFFT_CY_CANON[8] = {0x6d514258u, 0xd5063f88u, 0xe1bcc7d2u, 0x0be9eee3u,
                                        0x3f498728u, 0xb4efdffeu, 0x6bdc81edu, 0x00000055u};
constexpr int ECFFT_LOG_ORDER = 28;  // subgroup_adic, src/ec_fft.rs:205

static Fr fr_from_limbs_mont(const uint32_t* c) {
  Fr r;
  for (int i = 0; i < 8; ++i) r.v[i] = c[i];
  return fr_to_mont(r);
}

// ---- host-side short-Weierstrass affine arithmetic (only O(log N) of it) -----------------------
struct SwPt {
  Fr x, y;  // Montgomery
  bool inf;
};
static SwPt sw_add(const SwPt& p, const SwPt& q, const Fr& a) {
  if (p.inf) return q;
  if (q.inf) return p;
  Fr lam;
  if (fr_eq(p.x, q.x)) {
    if (fr_is_zero(fr_add(p.y, q.y))) return SwPt{fr_zero(), fr_zero(), true};
    Fr x2 = fr_sqr(p.x);
    Fr num = fr_add(fr_add(fr_dbl(x2), x2), a);
    lam = fr_mul(num, fr_inv(fr_dbl(p.y)));
  } else {
    lam = fr_mul(fr_sub(q.y, p.y), fr_inv(fr_sub(q.x, p.x)));
  }
  SwPt r;
  r.inf = false;
  r.x = fr_sub(fr_sub(fr_sqr(lam), p.x), q.x);
  r.y = fr_sub(fr_mul(lam, fr_sub(p.x, r.x)), p.y);
  return r;
}

// ---- device kernels ----------------------------------------------------------------------------

// leaves[i] = x(coset + i*g), i < n.  tab[j] = 2^j * g (affine, Montgomery).  Jacobian mixed adds
// (madd-2007-bl); no exceptional case can occur because coset is outside the subgroup <g>.
__global__ void __launch_bounds__(256) k_leaves(const Fr* __restrict__ tab_x, const Fr* __restrict__ tab_y, Fr cx, Fr cy,
                         int log_n, Fr* __restrict__ out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr X = cx, Y = cy, Z = fr_one_mont();
  for (int j = 0; j < log_n; ++j) {
    if (!((i >> j) & 1)) continue;
    Fr x2 = tab_x[j], y2 = tab_y[j];
    Fr z1z1 = fr_sqr(Z);
    Fr u2 = fr_mul(x2, z1z1);
    Fr s2 = fr_mul(fr_mul(y2, Z), z1z1);
    Fr h = fr_sub(u2, X);
    Fr hh = fr_sqr(h);
    Fr i4 = fr_dbl(fr_dbl(hh));
    Fr jj = fr_mul(h, i4);
    Fr r = fr_dbl(fr_sub(s2, Y));
    Fr v = fr_mul(X, i4);
    Fr x3 = fr_sub(fr_sub(fr_sqr(r), jj), fr_dbl(v));
    Fr y3 = fr_sub(fr_mul(r, fr_sub(v, x3)), fr_dbl(fr_mul(Y, jj)));
    Fr z3 = fr_sub(fr_sub(fr_sqr(fr_add(Z, h)), z1z1), hh);
    X = x3; Y = y3; Z = z3;
  }
  Fr zi = fr_inv(Z);
  out[i] = fr_mul(X, fr_sqr(zi));
}

// next[i] = psi(cur[i]) = cur[i] + t/(cur[i]-x0), i < half
__global__ void __launch_bounds__(256) k_next_layer(const Fr* __restrict__ cur, Fr x0, Fr t, Fr* __restrict__ next, uint32_t half) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= half) return;
  Fr x = cur[i];
  next[i] = fr_add(x, fr_mul(t, fr_inv(fr_sub(x, x0))));
}

// 2x2 matrices of one layer of one (strided) tree.  Ld = layer-d leaves of the full tree; the
// strided tree's leaf j is Ld[j << sl] and it has 2*nd leaves.  Pair i < nd/2:
//   decompose  (from leaves 2i+src, 2i+src+nd):  inverse of [[v0, s0 v0],[v1, s1 v1]]
//   recombine  (from leaves 2i+dst, 2i+dst+nd):            [[v0, s0 v0],[v1, s1 v1]]
// with v = (s - x0)^(nd/2 - 1).
// Matrix entries are consumed only by fr_dot2, so they are stored pre-sliced into 29-bit limbs (same 32 bytes).
__device__ __forceinline__ Fr29 mat_store(const Fr& a) { return fr29_from(a); }

__global__ void __launch_bounds__(256) k_build_mats(const Fr* __restrict__ Ld, int sl, uint32_t nd, Fr x0, int src, int dst,
                             Fr29* __restrict__ dec, Fr29* __restrict__ rec) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t h = nd >> 1;
  if (i >= h) return;
  uint64_t e = (uint64_t)h - 1;
  {
    Fr s0 = Ld[(size_t)(2 * i + src) << sl];
    Fr s1 = Ld[(size_t)(2 * i + src + nd) << sl];
    Fr v0 = fr_pow_u64(fr_sub(s0, x0), e);
    Fr v1 = fr_pow_u64(fr_sub(s1, x0), e);
    Fr det = fr_mul(fr_mul(v0, v1), fr_sub(s1, s0));
    Fr inv = fr_inv(det);
    Fr iv0 = fr_mul(inv, v0), iv1 = fr_mul(inv, v1);
    dec[4 * (size_t)i + 0] = mat_store(fr_mul(s1, iv1));
    dec[4 * (size_t)i + 1] = mat_store(fr_neg(fr_mul(s0, iv0)));
    dec[4 * (size_t)i + 2] = mat_store(fr_neg(iv1));
    dec[4 * (size_t)i + 3] = mat_store(iv0);
  }
  {
    Fr s0 = Ld[(size_t)(2 * i + dst) << sl];
    Fr s1 = Ld[(size_t)(2 * i + dst + nd) << sl];
    Fr v0 = fr_pow_u64(fr_sub(s0, x0), e);
    Fr v1 = fr_pow_u64(fr_sub(s1, x0), e);
    rec[4 * (size_t)i + 0] = mat_store(v0);
    rec[4 * (size_t)i + 1] = mat_store(fr_mul(s0, v0));
    rec[4 * (size_t)i + 2] = mat_store(v1);
    rec[4 * (size_t)i + 3] = mat_store(fr_mul(s1, v1));
  }
}

// ---- TWISTED butterflies (round 4) ---------------------------------------------------------------------------------------
// The matrices above are  recombine = diag(v0, v1) [[1, s0], [1, s1]]  with v = q(s)^(h - 1) (q = the isogeny's denominator) and
// decompose = its inverse.  Carry every value divided by the twist  W_d(s) = v_d(s) W_{d+1}(psi_d(s)),  W_bottom = 1  -- the
// product of the v factors along the point's orbit down the isogeny chain.  Both halves of a decomposition live at the same
// image point t, so they share ONE twist W_{d+1}(t), and in twisted values Q = P / W a butterfly is just
//      recombine   Q(s0) = Q0 + s0 Q1,   Q(s1) = Q0 + s1 Q1
//      decompose   Q1 = (Q(s0) - Q(s1)) / (s0 - s1),   Q0 = Q(s0) - s0 Q1
// TWO field products per pair (each with its addition folded into the Montgomery accumulation, fr_muladd29: 2 x 96 limb
// products) instead of the FOUR of a general 2x2 matrix (2 x fr_dot2: 2 x 160), and 64 bytes of constants per pair instead of
// 128.  The twist depends only on the point, i.e. on the position in the top-level vector: an extend multiplies its input by
// 1 / W^src once (fused into its first pass) and its output by W^dst once (fused into its last pass); at the bottom both twists
// are 1, which is where the source and destination halves meet.  Values are the same field elements as before: every parity
// test (oracle extend / enter / exit element for element, FFTR sections, proofs) is unchanged.
// Constants per pair (Montgomery form, pre-sliced): decompose (1 / (s0 - s1), -s0), recombine (s0, s1); layer d at offset
// 2 (n - (n >> d)).  The true 2x2 matrices are still built on demand for dvp_debug_ecfft_matrices / FFTR tree files (k_build_mats).
__global__ void __launch_bounds__(256) k_build_twiddles(const Fr* __restrict__ Ld, int sl, uint32_t nd, int src, int dst, Fr30* __restrict__ dec,
                                                        Fr30* __restrict__ rec) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t h = nd >> 1;
  if (i >= h) return;
  {
    Fr s0 = Ld[(size_t)(2 * i + src) << sl];
    Fr s1 = Ld[(size_t)(2 * i + src + nd) << sl];
    dec[2 * (size_t)i + 0] = fr30_const(fr_inv(fr_sub(s0, s1)));
    dec[2 * (size_t)i + 1] = fr30_const(fr_neg(s0));
  }
  rec[2 * (size_t)i + 0] = fr30_const(Ld[(size_t)(2 * i + dst) << sl]);
  rec[2 * (size_t)i + 1] = fr30_const(Ld[(size_t)(2 * i + dst + nd) << sl]);
}
// W_d[j] = v_d(point of position j in a block of nd values) * W_{d+1}[j mod (nd / 2)], bottom up; positions j < nd/2 sit at the
// first point of pair j, the others at the second point of pair j - nd/2 (that is where the in-place butterflies leave them)
__global__ void __launch_bounds__(256) k_twist_layer(const Fr* __restrict__ Ld, int sl, uint32_t nd, Fr x0, int parity, const Fr* __restrict__ w_next,
                                                     Fr* __restrict__ w_cur) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nd) return;
  const uint32_t h = nd >> 1, i = j & (h - 1);
  const Fr s = Ld[(size_t)(2 * i + parity + (j >= h ? nd : 0)) << sl];
  Fr v = fr_pow_u64(fr_sub(s, x0), (uint64_t)h - 1);
  if (w_next) v = fr_mul(v, w_next[i]);
  w_cur[j] = v;
}
__global__ void __launch_bounds__(256) k_twist_finish(const Fr* __restrict__ w_src, const Fr* __restrict__ w_dst, uint32_t n, Fr30* __restrict__ win,
                                                      Fr30* __restrict__ wout) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  win[j] = fr30_const(fr_inv(w_src[j]));
  wout[j] = fr30_const(w_dst[j]);
}

// ---- the butterflies on lazy 30-bit-limb values (fr.cuh: Fr30) ---------------------------------------------------------------
// Between the passes of an extend the data buffer holds Fr30 values (same 32 bytes as an Fr): the pass that carries the input twist
// (`pre`, always the first) slices its canonical input, the pass that carries the output twist (`post`, always the last) reduces and
// re-slices; everything in between is multiply-accumulate only.  Bounds (in units of p, fr.cuh): the input twist leaves < 1.01; a
// decompose step gives Q1 < 1.4 and Q0 < bound(e0) + 1.01; a recombine step bound(Q0) + 1 + bound(Q1)/512: at most +1.13 per layer,
// < 64 after the 2 x 27 layers of the largest tree -- the lent 128 p of fr30_sub_lazy always covers its subtrahend.
struct Tw {
  Fr30 a, b;
};
__device__ __forceinline__ Tw tw_load(const Fr30* __restrict__ t, uint32_t i) {
  Tw r;
  r.a = t[2 * (size_t)i];
  r.b = t[2 * (size_t)i + 1];
  return r;
}
__device__ __forceinline__ Fr30 ld30(const Fr* p) {  // a value a previous pass stored in Fr30 form
  const Fr v = *p;
  Fr30 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = v.v[i];
  return r;
}
__device__ __forceinline__ void st30(Fr* p, const Fr30& a) {
  Fr v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v.v[i] = a.l[i];
  *p = v;
}
// two independent decompose steps: (e0, e1) = (Q(s0), Q(s1)) -> (Q0, Q1); t = (1 / (s0 - s1), -s0)
__device__ __forceinline__ void bf_dec2(Fr30& a0, Fr30& a1, const Tw& ta, Fr30& b0, Fr30& b1, const Tw& tb) {
  const Fr30 z = fr30_zero();
  Fr30 qa, qb;
  fr30_muladd_x2(ta.a, fr30_sub_lazy(a0, a1), z, tb.a, fr30_sub_lazy(b0, b1), z, qa, qb);
  fr30_muladd_x2(ta.b, qa, a0, tb.b, qb, b0, a0, b0);
  a1 = qa;
  b1 = qb;
}
__device__ __forceinline__ void bf_dec(Fr30& e0, Fr30& e1, const Tw& t) {
  const Fr30 q1 = fr30_muladd(t.a, fr30_sub_lazy(e0, e1), fr30_zero());
  e0 = fr30_muladd(t.b, q1, e0);
  e1 = q1;
}
// (e0, e1) = (Q0, Q1) -> (Q(s0), Q(s1)); t = (s0, s1): the two products of one step are independent
__device__ __forceinline__ void bf_rec(Fr30& e0, Fr30& e1, const Tw& t) { fr30_muladd_x2(t.a, e1, e0, t.b, e1, e0, e0, e1); }
// the twists of the first / last pass, two values at a time
__device__ __forceinline__ void tw_in2(const Fr30* __restrict__ w, uint32_t p0, uint32_t p1, const Fr& x0, const Fr& x1, Fr30& r0, Fr30& r1) {
  const Fr30 z = fr30_zero();
  fr30_muladd_x2(w[p0], fr30_from(x0), z, w[p1], fr30_from(x1), z, r0, r1);
}
__device__ __forceinline__ void tw_out2(const Fr30* __restrict__ w, uint32_t p0, uint32_t p1, const Fr30& x0, const Fr30& x1, Fr& r0, Fr& r1) {
  const Fr30 z = fr30_zero();
  Fr30 y0, y1;
  fr30_muladd_x2(w[p0], x0, z, w[p1], x1, z, y0, y1);
  r0 = fr30_canon(y0);
  r1 = fr30_canon(y1);
}

// One butterfly pass over `batch` vectors of n values.  Blocks of size 2h; pair (i, i+h) inside each block uses constants i of
// this layer.  One thread = one pair for all batch vectors, so the constants are read once per pair and reused `batch` times.
// pre / post (nullptr = none): the twists of the first / last pass of an extend, indexed by the position in the vector.
template <int BATCH, bool DEC>
__global__ void __launch_bounds__(256) k_butterfly(const Fr* src /* == data, or the untouched input of the first pass */, Fr* data,
                                                   const Fr30* __restrict__ tws, int lh, uint32_t n, const Fr30* __restrict__ pre,
                                                   const Fr30* __restrict__ post, uint32_t nv /* values per vector (twist index range) */) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (n >> 1)) return;
  uint32_t h = 1u << lh;
  uint32_t i = tid & (h - 1);
  uint32_t i0 = ((tid >> lh) << (lh + 1)) | i;
  uint32_t i1 = i0 + h;
  const Tw t = tw_load(tws, i);
  const uint32_t p0 = i0 & (nv - 1), p1 = i1 & (nv - 1);
#pragma unroll
  for (int b = 0; b < BATCH; ++b) {
    Fr* v = data + (size_t)b * n;
    const Fr* u = src + (size_t)b * n;
    Fr30 e0, e1;
    if (pre) tw_in2(pre, p0, p1, u[i0], u[i1], e0, e1); else { e0 = ld30(u + i0); e1 = ld30(u + i1); }
    if (DEC) bf_dec(e0, e1, t); else bf_rec(e0, e1, t);
    if (post) {
      Fr o0, o1;
      tw_out2(post, p0, p1, e0, e1, o0, o1);
      v[i0] = o0;
      v[i1] = o1;
    } else {
      st30(v + i0, e0);
      st30(v + i1, e1);
    }
  }
}

// TWO butterfly layers in one pass (radix 4): layer d (pairs h1 = 2^lh apart) and layer d + 1 (pairs h2 = h1 / 2 apart) only mix
// the four values {j, j + h2, j + h1, j + h1 + h2} of a block of 4 h2, so one thread loads them once, applies both layers in
// registers and stores them once: half the HBM round trips of the data.  Constants: two pairs of layer d (j, j + h2) and one of
// layer d + 1 (j, shared by both of its pairs).  DEC = decompose order (wide layer first); recombine runs the narrow layer first.
// BATCH vectors: lane = (quad of values, vector) with the vector index fastest, so the BATCH lanes that need the same constants
// sit in the same wave and their loads are one request.
// MAP: where the constants of half-block position i sit in a layer's table -- i itself for data held in vector order (TwId); a tile
// of a long vector (k_extend_top) maps its rows and columns back to vector positions (TwTile)
struct TwId {
  __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return i; }
};
template <bool DEC, class MAP = TwId>
__device__ __forceinline__ void radix4(Fr30& x0, Fr30& x1, Fr30& x2, Fr30& x3, const Fr30* __restrict__ tw_wide, const Fr30* __restrict__ tw_narrow,
                                       uint32_t j, uint32_t h2, MAP f = MAP()) {
  if (DEC) {
    bf_dec2(x0, x2, tw_load(tw_wide, f(j)), x1, x3, tw_load(tw_wide, f(j + h2)));
    const Tw c = tw_load(tw_narrow, f(j));
    bf_dec2(x0, x1, c, x2, x3, c);
  } else {
    {
      const Tw c = tw_load(tw_narrow, f(j));
      bf_rec(x0, x1, c);
      bf_rec(x2, x3, c);
    }
    bf_rec(x0, x2, tw_load(tw_wide, f(j)));
    bf_rec(x1, x3, tw_load(tw_wide, f(j + h2)));
  }
}
template <int BATCH, bool DEC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k_butterfly4(const Fr* src /* == data, or the untouched input of the first pass */, Fr* data,
                                                    const Fr30* __restrict__ tw_wide, const Fr30* __restrict__ tw_narrow, int lh2, uint32_t n,
                                                    const Fr30* __restrict__ pre, const Fr30* __restrict__ post, uint32_t nv) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t tid = gid / BATCH, bv = gid - tid * BATCH;
  if (tid >= (n >> 2)) return;
  const uint32_t h2 = 1u << lh2, h1 = h2 << 1;
  const uint32_t j = tid & (h2 - 1);
  const uint32_t i0 = ((tid >> lh2) << (lh2 + 2)) | j;
  Fr* v = data + (size_t)bv * n + i0;
  const Fr* u = src + (size_t)bv * n + i0;
  const uint32_t p0 = i0 & (nv - 1);
  Fr30 x0, x1, x2, x3;
  if (pre) {
    tw_in2(pre, p0, p0 + h2, u[0], u[h2], x0, x1);
    tw_in2(pre, p0 + h1, p0 + h1 + h2, u[h1], u[h1 + h2], x2, x3);
  } else {
    x0 = ld30(u); x1 = ld30(u + h2); x2 = ld30(u + h1); x3 = ld30(u + h1 + h2);
  }
  radix4<DEC>(x0, x1, x2, x3, tw_wide, tw_narrow, j, h2);
  if (post) {
    Fr o0, o1, o2, o3;
    tw_out2(post, p0, p0 + h2, x0, x1, o0, o1);
    tw_out2(post, p0 + h1, p0 + h1 + h2, x2, x3, o2, o3);
    v[0] = o0; v[h2] = o1; v[h1] = o2; v[h1 + h2] = o3;
  } else {
    st30(v, x0); st30(v + h2, x1); st30(v + h1, x2); st30(v + h1 + h2, x3);
  }
}

// THREE layers in one pass (radix 8) for the top of long vectors: with the lazy multiplier a two-layer pass is HBM-bound (192 MB of
// data + up to 48 MB of constants in ~52 us at m = 2^20, batch 3), so a third layer per round trip is nearly free.  A thread owns the
// eight values i0 + {0, h3, h2, h2 + h3, h1, h1 + h3, h1 + h2, h1 + h2 + h3} (h1 = 2 h2 = 4 h3 = the widest pair distance); constants:
// four pairs of layer d (j, j + h3, j + h2, j + h2 + h3), two of layer d + 1 (j, j + h3), one of layer d + 2 (j).
#ifndef DVP_BF8_WAVES
#define DVP_BF8_WAVES 3
#endif
template <int BATCH, bool DEC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DVP_BF8_WAVES, DVP_BF8_WAVES))) k_butterfly8(const Fr* src, Fr* data, const Fr30* __restrict__ tw0, const Fr30* __restrict__ tw1,
                                                    const Fr30* __restrict__ tw2, int lh3, uint32_t n, const Fr30* __restrict__ pre,
                                                    const Fr30* __restrict__ post, uint32_t nv) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t tid = gid / BATCH, bv = gid - tid * BATCH;
  if (tid >= (n >> 3)) return;
  const uint32_t h3 = 1u << lh3, h2 = h3 << 1, h1 = h3 << 2;
  const uint32_t j = tid & (h3 - 1);
  const uint32_t i0 = ((tid >> lh3) << (lh3 + 3)) | j;
  Fr* v = data + (size_t)bv * n + i0;
  const Fr* u = src + (size_t)bv * n + i0;
  const uint32_t p0 = i0 & (nv - 1);
  Fr30 x[8];
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    const uint32_t o0 = (k & 4 ? h1 : 0) + (k & 2 ? h2 : 0), o1 = o0 + h3;
    if (pre) tw_in2(pre, p0 + o0, p0 + o1, u[o0], u[o1], x[k], x[k + 1]);
    else { x[k] = ld30(u + o0); x[k + 1] = ld30(u + o1); }
  }
  if (DEC) {
    bf_dec2(x[0], x[4], tw_load(tw0, j), x[1], x[5], tw_load(tw0, j + h3));
    bf_dec2(x[2], x[6], tw_load(tw0, j + h2), x[3], x[7], tw_load(tw0, j + h2 + h3));
    {
      const Tw a = tw_load(tw1, j), b = tw_load(tw1, j + h3);
      bf_dec2(x[0], x[2], a, x[1], x[3], b);
      bf_dec2(x[4], x[6], a, x[5], x[7], b);
    }
    const Tw c = tw_load(tw2, j);
    bf_dec2(x[0], x[1], c, x[2], x[3], c);
    bf_dec2(x[4], x[5], c, x[6], x[7], c);
  } else {
    {
      const Tw c = tw_load(tw2, j);
      bf_rec(x[0], x[1], c); bf_rec(x[2], x[3], c); bf_rec(x[4], x[5], c); bf_rec(x[6], x[7], c);
    }
    {
      const Tw a = tw_load(tw1, j), b = tw_load(tw1, j + h3);
      bf_rec(x[0], x[2], a); bf_rec(x[1], x[3], b); bf_rec(x[4], x[6], a); bf_rec(x[5], x[7], b);
    }
    bf_rec(x[0], x[4], tw_load(tw0, j));
    bf_rec(x[1], x[5], tw_load(tw0, j + h3));
    bf_rec(x[2], x[6], tw_load(tw0, j + h2));
    bf_rec(x[3], x[7], tw_load(tw0, j + h2 + h3));
  }
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    const uint32_t o0 = (k & 4 ? h1 : 0) + (k & 2 ? h2 : 0), o1 = o0 + h3;
    if (post) {
      Fr r0, r1;
      tw_out2(post, p0 + o0, p0 + o1, x[k], x[k + 1], r0, r1);
      v[o0] = r0;
      v[o1] = r1;
    } else {
      st30(v + o0, x[k]);
      st30(v + o1, x[k + 1]);
    }
  }
}

// ---- what the first launch of an extend reads and the last one writes (round 6) -----------------------------------------------
// SM = 0 is the extend proper: the input twist on the way in (x = pre[p] * src[g]), the output twist on the way out.  exit and enter
// used to wrap every extend in separate pointwise launches (k_exit_pre / _mid / _mulc / _post, a device-to-device copy, k_enter_combine:
// seven HBM passes and a copy around the four extends of an exit level); all of them are products with per-position tables, so they
// fold into the twists: the tables are multiplied together once (k_exit_fuse_tables) and the LDS kernels' first / last pass do the rest.
//   load (every SM):   x = pre[p] * src[g << lshift]       lshift = 1 reads the even entries of an interleaved vector
//   SM 1 ("mid"):      out[g] = tb[p] * aux[(g << ashift) + aoff] + post[p] * x                 (post = -(wout * xnn_odd * z0inv))
//   SM 2 ("post"):     u = post[p] * x;  out[(2c) h + i] = u;  out[(2c + 1) h + i] = tb[p] * (aux[2 g] - u),   g = c h + i
//   SM 3 ("combine"):  enter's recombination, vectors 2c (lo) and 2c + 1 (hi) of one workgroup:
//                      out[c 2h + 2i] = aux[lo] + tb[i] * aux[hi];  out[c 2h + 2i + 1] = post[i] * x_lo + tc[i] * x_hi
// Products are lazy (fr.cuh): a sum of two of them is below 2.2 p, two conditional subtractions bring it home.
template <int SM>
struct ExtIo {
  const Fr30* pre = nullptr;
  int lshift = 0;
  const Fr30* post = nullptr;
  Fr* out = nullptr;
  const Fr30* tb = nullptr;
  const Fr30* tc = nullptr;
  const Fr* aux = nullptr;
  int ashift = 0, aoff = 0, lh = 0;
};
__device__ __forceinline__ Fr fr30_canon2(const Fr30& a) { return fr_cond_sub_p(fr_cond_sub_p(fr30_to_fr(a))); }
template <int SM>
__device__ __forceinline__ void io_load2(const ExtIo<SM>& io, const Fr* src, size_t g0, size_t g1, uint32_t p0, uint32_t p1, Fr30& r0, Fr30& r1) {
  if (io.pre) tw_in2(io.pre, p0, p1, src[g0 << io.lshift], src[g1 << io.lshift], r0, r1);
  else {
    r0 = ld30(src + g0);
    r1 = ld30(src + g1);
  }
}
// the last launch's stores for SM 0 .. 2 (two values of one thread; SM 3 pairs values across vectors: k_extend_fused does it itself)
template <int SM>
__device__ __forceinline__ void io_store2(const ExtIo<SM>& io, Fr* data, size_t g0, size_t g1, uint32_t p0, uint32_t p1, const Fr30& x0, const Fr30& x1) {
  if (SM == 0) {
    if (io.post) {
      Fr o0, o1;
      tw_out2(io.post, p0, p1, x0, x1, o0, o1);
      data[g0] = o0;
      data[g1] = o1;
    } else {
      st30(data + g0, x0);
      st30(data + g1, x1);
    }
  } else if (SM == 1) {
    const Fr30 z = fr30_zero();
    Fr30 a0, a1, y0, y1;
    fr30_muladd_x2(io.tb[p0], fr30_from(io.aux[(g0 << io.ashift) + io.aoff]), z, io.tb[p1], fr30_from(io.aux[(g1 << io.ashift) + io.aoff]), z, a0, a1);
    fr30_muladd_x2(io.post[p0], x0, a0, io.post[p1], x1, a1, y0, y1);
    io.out[g0] = fr30_canon2(y0);
    io.out[g1] = fr30_canon2(y1);
  } else if (SM == 2) {
    Fr u0, u1;
    tw_out2(io.post, p0, p1, x0, x1, u0, u1);
    const Fr d0 = fr_sub(io.aux[g0 << 1], u0), d1 = fr_sub(io.aux[g1 << 1], u1);
    const uint32_t hm = (1u << io.lh) - 1u;
    const size_t o0 = ((g0 >> io.lh) << (io.lh + 1)) + (g0 & hm), o1 = ((g1 >> io.lh) << (io.lh + 1)) + (g1 & hm);
    io.out[o0] = u0;
    io.out[o1] = u1;
    Fr v0, v1;
    tw_out2(io.tb, p0, p1, fr30_from(d0), fr30_from(d1), v0, v1);
    io.out[o0 + (hm + 1u)] = v0;
    io.out[o1 + (hm + 1u)] = v1;
  }
}

// Fused bottom of an extend: the last `lb` decompose layers and the first `lb` recombine layers only mix
// elements inside aligned blocks of 2^lb <= 2048 values, so a workgroup keeps 2048 consecutive values
// (64 KB) in LDS and runs all 2*lb butterfly layers on them in ONE launch and ONE HBM round trip (the
// per-layer kernels above then cover only the top layers of long vectors).  The constants of these layers
// total < 0.25 MB per tree and are served from L2.  `data` is the concatenation of the batch vectors.
constexpr int FUSE_LOG = 11;
constexpr uint32_t FUSE_ELEMS = 1u << FUSE_LOG;
#ifndef DVP_FUSE_TPB
#define DVP_FUSE_TPB 512
#endif
constexpr int FUSE_TPB = DVP_FUSE_TPB;  // 64 KB of LDS per workgroup: two workgroups per CU, FUSE_TPB / 128 waves per SIMD

template <bool DEC, class MAP = TwId>
__device__ __forceinline__ void lds_bfly(Fr30* x, const Fr30* __restrict__ tws, int lh, uint32_t pairs, MAP f = MAP()) {
  const uint32_t h = 1u << lh;
  for (uint32_t q = threadIdx.x; q < pairs; q += blockDim.x) {
    uint32_t i = q & (h - 1);
    uint32_t i0 = ((q >> lh) << (lh + 1)) | i, i1 = i0 + h;
    Fr30 e0 = x[i0], e1 = x[i1];
    if (DEC) bf_dec(e0, e1, tw_load(tws, f(i))); else bf_rec(e0, e1, tw_load(tws, f(i)));
    x[i0] = e0;
    x[i1] = e1;
  }
  __syncthreads();
}
// two layers per barrier inside the block (the radix-4 step of k_butterfly4 on the LDS copy): half the barriers and half the LDS
// round trips of the per-layer loop.  lh2 = log2 of the narrow layer's pair distance; wide layer first when DEC.
// (Round 5 measured THREE layers per barrier, the radix-8 step of k_butterfly8 on the LDS copy, in this kernel and in k_extend_top:
// eight values and seven constant pairs per thread spill 250 B per lane at the 128 VGPRs of 512-thread workgroups and still cost a
// wave per SIMD at 384 threads -- the prover's three extends 0.77 -> 0.99 / 1.16 ms, enter 3.4 -> 3.9 / 4.3 ms: not kept.)
template <bool DEC, class MAP = TwId>
__device__ __forceinline__ void lds_bfly4(Fr30* x, const Fr30* __restrict__ tw_wide, const Fr30* __restrict__ tw_narrow, int lh2, uint32_t quads, MAP f = MAP()) {
  const uint32_t h2 = 1u << lh2, h1 = h2 << 1;
  for (uint32_t q = threadIdx.x; q < quads; q += blockDim.x) {
    const uint32_t j = q & (h2 - 1);
    const uint32_t i0 = ((q >> lh2) << (lh2 + 2)) | j;
    Fr30 x0 = x[i0], x1 = x[i0 + h2], x2 = x[i0 + h1], x3 = x[i0 + h1 + h2];
    radix4<DEC, MAP>(x0, x1, x2, x3, tw_wide, tw_narrow, j, h2, f);
    x[i0] = x0; x[i0 + h2] = x1; x[i0 + h1] = x2; x[i0 + h1 + h2] = x3;
  }
  __syncthreads();
}

template <int SM>
__global__ void __launch_bounds__(FUSE_TPB) __attribute__((amdgpu_waves_per_eu(FUSE_TPB / 128, FUSE_TPB / 128)))
k_extend_fused(const Fr* src /* == data unless this is the first pass of an out-of-place extend */, Fr* data, const Fr30* __restrict__ dec,
               const Fr30* __restrict__ rec, uint32_t n, int ln, int lb, size_t total, const ExtIo<SM> io) {
  __shared__ Fr30 x[FUSE_ELEMS];
  const size_t base = (size_t)blockIdx.x * FUSE_ELEMS;
  const uint32_t elems = (uint32_t)min((size_t)FUSE_ELEMS, total - base);
  const uint32_t pos0 = (uint32_t)(base & (size_t)(n - 1));  // position of the block's first value inside its vector (n is a power of two)
  // two values per thread and trip (k, k + elems / 2): their twist products are independent
  const uint32_t half = elems >> 1;  // elems is even: a batch of vectors of n >= 2 values
  for (uint32_t k = threadIdx.x; k < half; k += blockDim.x)
    io_load2(io, src, base + k, base + k + half, (pos0 + k) & (n - 1), (pos0 + k + half) & (n - 1), x[k], x[k + half]);
  __syncthreads();
  auto tw_of = [&](const Fr30* b, int L) { return b + 2 * (size_t)(n - (n >> (ln - lb + L))); };
  const bool whole = elems == FUSE_ELEMS || (elems & 3u) == 0;  // (a short last block still holds whole sub-blocks of every layer it runs)
  {
    int L = 0;  // decompose, sub-block size 2^(lb-L); two layers per barrier while two remain
    if (whole)
      for (; L + 1 < lb; L += 2) lds_bfly4<true>(x, tw_of(dec, L), tw_of(dec, L + 1), lb - L - 2, elems >> 2);
    for (; L < lb; ++L) lds_bfly<true>(x, tw_of(dec, L), lb - L - 1, elems >> 1);
  }
  {
    int L = lb - 1;  // recombine: the odd layer (the innermost one) first, as the decompose left it
    if (whole && (lb & 1)) { lds_bfly<false>(x, tw_of(rec, L), lb - L - 1, elems >> 1); --L; }
    if (whole)
      for (; L >= 1; L -= 2) lds_bfly4<false>(x, tw_of(rec, L - 1), tw_of(rec, L), lb - L - 1, elems >> 2);
    for (; L >= 0; --L) lds_bfly<false>(x, tw_of(rec, L), lb - L - 1, elems >> 1);
  }
  if (SM == 3) {
    // enter's combine: the block holds whole (lo, hi) pairs of vectors of n = 2^lh <= 1024 values (the host checks); thread q owns
    // position i of pair cl
    const int lh = io.lh;
    const uint32_t hm = n - 1u;
    for (uint32_t q = threadIdx.x; q < half; q += blockDim.x) {
      const uint32_t i = q & hm, e_lo = ((q >> lh) << (lh + 1)) + i, e_hi = e_lo + n;
      const Fr30 z = fr30_zero();
      Fr30 a, b, ye, yo;
      fr30_muladd_x2(io.post[i], x[e_lo], z, io.tb[i], fr30_from(io.aux[base + e_hi]), fr30_from(io.aux[base + e_lo]), a, ye);
      yo = fr30_muladd(io.tc[i], x[e_hi], a);
      const size_t o = base + ((size_t)(q >> lh) << (lh + 1)) + 2u * i;
      io.out[o] = fr30_canon2(ye);
      io.out[o + 1] = fr30_canon2(yo);
    }
    return;
  }
  for (uint32_t k = threadIdx.x; k < half; k += blockDim.x)
    io_store2(io, data, base + k, base + k + half, (pos0 + k) & (n - 1), (pos0 + k + half) & (n - 1), x[k], x[k + half]);
}

// Fused TOP of an extend (round 5): the `tl` layers d0 .. d0 + tl - 1 of a long vector mix values that sit S = n >> (d0 + tl) apart
// (the pair distance of the narrowest of them) inside blocks of n >> d0 values.  A workgroup takes a TILE of such a block -- R = 2^tl
// rows S apart, C = FUSE_ELEMS / R consecutive columns (a row of the tile is C x 32 B of consecutive memory: whole 128-byte lines
// from C = 4, i.e. tl <= 9) -- into LDS and runs all tl layers on it: ONE launch and one HBM round trip where k_butterfly8 took
// ceil(tl / 3).  The 2^20-constraint prover's extends (top = 9 layers above the 2048-value blocks of k_extend_fused) go from
// 3 + 1 + 3 launches to 1 + 1 + 1.  LDS position e = row * C + col; the constants of half-block position i of a layer sit at vector
// position (i / C) * S + col0 + i % C (TwTile).  DEC: wide layer first (the first launch of an extend, with the input twist `pre`);
// otherwise narrow first (the last launch, with the output twist `post`).  In place: a tile is read completely before it is written
// and no two workgroups share a value.
struct TwTile {
  uint32_t lc, ls, col0;
  __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return ((i >> lc) << ls) + col0 + (i & ((1u << lc) - 1u)); }
};
template <bool DEC, int SM>
__global__ void __launch_bounds__(FUSE_TPB) __attribute__((amdgpu_waves_per_eu(FUSE_TPB / 128, FUSE_TPB / 128)))
k_extend_top(const Fr* src, Fr* data, const Fr30* __restrict__ tw /* dec or rec, whole table */, uint32_t n, int ln, int d0, int tl,
             uint32_t batch, const ExtIo<SM> io) {
  __shared__ Fr30 x[FUSE_ELEMS];
  // SM 3 (enter's combine, last launch, d0 == 0): a tile takes its rows from vectors 2u AND 2u + 1 -- the rows of a vector are
  // S = n >> tl apart and 2^tl S = n, so row 2^tl + r of the "unit" (2u, 2u + 1) is row r of vector 2u + 1: twice the rows, half the
  // columns, and x[k], x[k + half] are the two vectors' values at the same position, which is what the combine pairs
  constexpr int PR = SM == 3 ? 1 : 0;
  const int lc = FUSE_LOG - tl - PR, ls = ln - d0 - tl;   // log2 of the tile's columns and of the row distance
  const uint32_t groups = 1u << (ls - lc);                // tiles per block of n >> d0 values (per unit of two vectors with SM 3)
  // workgroup -> (vector, tile of that vector).  The same tile of the `batch` vectors reads the same constants (64 bytes per pair
  // and layer: as many bytes as the data), and consecutive workgroups go to the 8 XCDs in turn: eight tiles of vector 0, the same
  // eight of vector 1, .. -- so that the workgroups sharing constants land on the SAME XCD one dispatch round apart and the second
  // and third find them in its L2 (a tile-major order put them on three different XCDs: three fetches from memory)
  // (a vector of fewer than eight tiles -- the many short vectors of an enter / exit level -- is walked tile by tile)
  const uint32_t tpv = (n << PR) >> FUSE_LOG;  // tiles per vector (per unit)
  const uint32_t per8 = 8u * (batch >> PR), grp8 = blockIdx.x / per8, rem = blockIdx.x - grp8 * per8;
  const uint32_t vec = tpv >= 8u ? rem >> 3 : blockIdx.x / tpv, tv = tpv >= 8u ? ((grp8 << 3) | (rem & 7u)) : blockIdx.x - vec * tpv;
  const uint32_t gb = (vec << d0) + (tv >> (ls - lc)), cg = tv & (groups - 1u);
  const size_t base = ((size_t)gb << (ln - d0 + PR)) + ((size_t)cg << lc);
  const uint32_t cmask = (1u << lc) - 1u, half = FUSE_ELEMS >> 1;
  auto gidx = [&](uint32_t e) -> size_t { return base + ((size_t)(e >> lc) << ls) + (e & cmask); };
  for (uint32_t k = threadIdx.x; k < half; k += blockDim.x) {
    const size_t g0 = gidx(k), g1 = gidx(k + half);
    io_load2(io, src, g0, g1, (uint32_t)(g0 & (size_t)(n - 1)), (uint32_t)(g1 & (size_t)(n - 1)), x[k], x[k + half]);
  }
  __syncthreads();
  const TwTile f{(uint32_t)lc, (uint32_t)ls, cg << lc};
  auto tw_of = [&](int L) { return tw + 2 * (size_t)(n - (n >> (d0 + L))); };  // layer d0 + L: rows 2^(tl - 1 - L) apart
  if (DEC) {
    int L = 0;
    for (; L + 1 < tl; L += 2) lds_bfly4<true, TwTile>(x, tw_of(L), tw_of(L + 1), tl - L - 2 + lc, FUSE_ELEMS >> 2, f);
    for (; L < tl; ++L) lds_bfly<true, TwTile>(x, tw_of(L), tl - L - 1 + lc, FUSE_ELEMS >> 1, f);
  } else {
    int L = tl - 1;
    if (tl & 1) { lds_bfly<false, TwTile>(x, tw_of(L), tl - L - 1 + lc, FUSE_ELEMS >> 1, f); --L; }
    for (; L >= 1; L -= 2) lds_bfly4<false, TwTile>(x, tw_of(L - 1), tw_of(L), tl - L - 1 + lc, FUSE_ELEMS >> 2, f);
  }
  if (SM == 3) {
    for (uint32_t k = threadIdx.x; k < half; k += blockDim.x) {
      const size_t g0 = gidx(k), g1 = g0 + n;  // == gidx(k + half): the same position of vector 2u + 1
      const uint32_t i = (uint32_t)(g0 & (size_t)(n - 1));
      const Fr30 z = fr30_zero();
      Fr30 a, ye, yo;
      fr30_muladd_x2(io.post[i], x[k], z, io.tb[i], fr30_from(io.aux[g1]), fr30_from(io.aux[g0]), a, ye);
      yo = fr30_muladd(io.tc[i], x[k + half], a);
      io.out[g0 + i] = fr30_canon2(ye);  // unit u starts at u 2n = g0 - i: out[u 2n + 2i], out[u 2n + 2i + 1]
      io.out[g0 + i + 1] = fr30_canon2(yo);
    }
    return;
  }
  for (uint32_t k = threadIdx.x; k < half; k += blockDim.x) {
    const size_t g0 = gidx(k), g1 = gidx(k + half);
    io_store2(io, data, g0, g1, (uint32_t)(g0 & (size_t)(n - 1)), (uint32_t)(g1 & (size_t)(n - 1)), x[k], x[k + half]);
  }
}

// enter combine (one recursion level, all sub-problems at once):
//   lo/hi evaluations u0,v0 (even leaves) and u1,v1 (odd leaves) -> res[2i]   = u0 + xnn[2i]  *v0
//                                                                   res[2i+1] = u1 + xnn[2i+1]*v1
// `even`/`odd` hold 2*nsub vectors of h values: [.. lo_c, hi_c ..]; out holds nsub vectors of 2h.
__global__ void __launch_bounds__(256) k_enter_combine(const Fr* __restrict__ even, const Fr* __restrict__ odd,
                                const Fr* __restrict__ xnn, Fr* __restrict__ out, uint32_t h, uint32_t total) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= total) return;  // total = nsub*h
  uint32_t c = tid / h, i = tid - c * h;
  size_t lo = (size_t)(2 * c) * h + i, hi = lo + h;
  Fr u0 = even[lo], v0 = even[hi], u1 = odd[lo], v1 = odd[hi];
  size_t o = (size_t)c * 2 * h + 2 * i;
  out[o] = fr_add(u0, fr_mul(xnn[2 * i], v0));
  out[o + 1] = fr_add(u1, fr_mul(xnn[2 * i + 1], v1));
}

// xnn[j] = leaf_j^h over the strided tree (Montgomery)
__global__ void __launch_bounds__(256) k_pow_table(const Fr* __restrict__ L0, int sl, uint32_t sz, uint64_t e, Fr* __restrict__ out) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= sz) return;
  out[j] = fr_pow_u64(L0[(size_t)j << sl], e);
}

__global__ void __launch_bounds__(256) k_from_mont(const Fr* __restrict__ in, Fr* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fr_from_mont(in[i]);
}
__global__ void __launch_bounds__(256) k_check_canonical(const Fr* __restrict__ in, size_t n, unsigned long long* bad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && !fr_is_canonical(in[i])) atomicMin(bad, (unsigned long long)i);
}

}  // namespace dvp

using namespace dvp;

#include "ecfft_internal.h"

static const int TPB = 256;

static int build_matset(dvp_ecfft* c, int sl, int to_even, MatSet** out, hipStream_t st) {
  std::lock_guard<std::mutex> g(c->mu);
  int key = sl * 2 + to_even;
  auto it = c->mats.find(key);
  if (it != c->mats.end()) {
    *out = &it->second;
    return DVP_OK;
  }
  uint32_t n = (c->n_leaves >> sl) >> 1;  // evaluations moved by extend on this subtree
  MatSet ms;
  if (n > 1) {
    size_t bytes = (size_t)(n - 1) * 2 * sizeof(Fr30);
    DVP_HIP(hipMalloc((void**)&ms.dec, bytes));
    DVP_HIP(hipMalloc((void**)&ms.rec, bytes));
    DVP_HIP(hipMalloc((void**)&ms.win, (size_t)n * sizeof(Fr30)));
    DVP_HIP(hipMalloc((void**)&ms.wout, (size_t)n * sizeof(Fr30)));
    int src = to_even ? 1 : 0, dst = to_even ? 0 : 1;
    int ln = 31 - __builtin_clz(n);
    for (int d = 0; d < ln; ++d) {
      uint32_t nd = n >> d;
      size_t off = 2 * (size_t)(n - nd);
      hipLaunchKernelGGL(k_build_twiddles, dim3(cdiv(nd >> 1, TPB)), dim3(TPB), 0, st, c->layer(d), sl, nd, src, dst, ms.dec + off, ms.rec + off);
    }
    // the twists, bottom up: W_d from W_{d+1} (ping-pong), once per parity
    DevBuf wa, wb, ws, wd;
    DVP_TRY(wa.alloc((size_t)n * sizeof(Fr)));
    DVP_TRY(wb.alloc((size_t)n * sizeof(Fr)));
    DVP_TRY(ws.alloc((size_t)n * sizeof(Fr)));
    DVP_TRY(wd.alloc((size_t)n * sizeof(Fr)));
    for (int par = 0; par < 2; ++par) {
      const int parity = par == 0 ? src : dst;
      Fr* cur = wa.as<Fr>();
      Fr* nxt = wb.as<Fr>();
      const Fr* prev = nullptr;
      for (int d = ln - 1; d >= 0; --d) {
        uint32_t nd = n >> d;
        Fr* outp = d == 0 ? (par == 0 ? ws.as<Fr>() : wd.as<Fr>()) : cur;
        hipLaunchKernelGGL(k_twist_layer, dim3(cdiv(nd, TPB)), dim3(TPB), 0, st, c->layer(d), sl, nd, c->x0[d], parity, prev, outp);
        prev = outp;
        Fr* t = cur; cur = nxt; nxt = t;
      }
    }
    hipLaunchKernelGGL(k_twist_finish, dim3(cdiv(n, TPB)), dim3(TPB), 0, st, ws.as<Fr>(), wd.as<Fr>(), n, ms.win, ms.wout);
    DVP_HIP(hipGetLastError());
    DVP_HIP(hipStreamSynchronize(st));  // the scratch buffers go out of scope
  }
  c->mats[key] = ms;
  *out = &c->mats[key];
  return DVP_OK;
}

// host-side description of an extend's folded-in pointwise stages (ExtIo above; SM is picked at run time)
struct ExtIoSpec {
  int sm = 0;
  const Fr30* pre = nullptr;   // replaces the input twist (a table that already contains it)
  int lshift = 0;
  const Fr30* post = nullptr;  // replaces the output twist
  Fr* out = nullptr;
  const Fr30* tb = nullptr;
  const Fr30* tc = nullptr;
  const Fr* aux = nullptr;
  int ashift = 0, aoff = 0, lh = 0;
};
constexpr int EXT_TOP_MAX = 9;  // == FUSE_LOG - 2 (asserted in extend_io): the layers one k_extend_top launch covers
// can an extend of vectors of n values carry an ExtIoSpec?  Its first and last launch must be the LDS kernels (k_extend_top /
// k_extend_fused): 2 <= n <= 2^(FUSE_LOG + EXT_TOP_MAX); the combine (sm 3) pairs two vectors inside one k_extend_fused block
// (sm 3 inside one k_extend_fused block: n <= 1024; across the two vectors of a k_extend_top tile: n >= 4096.  n = 2048 -- one
// vector per k_extend_fused block, no top -- keeps the separate k_enter_combine)
static bool ext_io_ok(uint32_t n, int sm) { return n >= 2 && n <= (1u << (11 + EXT_TOP_MAX)) && (sm != 3 || n != 2048u); }
static int extend_io(dvp_ecfft* c, int sl, int to_even, const Fr* src_in, Fr* data, uint32_t batch, hipStream_t st, const ExtIoSpec* io);

// in-place extend of `batch` vectors of n = (N>>sl)/2 values
int extend_inplace(dvp_ecfft* c, int sl, int to_even, Fr* data, uint32_t batch, hipStream_t st) { return extend_io(c, sl, to_even, data, data, batch, st, nullptr); }
// the same out of place: `src` is read by the first pass only and left untouched (the prover keeps a, b, c on D for its K scalars;
// a copy of 3 x 32 MB before every extend was one more HBM round trip)
int extend_from(dvp_ecfft* c, int sl, int to_even, const Fr* src_in, Fr* data, uint32_t batch, hipStream_t st) {
  return extend_io(c, sl, to_even, src_in, data, batch, st, nullptr);
}
static int extend_io(dvp_ecfft* c, int sl, int to_even, const Fr* src_in, Fr* data, uint32_t batch, hipStream_t st, const ExtIoSpec* io) {
  uint32_t n = (c->n_leaves >> sl) >> 1;
  if (io && !ext_io_ok(n, io->sm)) return DVP_EINVAL;
  if (n <= 1) {
    if (src_in != data && n == 1) DVP_HIP(hipMemcpyAsync(data, src_in, (size_t)batch * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    return DVP_OK;
  }
  const Fr* src = src_in;  // becomes `data` after the first launch
  MatSet* ms;
  DVP_TRY(build_matset(c, sl, to_even, &ms, st));
  int ln = 31 - __builtin_clz(n);
  // Blocks are contiguous, so `batch` vectors of n behave like one vector of batch*n for the block
  // structure; the BATCH template only buys matrix reuse when the batch is small and n is large.
  // The first launch multiplies its input by the source twist, the last one its output by the destination twist (see "TWISTED
  // butterflies" above); a pass knows which it is from `first` / the `last` flag of its caller.
  bool first = true;
  auto pre_of = [&]() { const Fr30* r = first ? (io && io->pre ? io->pre : ms->win) : nullptr; first = false; return r; };
  const Fr30* wout = io && io->post ? io->post : ms->wout;
  const int sm_last = io ? io->sm : 0;
  // the ExtIo of one LDS-kernel launch: `pre` set = it is the first one (and reads with the caller's stride), `last` = it carries the
  // output side (twist alone, or the caller's folded-in store)
  auto fill = [&](auto& e, const Fr30* pre, bool last) {
    e.pre = pre;
    e.lshift = pre && io ? io->lshift : 0;
    if (last) {
      e.post = wout;
      if (io) {
        e.out = io->out; e.tb = io->tb; e.tc = io->tc; e.aux = io->aux; e.ashift = io->ashift; e.aoff = io->aoff; e.lh = io->lh;
      }
    }
  };
  auto pass = [&](const Fr30* base, int d, bool dec, bool last) {
    const Fr30* tws = base + 2 * (size_t)(n - (n >> d));
    const int lh = ln - d - 1;
    const Fr30* pre = pre_of();
    const Fr30* post = last ? wout : nullptr;
    const uint32_t nn = batch <= 4 && batch >= 2 ? n : (uint32_t)((size_t)batch * n);
    const dim3 g(cdiv(nn >> 1, TPB)), b(TPB);
#define DVP_BF(B) \
  do { if (dec) hipLaunchKernelGGL((k_butterfly<B, true>), g, b, 0, st, src, data, tws, lh, nn, pre, post, n); \
       else hipLaunchKernelGGL((k_butterfly<B, false>), g, b, 0, st, src, data, tws, lh, nn, pre, post, n); } while (0)
    if (batch == 4) DVP_BF(4); else if (batch == 3) DVP_BF(3); else if (batch == 2) DVP_BF(2); else DVP_BF(1);
#undef DVP_BF
    src = data;
  };
  // two layers per pass (k_butterfly4) while two top layers remain; `wide` = layer d, `narrow` = layer d + 1
  auto pass4 = [&](const Fr30* base, int d, bool dec, bool last) {
    const Fr30* wide = base + 2 * (size_t)(n - (n >> d));
    const Fr30* narrow = base + 2 * (size_t)(n - (n >> (d + 1)));
    const int lh2 = ln - d - 2;
    const Fr30* pre = pre_of();
    const Fr30* post = last ? wout : nullptr;
    const dim3 g(cdiv((size_t)(n >> 2) * batch, TPB)), b(TPB);
    const uint32_t nn = batch <= 4 && batch >= 2 ? n : (uint32_t)((size_t)batch * n);
    const dim3 g1(cdiv(nn >> 2, TPB));
#define DVP_BF4(B, GRID) \
  do { if (dec) hipLaunchKernelGGL((k_butterfly4<B, true>), GRID, b, 0, st, src, data, wide, narrow, lh2, nn, pre, post, n); \
       else hipLaunchKernelGGL((k_butterfly4<B, false>), GRID, b, 0, st, src, data, wide, narrow, lh2, nn, pre, post, n); } while (0)
    if (batch == 4) DVP_BF4(4, g); else if (batch == 3) DVP_BF4(3, g); else if (batch == 2) DVP_BF4(2, g); else DVP_BF4(1, g1);
#undef DVP_BF4
    src = data;
  };
  // three layers per pass (k_butterfly8): layers d (widest), d + 1, d + 2
  auto pass8 = [&](const Fr30* base, int d, bool dec, bool last) {
    const Fr30* t0 = base + 2 * (size_t)(n - (n >> d));
    const Fr30* t1 = base + 2 * (size_t)(n - (n >> (d + 1)));
    const Fr30* t2 = base + 2 * (size_t)(n - (n >> (d + 2)));
    const int lh3 = ln - d - 3;
    const Fr30* pre = pre_of();
    const Fr30* post = last ? wout : nullptr;
    const dim3 g(cdiv((size_t)(n >> 3) * batch, TPB)), b(TPB);
    const uint32_t nn = batch <= 4 && batch >= 2 ? n : (uint32_t)((size_t)batch * n);
    const dim3 g1(cdiv(nn >> 3, TPB));
#define DVP_BF8(B, GRID) \
  do { if (dec) hipLaunchKernelGGL((k_butterfly8<B, true>), GRID, b, 0, st, src, data, t0, t1, t2, lh3, nn, pre, post, n); \
       else hipLaunchKernelGGL((k_butterfly8<B, false>), GRID, b, 0, st, src, data, t0, t1, t2, lh3, nn, pre, post, n); } while (0)
    if (batch == 4) DVP_BF8(4, g); else if (batch == 3) DVP_BF8(3, g); else if (batch == 2) DVP_BF8(2, g); else DVP_BF8(1, g1);
#undef DVP_BF8
    src = data;
  };
  const int lb = ln < FUSE_LOG ? ln : FUSE_LOG;  // layers handled inside LDS
  const int top = ln - lb;
  // the top layers in groups of 3 (radix 8), then one group of 2 or 1; the recombine direction mirrors the decompose's grouping
  const int radix = (int)tune().ecfft_radix4;  // 0: one layer per pass, 1: two, 2: three, 3 (default): the top in one LDS-tiled launch
  // k_extend_top covers the last tl <= TOP_MAX top layers (the ones next to the fused bottom), whatever is above them goes in groups
  constexpr int TOP_MAX = FUSE_LOG - 2;  // >= 4 columns per tile row (128-byte lines); below 4 layers k_butterfly8 / 4 do as well
  static_assert(TOP_MAX == EXT_TOP_MAX && FUSE_LOG == 11, "ext_io_ok restates these");
  const int TOP_MIN = io ? 1 : 4;        // an extend with folded-in stages needs its first and last launch to be LDS kernels
  const int tl = ((radix >= 3 || io) && top >= TOP_MIN) ? (top < TOP_MAX ? top : TOP_MAX) : 0;
  const int d0 = top - tl;  // layers 0 .. d0 - 1 in groups, d0 .. top - 1 tiled
  auto pass_top = [&](const Fr30* base, bool dec, bool last) {
    const uint32_t nn = (uint32_t)((size_t)batch * n);  // the batch vectors are contiguous: blocks of n >> d0 values all the way through
    const dim3 g(nn >> FUSE_LOG), b(FUSE_TPB);
    if (dec) {
      ExtIo<0> e;
      fill(e, pre_of(), false);
      hipLaunchKernelGGL((k_extend_top<true, 0>), g, b, 0, st, src, data, base, n, ln, d0, tl, batch, e);
    } else {
      (void)pre_of();
#define DVP_TOP_LAST(SM_) \
  do { ExtIo<SM_> e; fill(e, nullptr, last); hipLaunchKernelGGL((k_extend_top<false, SM_>), g, b, 0, st, src, data, base, n, ln, d0, tl, batch, e); } while (0)
      const int sm = last ? sm_last : 0;
      if (sm == 1) DVP_TOP_LAST(1); else if (sm == 2) DVP_TOP_LAST(2); else if (sm == 3) DVP_TOP_LAST(3); else DVP_TOP_LAST(0);
#undef DVP_TOP_LAST
    }
    src = data;
  };
  std::vector<int> groups;
  for (int left = d0; left > 0;) {
    const int g = radix >= 2 && left >= 3 ? 3 : (radix >= 1 && left >= 2 ? 2 : 1);
    groups.push_back(g);
    left -= g;
  }
  {
    int d = 0;
    for (int g : groups) {
      if (g == 3) pass8(ms->dec, d, true, false); else if (g == 2) pass4(ms->dec, d, true, false); else pass(ms->dec, d, true, false);
      d += g;
    }
    if (tl) pass_top(ms->dec, true, false);
  }
  {
    size_t total = (size_t)batch * n;
    const Fr30* pre = pre_of();
    const bool last = top == 0;
#define DVP_FUSED(SM_) \
  do { ExtIo<SM_> e; fill(e, pre, last); \
       hipLaunchKernelGGL((k_extend_fused<SM_>), dim3(cdiv(total, FUSE_ELEMS)), dim3(FUSE_TPB), 0, st, src, data, ms->dec, ms->rec, n, ln, lb, total, e); } while (0)
    const int sm = last ? sm_last : 0;
    if (sm == 1) DVP_FUSED(1); else if (sm == 2) DVP_FUSED(2); else if (sm == 3) DVP_FUSED(3); else DVP_FUSED(0);
#undef DVP_FUSED
    src = data;
  }
  {
    if (tl) pass_top(ms->rec, false, d0 == 0);
    int d = d0;
    for (size_t k = groups.size(); k-- > 0;) {
      const int g = groups[k];
      d -= g;
      if (g == 3) pass8(ms->rec, d, false, d == 0); else if (g == 2) pass4(ms->rec, d, false, d == 0); else pass(ms->rec, d, false, d == 0);
    }
  }
  DVP_HIP(hipGetLastError());
  return DVP_OK;
}

static int get_xnn(dvp_ecfft* c, int sl, Fr** out, hipStream_t st) {
  std::lock_guard<std::mutex> g(c->mu);
  auto it = c->xnn.find(sl);
  if (it != c->xnn.end()) {
    *out = it->second;
    return DVP_OK;
  }
  uint32_t sz = c->n_leaves >> sl;
  Fr* p;
  DVP_HIP(hipMalloc((void**)&p, (size_t)sz * sizeof(Fr)));
  hipLaunchKernelGGL(k_pow_table, dim3(cdiv(sz, TPB)), dim3(TPB), 0, st, c->layer(0), sl, sz, (uint64_t)(sz >> 1), p);
  DVP_HIP(hipGetLastError());
  c->xnn[sl] = p;
  *out = p;
  return DVP_OK;
}

// ---- C ABI ---------------------------------------------------------------------------------------
static int ecfft_init(dvp_ecfft* c, uint32_t log_n, int shifted, uint32_t base_log);
extern "C" void dvp_ecfft_destroy(dvp_ecfft* c);

extern "C" int dvp_ecfft_create(uint32_t log_n, int shifted, uint32_t base_log, dvp_ecfft** out) {
  if (!out || log_n < 1 || log_n > (uint32_t)ECFFT_LOG_ORDER) return DVP_EINVAL;
  if (base_log == 0) base_log = log_n;
  if (base_log < log_n || base_log > (uint32_t)ECFFT_LOG_ORDER) return DVP_EINVAL;
  dvp_ecfft* c = new dvp_ecfft();
  int rc = ecfft_init(c, log_n, shifted, base_log);
  if (rc != DVP_OK) {
    dvp_ecfft_destroy(c);
    return rc;
  }
  *out = c;
  return DVP_OK;
}

static int ecfft_init(dvp_ecfft* c, uint32_t log_n, int shifted, uint32_t base_log) {
  c->log_n = (int)log_n;
  c->n_leaves = 1u << log_n;
  DVP_HIP(hipGetDevice(&c->device));
  const uint32_t N = c->n_leaves;
  Fr a = fr_from_limbs_mont(ECFFT_A_CANON);
  SwPt G{fr_from_limbs_mont(ECFFT_GX_CANON), fr_from_limbs_mont(ECFFT_GY_CANON), false};
  SwPt C{fr_from_limbs_mont(ECFFT_CX_CANON), fr_from_limbs_mont(ECFFT_CY_CANON), false};
  // g = 2^(28-log_n) * G  (src/ec_fft.rs:116-119); base generator for the D' shift (:72-82,151-155)
  SwPt g = G;
  for (int i = 0; i < ECFFT_LOG_ORDER - (int)log_n; ++i) g = sw_add(g, g, a);
  if (shifted) {
    SwPt bg = G;
    for (int i = 0; i < ECFFT_LOG_ORDER - (int)base_log; ++i) bg = sw_add(bg, bg, a);
    C = sw_add(C, bg, a);
  }
  // tab[j] = 2^j g ; tab[log_n-j] has order 2^j
  std::vector<SwPt> tab(log_n);
  tab[0] = g;
  for (uint32_t j = 1; j < log_n; ++j) tab[j] = sw_add(tab[j - 1], tab[j - 1], a);
  if (!fr_is_zero(tab[log_n - 1].y)) return DVP_EINVAL;  // order-2 point must have y == 0
  // q[j-1] = x of the point of order 2^j; pushed through the isogenies as we go
  std::vector<Fr> q(log_n);
  for (uint32_t j = 1; j <= log_n; ++j) q[j - 1] = tab[log_n - j].x;
  c->x0.resize(log_n);
  c->t.resize(log_n);
  Fr acur = a;
  for (uint32_t d = 0; d < log_n; ++d) {
    Fr x0 = q[d];
    Fr x0sq = fr_sqr(x0);
    Fr t = fr_add(fr_add(fr_dbl(x0sq), x0sq), acur);  // 3 x0^2 + a_d
    c->x0[d] = x0;
    c->t[d] = t;
    for (uint32_t j = d + 1; j < log_n; ++j) q[j] = fr_add(q[j], fr_mul(t, fr_inv(fr_sub(q[j], x0))));
    Fr t5 = fr_add(fr_dbl(fr_dbl(t)), t);
    acur = fr_sub(acur, t5);  // a_{d+1} = a_d - 5t
  }
  // device: layers
  c->layer_off.resize(log_n + 1);
  size_t off = 0;
  for (uint32_t d = 0; d <= log_n; ++d) {
    c->layer_off[d] = off;
    off += (size_t)(N >> d);
  }
  DVP_HIP(hipMalloc((void**)&c->layers, off * sizeof(Fr)));
  {
    std::vector<Fr> tx(log_n), ty(log_n);
    for (uint32_t j = 0; j < log_n; ++j) { tx[j] = tab[j].x; ty[j] = tab[j].y; }
    DevBuf dx, dy;
    DVP_TRY(dx.alloc(log_n * sizeof(Fr)));
    DVP_TRY(dy.alloc(log_n * sizeof(Fr)));
    DVP_HIP(hipMemcpy(dx.p, tx.data(), log_n * sizeof(Fr), hipMemcpyHostToDevice));
    DVP_HIP(hipMemcpy(dy.p, ty.data(), log_n * sizeof(Fr), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_leaves, dim3(cdiv(N, TPB)), dim3(TPB), 0, 0, dx.as<Fr>(), dy.as<Fr>(), C.x, C.y, (int)log_n,
                       c->layer(0), N);
    for (uint32_t d = 0; d < log_n; ++d) {
      uint32_t half = (N >> d) >> 1;
      hipLaunchKernelGGL(k_next_layer, dim3(cdiv(half, TPB)), dim3(TPB), 0, 0, c->layer(d), c->x0[d], c->t[d],
                         c->layer(d + 1), half);
    }
    DVP_HIP(hipGetLastError());
    DVP_HIP(hipDeviceSynchronize());
  }
  return DVP_OK;
}

static void free_exit_tables(dvp_ecfft* c);
static void free_enter_tables(dvp_ecfft* c);
static void free_leaf8(dvp_ecfft* c);
extern "C" void dvp_ecfft_destroy(dvp_ecfft* c) {
  if (!c) return;
  (void)hipFree(c->layers);
  for (auto& kv : c->mats) {
    (void)hipFree(kv.second.dec);
    (void)hipFree(kv.second.rec);
    (void)hipFree(kv.second.win);
    (void)hipFree(kv.second.wout);
  }
  for (auto& kv : c->xnn) (void)hipFree(kv.second);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->d_x0) (void)hipFree(c->d_x0);
  if (c->d_t) (void)hipFree(c->d_t);
  free_exit_tables(c);
  free_enter_tables(c);
  free_leaf8(c);
  delete c;
}

extern "C" uint32_t dvp_ecfft_log2_leaves(const dvp_ecfft* c) { return c ? (uint32_t)c->log_n : 0; }

extern "C" int dvp_ecfft_leaves(const dvp_ecfft* c, uint64_t* out) {
  if (!c || !out) return DVP_EINVAL;
  size_t n = c->n_leaves;
  DevBuf tmp;
  DVP_TRY(tmp.alloc(n * sizeof(Fr)));
  hipLaunchKernelGGL(k_from_mont, dim3(cdiv(n, TPB)), dim3(TPB), 0, 0, c->layer(0), tmp.as<Fr>(), n);
  DVP_HIP(hipGetLastError());
  DVP_HIP(hipMemcpy(out, tmp.p, n * sizeof(Fr), hipMemcpyDeviceToHost));
  return DVP_OK;
}

static int check_canonical_dev(const Fr* d, size_t n, hipStream_t st) {
  DevBuf bad;
  DVP_TRY(bad.alloc(8));
  unsigned long long init = ~0ull;
  DVP_HIP(hipMemcpyAsync(bad.p, &init, 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_check_canonical, dim3(cdiv(n, TPB)), dim3(TPB), 0, st, d, n, bad.as<unsigned long long>());
  DVP_HIP(hipMemcpyAsync(&init, bad.p, 8, hipMemcpyDeviceToHost, st));
  DVP_HIP(hipStreamSynchronize(st));
  if (init != ~0ull) {
    g_last_error_index = (int64_t)init;
    return DVP_EINVAL;
  }
  return DVP_OK;
}

extern "C" int dvp_ecfft_extend_dev(dvp_ecfft* c, const void* d_in, uint32_t batch, void* d_out, void* stream) {
  if (!c || !d_in || !d_out || batch == 0) return DVP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  size_t n = c->n_leaves >> 1;
  (void)n;
  return extend_from(c, 0, 0, (const Fr*)d_in, (Fr*)d_out, batch, st);  // the first pass reads d_in, everything after it d_out
}

extern "C" int dvp_ecfft_extend(dvp_ecfft* c, const uint64_t* evals, uint32_t batch, uint64_t* out) {
  if (!c || !evals || !out || batch == 0) return DVP_EINVAL;
  size_t bytes = (size_t)batch * (c->n_leaves >> 1) * sizeof(Fr);
  DevBuf buf;
  DVP_TRY(buf.alloc(bytes));
  DVP_HIP(hipMemcpy(buf.p, evals, bytes, hipMemcpyHostToDevice));
  DVP_TRY(check_canonical_dev(buf.as<Fr>(), bytes / sizeof(Fr), 0));
  DVP_TRY(extend_inplace(c, 0, 0, buf.as<Fr>(), batch, 0));
  DVP_HIP(hipMemcpy(out, buf.p, bytes, hipMemcpyDeviceToHost));
  return DVP_OK;
}

// ---- enter / exit on the stride-2^sl0 subtree (M = N >> sl0 leaves) ------------------------------------
static int ensure_scratch(dvp_ecfft* c) {
  if (!c->scratch) DVP_HIP(hipMalloc((void**)&c->scratch, (size_t)6 * c->n_leaves * sizeof(Fr)));
  return DVP_OK;
}

// enter: bottom-up over recursion depth k (sub-problem size sz = N>>k on the stride-2^k subtree).
// All sub-problems of a depth share the subtree, so each depth is ONE batched extend
// (batch 2*nsub, vectors of sz/2) plus one combine kernel.
namespace dvp {
// enter's combine tables of the stride-2^sl subtree (ExtIo SM 3): xe[i] = xnn[2i], t2[i] = wout[i] * xnn[2i + 1], multiplier-constant form
__global__ void __launch_bounds__(256) k_enter_fuse_tables(const Fr* __restrict__ xnn, const Fr30* __restrict__ wout, Fr30* __restrict__ xe,
                                                           Fr30* __restrict__ t2, uint32_t h) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= h) return;
  xe[i] = fr30_const(xnn[2 * i]);
  // (w R') * (y R) / R' = w y R: the Montgomery form of the product, lazily reduced
  t2[i] = fr30_const(fr30_canon(fr30_muladd(wout[i], fr30_from(xnn[2 * i + 1]), fr30_zero())));
}
}  // namespace dvp
namespace dvp {
// The three deepest levels of enter / exit (sub-problems of 8 values: h = 4, 2, 1) as ONE pass with a constant 8 x 8 matrix (round 6):
// every sub-problem of a level lives on the same strided subtree, so enter on 8 coefficients is the Vandermonde matrix V[j][e] =
// leaf_j^e of that subtree's 8 leaves and exit is its inverse -- 72 multiply-adds per sub-problem (8 x 8 plus one per output that
// brings the lazy sum back below 2 p), one read and one write of the vector, where the recursion ran three levels of extends and
// pointwise passes (enter 3 launches, exit 9).  Same field elements: the interpolating polynomial is unique.  The matrix is uniform
// over the launch (scalar loads); a thread owns one sub-problem.
__global__ void __launch_bounds__(256) k_leaf_matmul8(const Fr* __restrict__ in, Fr* __restrict__ out, const Fr30* __restrict__ mat /* [8][8] + one */,
                                                      uint32_t nsub) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nsub) return;
  Fr30 x[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) x[q] = fr30_from(in[8 * (size_t)c + q]);
  const Fr30 one = mat[64];
#pragma unroll 1
  for (int r = 0; r < 8; r += 2) {
    Fr30 a0 = fr30_zero(), a1 = fr30_zero();
#pragma unroll
    for (int q = 0; q < 8; ++q) fr30_muladd_x2(mat[8 * r + q], x[q], a0, mat[8 * r + 8 + q], x[q], a1, a0, a1);
    const Fr30 z = fr30_zero();
    fr30_muladd_x2(one, a0, z, one, a1, z, a0, a1);  // sums of eight lazy products (< 8.1 p) -> below 1.02 p
    out[8 * (size_t)c + r] = fr30_canon(a0);
    out[8 * (size_t)c + r + 1] = fr30_canon(a1);
  }
}
}  // namespace dvp
// V and V^-1 on the 8 leaves of the stride-2^(log_n - 3) subtree, in the multiplier's constant form, + the constant 1 (k_leaf_matmul8)
struct Leaf8 {
  Fr30* enter = nullptr;  // 65 entries
  Fr30* exit = nullptr;
};
static std::map<dvp_ecfft*, Leaf8> g_leaf8;
static std::mutex g_leaf8_mu;
static int get_leaf8(dvp_ecfft* c, Leaf8* out, hipStream_t st) {
  std::lock_guard<std::mutex> g(g_leaf8_mu);
  auto it = g_leaf8.find(c);
  if (it != g_leaf8.end()) {
    *out = it->second;
    return DVP_OK;
  }
  if (c->log_n < 3) return DVP_EINVAL;
  Fr leaf[8];
  const int k8 = c->log_n - 3;
  DVP_HIP(hipStreamSynchronize(st));
  for (int j = 0; j < 8; ++j) DVP_HIP(hipMemcpy(&leaf[j], c->layer(0) + ((size_t)j << k8), sizeof(Fr), hipMemcpyDeviceToHost));
  // Montgomery arithmetic on the host: V[j][e] = leaf_j^e, W = V^-1 by Gauss-Jordan (the leaves are distinct: V is invertible)
  Fr V[8][8], A[8][8], W[8][8];
  for (int j = 0; j < 8; ++j) {
    V[j][0] = fr_one_mont();
    for (int e = 1; e < 8; ++e) V[j][e] = fr_mul(V[j][e - 1], leaf[j]);
    for (int e = 0; e < 8; ++e) {
      A[j][e] = V[j][e];
      W[j][e] = j == e ? fr_one_mont() : fr_zero();
    }
  }
  for (int col = 0; col < 8; ++col) {
    int piv = col;
    while (piv < 8 && fr_is_zero(A[piv][col])) ++piv;
    if (piv == 8) return DVP_EINVAL;
    if (piv != col)
      for (int e = 0; e < 8; ++e) {
        Fr t = A[piv][e]; A[piv][e] = A[col][e]; A[col][e] = t;
        t = W[piv][e]; W[piv][e] = W[col][e]; W[col][e] = t;
      }
    const Fr inv = fr_inv(A[col][col]);
    for (int e = 0; e < 8; ++e) {
      A[col][e] = fr_mul(A[col][e], inv);
      W[col][e] = fr_mul(W[col][e], inv);
    }
    for (int r = 0; r < 8; ++r) {
      if (r == col || fr_is_zero(A[r][col])) continue;
      const Fr f = A[r][col];
      for (int e = 0; e < 8; ++e) {
        A[r][e] = fr_sub(A[r][e], fr_mul(f, A[col][e]));
        W[r][e] = fr_sub(W[r][e], fr_mul(f, W[col][e]));
      }
    }
  }
  Fr30 he[65], hx[65];
  for (int r = 0; r < 8; ++r)
    for (int q = 0; q < 8; ++q) {
      he[8 * r + q] = fr30_const(V[r][q]);
      hx[8 * r + q] = fr30_const(W[r][q]);
    }
  he[64] = hx[64] = fr30_const(fr_one_mont());
  Leaf8 t;
  DVP_HIP(hipMalloc((void**)&t.enter, sizeof(he)));
  DVP_HIP(hipMalloc((void**)&t.exit, sizeof(hx)));
  DVP_HIP(hipMemcpy(t.enter, he, sizeof(he), hipMemcpyHostToDevice));
  DVP_HIP(hipMemcpy(t.exit, hx, sizeof(hx), hipMemcpyHostToDevice));
  g_leaf8[c] = t;
  *out = t;
  return DVP_OK;
}
static void free_leaf8(dvp_ecfft* c) {
  std::lock_guard<std::mutex> g(g_leaf8_mu);
  auto it = g_leaf8.find(c);
  if (it == g_leaf8.end()) return;
  (void)hipFree(it->second.enter);
  (void)hipFree(it->second.exit);
  g_leaf8.erase(it);
}

struct EnterTab {
  Fr30* xe = nullptr;
  Fr30* t2 = nullptr;
};
static std::map<dvp_ecfft*, std::map<int, EnterTab>> g_enter_tabs;  // per ctx, per stride level
static std::mutex g_enter_mu;
static int get_enter_tab(dvp_ecfft* c, int sl, EnterTab* out, hipStream_t st) {
  {
    std::lock_guard<std::mutex> g(g_enter_mu);
    auto& m = g_enter_tabs[c];
    auto it = m.find(sl);
    if (it != m.end()) {
      *out = it->second;
      return DVP_OK;
    }
  }
  const uint32_t h = (c->n_leaves >> sl) >> 1;
  Fr* xnn;
  DVP_TRY(get_xnn(c, sl, &xnn, st));
  MatSet* ms;
  DVP_TRY(build_matset(c, sl, 0, &ms, st));
  EnterTab tb;
  DVP_HIP(hipMalloc((void**)&tb.xe, (size_t)h * sizeof(Fr30)));
  DVP_HIP(hipMalloc((void**)&tb.t2, (size_t)h * sizeof(Fr30)));
  hipLaunchKernelGGL(k_enter_fuse_tables, dim3(cdiv(h, TPB)), dim3(TPB), 0, st, xnn, ms->wout, tb.xe, tb.t2, h);
  DVP_HIP(hipGetLastError());
  std::lock_guard<std::mutex> g(g_enter_mu);
  g_enter_tabs[c][sl] = tb;
  *out = tb;
  return DVP_OK;
}

static int enter_core(dvp_ecfft* c, int sl0, const Fr* d_coeffs, Fr* d_out, hipStream_t st) {
  const uint32_t N = c->n_leaves, M = N >> sl0;
  DVP_TRY(ensure_scratch(c));
  Fr* odd = c->scratch + M;
  Fr* bufs[2] = {c->scratch, c->scratch + 2 * (size_t)M};
  const bool fold = tune().ecfft_fold != 0;
  const Fr* even = d_coeffs;  // read in place: the first level's extend and combine only read it
  int nb = 0;
  int k_first = c->log_n - 1;
  if (fold && M >= 8) {  // levels log_n - 1 .. log_n - 3 in one pass (k_leaf_matmul8)
    Leaf8 l8;
    DVP_TRY(get_leaf8(c, &l8, st));
    Fr* dst = (c->log_n - 3 == sl0) ? d_out : bufs[nb];
    hipLaunchKernelGGL(k_leaf_matmul8, dim3(cdiv(M / 8, TPB)), dim3(TPB), 0, st, d_coeffs, dst, l8.enter, M / 8);
    DVP_HIP(hipGetLastError());
    even = dst;
    nb ^= 1;
    k_first = c->log_n - 4;
  }
  for (int k = k_first; k >= sl0; --k) {
    uint32_t sz = N >> k, h = sz >> 1, nsub = 1u << (k - sl0);
    Fr* dst = (k == sl0) ? d_out : bufs[nb];
    if (fold && ext_io_ok(h, 3)) {
      // the extend reads `even`, works in `odd`, and its last launch recombines straight into dst
      EnterTab tb;
      DVP_TRY(get_enter_tab(c, k, &tb, st));
      ExtIoSpec io;
      io.sm = 3;
      io.out = dst;
      io.tb = tb.xe;
      io.tc = tb.t2;
      io.aux = even;
      io.lh = 31 - __builtin_clz(h);
      DVP_TRY(extend_io(c, k, 0, even, odd, 2 * nsub, st, &io));
    } else {
      if (h > 1) DVP_TRY(extend_from(c, k, 0, even, odd, 2 * nsub, st));  // (h == 1: a constant is its own extension)
      Fr* xnn;
      DVP_TRY(get_xnn(c, k, &xnn, st));
      hipLaunchKernelGGL(k_enter_combine, dim3(cdiv(nsub * h, TPB)), dim3(TPB), 0, st, even, h > 1 ? odd : even, xnn, dst, h, nsub * h);
      DVP_HIP(hipGetLastError());
    }
    even = dst;
    nb ^= 1;
  }
  if (c->log_n == sl0) DVP_HIP(hipMemcpyAsync(d_out, d_coeffs, sizeof(Fr), hipMemcpyDeviceToDevice, st));
  return DVP_OK;
}

namespace dvp {
// exit tables of the stride-2^sl subtree: xinv[i] = 1/xnn[2i], z0inv[i] = 1/Z_0(leaf 2i+1)
__global__ void __launch_bounds__(256)
k_exit_tables(const Fr* __restrict__ L0, int sl, uint32_t h, const Fr* __restrict__ xnn, const Fr* __restrict__ x0s,
              const Fr* __restrict__ ts, int kk, const Fr* __restrict__ c0p, Fr* __restrict__ xinv, Fr* __restrict__ z0inv) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= h) return;
  xinv[i] = fr_inv(xnn[2 * i]);
  Fr leaf = L0[(size_t)(2 * i + 1) << sl];
  z0inv[i] = fr_inv(vanish_chain(leaf, x0s, ts, kk, *c0p));
}

// t0[c*h+i] = ev[c*sz+2i] * xinv[i]
__global__ void __launch_bounds__(256)
k_exit_pre(const Fr* __restrict__ ev, const Fr* __restrict__ xinv, Fr* __restrict__ t0, uint32_t h, uint32_t total) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= total) return;
  uint32_t c = tid / h, i = tid - c * h;
  t0[tid] = fr_mul(xinv[i], ev[(size_t)c * 2 * h + 2 * i]);
}
// h1[c*h+i] = (ev[c*sz+2i+1] - g1[c*h+i]*xnn[2i+1]) * z0inv[i]
__global__ void __launch_bounds__(256)
k_exit_mid(const Fr* __restrict__ ev, const Fr* __restrict__ g1, const Fr* __restrict__ xnn, const Fr* __restrict__ z0inv,
           Fr* __restrict__ h1, uint32_t h, uint32_t total) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= total) return;
  uint32_t c = tid / h, i = tid - c * h;
  Fr d = fr_sub(ev[(size_t)c * 2 * h + 2 * i + 1], fr_mul(xnn[2 * i + 1], g1[tid]));
  h1[tid] = fr_mul(z0inv[i], d);
}
// out[c*sz+2i] = h0*ctab[2i], out[c*sz+2i+1] = h1*ctab[2i+1]
__global__ void __launch_bounds__(256)
k_exit_mulc(const Fr* __restrict__ h0, const Fr* __restrict__ h1, const Fr* __restrict__ ctab, Fr* __restrict__ out,
            uint32_t h, uint32_t total) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= total) return;
  uint32_t c = tid / h, i = tid - c * h;
  size_t o = (size_t)c * 2 * h + 2 * i;
  out[o] = fr_mul(ctab[2 * i], h0[tid]);
  out[o + 1] = fr_mul(ctab[2 * i + 1], h1[tid]);
}
// next[(2c)*h+i] = u0 ; next[(2c+1)*h+i] = (ev[c*sz+2i] - u0) * xinv[i]
__global__ void __launch_bounds__(256)
k_exit_post(const Fr* __restrict__ ev, const Fr* __restrict__ u0, const Fr* __restrict__ xinv, Fr* __restrict__ next,
            uint32_t h, uint32_t total) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= total) return;
  uint32_t c = tid / h, i = tid - c * h;
  Fr u = u0[tid];
  next[(size_t)(2 * c) * h + i] = u;
  next[(size_t)(2 * c + 1) * h + i] = fr_mul(xinv[i], fr_sub(ev[(size_t)c * 2 * h + 2 * i], u));
}

// exit's folded tables of one stride level (ExtIo; all in the multiplier's constant form, indexed by the position i < h):
//   P  = win * xinv                      first extend of a redc reads the even entries:      x = P[i] * ev[2 (c h + i)]
//   R  = P * ctab[2i]                    ... or the previous redc's even half h0:            x = R[i] * h0[c h + i]
//   nA = -(wout * xnn[2i+1] * z0inv)     its last pass stores h1 = B * ev_odd + nA * x       (B = z0inv; B2 = z0inv * ctab[2i+1] on h1)
//   XI = xinv                            the level's last extend stores (u, XI * (ev_even - u))
// win / wout are the twists of the even -> odd extend of this level (MatSet to_even = 0).
__global__ void __launch_bounds__(256)
k_exit_fuse_tables(const Fr30* __restrict__ win, const Fr30* __restrict__ wout, const Fr* __restrict__ xinv, const Fr* __restrict__ z0inv,
                   const Fr* __restrict__ xnn, const Fr* __restrict__ ctab, Fr30* __restrict__ P, Fr30* __restrict__ R, Fr30* __restrict__ nA,
                   Fr30* __restrict__ B, Fr30* __restrict__ B2, Fr30* __restrict__ XI, uint32_t h) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= h) return;
  const Fr30 z = fr30_zero();
  const Fr xi = xinv[i], zi = z0inv[i];
  // (w R') * (y R) / R' = w y R: the Montgomery form of the product
  const Fr p = fr30_canon(fr30_muladd(win[i], fr30_from(xi), z));
  const Fr a = fr30_canon(fr30_muladd(wout[i], fr30_from(xnn[2 * i + 1]), z));
  P[i] = fr30_const(p);
  R[i] = fr30_const(fr_mul(p, ctab[2 * i]));
  nA[i] = fr30_const(fr_neg(fr_mul(a, zi)));
  B[i] = fr30_const(zi);
  B2[i] = fr30_const(fr_mul(zi, ctab[2 * i + 1]));
  XI[i] = fr30_const(xi);
}

// the deepest exit level (h = 1: two evaluations per sub-problem, every extend the identity) in one pass: what k_exit_pre / _mid /
// _mulc / _pre / _mid / _post do with six launches and two copies.  out[2c] = u, out[2c + 1] = xinv (ev0 - u)
__global__ void __launch_bounds__(256)
k_exit_leaf(const Fr* __restrict__ ev, const Fr* __restrict__ xinv, const Fr* __restrict__ z0inv, const Fr* __restrict__ xnn,
            const Fr* __restrict__ ctab, Fr* __restrict__ out, uint32_t nsub) {
  uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nsub) return;
  const Fr xi = xinv[0], zi = z0inv[0], xo = xnn[1], ce = ctab[0], co = ctab[1];
  const Fr e0 = ev[2 * (size_t)c], e1 = ev[2 * (size_t)c + 1];
  const Fr g1 = fr_mul(xi, e0);
  const Fr h1 = fr_mul(zi, fr_sub(e1, fr_mul(xo, g1)));              // = h0
  const Fr g2 = fr_mul(xi, fr_mul(ce, h1));
  const Fr u = fr_mul(zi, fr_sub(fr_mul(co, h1), fr_mul(xo, g2)));
  out[2 * (size_t)c] = u;
  out[2 * (size_t)c + 1] = fr_mul(xi, fr_sub(e0, u));
}

// helpers for the z0z0 bootstrap
__global__ void __launch_bounds__(256) k_neg_from_mont_even(const Fr* __restrict__ xnn, Fr* __restrict__ out, uint32_t h) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < h) out[i] = fr_neg(fr_from_mont(xnn[2 * i]));
}
__global__ void __launch_bounds__(256) k_mul_canon(const Fr* __restrict__ a, const Fr* __restrict__ b, Fr* __restrict__ out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fr_mul(fr_to_mont(a[i]), b[i]);  // canonical product
}
// rho[j] = A[j] + (j >= h/2 ? 2*B[j-h/2] : 0), j < h ; rho[h..2h) = 0
__global__ void __launch_bounds__(256) k_rho(const Fr* __restrict__ A, const Fr* __restrict__ B, Fr* __restrict__ rho, uint32_t h) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= 2 * h) return;
  Fr r = fr_zero();
  if (j < h) {
    r = A[j];
    if (j >= h / 2) r = fr_add(r, fr_dbl(B[j - h / 2]));
  }
  rho[j] = r;
}
__global__ void __launch_bounds__(256) k_to_mont(const Fr* __restrict__ in, Fr* __restrict__ out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fr_to_mont(in[i]);
}
// h == 1: c[0] = c[1] = s0^2 (Montgomery)
__global__ void k_c_base(const Fr* __restrict__ L0, Fr* __restrict__ ctab) {
  Fr s = fr_sqr(L0[0]);
  ctab[0] = s;
  ctab[1] = s;
}
}  // namespace dvp

// device copies of the per-layer isogeny constants
int ecfft_device_consts(dvp_ecfft* c) {
  std::lock_guard<std::mutex> g(c->mu);
  if (!c->d_x0) {
    DVP_HIP(hipMalloc((void**)&c->d_x0, c->log_n * sizeof(Fr)));
    DVP_HIP(hipMalloc((void**)&c->d_t, c->log_n * sizeof(Fr)));
    DVP_HIP(hipMemcpy(c->d_x0, c->x0.data(), c->log_n * sizeof(Fr), hipMemcpyHostToDevice));
    DVP_HIP(hipMemcpy(c->d_t, c->t.data(), c->log_n * sizeof(Fr), hipMemcpyHostToDevice));
  }
  return DVP_OK;
}

struct ExitTab {
  Fr* xinv = nullptr;   // h
  Fr* z0inv = nullptr;  // h
  Fr* ctab = nullptr;   // sz, <Z_0^2 mod X^h> on the subtree (Montgomery)
  Fr30* fused = nullptr;  // 6 h: P | R | nA | B | B2 | XI (k_exit_fuse_tables); nullptr where the level is not folded (h == 1, or above the LDS kernels' reach)
};
static std::map<dvp_ecfft*, std::map<int, ExitTab>> g_exit_tabs;  // per ctx, per stride level
static std::mutex g_exit_mu;

static int exit_core(dvp_ecfft* c, int sl0, const Fr* d_evals, Fr* d_out, hipStream_t st);

// Build (once) the exit tables of every stride level >= sl, deepest first.
static int ensure_exit_tables(dvp_ecfft* c, int sl_min, hipStream_t st) {
  const uint32_t N = c->n_leaves;
  DVP_TRY(ecfft_device_consts(c));
  for (int sl = c->log_n - 1; sl >= sl_min; --sl) {
    {
      std::lock_guard<std::mutex> g(g_exit_mu);
      if (g_exit_tabs[c].count(sl)) continue;
    }
    uint32_t sz = N >> sl, h = sz >> 1;
    int kk = 31 - __builtin_clz(h);  // h = 2^kk even leaves -> kk isogenies collapse them
    ExitTab tb;
    DVP_HIP(hipMalloc((void**)&tb.xinv, (size_t)h * sizeof(Fr)));
    DVP_HIP(hipMalloc((void**)&tb.z0inv, (size_t)h * sizeof(Fr)));
    DVP_HIP(hipMalloc((void**)&tb.ctab, (size_t)sz * sizeof(Fr)));
    Fr* xnn;
    DVP_TRY(get_xnn(c, sl, &xnn, st));
    // the even leaves of the stride-2^sl tree all map to layer-kk leaf 0 of the strided tree = layers[kk][0]
    hipLaunchKernelGGL(k_exit_tables, dim3(cdiv(h, TPB)), dim3(TPB), 0, st, c->layer(0), sl, h, xnn, c->d_x0, c->d_t, kk,
                       c->layer(kk), tb.xinv, tb.z0inv);
    if (h == 1) {
      hipLaunchKernelGGL(k_c_base, dim3(1), dim3(1), 0, st, c->layer(0), tb.ctab);
    } else {
      // (1) z = Z_0 - X^h in coefficient form: exit on the even subtree of the evaluations -leaf^h
      DevBuf zlow, lo, hi, A, B, rho;
      DVP_TRY(zlow.alloc((size_t)h * sizeof(Fr)));
      DVP_TRY(lo.alloc((size_t)h * sizeof(Fr)));
      DVP_TRY(hi.alloc((size_t)h * sizeof(Fr)));
      DVP_TRY(A.alloc((size_t)h * sizeof(Fr)));
      DVP_TRY(B.alloc((size_t)h * sizeof(Fr)));
      DVP_TRY(rho.alloc((size_t)sz * sizeof(Fr)));
      hipLaunchKernelGGL(k_neg_from_mont_even, dim3(cdiv(h, TPB)), dim3(TPB), 0, st, xnn, lo.as<Fr>(), h);
      DVP_TRY(exit_core(c, sl + 1, lo.as<Fr>(), zlow.as<Fr>(), st));
      // (2) z = z_lo + X^(h/2) z_hi;  Z_0^2 mod X^h = z_lo^2 + 2 X^(h/2) (z_lo z_hi mod X^(h/2))
      DVP_HIP(hipMemsetAsync(lo.p, 0, (size_t)h * sizeof(Fr), st));
      DVP_HIP(hipMemsetAsync(hi.p, 0, (size_t)h * sizeof(Fr), st));
      DVP_HIP(hipMemcpyAsync(lo.p, zlow.p, (size_t)(h / 2) * sizeof(Fr), hipMemcpyDeviceToDevice, st));
      DVP_HIP(hipMemcpyAsync(hi.p, zlow.as<Fr>() + h / 2, (size_t)(h / 2) * sizeof(Fr), hipMemcpyDeviceToDevice, st));
      DVP_TRY(enter_core(c, sl + 1, lo.as<Fr>(), A.as<Fr>(), st));  // <z_lo> on the even subtree
      DVP_TRY(enter_core(c, sl + 1, hi.as<Fr>(), B.as<Fr>(), st));  // <z_hi>
      hipLaunchKernelGGL(k_mul_canon, dim3(cdiv(h, TPB)), dim3(TPB), 0, st, A.as<Fr>(), B.as<Fr>(), hi.as<Fr>(), h);  // <z_lo z_hi>
      hipLaunchKernelGGL(k_mul_canon, dim3(cdiv(h, TPB)), dim3(TPB), 0, st, A.as<Fr>(), A.as<Fr>(), lo.as<Fr>(), h);  // <z_lo^2>
      DVP_TRY(exit_core(c, sl + 1, lo.as<Fr>(), A.as<Fr>(), st));
      DVP_TRY(exit_core(c, sl + 1, hi.as<Fr>(), B.as<Fr>(), st));
      hipLaunchKernelGGL(k_rho, dim3(cdiv(sz, TPB)), dim3(TPB), 0, st, A.as<Fr>(), B.as<Fr>(), rho.as<Fr>(), h);
      // (3) evaluate on the whole subtree, keep in Montgomery form
      DevBuf ev;
      DVP_TRY(ev.alloc((size_t)sz * sizeof(Fr)));
      DVP_TRY(enter_core(c, sl, rho.as<Fr>(), ev.as<Fr>(), st));
      hipLaunchKernelGGL(k_to_mont, dim3(cdiv(sz, TPB)), dim3(TPB), 0, st, ev.as<Fr>(), tb.ctab, sz);
      DVP_HIP(hipGetLastError());
      DVP_HIP(hipStreamSynchronize(st));  // DevBufs go out of scope
    }
    DVP_HIP(hipGetLastError());
    if (ext_io_ok(h, 1)) {
      MatSet* ms;
      DVP_TRY(build_matset(c, sl, 0, &ms, st));
      DVP_HIP(hipMalloc((void**)&tb.fused, (size_t)6 * h * sizeof(Fr30)));
      Fr30* f = tb.fused;
      hipLaunchKernelGGL(k_exit_fuse_tables, dim3(cdiv(h, TPB)), dim3(TPB), 0, st, ms->win, ms->wout, tb.xinv, tb.z0inv, xnn, tb.ctab, f, f + h, f + 2 * (size_t)h,
                         f + 3 * (size_t)h, f + 4 * (size_t)h, f + 5 * (size_t)h, h);
      DVP_HIP(hipGetLastError());
    }
    std::lock_guard<std::mutex> g(g_exit_mu);
    g_exit_tabs[c][sl] = tb;
  }
  return DVP_OK;
}

// exit: top-down.  At depth k every sub-problem (sz evaluations on the stride-2^k subtree) is split into
// the evaluations of its low and high coefficient halves on the even subtree:
//   u = <P mod X^h> = redc(redc(P) * <Z_0^2 mod X^h>),   redc(P) = <P Z_0^-1 mod X^h>  (Montgomery-style)
//   lo <- u|even,   hi <- (P|even - u|even) / X^h|even
static int exit_core(dvp_ecfft* c, int sl0, const Fr* d_evals, Fr* d_out, hipStream_t st) {
  const uint32_t N = c->n_leaves, M = N >> sl0;
  if (M == 1) {
    DVP_HIP(hipMemcpyAsync(d_out, d_evals, sizeof(Fr), hipMemcpyDeviceToDevice, st));
    return DVP_OK;
  }
  DVP_TRY(ensure_exit_tables(c, sl0, st));
  DVP_TRY(ensure_scratch(c));
  Fr* S = c->scratch;
  Fr* pp[2] = {S + 4 * (size_t)M, S};  // M each: the levels' outputs, in turn (the last level writes d_out itself)
  Fr* t0 = S + (size_t)M;             // M/2  (also g1)
  Fr* h1 = S + (size_t)M + M / 2;     // M/2
  Fr* h0 = S + 2 * (size_t)M;         // M/2
  Fr* r1 = S + 3 * (size_t)M;         // M
  const Fr* cur = d_evals;  // the first level reads the caller's vector in place
  int w = 0;
  const bool leaf8 = tune().ecfft_fold != 0 && M >= 8;  // the levels below sz = 8 in one pass (k_leaf_matmul8)
  const int k_end = leaf8 ? c->log_n - 3 : c->log_n;
  for (int k = sl0; k < k_end; ++k) {
    uint32_t sz = N >> k, h = sz >> 1, nsub = 1u << (k - sl0), total = nsub * h;
    Fr* nxt = k == c->log_n - 1 ? d_out : pp[w];
    ExitTab tb;
    {
      std::lock_guard<std::mutex> g(g_exit_mu);
      tb = g_exit_tabs[c][k];
    }
    Fr* xnn;
    DVP_TRY(get_xnn(c, k, &xnn, st));
    dim3 grid(cdiv(total, TPB)), blk(TPB);
    auto redc = [&](const Fr* ev) -> int {  // leaves h0 (even part) and h1 (odd part)
      hipLaunchKernelGGL(k_exit_pre, grid, blk, 0, st, ev, tb.xinv, t0, h, total);
      DVP_TRY(extend_inplace(c, k, 0, t0, nsub, st));
      hipLaunchKernelGGL(k_exit_mid, grid, blk, 0, st, ev, t0, xnn, tb.z0inv, h1, h, total);
      DVP_TRY(extend_from(c, k, 1, h1, h0, nsub, st));  // (h == 1: a copy)
      return DVP_OK;
    };
    if (tb.fused && tune().ecfft_fold != 0) {
      // four extends and nothing else (round 6): every pointwise stage rides in the first / last pass of an extend (ExtIo), the
      // copy h1 -> h0 is the out-of-place first pass, and r1 = (ctab_even h0 | ctab_odd h1) is never written
      const Fr30 *P = tb.fused, *R = P + h, *nA = P + 2 * (size_t)h, *B = P + 3 * (size_t)h, *B2 = P + 4 * (size_t)h, *XI = P + 5 * (size_t)h;
      Fr* h1b = r1;  // the second redc's odd half (r1 itself is not needed any more)
      const int lh = 31 - __builtin_clz(h);
      ExtIoSpec a1;  // redc 1, even -> odd: t = xinv * ev_even in, h1 = z0inv * (ev_odd - xnn_odd * g1) out
      a1.sm = 1; a1.pre = P; a1.lshift = 1; a1.post = nA; a1.out = h1; a1.tb = B; a1.aux = cur; a1.ashift = 1; a1.aoff = 1;
      DVP_TRY(extend_io(c, k, 0, cur, t0, nsub, st, &a1));
      DVP_TRY(extend_from(c, k, 1, h1, h0, nsub, st));  // odd -> even: h0
      ExtIoSpec a2;  // redc 2 on r1 = (ctab_even h0 | ctab_odd h1), never materialised
      a2.sm = 1; a2.pre = R; a2.post = nA; a2.out = h1b; a2.tb = B2; a2.aux = h1;
      DVP_TRY(extend_io(c, k, 0, h0, t0, nsub, st, &a2));
      ExtIoSpec b2;  // odd -> even: u; next level's (lo, hi) = (u, xinv * (ev_even - u))
      b2.sm = 2; b2.out = nxt; b2.tb = XI; b2.aux = cur; b2.lh = lh;
      DVP_TRY(extend_io(c, k, 1, h1b, h0, nsub, st, &b2));
    } else if (h == 1 && tune().ecfft_fold != 0) {
      hipLaunchKernelGGL(k_exit_leaf, grid, blk, 0, st, cur, tb.xinv, tb.z0inv, xnn, tb.ctab, nxt, nsub);
    } else {
      DVP_TRY(redc(cur));
      hipLaunchKernelGGL(k_exit_mulc, grid, blk, 0, st, h0, h1, tb.ctab, r1, h, total);
      DVP_TRY(redc(r1));
      hipLaunchKernelGGL(k_exit_post, grid, blk, 0, st, cur, h0, tb.xinv, nxt, h, total);
    }
    DVP_HIP(hipGetLastError());
    cur = nxt;
    w ^= 1;
  }
  if (leaf8) {
    Leaf8 l8;
    DVP_TRY(get_leaf8(c, &l8, st));
    hipLaunchKernelGGL(k_leaf_matmul8, dim3(cdiv(M / 8, TPB)), dim3(TPB), 0, st, cur, d_out, l8.exit, M / 8);
    DVP_HIP(hipGetLastError());
  }
  return DVP_OK;
}

extern "C" int dvp_ecfft_enter_dev(dvp_ecfft* c, const void* d_coeffs, void* d_out, void* stream) {
  if (!c || !d_coeffs || !d_out) return DVP_EINVAL;
  return enter_core(c, 0, (const Fr*)d_coeffs, (Fr*)d_out, (hipStream_t)stream);
}
extern "C" int dvp_ecfft_exit_dev(dvp_ecfft* c, const void* d_evals, void* d_out, void* stream) {
  if (!c || !d_evals || !d_out) return DVP_EINVAL;
  return exit_core(c, 0, (const Fr*)d_evals, (Fr*)d_out, (hipStream_t)stream);
}

static int host_roundtrip(dvp_ecfft* c, const uint64_t* in, uint64_t* out, bool is_exit) {
  if (!c || !in || !out) return DVP_EINVAL;
  size_t bytes = (size_t)c->n_leaves * sizeof(Fr);
  DevBuf i, o;
  DVP_TRY(i.alloc(bytes));
  DVP_TRY(o.alloc(bytes));
  DVP_HIP(hipMemcpy(i.p, in, bytes, hipMemcpyHostToDevice));
  DVP_TRY(check_canonical_dev(i.as<Fr>(), c->n_leaves, 0));
  DVP_TRY(is_exit ? exit_core(c, 0, i.as<Fr>(), o.as<Fr>(), 0) : enter_core(c, 0, i.as<Fr>(), o.as<Fr>(), 0));
  DVP_HIP(hipMemcpy(out, o.p, bytes, hipMemcpyDeviceToHost));
  return DVP_OK;
}
extern "C" int dvp_ecfft_enter(dvp_ecfft* c, const uint64_t* coeffs, uint64_t* out) { return host_roundtrip(c, coeffs, out, false); }
extern "C" int dvp_ecfft_exit(dvp_ecfft* c, const uint64_t* evals, uint64_t* out) { return host_roundtrip(c, evals, out, true); }

// Z_D(x) (which = 0: even leaves) or Z_D'(x) (which = 1: odd leaves); host-side chain, O(log N) field ops
extern "C" int dvp_ecfft_vanish_at(const dvp_ecfft* c, int which, const uint64_t x[4], uint64_t out[4]) {
  if (!c || !x || !out || (which != 0 && which != 1)) return DVP_EINVAL;
  Fr xc;
  memcpy(xc.v, x, 32);
  if (!fr_is_canonical(xc)) return DVP_EINVAL;
  int kk = c->log_n - 1;
  Fr cpt[2];  // the two leaves of layer kk (Montgomery): even leaves collapse to [0], odd leaves to [1]
  DVP_HIP(hipMemcpy(cpt, c->layer(kk), 2 * sizeof(Fr), hipMemcpyDeviceToHost));
  Fr u = fr_to_mont(xc), v = fr_one_mont();
  for (int d = 0; d < kk; ++d) {
    Fr uv = fr_mul(u, v), vv = fr_sqr(v);
    Fr nu = fr_add(fr_sub(fr_sqr(u), fr_mul(c->x0[d], uv)), fr_mul(c->t[d], vv));
    Fr nv = fr_sub(uv, fr_mul(c->x0[d], vv));
    u = nu;
    v = nv;
  }
  Fr z = fr_from_mont(fr_sub(u, fr_mul(cpt[which], v)));
  memcpy(out, z.v, 32);
  return DVP_OK;
}

static void free_exit_tables(dvp_ecfft* c) {
  std::lock_guard<std::mutex> g(g_exit_mu);
  auto it = g_exit_tabs.find(c);
  if (it == g_exit_tabs.end()) return;
  for (auto& kv : it->second) {
    (void)hipFree(kv.second.xinv);
    (void)hipFree(kv.second.z0inv);
    (void)hipFree(kv.second.ctab);
    if (kv.second.fused) (void)hipFree(kv.second.fused);
  }
  g_exit_tabs.erase(it);
}
static void free_enter_tables(dvp_ecfft* c) {
  std::lock_guard<std::mutex> g(g_enter_mu);
  auto it = g_enter_tabs.find(c);
  if (it == g_enter_tabs.end()) return;
  for (auto& kv : it->second) {
    (void)hipFree(kv.second.xe);
    (void)hipFree(kv.second.t2);
  }
  g_enter_tabs.erase(it);
}

// Parity-test read-out of the butterfly matrices extend() runs on (include/dvpari_internal.h): what the reference keeps
// in FFTree::{decompose,recombine}_matrices and stores in sections 2 / 1 of its FFTR tree files (src/tree_io.rs:353-433).
namespace dvp {
__global__ void __launch_bounds__(256) k_mats_export(const Fr29* __restrict__ in, Fr* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t l[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) l[k] = in[i].l[k];
  out[i] = fr_from_mont(fr_from29(l));
}
}  // namespace dvp
extern "C" int dvp_debug_ecfft_matrices(dvp_ecfft* c, int to_even, int which, uint64_t* out) {
  if (!c || !out || (to_even != 0 && to_even != 1) || (which != 0 && which != 1) || c->log_n < 2) return DVP_EINVAL;
  // the TRUE 2x2 matrices (what the reference stores), rebuilt on demand: extend() itself runs on the twisted constants
  const uint32_t nv = c->n_leaves >> 1;
  const size_t n = ((size_t)nv - 1) * 4;
  DevBuf dec, rec, tmp;
  DVP_TRY(dec.alloc(n * sizeof(Fr29)));
  DVP_TRY(rec.alloc(n * sizeof(Fr29)));
  DVP_TRY(tmp.alloc(n * sizeof(Fr)));
  const int src = to_even ? 1 : 0, dst = to_even ? 0 : 1;
  const int ln = 31 - __builtin_clz(nv);
  for (int d = 0; d < ln; ++d) {
    const uint32_t nd = nv >> d;
    const size_t off = 4 * (size_t)(nv - nd);
    hipLaunchKernelGGL(k_build_mats, dim3(cdiv(nd >> 1, TPB)), dim3(TPB), 0, 0, c->layer(d), 0, nd, c->x0[d], src, dst, dec.as<Fr29>() + off,
                       rec.as<Fr29>() + off);
  }
  hipLaunchKernelGGL(k_mats_export, dim3(cdiv(n, TPB)), dim3(TPB), 0, 0, which ? rec.as<Fr29>() : dec.as<Fr29>(), tmp.as<Fr>(), n);
  DVP_HIP(hipGetLastError());
  DVP_HIP(hipMemcpy(out, tmp.p, n * sizeof(Fr), hipMemcpyDeviceToHost));
  return DVP_OK;
}
extern "C" int dvp_debug_ecfft_layer(const dvp_ecfft* c, uint32_t d, uint64_t* out) {
  if (!c || !out || d > (uint32_t)c->log_n) return DVP_EINVAL;
  const size_t n = c->n_leaves >> d;
  DevBuf tmp;
  DVP_TRY(tmp.alloc(n * sizeof(Fr)));
  hipLaunchKernelGGL(k_from_mont, dim3(cdiv(n, TPB)), dim3(TPB), 0, 0, c->layer((int)d), tmp.as<Fr>(), n);
  DVP_HIP(hipGetLastError());
  DVP_HIP(hipMemcpy(out, tmp.p, n * sizeof(Fr), hipMemcpyDeviceToHost));
  return DVP_OK;
}
