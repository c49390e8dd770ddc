// Proof::prove on gfx950 -- the orchestration of src/proving.rs:426-688 with every vector stage on
// the device; since round 5 the BLAKE3 transcript (src/proving.rs:79-198) runs there too (k_transcript; the host
// flavour stays behind DVP_PROVE_HOST_TRANSCRIPT) and the host only assembles the 118 bytes.  Stage map (reference line -> kernel):
//   get_matrix_evaluations_from_witness   :348-403  -> k_r1cs_eval  (CSR rows, one thread per row; the
//                                                      Vandermonde fold C' = C - D is applied on the fly)
//   multi_scalar_mul(assignment, g_m)     :462-463  \  one MSM over [w | q2] x [g_m | g_q]
//   multi_scalar_mul(q_vals2, g_q)        :511-512  /  (commit_p = msm_q + msm_gm, :515)
//   extend_evals                          :410-422  -> dvp_ecfft batched extend (4 vectors)
//   r_vals2 / q_vals2                     :492-508  -> k_quotient
//   transcript -> alpha                   :517-558  -> k_transcript (or host BLAKE3) + k_alpha_denoms (domain check)
//   barycentric evaluations a0,b0,i0      :561-594  -> k_bary3_partial / k_bary3_final
//     (Z(alpha) comes from the isogeny chain instead of a Horner pass over z_poly)
//   denom_invs, denom_invs2               :599-616  -> k_alpha_denoms + batch inverse
//   k_scalar_a / _b / r_vals / k_scalar_r :619-654  -> k_kscalars (interleaved [D_i, D'_i])
//   multi_scalar_mul(srs_s_k, srs_g_k)    :666-680  -> MSM over 4m points
// The prover-side precomputes of prover_prepares_precomputes (:225-325) -- barycentric weights and
// 1/Z_D on D' -- are regenerated on the device from the isogeny chain (k_domain_tables), so neither
// z_poly nor the FFTR tree cache is needed.
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "blake3.h"
#include "common.h"
#include "ecfft_internal.h"
#include "k233.cuh"

extern "C" int dvp_ecfft_create(uint32_t log_n, int shifted, uint32_t base_log, dvp_ecfft** out);
extern "C" void dvp_ecfft_destroy(dvp_ecfft* c);
int ecfft_device_consts(dvp_ecfft* c);

namespace dvp {

int msm_affine_dev(const void* d_scalars, const void* d_bases, const void* d_inf, size_t n, void* d_out_xy,
                   void* d_out_inf, hipStream_t st);
int decode_dev(const uint8_t* d_enc, size_t n, Aff* d_out, uint8_t* d_inf, hipStream_t st);
int encode_dev(const Aff* d_pts, const uint8_t* d_inf, size_t n, uint8_t* d_out, hipStream_t st);
int encode_point_dev(const Aff* d_pt, const uint32_t* d_inf32, uint8_t* d_out, hipStream_t st);
int batch_inverse_dev(Fr* d, size_t n, hipStream_t st);
struct MsmFixedCtx;
int msm_fixed_create(const Aff* d_bases, uint32_t n_total, size_t range_hint, MsmFixedCtx** out);
int msm_sum_points_dev(const void* d_pts, uint32_t pt_stride_bytes, const void* d_inf32, uint32_t n, uint32_t inf_stride, void* d_out_xy,
                       void* d_out_inf, hipStream_t st);
void msm_fixed_destroy(MsmFixedCtx* c);
int msm_fixed_info(const MsmFixedCtx* c, int* cbits, int* windows);
uint64_t msm_fixed_table_bytes(const MsmFixedCtx* c, int* signed_flavour);
const void* msm_fixed_table_ptr(const MsmFixedCtx* c);
int msm_fixed_dev_enc(const MsmFixedCtx* c, const void* d_scalars, const void* d_inf, uint32_t lo, uint32_t hi, void* d_out_xy,
                      void* d_out_inf, void* d_out_enc, void* h_copy, const void* d_copy, size_t copy_bytes, hipStream_t st,
                      unsigned long long* d_err_defer);
int msm_affine_dev_enc(const void* d_scalars, const void* d_bases, const void* d_inf, size_t n, void* d_out_xy, void* d_out_inf,
                       void* d_out_enc, void* h_copy, const void* d_copy, size_t copy_bytes, hipStream_t st, unsigned long long* d_err_defer);
int msm_fixed_dev(const MsmFixedCtx* c, const void* d_scalars, const void* d_inf, uint32_t lo, uint32_t hi, void* d_out_xy,
                  void* d_out_inf, hipStream_t st);

// ---- domain tables from the isogeny chain ---------------------------------------------------------
// For S = even leaves (D) of the 2m-leaf tree, Z_S = U - c0 V with (U,V) the projective image under
// the first kk = log2(m) isogenies.  Forward-mode derivative gives Z_S'(s_i) for the barycentric
// weights; Z_S on the other half gives z_vals2inv.
//   bar_w[i]   = 1 / Z_S'(S[i])          (Montgomery)
//   zinv_o[i]  = 1 / Z_S(other[i])       (Montgomery)
// which = 0: S = even leaves, other = odd leaves; which = 1: mirrored.
__global__ void __launch_bounds__(256)
k_domain_tables(const Fr* __restrict__ L0, uint32_t m, const Fr* __restrict__ x0s, const Fr* __restrict__ ts, int kk,
                const Fr* __restrict__ ctop /* layer kk: 2 leaves */, int which, Fr* __restrict__ bar_w, Fr* __restrict__ zinv_o) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  Fr c0 = ctop[which];
  Fr s = L0[2 * (size_t)i + which];
  Fr o = L0[2 * (size_t)i + 1 - which];
  zinv_o[i] = fr_inv(vanish_chain(o, x0s, ts, kk, c0));
  // derivative at s
  Fr u = s, v = fr_one_mont(), du = fr_one_mont(), dv = fr_zero();
  for (int d = 0; d < kk; ++d) {
    Fr x0 = x0s[d], t = ts[d];
    Fr uv = fr_mul(u, v), vv = fr_sqr(v);
    Fr duv = fr_add(fr_mul(du, v), fr_mul(u, dv));  // (uv)'
    Fr dvv = fr_dbl(fr_mul(v, dv));                 // (v^2)'
    Fr nu = fr_add(fr_sub(fr_sqr(u), fr_mul(x0, uv)), fr_mul(t, vv));
    Fr nv = fr_sub(uv, fr_mul(x0, vv));
    Fr ndu = fr_add(fr_sub(fr_dbl(fr_mul(u, du)), fr_mul(x0, duv)), fr_mul(t, dvv));
    Fr ndv = fr_sub(duv, fr_mul(x0, dvv));
    u = nu; v = nv; du = ndu; dv = ndv;
  }
  bar_w[i] = fr_inv(fr_sub(du, fr_mul(c0, dv)));
}

// contiguous Montgomery copies of D and D'
__global__ void __launch_bounds__(256) k_split_domains(const Fr* __restrict__ L0, uint32_t m, Fr* __restrict__ d, Fr* __restrict__ d2) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  d[i] = L0[2 * (size_t)i];
  d2[i] = L0[2 * (size_t)i + 1];
}

// ---- stage kernels ---------------------------------------------------------------------------------
struct Csr {
  const uint32_t* row_ptr;  // n_rows + 1
  const uint32_t* wire;
  const uint32_t* coeff;
  uint32_t n_rows;
};

__device__ __forceinline__ Fr csr_row(const Csr& m, uint32_t row, const Fr* __restrict__ coeffs_m, const Fr* __restrict__ w) {
  Fr acc = fr_zero();
  if (row >= m.n_rows) return acc;
  for (uint32_t k = m.row_ptr[row]; k < m.row_ptr[row + 1]; ++k)
    acc = fr_add(acc, fr_mul(coeffs_m[m.coeff[k]], w[m.wire[k]]));  // eval_row, src/gnark_r1cs.rs:273-280
  return acc;
}

// E = [a | b | c' | i] (4 x m, canonical).  c' = C w - i(d), i(d) = sum_j pub_j d^j (src/gnark_r1cs.rs:333-399)
__global__ void __launch_bounds__(256)
k_r1cs_eval(Csr A, Csr B, Csr C, const Fr* __restrict__ coeffs_m, const Fr* __restrict__ w, const Fr* __restrict__ dom_m,
            uint32_t npub, uint32_t m, Fr* __restrict__ E, unsigned long long* __restrict__ unsat) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  Fr a = csr_row(A, i, coeffs_m, w), b = csr_row(B, i, coeffs_m, w), cw = csr_row(C, i, coeffs_m, w);
  Fr d = dom_m[i];
  Fr iv = fr_zero();
  for (int j = (int)npub - 1; j >= 0; --j) iv = fr_add(fr_mul(d, iv), w[1 + j]);  // Horner, canonical
  // a*b == c' + i  <=>  a*b == C w      (assert_eq!, src/proving.rs:389-395)
  if (!fr_eq(fr_mul(fr_to_mont(a), b), cw)) atomicMin(unsat, (unsigned long long)i);
  E[i] = a;
  E[(size_t)m + i] = b;
  E[2 * (size_t)m + i] = fr_sub(cw, iv);
  E[3 * (size_t)m + i] = iv;
}

// r2 = a2 b2 - i2 ; q2 = (r2 - c2) * z2inv        (src/proving.rs:492-508)
// HORNER: i has only npub coefficients, so for a short public input i2 = i(d'_i) is evaluated directly (the same
// canonical value the extend of i's evaluations yields, at npub products instead of an ECFFT pass).
template <bool HORNER>
__global__ void __launch_bounds__(256)
k_quotient(Fr* __restrict__ E2, const Fr* __restrict__ z2inv_m, uint32_t m, const Fr* __restrict__ w, const Fr* __restrict__ dom2_m,
           uint32_t npub, Fr* __restrict__ r2, Fr* __restrict__ q2) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  Fr a = E2[i], b = E2[(size_t)m + i], c = E2[2 * (size_t)m + i], iv;
  if (HORNER) {
    Fr d = dom2_m[i];
    iv = fr_zero();
    for (int j = (int)npub - 1; j >= 0; --j) iv = fr_add(fr_mul(d, iv), w[1 + j]);
    E2[3 * (size_t)m + i] = iv;
  } else {
    iv = E2[3 * (size_t)m + i];
  }
  Fr r = fr_sub(fr_mul(fr_to_mont(a), b), iv);
  r2[i] = r;
  q2[i] = fr_mul(z2inv_m[i], fr_sub(r, c));
}

// den[i] = d_i - alpha, den2[i] = d'_i - alpha (canonical); flags alpha in D u D' (src/proving.rs:548-556)
__global__ void __launch_bounds__(256)
k_alpha_denoms(const Fr* __restrict__ d_m, const Fr* __restrict__ d2_m, const Fr* __restrict__ alpha_m_p, uint32_t m, Fr* __restrict__ den,
               Fr* __restrict__ den2, unsigned long long* __restrict__ hit) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const Fr alpha_m = *alpha_m_p;  // from the host (dvp_prove_challenge) or from k_transcript, in the prover's own block
  Fr x = fr_sub(d_m[i], alpha_m), y = fr_sub(d2_m[i], alpha_m);
  if (fr_is_zero(x) || fr_is_zero(y)) atomicMin(hit, (unsigned long long)i);
  den[i] = fr_from_mont(x);
  den2[i] = fr_from_mont(y);
}

// partial sums of  y_i * w_i * dinv_i  for y in {a, b, i}   (dinv = 1/(d_i - alpha), canonical)
__global__ void __launch_bounds__(256)
k_bary3_partial(const Fr* __restrict__ E, const Fr* __restrict__ barw_m, const Fr* __restrict__ dinv, uint32_t m,
                Fr* __restrict__ partial /* [3][nb] Montgomery */) {
  __shared__ Fr sh[256];
  Fr s[3] = {fr_zero(), fr_zero(), fr_zero()};
  // four products per element (round 6; eight before): barw (Montgomery) times the canonical dinv is the canonical product k, and
  // fr_mul(k, y) of two canonical values is k y / R -- the sums carry one factor 1 / R, which the block's ONE thread that writes
  // the partial takes back out (two products by R^2: x / R -> x -> x R, the Montgomery partial k_bary3_final expects)
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
    Fr k = fr_mul(barw_m[i], dinv[i]);
    s[0] = fr_add(s[0], fr_mul(k, E[i]));
    s[1] = fr_add(s[1], fr_mul(k, E[(size_t)m + i]));
    s[2] = fr_add(s[2], fr_mul(k, E[3 * (size_t)m + i]));
  }
#pragma unroll
  for (int v = 0; v < 3; ++v) {  // (unrolled: a runtime index put s[] / res[] into 112 B of scratch per lane)
    sh[threadIdx.x] = s[v];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) sh[threadIdx.x] = fr_add(sh[threadIdx.x], sh[threadIdx.x + o]);
      __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)v * gridDim.x + blockIdx.x] = fr_mul(fr_mul(sh[0], fr_r2()), fr_r2());
    __syncthreads();
  }
}
// out[0..2] = a0, b0, i0 (canonical); out[3] = r0 = a0 b0 - i0.   P(alpha) = -Z(alpha) * sum y w / (d - alpha)
__global__ void __launch_bounds__(256) k_bary3_final(const Fr* __restrict__ partial, uint32_t nb, const Fr* __restrict__ neg_z_alpha_m_p, Fr* __restrict__ out) {
  __shared__ Fr sh[256];
  Fr res[3];
  const Fr neg_z_alpha_m = *neg_z_alpha_m_p;
#pragma unroll
  for (int v = 0; v < 3; ++v) {  // (unrolled: a runtime index put s[] / res[] into 112 B of scratch per lane)
    Fr s = fr_zero();
    for (uint32_t i = threadIdx.x; i < nb; i += 256) s = fr_add(s, partial[(size_t)v * nb + i]);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) sh[threadIdx.x] = fr_add(sh[threadIdx.x], sh[threadIdx.x + o]);
      __syncthreads();
    }
    res[v] = fr_mul(sh[0], neg_z_alpha_m);  // Montgomery
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = fr_from_mont(res[0]);
    out[1] = fr_from_mont(res[1]);
    out[2] = fr_from_mont(res[2]);
    out[3] = fr_from_mont(fr_sub(fr_mul(res[0], res[1]), res[2]));
  }
}

// S = [k_a | k_b | k_r] with k_r interleaved [D_i, D'_i]   (src/proving.rs:619-654)
__global__ void __launch_bounds__(256)
k_kscalars(const Fr* __restrict__ E, const Fr* __restrict__ r2, const Fr* __restrict__ dinv, const Fr* __restrict__ dinv2,
           const Fr* __restrict__ abir0 /* a0,b0,i0,r0 canonical */, uint32_t m, Fr* __restrict__ S) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  Fr a0 = abir0[0], b0 = abir0[1], r0 = abir0[3];
  Fr a = E[i], b = E[(size_t)m + i], iv = E[3 * (size_t)m + i];
  Fr di = fr_to_mont(dinv[i]), di2 = fr_to_mont(dinv2[i]);
  S[i] = fr_mul(di, fr_sub(a, a0));
  S[(size_t)m + i] = fr_mul(di, fr_sub(b, b0));
  Fr r = fr_sub(fr_mul(fr_to_mont(a), b), iv);
  S[2 * (size_t)m + 2 * (size_t)i] = fr_mul(di, fr_sub(r, r0));
  S[2 * (size_t)m + 2 * (size_t)i + 1] = fr_mul(di2, fr_sub(r2[i], r0));
}

// ---- index-sharded challenge phase (pointwise / barycentric row): every stage below works on a slice ----
// den[i] = dom[i] - alpha for i in [lo, hi) of ONE domain (canonical); flags alpha in that slice of the domain
__global__ void __launch_bounds__(256)
k_alpha_denoms_range(const Fr* __restrict__ dom_m, Fr alpha_m, uint32_t lo, uint32_t hi, Fr* __restrict__ den, unsigned long long* __restrict__ hit) {
  uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hi) return;
  Fr x = fr_sub(dom_m[i], alpha_m);
  if (fr_is_zero(x)) atomicMin(hit, (unsigned long long)i);
  den[i] = fr_from_mont(x);
}
// the partial sums of k_bary3_partial over [lo, hi) only
__global__ void __launch_bounds__(256)
k_bary3_partial_range(const Fr* __restrict__ E, const Fr* __restrict__ barw_m, const Fr* __restrict__ dinv, uint32_t m, uint32_t lo,
                      uint32_t hi, Fr* __restrict__ partial /* [3][nb] Montgomery */) {
  __shared__ Fr sh[256];
  Fr s[3] = {fr_zero(), fr_zero(), fr_zero()};
  for (uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += gridDim.x * blockDim.x) {
    Fr k = fr_mul(barw_m[i], dinv[i]);  // (four products per element: see k_bary3_partial)
    s[0] = fr_add(s[0], fr_mul(k, E[i]));
    s[1] = fr_add(s[1], fr_mul(k, E[(size_t)m + i]));
    s[2] = fr_add(s[2], fr_mul(k, E[3 * (size_t)m + i]));
  }
#pragma unroll
  for (int v = 0; v < 3; ++v) {  // (unrolled: a runtime index put s[] / res[] into 112 B of scratch per lane)
    sh[threadIdx.x] = s[v];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) sh[threadIdx.x] = fr_add(sh[threadIdx.x], sh[threadIdx.x + o]);
      __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)v * gridDim.x + blockIdx.x] = fr_mul(fr_mul(sh[0], fr_r2()), fr_r2());
    __syncthreads();
  }
}
// block partials -> one 128-byte record: 3 Montgomery sums (this rank's share of sum y w / (d - alpha), y in {a, b, i}),
// then the rank's alpha-in-domain flag
__global__ void __launch_bounds__(256)
k_bary3_record(const Fr* __restrict__ partial, uint32_t nb, const unsigned long long* __restrict__ hit, Fr* __restrict__ rec) {
  __shared__ Fr sh[256];
#pragma unroll
  for (int v = 0; v < 3; ++v) {  // (unrolled: a runtime index put s[] / res[] into 112 B of scratch per lane)
    Fr s = fr_zero();
    for (uint32_t i = threadIdx.x; i < nb; i += 256) s = fr_add(s, partial[(size_t)v * nb + i]);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) sh[threadIdx.x] = fr_add(sh[threadIdx.x], sh[threadIdx.x + o]);
      __syncthreads();
    }
    if (threadIdx.x == 0) rec[v] = sh[0];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    Fr f = fr_zero();
    const unsigned long long h = *hit;
    f.v[0] = (uint32_t)h;
    f.v[1] = (uint32_t)(h >> 32);
    rec[3] = f;
  }
}
// n gathered records -> a0, b0, i0, r0 (what k_bary3_final does with block partials) and the combined flag
__global__ void __launch_bounds__(64)
k_bary3_from_records(const Fr* __restrict__ recs /* n x 4 Fr */, uint32_t n, Fr neg_z_alpha_m, Fr* __restrict__ out,
                     unsigned long long* __restrict__ hit) {
  if (threadIdx.x != 0) return;
  Fr res[3];
  unsigned long long h = ~0ull;
#pragma unroll
  for (int v = 0; v < 3; ++v) {  // (unrolled: a runtime index put s[] / res[] into 112 B of scratch per lane)
    Fr s = fr_zero();
    for (uint32_t r = 0; r < n; ++r) s = fr_add(s, recs[(size_t)r * 4 + v]);
    res[v] = fr_mul(s, neg_z_alpha_m);
  }
  for (uint32_t r = 0; r < n; ++r) {
    const Fr& f = recs[(size_t)r * 4 + 3];
    h = min(h, (unsigned long long)f.v[0] | ((unsigned long long)f.v[1] << 32));
  }
  out[0] = fr_from_mont(res[0]);
  out[1] = fr_from_mont(res[1]);
  out[2] = fr_from_mont(res[2]);
  out[3] = fr_from_mont(fr_sub(fr_mul(res[0], res[1]), res[2]));
  *hit = h;
}
// S[k] for k in [k_lo, k_hi) of S = [k_a | k_b | k_r] (k_kscalars restricted to a range of the OUTPUT index)
__global__ void __launch_bounds__(256)
k_kscalars_range(const Fr* __restrict__ E, const Fr* __restrict__ r2, const Fr* __restrict__ dinv, const Fr* __restrict__ dinv2,
                 const Fr* __restrict__ abir0, uint32_t m, size_t k_lo, size_t k_hi, Fr* __restrict__ S) {
  size_t k = k_lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= k_hi) return;
  if (k < m) {
    S[k] = fr_mul(fr_to_mont(dinv[k]), fr_sub(E[k], abir0[0]));
  } else if (k < 2 * (size_t)m) {
    size_t i = k - m;
    S[k] = fr_mul(fr_to_mont(dinv[i]), fr_sub(E[(size_t)m + i], abir0[1]));
  } else {
    size_t j = k - 2 * (size_t)m, i = j >> 1;
    Fr r0 = abir0[3];
    if (j & 1) {
      S[k] = fr_mul(fr_to_mont(dinv2[i]), fr_sub(r2[i], r0));
    } else {
      Fr r = fr_sub(fr_mul(fr_to_mont(E[i]), E[(size_t)m + i]), E[3 * (size_t)m + i]);
      S[k] = fr_mul(fr_to_mont(dinv[i]), fr_sub(r, r0));
    }
  }
}


// ---- the Fiat-Shamir transcript on the device (round 5) ----------------------------------------------------------------
// Transcript::output (src/proving.rs:164-197) needs four single-chunk BLAKE3 hashes once commit_p exists -- H(commit_p),
// H(public inputs as 29-byte LE), H(H_wc || H_pi), H(H_ct || H_rt); H_ct = H(H(empty) || H(empty)) is a constant the host passes in --
// and alpha is the result with its top four bytes cleared (:192).  One lane of one wave does that right behind the commitment MSM's
// tail kernel, so dvp_prove_dev has no host round trip between its two MSMs (the host is already enqueueing the second MSM's sort while
// the first one's rounds run).  Written from the BLAKE3 specification like blake3.h (the host flavour, which the phased entries and
// n_public > 35 keep using); the two are compared in tests/ (dvp_debug_transcript_dev).
namespace b3d {
struct Words8 { uint32_t w[8]; };
__device__ __forceinline__ uint32_t rotr(uint32_t x, uint32_t n) { return (x >> n) | (x << (32 - n)); }
#define DVP_B3G(a, b, c, d, mx, my)                      \
  do {                                                   \
    s[a] = s[a] + s[b] + (mx); s[d] = rotr(s[d] ^ s[a], 16); \
    s[c] = s[c] + s[d];        s[b] = rotr(s[b] ^ s[c], 12); \
    s[a] = s[a] + s[b] + (my); s[d] = rotr(s[d] ^ s[a], 8);  \
    s[c] = s[c] + s[d];        s[b] = rotr(s[b] ^ s[c], 7);  \
  } while (0)
// one compression of a single-chunk hash (counter 0): cv' = first eight output words
__device__ __forceinline__ void compress(uint32_t cv[8], const uint32_t blk[16], uint32_t block_len, uint32_t flags) {
  constexpr uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
  constexpr int PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
  uint32_t s[16], m[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = cv[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s[8 + i] = IV[i];
  s[12] = 0; s[13] = 0; s[14] = block_len; s[15] = flags;
#pragma unroll
  for (int i = 0; i < 16; ++i) m[i] = blk[i];
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    DVP_B3G(0, 4, 8, 12, m[0], m[1]); DVP_B3G(1, 5, 9, 13, m[2], m[3]); DVP_B3G(2, 6, 10, 14, m[4], m[5]); DVP_B3G(3, 7, 11, 15, m[6], m[7]);
    DVP_B3G(0, 5, 10, 15, m[8], m[9]); DVP_B3G(1, 6, 11, 12, m[10], m[11]); DVP_B3G(2, 7, 8, 13, m[12], m[13]); DVP_B3G(3, 4, 9, 14, m[14], m[15]);
    if (r < 6) {
      uint32_t t[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] = m[PERM[i]];
#pragma unroll
      for (int i = 0; i < 16; ++i) m[i] = t[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) cv[i] = s[i] ^ s[i + 8];
}
#undef DVP_B3G
// BLAKE3 of `len` <= 1024 bytes held as zero-padded little-endian words
__device__ __forceinline__ void hash_chunk(const uint32_t* words, uint32_t len, uint32_t out[8]) {
  constexpr uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = IV[i];
  const uint32_t nblocks = len ? (len + 63) / 64 : 1;
#pragma unroll 1
  for (uint32_t b = 0; b < nblocks; ++b) {
    uint32_t blk[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) blk[i] = words[16 * b + i];
    const bool last = b + 1 == nblocks;
    compress(out, blk, last ? len - 64 * b : 64u, (b == 0 ? 1u : 0u) | (last ? (2u | 8u) : 0u));  // CHUNK_START | CHUNK_END | ROOT
  }
}
}  // namespace b3d
constexpr uint32_t TRANSCRIPT_DEV_MAX_PUB = 35;  // 35 x 29 bytes = 1015 <= one 1024-byte chunk

// alpha from (commit_p, public inputs), both resident: enc30 = the commitment's 30 bytes (16-byte aligned, the two bytes behind them
// belong to the next encoding), pub = n_pub canonical Fr.  Writes alpha (canonical) and alpha (Montgomery).
__global__ void __launch_bounds__(64) k_transcript(const uint32_t* __restrict__ enc30, const Fr* __restrict__ pub, uint32_t npub, b3d::Words8 h_ct,
                                                   Fr* __restrict__ alpha_canon, Fr* __restrict__ alpha_m) {
  __shared__ uint32_t msg[256];
  for (uint32_t i = threadIdx.x; i < 256; i += 64) msg[i] = 0;
  __syncthreads();
  const uint32_t len = 29u * npub;  // <= 1015
  uint8_t* mb = (uint8_t*)msg;
  for (uint32_t t = threadIdx.x; t < len; t += 64) {
    const uint32_t j = t / 29u, b = t - 29u * j;
    mb[t] = ((const uint8_t*)(pub + j))[b];  // 29-byte LE of a canonical Fr (src/proving.rs:147-150)
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  uint32_t blk[16], h_wc[8], h_pi[8], h_rt[8], out[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) blk[i] = i < 8 ? enc30[i] : 0u;
  blk[7] &= 0xffffu;  // 30 bytes
  b3d::hash_chunk(blk, 30, h_wc);
  b3d::hash_chunk(msg, len, h_pi);
#pragma unroll
  for (int i = 0; i < 8; ++i) { blk[i] = h_wc[i]; blk[8 + i] = h_pi[i]; }
  b3d::hash_chunk(blk, 64, h_rt);
#pragma unroll
  for (int i = 0; i < 8; ++i) { blk[i] = h_ct.w[i]; blk[8 + i] = h_rt[i]; }
  b3d::hash_chunk(blk, 64, out);
  Fr a;
#pragma unroll
  for (int i = 0; i < 7; ++i) a.v[i] = out[i];
  a.v[7] = 0;  // 224-bit challenge: the top four bytes cleared (src/proving.rs:192), always < p
  *alpha_canon = a;
  *alpha_m = fr_to_mont(a);
}
// -Z_D(alpha) from the isogeny chain (what host_vanish computes on the host): one lane, ~7 log2(m) dependent products -- run on the
// prover's side stream beside the denominators / batch inversion / barycentric partial sums, which do not need it
__global__ void __launch_bounds__(64) k_zalpha(const Fr* __restrict__ alpha_m, const Fr* __restrict__ x0s, const Fr* __restrict__ ts, int kk,
                                               const Fr* __restrict__ ctop, Fr* __restrict__ neg_z_alpha_m) {
  if (threadIdx.x != 0) return;
  *neg_z_alpha_m = fr_neg(vanish_chain(*alpha_m, x0s, ts, kk, ctop[0]));
}

__global__ void __launch_bounds__(256) k_to_mont_vec(const Fr* __restrict__ in, Fr* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fr_to_mont(in[i]);
}
__global__ void __launch_bounds__(256) k_from_mont_vec(const Fr* __restrict__ in, Fr* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fr_from_mont(in[i]);
}

}  // namespace dvp

using namespace dvp;

struct DevCsr {
  uint32_t* row_ptr = nullptr;
  uint32_t* wire = nullptr;
  uint32_t* coeff = nullptr;
  uint32_t n_rows = 0;
  size_t nnz = 0;
  uint32_t max_coeff_id = 0;  // largest coefficient id stored (re-checked against the table in prover_ready)
};

// the prover's result block on the device; the first FIN_BYTES go to the host (pinned) in ONE copy at the end of a proof
static constexpr size_t FIN_OFF_FLAGS = 4 * sizeof(Fr);        // a0 b0 i0 r0, then [0] unsatisfied row, [1] alpha in D u D'
static constexpr size_t FIN_OFF_ENC = FIN_OFF_FLAGS + 16;      // commit_p (30) | kzg_k (30) | 4 spare
static constexpr size_t FIN_OFF_ALPHA = FIN_OFF_ENC + 64;      // alpha, canonical (written by k_transcript; the host flavour keeps its own copy)
static constexpr size_t FIN_OFF_MSMERR = FIN_OFF_ALPHA + 32;   // scalar-range words of deferred MSMs: [0] the commitment MSM's
static constexpr size_t FIN_BYTES = FIN_OFF_MSMERR + 16;
static constexpr size_t DEV_OFF_ALPHA_M = FIN_BYTES;           // device only: alpha (Montgomery), -Z_D(alpha) (Montgomery)
static constexpr size_t DEV_OFF_NEGZ = DEV_OFF_ALPHA_M + 32;
static constexpr size_t DEV_BLOCK_BYTES = DEV_OFF_NEGZ + 32;
struct dvp_prover {
  uint32_t log_m = 0, m = 0, n_pub = 0, n_wires = 0;
  dvp_ecfft* tree = nullptr;  // 2m leaves (tree2n)
  Fr *dD = nullptr, *dD2 = nullptr, *barw = nullptr, *z2inv = nullptr;  // Montgomery, m each
  DevCsr mat[3];
  Fr* coeffs_m = nullptr;
  uint32_t n_coeffs = 0;
  Aff* bases_a = nullptr;      // [g_m | g_q]        n_wires + m
  uint8_t* inf_a = nullptr;
  Aff* bases_k = nullptr;      // [g_k0 | g_k1 | g_k2] 4m
  uint8_t* inf_k = nullptr;
  bool have_srs[5] = {false, false, false, false, false};
  // stream of the host-pointer seam (dvp_prove / dvp_prove_cache_dir, which carry no stream argument): one per prover, so that
  // two provers driven from two host threads overlap on the GPU instead of meeting on the default stream
  hipStream_t own_stream = nullptr;
  // fixed-base MSM contexts over [g_m | g_q] and [g_k_0 | g_k_1 | g_k_2]: bases pre-rotated by tau^(20 w), built
  // lazily once the SRS is complete (12 x the base storage: 4.8 GB at m = 2^20 -- HBM is not the scarce resource)
  MsmFixedCtx* fx[2] = {nullptr, nullptr};
  size_t fx_lo[2] = {0, 0}, fx_hi[2] = {0, 0};  // index range of the bases the context of MSM `which` was built for
  bool last_begin_extended = false;            // dvp_prove_begin_partial(need_extend): q2 / k_r are valid only if true
  uint32_t ext_filled = 0;                     // bit v: extended vector v of this proof is in E2 (extended here, or received and marked)
  // in-library multi-GPU (dvp_set_devices): one shard per listed device, each with its slice of both base vectors
  struct Shard {
    int device = 0;
    size_t lo[2] = {0, 0}, hi[2] = {0, 0};
    Aff* bases[2] = {nullptr, nullptr};
    uint8_t* inf[2] = {nullptr, nullptr};
    Fr* sc[2] = {nullptr, nullptr};
    MsmFixedCtx* fx[2] = {nullptr, nullptr};
    uint32_t* out = nullptr;  // 2 x (16 words x||y + 1 word flag + pad)
    hipStream_t st = nullptr;
  };
  std::vector<Shard> shards;
  std::vector<int> shard_devices;
  int home_device = 0;
  uint32_t* mg_parts = nullptr;  // home: 64 records of 80 B (packed for msm_sum_points_dev: 64 x 64 B points, then 64 flags)
  // work buffers
  Fr *w = nullptr, *E = nullptr, *E2 = nullptr, *r2 = nullptr, *SA = nullptr /* [w | q2] */, *den = nullptr, *den2 = nullptr,
     *SK = nullptr, *partial = nullptr, *abir0 = nullptr;
  unsigned long long* flags = nullptr;  // [0] unsat row, [1] alpha hit
  Aff* pts = nullptr;                   // 2 result points
  uint32_t* pts_inf32 = nullptr;
  uint8_t* pts_inf8 = nullptr;
  uint8_t* enc = nullptr;               // 60 bytes
  bool enc_fused[2] = {false, false};   // enc + 30 * which already holds the encoding of pts[which] (written by the MSM's own tail kernel)
                                        // AND fin_host mirrors [abir0 | flags | enc] as of that MSM's end (copied before its final sync)
  Fr ctop_host[2];                      // layer log_m leaves (collapse points of D, D'), Montgomery
  // last-proof intermediates kept for parity tests
  Fr alpha_canon, abir0_host[4];
  Fr neg_z_alpha_m;  // -Z_D(alpha), kept between dvp_prove_challenge_partial and dvp_prove_challenge_finish
  std::vector<uint64_t> pub_host;
  uint8_t commit_p_host[30];
  uint8_t* fin_host = nullptr;  // pinned mirror of the device block [abir0 | flags | enc | alpha | msm err] + 64 bytes of staging for (alpha_m, -Z(alpha))
  Fr* chal_dev = nullptr;       // device: [alpha_m, -Z_D(alpha)] (Montgomery), behind the block the host reads
  Fr* alpha_dev = nullptr;      // device: alpha, canonical (inside the block)
  unsigned long long* msm_err = nullptr;  // device: scalar-range words of deferred MSMs (inside the block)
  // side stream + events of the device transcript: -Z(alpha) is computed beside the challenge phase's vector kernels
  hipStream_t side = nullptr;
  hipEvent_t ev_alpha = nullptr, ev_negz = nullptr;
  b3d::Words8 h_ct;             // H(H(empty) || H(empty)): srs / circuit hashes are hashes of empty buffers (src/proving.rs:86-105,113-132)
};

static const int PT = 256;
static const uint32_t HORNER_MAX_PUB = 48;  // ~ the Fr products per element of one extend pass

static Fr host_vanish(const dvp_prover* p, int which, const Fr& x_m) {
  const dvp_ecfft* c = p->tree;
  int kk = (int)p->log_m;
  Fr u = x_m, v = fr_one_mont();
  for (int d = 0; d < kk; ++d) {
    Fr uv = fr_mul(u, v), vv = fr_sqr(v);
    Fr nu = fr_add(fr_sub(fr_sqr(u), fr_mul(c->x0[d], uv)), fr_mul(c->t[d], vv));
    Fr nv = fr_sub(uv, fr_mul(c->x0[d], vv));
    u = nu;
    v = nv;
  }
  return fr_sub(u, fr_mul(p->ctop_host[which], v));
}

static int prover_init(dvp_prover* p, uint32_t log2_m, uint32_t n_public, uint32_t n_wires);
static void shards_release(dvp_prover* p);
extern "C" void dvp_prover_destroy(dvp_prover* p);

extern "C" int dvp_prover_create(uint32_t log2_m, uint32_t n_public, uint32_t n_wires, dvp_prover** out) {
  // 2^24 constraints is the largest size the MSM index ranges cover (4m bases x 14 pre-rotated windows must stay below
  // 2^32, msm.hip) -- the reference's own tree constants stop at 2^27 leaves (src/ec_fft.rs:205), its SP1 circuit is 2^23
  if (!out || log2_m < 1 || log2_m > DVP_MAX_LOG2_CONSTRAINTS || n_wires < 1 + n_public || (uint64_t)n_wires > (1ull << 25)) return DVP_EINVAL;
  dvp_prover* p = new dvp_prover();
  int rc = prover_init(p, log2_m, n_public, n_wires);
  if (rc != DVP_OK) {
    dvp_prover_destroy(p);  // releases whatever was allocated before the failure
    return rc;
  }
  *out = p;
  return DVP_OK;
}

static int prover_init(dvp_prover* p, uint32_t log2_m, uint32_t n_public, uint32_t n_wires) {
  p->log_m = log2_m;
  p->m = 1u << log2_m;
  p->n_pub = n_public;
  p->n_wires = n_wires;
  const size_t m = p->m;
  DVP_HIP(hipGetDevice(&p->home_device));  // every buffer of this prover lives on the device that is current now
  DVP_TRY(dvp_ecfft_create(log2_m + 1, 0, log2_m + 1, &p->tree));  // TREE_2N, src/proving.rs:274
  DVP_TRY(ecfft_device_consts(p->tree));
  auto A = [&](void** q, size_t bytes) -> int { DVP_HIP(hipMalloc(q, bytes ? bytes : 16)); return DVP_OK; };
  DVP_TRY(A((void**)&p->dD, m * sizeof(Fr)));
  DVP_TRY(A((void**)&p->dD2, m * sizeof(Fr)));
  DVP_TRY(A((void**)&p->barw, m * sizeof(Fr)));
  DVP_TRY(A((void**)&p->z2inv, m * sizeof(Fr)));
  DVP_TRY(A((void**)&p->bases_a, (n_wires + m) * sizeof(Aff)));
  DVP_TRY(A((void**)&p->inf_a, n_wires + m));
  DVP_TRY(A((void**)&p->bases_k, 4 * m * sizeof(Aff)));
  DVP_TRY(A((void**)&p->inf_k, 4 * m));
  DVP_TRY(A((void**)&p->w, n_wires * sizeof(Fr)));
  DVP_TRY(A((void**)&p->E, 4 * m * sizeof(Fr)));
  DVP_TRY(A((void**)&p->E2, 4 * m * sizeof(Fr)));
  DVP_TRY(A((void**)&p->r2, m * sizeof(Fr)));
  DVP_TRY(A((void**)&p->SA, (n_wires + m) * sizeof(Fr)));
  DVP_TRY(A((void**)&p->den, 2 * m * sizeof(Fr)));  // [den | den2]: one batch inversion covers both
  p->den2 = p->den + m;
  DVP_TRY(A((void**)&p->SK, 4 * m * sizeof(Fr)));
  DVP_TRY(A((void**)&p->partial, 3 * 1024 * sizeof(Fr)));
  // what the host reads back at the end of a proof, in ONE block and one copy: a0 b0 i0 r0 | flags | encoded points
  DVP_TRY(A((void**)&p->abir0, DEV_BLOCK_BYTES));
  DVP_HIP(hipMemset(p->abir0, 0, DEV_BLOCK_BYTES));
  p->flags = (unsigned long long*)((char*)p->abir0 + FIN_OFF_FLAGS);
  p->enc = (uint8_t*)p->abir0 + FIN_OFF_ENC;
  p->alpha_dev = (Fr*)((char*)p->abir0 + FIN_OFF_ALPHA);
  p->msm_err = (unsigned long long*)((char*)p->abir0 + FIN_OFF_MSMERR);
  p->chal_dev = (Fr*)((char*)p->abir0 + DEV_OFF_ALPHA_M);
  DVP_HIP(hipHostMalloc((void**)&p->fin_host, FIN_BYTES + 64, hipHostMallocDefault));
  DVP_HIP(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
  DVP_HIP(hipEventCreateWithFlags(&p->ev_alpha, hipEventDisableTiming));
  DVP_HIP(hipEventCreateWithFlags(&p->ev_negz, hipEventDisableTiming));
  {
    uint8_t h_empty[32], buf[64], h[32];
    b3::hash(nullptr, 0, h_empty);
    memcpy(buf, h_empty, 32); memcpy(buf + 32, h_empty, 32);
    b3::hash(buf, 64, h);
    memcpy(p->h_ct.w, h, 32);
  }
  DVP_TRY(A((void**)&p->pts, 2 * sizeof(Aff)));
  DVP_TRY(A((void**)&p->pts_inf32, 8));
  DVP_TRY(A((void**)&p->pts_inf8, 8));
  DVP_HIP(hipMemset(p->inf_a, 0, n_wires + m));
  DVP_HIP(hipMemset(p->inf_k, 0, 4 * m));
  dvp_ecfft* t = p->tree;
  int kk = (int)log2_m;
  hipLaunchKernelGGL(k_split_domains, dim3(cdiv(m, PT)), dim3(PT), 0, 0, t->layer(0), (uint32_t)m, p->dD, p->dD2);
  // bar weights of D and 1/Z_D on D'
  hipLaunchKernelGGL(k_domain_tables, dim3(cdiv(m, PT)), dim3(PT), 0, 0, t->layer(0), (uint32_t)m, t->d_x0, t->d_t, kk,
                     t->layer(kk), 0, p->barw, p->z2inv);
  DVP_HIP(hipGetLastError());
  DVP_HIP(hipMemcpy(p->ctop_host, t->layer(kk), 2 * sizeof(Fr), hipMemcpyDeviceToHost));
  DVP_HIP(hipDeviceSynchronize());
  return DVP_OK;
}

extern "C" void dvp_prover_destroy(dvp_prover* p) {
  if (!p) return;
  void* ptrs[] = {p->dD, p->dD2, p->barw, p->z2inv, p->coeffs_m, p->bases_a, p->inf_a, p->bases_k, p->inf_k, p->w, p->E,
                  p->E2, p->r2, p->SA, p->den, p->SK, p->partial, p->abir0 /* + flags + enc */, p->pts, p->pts_inf32,
                  p->pts_inf8};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  if (p->fin_host) (void)hipHostFree(p->fin_host);
  if (p->own_stream) (void)hipStreamDestroy(p->own_stream);
  if (p->side) (void)hipStreamDestroy(p->side);
  if (p->ev_alpha) (void)hipEventDestroy(p->ev_alpha);
  if (p->ev_negz) (void)hipEventDestroy(p->ev_negz);
  for (auto& mt : p->mat) {
    if (mt.row_ptr) (void)hipFree(mt.row_ptr);
    if (mt.wire) (void)hipFree(mt.wire);
    if (mt.coeff) (void)hipFree(mt.coeff);
  }
  msm_fixed_destroy(p->fx[0]);
  msm_fixed_destroy(p->fx[1]);
  shards_release(p);
  if (p->mg_parts) (void)hipFree(p->mg_parts);
  if (p->tree) dvp_ecfft_destroy(p->tree);
  delete p;
}

extern "C" int dvp_prover_set_coeffs(dvp_prover* p, const uint64_t* coeffs, uint32_t n) {
  if (!p || !coeffs || !n) return DVP_EINVAL;
  if (p->coeffs_m) (void)hipFree(p->coeffs_m);
  DVP_HIP(hipMalloc((void**)&p->coeffs_m, (size_t)n * sizeof(Fr)));
  DevBuf tmp;
  DVP_TRY(tmp.alloc((size_t)n * sizeof(Fr)));
  DVP_HIP(hipMemcpy(tmp.p, coeffs, (size_t)n * sizeof(Fr), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_to_mont_vec, dim3(cdiv(n, PT)), dim3(PT), 0, 0, tmp.as<Fr>(), p->coeffs_m, (size_t)n);
  DVP_HIP(hipDeviceSynchronize());
  p->n_coeffs = n;
  return DVP_OK;
}

// which: 0 = L (A), 1 = R (B), 2 = O (C) of src/gnark_r1cs.rs:112-119; CSR over n_rows <= m rows
extern "C" int dvp_prover_set_matrix(dvp_prover* p, int which, uint32_t n_rows, const uint32_t* row_ptr,
                                     const uint32_t* wire_ids, const uint32_t* coeff_ids) {
  if (!p || which < 0 || which > 2 || !row_ptr || n_rows > p->m) return DVP_EINVAL;
  // a well-formed CSR: row_ptr[0] == 0 and non-decreasing (k_r1cs_eval walks [row_ptr[i], row_ptr[i+1]) unchecked)
  if (row_ptr[0] != 0) { g_last_error_index = 0; return DVP_EINVAL; }
  for (uint32_t i = 0; i < n_rows; ++i)
    if (row_ptr[i + 1] < row_ptr[i]) { g_last_error_index = (int64_t)i; return DVP_EINVAL; }
  size_t nnz = row_ptr[n_rows];
  if (nnz && (!wire_ids || !coeff_ids)) return DVP_EINVAL;
  uint32_t max_cid = 0;
  for (size_t k = 0; k < nnz; ++k) {
    if (wire_ids[k] >= p->n_wires || (p->n_coeffs && coeff_ids[k] >= p->n_coeffs)) {
      g_last_error_index = (int64_t)k;
      return DVP_EINVAL;
    }
    if (coeff_ids[k] > max_cid) max_cid = coeff_ids[k];
  }
  DevCsr& mt = p->mat[which];
  if (mt.row_ptr) { (void)hipFree(mt.row_ptr); (void)hipFree(mt.wire); (void)hipFree(mt.coeff); }
  DVP_HIP(hipMalloc((void**)&mt.row_ptr, ((size_t)n_rows + 1) * 4));
  DVP_HIP(hipMalloc((void**)&mt.wire, (nnz ? nnz : 1) * 4));
  DVP_HIP(hipMalloc((void**)&mt.coeff, (nnz ? nnz : 1) * 4));
  DVP_HIP(hipMemcpy(mt.row_ptr, row_ptr, ((size_t)n_rows + 1) * 4, hipMemcpyHostToDevice));
  if (nnz) {
    DVP_HIP(hipMemcpy(mt.wire, wire_ids, nnz * 4, hipMemcpyHostToDevice));
    DVP_HIP(hipMemcpy(mt.coeff, coeff_ids, nnz * 4, hipMemcpyHostToDevice));
  }
  mt.n_rows = n_rows;
  mt.nnz = nnz;
  mt.max_coeff_id = max_cid;
  return DVP_OK;
}

// which: 0 g_m (n_wires), 1 g_q (m), 2 g_k_0 (m), 3 g_k_1 (m), 4 g_k_2 (2m) -- src/artifacts.rs names;
// enc = payload of the reference's point-vector file (n x 30 B, src/io_utils.rs:83-111), decoded ONCE
// into HBM-resident affine bases (the reference re-reads and re-decodes them on every prove()).
static void shards_release(dvp_prover* p);
// a base vector of MSM `slot` changed: its fixed-base tables (and every device shard) are stale
static void srs_changed(dvp_prover* p, int slot) {
  msm_fixed_destroy(p->fx[slot]);
  p->fx[slot] = nullptr;
  shards_release(p);
}
static int srs_slot(dvp_prover* p, int which, Aff** base, uint8_t** inf, size_t* n) {
  size_t m = p->m;
  switch (which) {
    case 0: *base = p->bases_a; *inf = p->inf_a; *n = p->n_wires; return DVP_OK;
    case 1: *base = p->bases_a + p->n_wires; *inf = p->inf_a + p->n_wires; *n = m; return DVP_OK;
    case 2: *base = p->bases_k; *inf = p->inf_k; *n = m; return DVP_OK;
    case 3: *base = p->bases_k + m; *inf = p->inf_k + m; *n = m; return DVP_OK;
    case 4: *base = p->bases_k + 2 * m; *inf = p->inf_k + 2 * m; *n = 2 * m; return DVP_OK;
  }
  return DVP_EINVAL;
}
extern "C" int dvp_prover_set_srs_encoded(dvp_prover* p, int which, const uint8_t* enc, size_t n) {
  if (!p || !enc) return DVP_EINVAL;
  Aff* base; uint8_t* inf; size_t want;
  DVP_TRY(srs_slot(p, which, &base, &inf, &want));
  if (n != want) return DVP_EINVAL;
  DevBuf de;
  DVP_TRY(de.alloc(n * 30));
  DVP_HIP(hipMemcpy(de.p, enc, n * 30, hipMemcpyHostToDevice));
  DVP_TRY(decode_dev(de.as<uint8_t>(), n, base, inf, 0));
  p->have_srs[which] = true;
  srs_changed(p, which < 2 ? 0 : 1);
  // the copies / fills above ran on the default stream; proofs run on other (non-blocking) streams, which do not order themselves
  // behind it: nothing of this setter may still be in flight when it returns
  DVP_HIP(hipStreamSynchronize(nullptr));
  return DVP_OK;
}
extern "C" int dvp_prover_set_srs_affine(dvp_prover* p, int which, const uint64_t* xy, const uint8_t* inf_in, size_t n) {
  if (!p || !xy) return DVP_EINVAL;
  Aff* base; uint8_t* inf; size_t want;
  DVP_TRY(srs_slot(p, which, &base, &inf, &want));
  if (n != want) return DVP_EINVAL;
  DVP_HIP(hipMemcpy(base, xy, n * sizeof(Aff), hipMemcpyHostToDevice));
  if (inf_in) DVP_HIP(hipMemcpy(inf, inf_in, n, hipMemcpyHostToDevice));
  else DVP_HIP(hipMemset(inf, 0, n));
  p->have_srs[which] = true;
  srs_changed(p, which < 2 ? 0 : 1);
  // the copies / fills above ran on the default stream; proofs run on other (non-blocking) streams, which do not order themselves
  // behind it: nothing of this setter may still be in flight when it returns
  DVP_HIP(hipStreamSynchronize(nullptr));
  return DVP_OK;
}
// device-to-device flavour (bases produced on the GPU, e.g. by dvp_mulgen on the same device)
extern "C" int dvp_prover_set_srs_affine_dev(dvp_prover* p, int which, const void* d_xy, const void* d_inf, size_t n) {
  if (!p || !d_xy) return DVP_EINVAL;
  Aff* base; uint8_t* inf; size_t want;
  DVP_TRY(srs_slot(p, which, &base, &inf, &want));
  if (n != want) return DVP_EINVAL;
  DVP_HIP(hipMemcpy(base, d_xy, n * sizeof(Aff), hipMemcpyDeviceToDevice));
  if (d_inf) DVP_HIP(hipMemcpy(inf, d_inf, n, hipMemcpyDeviceToDevice));
  else DVP_HIP(hipMemset(inf, 0, n));
  p->have_srs[which] = true;
  srs_changed(p, which < 2 ? 0 : 1);
  // the copies / fills above ran on the default stream; proofs run on other (non-blocking) streams, which do not order themselves
  // behind it: nothing of this setter may still be in flight when it returns
  DVP_HIP(hipStreamSynchronize(nullptr));
  return DVP_OK;
}

// Transcript::output, src/proving.rs:164-197 (srs/circuit hashes are hashes of EMPTY buffers, :86-105,113-132)
static void transcript_challenge(const uint8_t commit_p[30], const uint64_t* pub, uint32_t npub, uint8_t out32[32]) {
  uint8_t h_empty[32], h_wc[32], h_pi[32], h_ct[32], h_rt[32], buf[64];
  b3::hash(nullptr, 0, h_empty);
  b3::hash(commit_p, 30, h_wc);
  std::vector<uint8_t> pi((size_t)npub * 29);
  for (uint32_t j = 0; j < npub; ++j) memcpy(pi.data() + 29 * (size_t)j, (const uint8_t*)(pub + 4 * (size_t)j), 29);
  b3::hash(pi.data(), pi.size(), h_pi);
  memcpy(buf, h_empty, 32); memcpy(buf + 32, h_empty, 32);
  b3::hash(buf, 64, h_ct);
  memcpy(buf, h_wc, 32); memcpy(buf + 32, h_pi, 32);
  b3::hash(buf, 64, h_rt);
  memcpy(buf, h_ct, 32); memcpy(buf + 32, h_rt, 32);
  b3::hash(buf, 64, out32);
  out32[28] = out32[29] = out32[30] = out32[31] = 0;  // 224-bit challenge
}

extern "C" int dvp_transcript_challenge(const uint8_t commit_p[30], const uint64_t* public_inputs, uint32_t n_public, uint64_t out[4]) {
  if (!commit_p || (n_public && !public_inputs) || !out) return DVP_EINVAL;
  uint8_t h[32];
  transcript_challenge(commit_p, public_inputs, n_public, h);
  memcpy(out, h, 32);
  return DVP_OK;
}

// parity-test access to the DEVICE flavour of the transcript (k_transcript, k_zalpha) on caller-supplied inputs: alpha (canonical) and
// -Z_D(alpha) (canonical) for this prover's domain, to be compared with dvp_transcript_challenge / the oracle
extern "C" int dvp_prover_debug_transcript_dev(dvp_prover* p, const uint8_t commit_p[30], const uint64_t* public_inputs, uint32_t n_public,
                                               uint64_t out_alpha[4], uint64_t out_neg_z_alpha[4]) {
  if (!p || !commit_p || (n_public && !public_inputs) || !out_alpha || !out_neg_z_alpha || n_public > TRANSCRIPT_DEV_MAX_PUB) return DVP_EINVAL;
  DevBuf in, out;
  DVP_TRY(in.alloc(64 + (size_t)(n_public ? n_public : 1) * sizeof(Fr)));
  DVP_TRY(out.alloc(4 * sizeof(Fr)));
  uint8_t head[64];
  memset(head, 0xa5, sizeof(head));  // the bytes behind the 30 must not matter
  memcpy(head, commit_p, 30);
  DVP_HIP(hipMemcpy(in.p, head, 64, hipMemcpyHostToDevice));
  if (n_public) DVP_HIP(hipMemcpy((char*)in.p + 64, public_inputs, (size_t)n_public * sizeof(Fr), hipMemcpyHostToDevice));
  Fr* o = out.as<Fr>();
  hipLaunchKernelGGL(k_transcript, dim3(1), dim3(64), 0, 0, (const uint32_t*)in.p, (const Fr*)((char*)in.p + 64), n_public, p->h_ct, o, o + 1);
  hipLaunchKernelGGL(k_zalpha, dim3(1), dim3(64), 0, 0, o + 1, p->tree->d_x0, p->tree->d_t, (int)p->log_m, p->tree->layer((int)p->log_m), o + 2);
  hipLaunchKernelGGL(k_from_mont_vec, dim3(1), dim3(PT), 0, 0, o + 2, o + 3, (size_t)1);
  DVP_HIP(hipGetLastError());
  DVP_HIP(hipDeviceSynchronize());
  DVP_HIP(hipMemcpy(out_alpha, o, sizeof(Fr), hipMemcpyDeviceToHost));
  DVP_HIP(hipMemcpy(out_neg_z_alpha, o + 3, sizeof(Fr), hipMemcpyDeviceToHost));
  return DVP_OK;
}

extern "C" int dvp_blake3(const uint8_t* data, size_t len, uint8_t out[32]) {
  if ((len && !data) || !out) return DVP_EINVAL;
  b3::hash(data, len, out);
  return DVP_OK;
}

// ---- Proof::prove in phases ------------------------------------------------------------------------
// The two MSMs are the only stages that shard across GPUs, so prove is exposed as
//   begin -> [commitment MSM, possibly partial per rank] -> challenge -> [K MSM] -> finish
// and dvp_prove / dvp_prove_dev run the phases back to back on one device.
static bool prover_ready(dvp_prover* p) {
  for (int k = 0; k < 5; ++k)
    if (!p->have_srs[k]) return false;
  if (!(p->coeffs_m && p->mat[0].row_ptr && p->mat[1].row_ptr && p->mat[2].row_ptr)) return false;
  for (int k = 0; k < 3; ++k)  // whatever order set_coeffs / set_matrix were called in, every stored id must be in the table
    if (p->mat[k].nnz && p->mat[k].max_coeff_id >= p->n_coeffs) return false;
  return true;
}

// phase 1 (src/proving.rs:434-508): assignment -> a,b,c',i -> extend -> q2; leaves SA = [w | q2].
// d_assignment: n_wires canonical Fr = [1, public.., private..] already in HBM.
static uint32_t prover_n_ext(const dvp_prover* p);
extern "C" int dvp_prove_extend_vectors(dvp_prover* p, uint32_t mask, void* stream);
extern "C" int dvp_prove_quotient(dvp_prover* p, void* stream);
extern "C" int dvp_prove_begin(dvp_prover* p, const void* d_assignment, void* stream) {
  return dvp_prove_begin_partial(p, d_assignment, 1, stream);
}

// need_extend == 0: the caller's MSM shards lie inside [w] and [k_a | k_b] only, which need neither q2 nor r2, so the
// three extends and the quotient are skipped (multi-GPU load balancing, distributed.py::shard_plan).  q2 / r2 and the
// k_r part of the second MSM's scalars are then NOT valid on this prover until the next full begin.
static int prove_begin_impl(dvp_prover* p, const void* d_assignment, int need_extend, void* stream, bool defer_unsat, bool pub_to_host = true);
extern "C" int dvp_prove_begin_partial(dvp_prover* p, const void* d_assignment, int need_extend, void* stream) {
  return prove_begin_impl(p, d_assignment, need_extend, stream, false);
}
// the host's look at the unsatisfied-row flag (assert_eq!(a*b, c+i), src/proving.rs:389-395)
static int prove_check_unsat(dvp_prover* p, hipStream_t st) {
  unsigned long long f0[2];
  DVP_HIP(hipMemcpyAsync(f0, p->flags, 16, hipMemcpyDeviceToHost, st));
  DVP_HIP(hipStreamSynchronize(st));
  count_host_wait(0);
  if (f0[0] != ~0ull) {
    g_last_error_index = (int64_t)f0[0];
    return DVP_EUNSAT;
  }
  return DVP_OK;
}
// defer_unsat (dvp_prove_dev only): the flag is NOT waited for here -- the commitment MSM does not need the host to know it, and its
// own final synchronisation brings the flag block along (one host round trip less per proof; an unsatisfied witness then costs an
// MSM before it is reported)
static int prove_begin_impl(dvp_prover* p, const void* d_assignment, int need_extend, void* stream, bool defer_unsat, bool pub_to_host) {
  if (!p || !d_assignment || !prover_ready(p)) return DVP_EINVAL;
  p->last_begin_extended = false;  // set by dvp_prove_quotient once every extended vector is in place
  p->ext_filled = 0;
  p->enc_fused[0] = p->enc_fused[1] = false;  // a staged caller may have abandoned a proof after an MSM into the prover's own slot
  hipStream_t st = (hipStream_t)stream;
  const uint32_t m = p->m;
  const size_t nw = p->n_wires;
  // the witness goes straight into the scalar vector of the first MSM, [w | q2], and every stage reads it there (p->w is only the
  // staging buffer of the host-pointer seam): one 32 B / wire copy per proof instead of two
  if (d_assignment != p->SA) DVP_HIP(hipMemcpyAsync(p->SA, d_assignment, nw * sizeof(Fr), hipMemcpyDeviceToDevice, st));
  p->pub_host.resize((size_t)p->n_pub * 4);
  if (p->n_pub && pub_to_host) DVP_HIP(hipMemcpyAsync(p->pub_host.data(), p->SA + 1, (size_t)p->n_pub * 32, hipMemcpyDeviceToHost, st));
  DVP_HIP(hipMemsetAsync(p->flags, 0xff, 16, st));
  Csr A{p->mat[0].row_ptr, p->mat[0].wire, p->mat[0].coeff, p->mat[0].n_rows};
  Csr B{p->mat[1].row_ptr, p->mat[1].wire, p->mat[1].coeff, p->mat[1].n_rows};
  Csr C{p->mat[2].row_ptr, p->mat[2].wire, p->mat[2].coeff, p->mat[2].n_rows};
  dim3 gm(cdiv(m, PT)), bt(PT);
  hipLaunchKernelGGL(k_r1cs_eval, gm, bt, 0, st, A, B, C, p->coeffs_m, p->SA, p->dD, p->n_pub, m, p->E, p->flags);
  DVP_HIP(hipGetLastError());
  if (need_extend) {
    DVP_TRY(dvp_prove_extend_vectors(p, (1u << prover_n_ext(p)) - 1, stream));
    DVP_TRY(dvp_prove_quotient(p, stream));
  }
  if (defer_unsat) return DVP_OK;
  return prove_check_unsat(p, st);
}

// ---- the extends by VECTOR (option A) ------------------------------------------------------------------------
// The extends of a, b, c' (and i, when it is not evaluated by Horner) are independent (src/proving.rs:410-422), so the ranks
// of a multi-process prove that need q2 / r2 each extend only the vectors of `mask` (bit v = vector v of [a, b, c', i]),
// exchange the extended vectors (distributed.py: one broadcast per vector inside the extender group, straight out of and
// into dvp_prover_extended_ptr) and then run the quotient.  dvp_prove_begin_partial(need_extend = 1) is
// begin(need_extend = 0) + extend_vectors(all) + quotient.
static uint32_t prover_n_ext(const dvp_prover* p) {
  uint32_t hmax = HORNER_MAX_PUB;
  if (tune().horner_max_pub >= 0) hmax = (uint32_t)tune().horner_max_pub;  // tests force either route
  return p->n_pub <= hmax ? 3u : 4u;
}
extern "C" uint32_t dvp_prover_extend_count(const dvp_prover* p) { return p ? prover_n_ext(p) : 0; }
extern "C" int dvp_prover_extended_ptr(dvp_prover* p, uint32_t v, void** d_ptr) {
  if (!p || v > 3 || !d_ptr) return DVP_EINVAL;
  *d_ptr = p->E2 + (size_t)v * p->m;
  return DVP_OK;
}
extern "C" int dvp_prove_extend_vectors(dvp_prover* p, uint32_t mask, void* stream) {
  if (!p || !prover_ready(p)) return DVP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const uint32_t m = p->m, n_ext = prover_n_ext(p);
  if (mask >> n_ext) return DVP_EINVAL;
  // runs of consecutive selected vectors share one batched extend (the matrices are read once per run)
  for (uint32_t v = 0; v < n_ext;) {
    if (!((mask >> v) & 1)) { ++v; continue; }
    uint32_t e = v;
    while (e < n_ext && ((mask >> e) & 1)) ++e;
    ProfScope pe(PROF_EXTEND_TOTAL, st);
    DVP_TRY(extend_from(p->tree, 0, 0, p->E + (size_t)v * m, p->E2 + (size_t)v * m, e - v, st));  // E (a, b, c' on D) stays as it is
    pe.stop();
    v = e;
  }
  p->ext_filled |= mask;
  return DVP_OK;
}
// extended vector v of the CURRENT proof has been written into dvp_prover_extended_ptr(v) by the caller (a broadcast from the
// rank that extended it): dvp_prove_quotient refuses to run until every vector is either extended here or marked
extern "C" int dvp_prove_mark_extended(dvp_prover* p, uint32_t v) {
  if (!p || v >= prover_n_ext(p)) return DVP_EINVAL;
  p->ext_filled |= 1u << v;
  return DVP_OK;
}
// r2 = a2 b2 - i2, q2 = (r2 - c2) / Z_D on D' (src/proving.rs:492-508) from the extended vectors in place
extern "C" int dvp_prove_quotient(dvp_prover* p, void* stream) {
  if (!p || !prover_ready(p)) return DVP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const uint32_t m = p->m;
  const size_t nw = p->n_wires;
  if (p->ext_filled != (1u << prover_n_ext(p)) - 1) return DVP_EINVAL;  // a missed broadcast would otherwise give a silently wrong q2
  dim3 gm(cdiv(m, PT)), bt(PT);
  if (prover_n_ext(p) == 3)
    hipLaunchKernelGGL(k_quotient<true>, gm, bt, 0, st, p->E2, p->z2inv, m, p->SA, p->dD2, p->n_pub, p->r2, p->SA + nw);
  else
    hipLaunchKernelGGL(k_quotient<false>, gm, bt, 0, st, p->E2, p->z2inv, m, p->SA, p->dD2, p->n_pub, p->r2, p->SA + nw);
  DVP_HIP(hipGetLastError());
  p->last_begin_extended = true;
  return DVP_OK;
}

// which = 0: <[w | q2], [g_m | g_q]> (n_wires + m terms; commit_p = msm_q + msm_gm, src/proving.rs:463,512,515)
// which = 1: <[k_a | k_b | k_r], [g_k_0 | g_k_1 | g_k_2]> (4m terms, src/proving.rs:666-680)
// restricted to the index range [lo, hi): the per-GPU shard of the sum.
// d_out_enc (optional): the MSM's tail kernel also writes the 30-byte encoding of its result there
// d_err_defer (optional): deferred completion -- the call returns with the MSM enqueued (msm.hip: msm_core), its scalar-range word
// goes to that device word and nothing is copied to the host
static int prover_msm_partial(dvp_prover* p, int which, size_t lo, size_t hi, void* d_out_xy, void* d_out_inf, void* d_out_enc, void* stream,
                              unsigned long long* d_err_defer = nullptr) {
  if (!p || (which != 0 && which != 1) || !d_out_xy || !d_out_inf) return DVP_EINVAL;
  size_t total = which ? 4 * (size_t)p->m : (size_t)p->n_wires + p->m;
  if (lo > hi || hi > total) return DVP_EINVAL;
  const Fr* sc = which ? p->SK : p->SA;
  const Aff* bs = which ? p->bases_k : p->bases_a;
  const uint8_t* inf = which ? p->inf_k : p->inf_a;
  // after an extend-free begin (dvp_prove_begin_partial(need_extend = 0)) q2 and the r2 inside k_r were never computed:
  // a range that reads them would silently sum garbage
  if (!p->last_begin_extended && hi > lo && hi > (which ? 2 * (size_t)p->m : (size_t)p->n_wires)) return DVP_EINVAL;
  // fixed-base mode pays off once the shared 2^c-bucket set is well filled.  The pre-rotated table covers only the index
  // range this prover is asked for (a rank of a multi-process prove always asks for the same slice: W x slice x 64 B of
  // HBM instead of W x everything); a call outside the covered range rebuilds it for the new range.
  const size_t fixed_min = (size_t)(tune().msm_fixed_min > 0 ? tune().msm_fixed_min : 1);
  if (hi - lo >= fixed_min && total < ((size_t)1 << 27)) {
    if (p->fx[which] && (lo < p->fx_lo[which] || hi > p->fx_hi[which])) {
      msm_fixed_destroy(p->fx[which]);
      p->fx[which] = nullptr;
    }
    if (!p->fx[which]) {
      DVP_TRY(msm_fixed_create(bs + lo, (uint32_t)(hi - lo), hi - lo, &p->fx[which]));
      p->fx_lo[which] = lo;
      p->fx_hi[which] = hi;
    }
    const size_t o = p->fx_lo[which];
    return msm_fixed_dev_enc(p->fx[which], sc + lo, inf + lo, (uint32_t)(lo - o), (uint32_t)(hi - o), d_out_xy, d_out_inf, d_out_enc,
                             d_out_enc && !d_err_defer ? p->fin_host : nullptr, p->abir0, FIN_BYTES, (hipStream_t)stream, d_err_defer);
  }
  return msm_affine_dev_enc(sc + lo, bs + lo, inf + lo, hi - lo, d_out_xy, d_out_inf, d_out_enc, d_out_enc && !d_err_defer ? p->fin_host : nullptr, p->abir0,
                            FIN_BYTES, (hipStream_t)stream, d_err_defer);
}
extern "C" int dvp_prover_msm_partial(dvp_prover* p, int which, size_t lo, size_t hi, void* d_out_xy, void* d_out_inf, void* stream) {
  if (p && (which == 0 || which == 1) && d_out_xy == (void*)(p->pts + which)) p->enc_fused[which] = false;  // the point changes, its encoding does not follow
  return prover_msm_partial(p, which, lo, hi, d_out_xy, d_out_inf, nullptr, stream);
}

// ---- in-library multi-GPU (dvp_set_devices) ------------------------------------------------------------------------
// Only the MSMs shard: they are sums over disjoint index ranges and > 85 % of a proof.  Shard k lives on
// device shard_devices[k]: its slice [lo, hi) of both base vectors (copied once, device to device), the fixed-base
// tables of that slice, a scalar buffer and a stream.  A proof sends each shard its slice of the scalars
// (hipMemcpyPeerAsync), runs the partial MSMs concurrently -- one host thread per device -- and adds the partial points
// on the home device (k_sum_points).  RCCL is not involved: the exchange is N x 8..32 MB of scalars one way and N x 80
// bytes back, both point-to-point copies over xGMI.
static void shards_release(dvp_prover* p) {
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (auto& sh : p->shards) {
    (void)hipSetDevice(sh.device);
    for (int w = 0; w < 2; ++w) {
      if (sh.bases[w]) (void)hipFree(sh.bases[w]);
      if (sh.inf[w]) (void)hipFree(sh.inf[w]);
      if (sh.sc[w]) (void)hipFree(sh.sc[w]);
      msm_fixed_destroy(sh.fx[w]);
    }
    if (sh.out) (void)hipFree(sh.out);
    if (sh.st) (void)hipStreamDestroy(sh.st);
  }
  p->shards.clear();
  p->shard_devices.clear();
  (void)hipSetDevice(cur);
}
static void even_range(size_t total, size_t k, size_t n, size_t* lo, size_t* hi) {
  size_t base = total / n, rem = total % n;
  *lo = k * base + (k < rem ? k : rem);
  *hi = *lo + base + (k < rem ? 1 : 0);
}
static int shards_build(dvp_prover* p, const std::vector<int>& devs) {
  shards_release(p);
  const size_t n = devs.size();
  if (n > 64 || devs[0] != p->home_device) return DVP_EINVAL;  // ids[0] must be the device the prover's buffers live on
  // the single-device tables (up to 97 GB of rotations at 2^20 constraints) would sit beside the shard tables: shard 0's
  // msm_fixed_create would see less free HBM and fall back to the aligned windows
  for (int w = 0; w < 2; ++w) {
    msm_fixed_destroy(p->fx[w]);
    p->fx[w] = nullptr;
  }
  if (!p->mg_parts) DVP_HIP(hipMalloc((void**)&p->mg_parts, 64 * 68));
  p->shards.resize(n);
  int rc = DVP_OK;
  for (size_t k = 0; k < n && rc == DVP_OK; ++k) {
    dvp_prover::Shard& sh = p->shards[k];
    sh.device = devs[k];
    auto body = [&]() -> int {
      DVP_HIP(hipSetDevice(sh.device));
      if (sh.device != p->home_device) {
        // already-enabled / unsupported are fine (peer copies are staged then); clear the sticky error so that the next
        // hipGetLastError() check after a kernel launch does not report it
        if (hipDeviceEnablePeerAccess(p->home_device, 0) != hipSuccess) (void)hipGetLastError();
      }
      DVP_HIP(hipStreamCreateWithFlags(&sh.st, hipStreamNonBlocking));
      DVP_HIP(hipMalloc((void**)&sh.out, 2 * 80));
      for (int w = 0; w < 2; ++w) {
        const size_t total = w ? 4 * (size_t)p->m : (size_t)p->n_wires + p->m;
        even_range(total, k, n, &sh.lo[w], &sh.hi[w]);
        const size_t cnt = sh.hi[w] - sh.lo[w];
        const Aff* src_b = (w ? p->bases_k : p->bases_a) + sh.lo[w];
        const uint8_t* src_i = (w ? p->inf_k : p->inf_a) + sh.lo[w];
        DVP_HIP(hipMalloc((void**)&sh.bases[w], (cnt ? cnt : 1) * sizeof(Aff)));
        DVP_HIP(hipMalloc((void**)&sh.inf[w], cnt ? cnt : 1));
        DVP_HIP(hipMalloc((void**)&sh.sc[w], (cnt ? cnt : 1) * sizeof(Fr)));
        if (cnt) {
          DVP_HIP(hipMemcpyPeer(sh.bases[w], sh.device, src_b, p->home_device, cnt * sizeof(Aff)));
          DVP_HIP(hipMemcpyPeer(sh.inf[w], sh.device, src_i, p->home_device, cnt));
        }
        const size_t fixed_min = (size_t)(tune().msm_fixed_min > 0 ? tune().msm_fixed_min : 1);
        if (cnt >= fixed_min) DVP_TRY(msm_fixed_create(sh.bases[w], (uint32_t)cnt, cnt, &sh.fx[w]));
      }
      return DVP_OK;
    };
    rc = body();
  }
  (void)hipSetDevice(p->home_device);
  if (rc != DVP_OK) {
    shards_release(p);
    return rc;
  }
  p->shard_devices = devs;
  return DVP_OK;
}

// MSM `which` over all shards -> home buffers d_out_xy / d_out_inf.  The home stream must have produced the scalars.
static int mgpu_msm(dvp_prover* p, int which, void* d_out_xy, void* d_out_inf, hipStream_t home_st) {
  const Fr* sc = which ? p->SK : p->SA;
  DVP_HIP(hipStreamSynchronize(home_st));
  const size_t n = p->shards.size();
  std::vector<int> rcs(n, DVP_OK);
  std::vector<int64_t> err_idx(n, -1);
  std::vector<std::thread> th;
  struct Rec { uint32_t xy[16]; uint32_t inf; };
  std::vector<Rec> recs(n);
  for (size_t k = 0; k < n; ++k) {
    th.emplace_back([&, k]() {
      dvp_prover::Shard& sh = p->shards[k];
      auto body = [&]() -> int {
        DVP_HIP(hipSetDevice(sh.device));
        const size_t cnt = sh.hi[which] - sh.lo[which];
        uint32_t* out = sh.out + 20 * which;
        if (cnt) DVP_HIP(hipMemcpyPeerAsync(sh.sc[which], sh.device, sc + sh.lo[which], p->home_device, cnt * sizeof(Fr), sh.st));
        if (sh.fx[which])
          DVP_TRY(msm_fixed_dev(sh.fx[which], sh.sc[which], sh.inf[which], 0, (uint32_t)cnt, out, out + 16, sh.st));
        else
          DVP_TRY(msm_affine_dev(sh.sc[which], sh.bases[which], sh.inf[which], cnt, out, out + 16, sh.st));
        DVP_HIP(hipMemcpyAsync(&recs[k], out, 68, hipMemcpyDeviceToHost, sh.st));
        DVP_HIP(hipStreamSynchronize(sh.st));
        return DVP_OK;
      };
      rcs[k] = body();
      err_idx[k] = g_last_error_index;  // thread-local in the worker: carry it back
    });
  }
  for (auto& t : th) t.join();
  DVP_HIP(hipSetDevice(p->home_device));
  for (size_t k = 0; k < n; ++k)
    if (rcs[k] != DVP_OK) {
      if (err_idx[k] >= 0) g_last_error_index = err_idx[k] + (int64_t)p->shards[k].lo[which];  // index in the whole vector
      return rcs[k];
    }
  // partial points -> home: packed as n points then n flags
  std::vector<uint32_t> pk(n * 17);
  for (size_t k = 0; k < n; ++k) {
    memcpy(&pk[16 * k], recs[k].xy, 64);
    pk[16 * n + k] = recs[k].inf;
  }
  DVP_HIP(hipMemcpyAsync(p->mg_parts, pk.data(), n * 68, hipMemcpyHostToDevice, home_st));
  DVP_TRY(msm_sum_points_dev(p->mg_parts, 64, p->mg_parts + 16 * n, (uint32_t)n, 1, d_out_xy, d_out_inf, home_st));
  DVP_HIP(hipStreamSynchronize(home_st));  // pk goes out of scope
  return DVP_OK;
}
// the MSM of a full proof: over the shards when a device list is set, else on the home device alone
static int prove_msm(dvp_prover* p, int which, void* d_out_xy, void* d_out_inf, void* stream) {
  const std::vector<int> devs = mgpu_devices();
  if (devs.size() > 1) {
    if (devs != p->shard_devices) DVP_TRY(shards_build(p, devs));
    p->enc_fused[which] = false;
    return mgpu_msm(p, which, d_out_xy, d_out_inf, (hipStream_t)stream);
  }
  if (!p->shards.empty()) shards_release(p);
  // the whole sum on this device, into the prover's own slot: let the tail kernel encode it too (dvp_prove_challenge / _finish then
  // skip their k_encode_point: one launch, one host round trip and one inversion chain less per commitment)
  const bool own = d_out_xy == (void*)(p->pts + which) && d_out_inf == (void*)(p->pts_inf32 + which);
  p->enc_fused[which] = false;
  DVP_TRY(prover_msm_partial(p, which, 0, dvp_prover_msm_size(p, which), d_out_xy, d_out_inf, own ? p->enc + 30 * which : nullptr, stream));
  p->enc_fused[which] = own;
  return DVP_OK;
}
// window size / window count the fixed-base context of MSM `which` settled on (0,0 before its first use)
extern "C" int dvp_prover_msm_plan(const dvp_prover* p, int which, int* c_bits, int* windows) {
  if (!p || (which != 0 && which != 1) || !c_bits || !windows) return DVP_EINVAL;
  *c_bits = 0;
  *windows = 0;
  if (p->fx[which]) return msm_fixed_info(p->fx[which], c_bits, windows);
  if (!p->shards.empty() && p->shards[0].fx[which]) return msm_fixed_info(p->shards[0].fx[which], c_bits, windows);  // multi-GPU: shard 0's
  return DVP_OK;
}
// HBM held by the fixed-base table(s) of MSM `which` (all shards); *signed_windows = 1 for the default signed binary windows
extern "C" uint64_t dvp_prover_msm_table_bytes(const dvp_prover* p, int which, int* signed_windows) {
  if (signed_windows) *signed_windows = 0;
  if (!p || (which != 0 && which != 1)) return 0;
  uint64_t total = msm_fixed_table_bytes(p->fx[which], signed_windows);
  for (const auto& sh : p->shards) {
    int s = 0;
    total += msm_fixed_table_bytes(sh.fx[which], &s);
    if (s && signed_windows) *signed_windows = 1;
  }
  return total;
}
// the table the first pair round of MSM `which` gathers from on the home device (bench.py points dvp_ubench_gather at it)
extern "C" int dvp_prover_msm_table_ptr(const dvp_prover* p, int which, const void** d_table, uint64_t* bytes) {
  if (!p || (which != 0 && which != 1) || !d_table || !bytes) return DVP_EINVAL;
  *d_table = msm_fixed_table_ptr(p->fx[which]);
  *bytes = msm_fixed_table_bytes(p->fx[which], nullptr);
  return DVP_OK;
}
extern "C" size_t dvp_prover_msm_size(const dvp_prover* p, int which) {
  if (!p) return 0;
  return which ? 4 * (size_t)p->m : (size_t)p->n_wires + p->m;
}

// the vector stages of the challenge phase (src/proving.rs:561-654) once alpha (Montgomery) is in p->chal_dev[0]: denominators,
// one batch inversion over D and D', the three barycentric sums, a0 b0 i0 r0, the K scalars.  -Z_D(alpha) (p->chal_dev[1]) is needed
// by k_bary3_final only: the host flavour copies it in with alpha; the device flavour computes it on the prover's side stream
// meanwhile (z_on_side_stream: the stream joins the side stream in front of k_bary3_final).
static int challenge_vector_stages(dvp_prover* p, hipStream_t st, bool z_on_side_stream) {
  const uint32_t m = p->m;
  dim3 gm(cdiv(m, PT)), bt(PT);
  hipLaunchKernelGGL(k_alpha_denoms, gm, bt, 0, st, p->dD, p->dD2, p->chal_dev, m, p->den, p->den2, p->flags + 1);
  DVP_TRY(batch_inverse_dev(p->den, 2 * (size_t)m, st));  // den2 = den + m
  uint32_t nb = cdiv(m, PT);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(k_bary3_partial, dim3(nb), bt, 0, st, p->E, p->barw, p->den, m, p->partial);
  if (z_on_side_stream) DVP_HIP(hipStreamWaitEvent(st, p->ev_negz, 0));
  hipLaunchKernelGGL(k_bary3_final, dim3(1), bt, 0, st, p->partial, nb, p->chal_dev + 1, p->abir0);
  hipLaunchKernelGGL(k_kscalars, gm, bt, 0, st, p->E, p->r2, p->den, p->den2, p->abir0, m, p->SK);
  DVP_HIP(hipGetLastError());
  return DVP_OK;
}
// the same phase with the transcript on the device: commit_p's 30 bytes are where the commitment MSM's tail kernel wrote them
// (p->enc), the public inputs in the witness vector; no host round trip
static int challenge_dev(dvp_prover* p, hipStream_t st) {
  hipLaunchKernelGGL(k_transcript, dim3(1), dim3(64), 0, st, (const uint32_t*)p->enc, p->SA + 1, p->n_pub, p->h_ct, p->alpha_dev, p->chal_dev);
  DVP_HIP(hipEventRecord(p->ev_alpha, st));
  DVP_HIP(hipStreamWaitEvent(p->side, p->ev_alpha, 0));
  hipLaunchKernelGGL(k_zalpha, dim3(1), dim3(64), 0, p->side, p->chal_dev, p->tree->d_x0, p->tree->d_t, (int)p->log_m, p->tree->layer((int)p->log_m),
                     p->chal_dev + 1);
  DVP_HIP(hipEventRecord(p->ev_negz, p->side));
  return challenge_vector_stages(p, st, /*z_on_side_stream=*/true);
}

// phase 2 (src/proving.rs:515-654): commit_p -> alpha -> a0,b0,i0,r0 -> K scalars SK.
extern "C" int dvp_prove_challenge(dvp_prover* p, const void* d_commit_xy, const void* d_commit_inf, void* stream) {
  if (!p || !d_commit_xy || !d_commit_inf) return DVP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (d_commit_xy != p->pts) DVP_HIP(hipMemcpyAsync(p->pts, d_commit_xy, sizeof(Aff), hipMemcpyDeviceToDevice, st));
  if (d_commit_inf != p->pts_inf32) DVP_HIP(hipMemcpyAsync(p->pts_inf32, d_commit_inf, 4, hipMemcpyDeviceToDevice, st));
  if (d_commit_xy == p->pts && d_commit_inf == p->pts_inf32 && p->enc_fused[0]) {
    memcpy(p->commit_p_host, p->fin_host + FIN_OFF_ENC, 30);  // encoded by the MSM's tail, on the host since the MSM's own sync
  } else {
    DVP_TRY(encode_point_dev(p->pts, p->pts_inf32, p->enc, st));
    DVP_HIP(hipMemcpyAsync(p->commit_p_host, p->enc, 30, hipMemcpyDeviceToHost, st));
    DVP_HIP(hipStreamSynchronize(st));
    count_host_wait(0);
  }
  p->enc_fused[0] = false;
  uint8_t ch[32];
  transcript_challenge(p->commit_p_host, p->pub_host.data(), p->n_pub, ch);
  Fr alpha;
  memcpy(alpha.v, ch, 32);
  p->alpha_canon = alpha;
  Fr* stage = (Fr*)(p->fin_host + FIN_BYTES);  // pinned; rewritten only after this proof's final synchronisation
  stage[0] = fr_to_mont(alpha);
  stage[1] = fr_neg(host_vanish(p, 0, stage[0]));
  DVP_HIP(hipMemcpyAsync(p->chal_dev, stage, 2 * sizeof(Fr), hipMemcpyHostToDevice, st));
  return challenge_vector_stages(p, st, /*z_on_side_stream=*/false);
}

// ---- phase 2 for index-sharded provers (one process per GPU; SURVEY 8e "pointwise / batch-inverse / barycentric": slice by
// index, all-gather of the partial Fr sums + local add, batch inversion per shard) ------------------------------------------
// part 1: alpha from the commitment; 1/(d - alpha) only where this rank needs it -- its slice [d_lo, d_hi) of the barycentric
// sums and the part of D / D' its K-scalar range [k_lo, k_hi) reads -- and the rank's 128-byte record: three partial sums
// (Montgomery) + its alpha-in-domain flag.  part 2 (after the ranks exchanged their records, any order): a0 b0 i0 r0 from
// the n records, then the K scalars of [k_lo, k_hi) only.  dvp_prove_challenge == part 1 over everything + part 2 on the own
// record; the rest of S and of den / den2 is NOT valid on this prover afterwards.
extern "C" int dvp_prove_challenge_partial(dvp_prover* p, const void* d_commit_xy, const void* d_commit_inf, size_t d_lo, size_t d_hi,
                                           size_t k_lo, size_t k_hi, void* d_record_out, void* stream) {
  if (!p || !d_commit_xy || !d_commit_inf || !d_record_out) return DVP_EINVAL;
  const uint32_t m = p->m;
  if (d_lo > d_hi || d_hi > m || k_lo > k_hi || k_hi > 4 * (size_t)m) return DVP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  dim3 bt(PT);
  if (d_commit_xy != p->pts) DVP_HIP(hipMemcpyAsync(p->pts, d_commit_xy, sizeof(Aff), hipMemcpyDeviceToDevice, st));
  if (d_commit_inf != p->pts_inf32) DVP_HIP(hipMemcpyAsync(p->pts_inf32, d_commit_inf, 4, hipMemcpyDeviceToDevice, st));
  if (d_commit_xy == p->pts && d_commit_inf == p->pts_inf32 && p->enc_fused[0]) {
    memcpy(p->commit_p_host, p->fin_host + FIN_OFF_ENC, 30);  // encoded by the MSM's tail, on the host since the MSM's own sync
  } else {
    DVP_TRY(encode_point_dev(p->pts, p->pts_inf32, p->enc, st));
    DVP_HIP(hipMemcpyAsync(p->commit_p_host, p->enc, 30, hipMemcpyDeviceToHost, st));
    DVP_HIP(hipStreamSynchronize(st));
    count_host_wait(0);
  }
  p->enc_fused[0] = false;
  uint8_t ch[32];
  transcript_challenge(p->commit_p_host, p->pub_host.data(), p->n_pub, ch);
  Fr alpha;
  memcpy(alpha.v, ch, 32);
  p->alpha_canon = alpha;
  const Fr alpha_m = fr_to_mont(alpha);
  p->neg_z_alpha_m = fr_neg(host_vanish(p, 0, alpha_m));
  // index intervals of D whose inverse denominators this rank reads: its barycentric slice, k_a, k_b and the even half of
  // k_r ([D_i, D'_i] interleaved); of D': the odd half of k_r.  Overlaps are merged (inverting twice would undo it).
  std::vector<std::pair<size_t, size_t>> iv;
  auto clip = [&](size_t a, size_t b, size_t base) {  // [k_lo, k_hi) n [a, b), rebased
    size_t lo = std::max(k_lo, a), hi = std::min(k_hi, b);
    return lo < hi ? std::make_pair(lo - base, hi - base) : std::make_pair((size_t)0, (size_t)0);
  };
  iv.push_back({d_lo, d_hi});
  iv.push_back(clip(0, m, 0));
  iv.push_back(clip(m, 2 * (size_t)m, m));
  std::pair<size_t, size_t> kr = clip(2 * (size_t)m, 4 * (size_t)m, 2 * (size_t)m);
  std::pair<size_t, size_t> kri = {kr.first >> 1, (kr.second + 1) >> 1};
  if (kr.first == kr.second) kri = {0, 0};
  iv.push_back(kri);
  std::sort(iv.begin(), iv.end());
  std::vector<std::pair<size_t, size_t>> merged;
  for (auto& x : iv) {
    if (x.first == x.second) continue;
    if (!merged.empty() && x.first <= merged.back().second) merged.back().second = std::max(merged.back().second, x.second);
    else merged.push_back(x);
  }
  for (auto& x : merged) {
    const uint32_t cnt = (uint32_t)(x.second - x.first);
    hipLaunchKernelGGL(k_alpha_denoms_range, dim3(cdiv(cnt, PT)), bt, 0, st, p->dD, alpha_m, (uint32_t)x.first, (uint32_t)x.second, p->den, p->flags + 1);
    DVP_TRY(batch_inverse_dev(p->den + x.first, cnt, st));
  }
  if (kri.first < kri.second) {
    const uint32_t cnt = (uint32_t)(kri.second - kri.first);
    hipLaunchKernelGGL(k_alpha_denoms_range, dim3(cdiv(cnt, PT)), bt, 0, st, p->dD2, alpha_m, (uint32_t)kri.first, (uint32_t)kri.second, p->den2, p->flags + 1);
    DVP_TRY(batch_inverse_dev(p->den2 + kri.first, cnt, st));
  }
  uint32_t nb = 0;
  if (d_lo < d_hi) {
    nb = cdiv((uint32_t)(d_hi - d_lo), PT);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_bary3_partial_range, dim3(nb), bt, 0, st, p->E, p->barw, p->den, m, (uint32_t)d_lo, (uint32_t)d_hi, p->partial);
  }
  hipLaunchKernelGGL(k_bary3_record, dim3(1), bt, 0, st, p->partial, nb, p->flags + 1, (Fr*)d_record_out);
  DVP_HIP(hipGetLastError());
  return DVP_OK;
}
extern "C" int dvp_prove_challenge_finish(dvp_prover* p, const void* d_records, uint32_t n_records, size_t k_lo, size_t k_hi, void* stream) {
  if (!p || !d_records || !n_records) return DVP_EINVAL;
  const uint32_t m = p->m;
  if (k_lo > k_hi || k_hi > 4 * (size_t)m) return DVP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_bary3_from_records, dim3(1), dim3(64), 0, st, (const Fr*)d_records, n_records, p->neg_z_alpha_m, p->abir0, p->flags + 1);
  if (k_lo < k_hi)
    hipLaunchKernelGGL(k_kscalars_range, dim3(cdiv(k_hi - k_lo, PT)), dim3(PT), 0, st, p->E, p->r2, p->den, p->den2, p->abir0, m, k_lo, k_hi, p->SK);
  DVP_HIP(hipGetLastError());
  return DVP_OK;
}

// phase 3 (src/proving.rs:682-687): kzg_k -> bytes; proof = commit_p | kzg_k | a0 | b0.
extern "C" int dvp_prove_finish(dvp_prover* p, const void* d_kzg_xy, const void* d_kzg_inf, uint8_t proof[118], void* stream) {
  if (!p || !d_kzg_xy || !d_kzg_inf || !proof) return DVP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (d_kzg_xy != p->pts + 1) DVP_HIP(hipMemcpyAsync(p->pts + 1, d_kzg_xy, sizeof(Aff), hipMemcpyDeviceToDevice, st));
  if (d_kzg_inf != p->pts_inf32 + 1) DVP_HIP(hipMemcpyAsync(p->pts_inf32 + 1, d_kzg_inf, 4, hipMemcpyDeviceToDevice, st));
  if (!(d_kzg_xy == p->pts + 1 && d_kzg_inf == p->pts_inf32 + 1 && p->enc_fused[1])) {
    DVP_TRY(encode_point_dev(p->pts + 1, p->pts_inf32 + 1, p->enc + 30, st));
    DVP_HIP(hipMemcpyAsync(p->fin_host, p->abir0, FIN_BYTES, hipMemcpyDeviceToHost, st));  // pinned: one DMA, no staging
    DVP_HIP(hipStreamSynchronize(st));
    count_host_wait(0);
  }  // else: the K MSM's tail encoded kzg_k and fin_host was filled before that MSM's final sync (a0 b0 i0 r0 and the flags were final by then)
  p->enc_fused[1] = false;
  unsigned long long f[2];
  memcpy(p->abir0_host, p->fin_host, 4 * sizeof(Fr));
  memcpy(f, p->fin_host + FIN_OFF_FLAGS, 16);
  const uint8_t* kz = p->fin_host + FIN_OFF_ENC + 30;
  if (f[1] != ~0ull) {
    g_last_error_index = (int64_t)f[1];
    return DVP_ECHALLENGE;  // alpha in D u D', src/proving.rs:548-556
  }
  memcpy(proof, p->commit_p_host, 30);
  memcpy(proof + 30, kz, 30);
  memcpy(proof + 60, p->abir0_host[0].v, 29);  // FrBits::from_fr(a0): 232 LE bits, src/curve.rs:30-40
  memcpy(proof + 89, p->abir0_host[1].v, 29);
  return DVP_OK;
}

// Proof::prove with the assignment already resident in HBM (the timed configuration of bench.py)
// Host-transcript flavour (rounds 1-4; still what a device list, n_public > 35 or DVP_PROVE_HOST_TRANSCRIPT=1 gets): the host waits for
// the commitment MSM, hashes, and enqueues the challenge phase.
static int prove_dev_host_transcript(dvp_prover* p, const void* d_assignment, uint8_t proof[118], void* stream) {
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(PROF_PROVE_TOTAL, st);
  DVP_TRY(prove_begin_impl(p, d_assignment, 1, stream, true));
  {
    const int rc = prove_msm(p, 0, p->pts, p->pts_inf32, stream);
    if (rc == DVP_OK && p->enc_fused[0]) {  // fin_host = [a0 b0 i0 r0 | flags | encodings] as of the MSM's end
      unsigned long long f0;
      memcpy(&f0, p->fin_host + FIN_OFF_FLAGS, 8);
      if (f0 != ~0ull) {
        g_last_error_index = (int64_t)f0;
        return DVP_EUNSAT;
      }
    } else {  // sharded MSM (no block on the host), or an MSM error: the unsatisfied row is reported first, as before
      DVP_TRY(prove_check_unsat(p, st));
      DVP_TRY(rc);
    }
  }
  DVP_TRY(dvp_prove_challenge(p, p->pts, p->pts_inf32, stream));
  DVP_TRY(prove_msm(p, 1, p->pts + 1, p->pts_inf32 + 1, stream));
  ps.stop();
  return dvp_prove_finish(p, p->pts + 1, p->pts_inf32 + 1, proof, stream);
}
// Device-transcript flavour (round 5, the default on one device): begin -> commitment MSM (deferred: no final synchronisation, its
// tail kernel leaves commit_p's 30 bytes and its scalar-range word in the prover's block) -> k_transcript -> challenge phase -> K MSM,
// whose own final synchronisation -- the ONE wait of the proof on its stream -- brings the block
// [a0 b0 i0 r0 | flags | commit_p kzg_k | alpha | commitment MSM's word] to the host.  (Inside each MSM the host still reads the largest
// bucket on a side stream while the first pair round runs; that read never idles the GPU.)  Errors are reported in the reference's
// order from that block: unsatisfied row (src/proving.rs:389-395), scalar >= p in the witness (multi_scalar_mul's fr_to_le_bytes),
// alpha in D u D' (:548-556), then the K MSM's own status.
static int prove_dev_device_transcript(dvp_prover* p, const void* d_assignment, uint8_t proof[118], void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!p->shards.empty()) shards_release(p);
  ProfScope ps(PROF_PROVE_TOTAL, st);
  DVP_TRY(prove_begin_impl(p, d_assignment, 1, stream, true, /*pub_to_host=*/false));
  DVP_TRY(prover_msm_partial(p, 0, 0, dvp_prover_msm_size(p, 0), p->pts, p->pts_inf32, p->enc, stream, p->msm_err));
  // from here on the commitment MSM is in flight on `st` and in its workspace: every error return below drains the stream first
  {
    const int rc_c = challenge_dev(p, st);
    if (rc_c != DVP_OK) {
      (void)hipStreamSynchronize(st);
      return rc_c;
    }
  }
  // the block in pinned host memory still holds the PREVIOUS proof's words: mark the flags so that a K MSM which returns before its
  // copy (plan limits, an allocation failure) is not mistaken for one that brought this proof's block along
  constexpr unsigned long long FIN_STALE = ~0ull - 1;  // the device writes ~0 (fine) or an index < 2^32
  {
    const unsigned long long stale[2] = {FIN_STALE, FIN_STALE};
    memcpy(p->fin_host + FIN_OFF_FLAGS, stale, 16);
  }
  const int rc_k = prover_msm_partial(p, 1, 0, dvp_prover_msm_size(p, 1), p->pts + 1, p->pts_inf32 + 1, p->enc + 30, stream);
  ps.stop();
  const int64_t k_idx = g_last_error_index;
  unsigned long long f[2], e0;
  memcpy(f, p->fin_host + FIN_OFF_FLAGS, 16);
  if (f[0] == FIN_STALE || f[1] == FIN_STALE) {  // nothing was brought to the host
    (void)hipStreamSynchronize(st);
    g_last_error_index = k_idx;
    return rc_k != DVP_OK ? rc_k : DVP_EHIP;
  }
  memcpy(&e0, p->fin_host + FIN_OFF_MSMERR, 8);
  memcpy(p->abir0_host, p->fin_host, 4 * sizeof(Fr));
  memcpy(p->commit_p_host, p->fin_host + FIN_OFF_ENC, 30);
  memcpy(p->alpha_canon.v, p->fin_host + FIN_OFF_ALPHA, 32);
  if (f[0] != ~0ull) { g_last_error_index = (int64_t)f[0]; return DVP_EUNSAT; }
  if (e0 != ~0ull) { g_last_error_index = (int64_t)(e0 & 0xffffffffull); return DVP_EINVAL; }
  if (f[1] != ~0ull) { g_last_error_index = (int64_t)f[1]; return DVP_ECHALLENGE; }
  if (rc_k != DVP_OK) { g_last_error_index = k_idx; return rc_k; }
  memcpy(proof, p->commit_p_host, 30);
  memcpy(proof + 30, p->fin_host + FIN_OFF_ENC + 30, 30);
  memcpy(proof + 60, p->abir0_host[0].v, 29);  // FrBits::from_fr(a0): 232 LE bits, src/curve.rs:30-40
  memcpy(proof + 89, p->abir0_host[1].v, 29);
  return DVP_OK;
}
extern "C" int dvp_prove_dev(dvp_prover* p, const void* d_assignment, uint8_t proof[118], void* stream) {
  if (!p || !d_assignment || !proof) return DVP_EINVAL;
  if (mgpu_devices().size() <= 1 && p->n_pub <= TRANSCRIPT_DEV_MAX_PUB && !tune().prove_host_transcript)
    return prove_dev_device_transcript(p, d_assignment, proof, stream);
  return prove_dev_host_transcript(p, d_assignment, proof, stream);
}

// Proof::prove(cache_dir, public_inputs, private_inputs) -> Proof, src/proving.rs:426-688 (host witness).
// proof = commit_p[30] | kzg_k[30] | a0[29 LE] | b0[29 LE]  (the byte image of Proof::to_bits, :691-718)
extern "C" int dvp_prove(dvp_prover* p, const uint64_t* public_inputs, uint32_t n_public, const uint64_t* private_inputs,
                         uint32_t n_private, uint8_t proof[118]) {
  if (!p || !proof || n_public != p->n_pub || 1 + n_public + n_private != p->n_wires) return DVP_EINVAL;
  if ((n_public && !public_inputs) || (n_private && !private_inputs)) return DVP_EINVAL;
  if (!prover_ready(p)) return DVP_EINVAL;
  // assignment = [1, public, private]  (src/proving.rs:449-452)
  if (!p->own_stream) DVP_HIP(hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking));
  Fr one = fr_one_canon();
  DVP_HIP(hipMemcpyAsync(p->w, &one, sizeof(Fr), hipMemcpyHostToDevice, p->own_stream));
  if (n_public) DVP_HIP(hipMemcpyAsync(p->w + 1, public_inputs, (size_t)n_public * 32, hipMemcpyHostToDevice, p->own_stream));
  if (n_private) DVP_HIP(hipMemcpyAsync(p->w + 1 + n_public, private_inputs, (size_t)n_private * 32, hipMemcpyHostToDevice, p->own_stream));
  DVP_HIP(hipStreamSynchronize(p->own_stream));  // `one` lives on this stack frame, and the caller's buffers are free again on return anyway
  return dvp_prove_dev(p, p->w, proof, p->own_stream);
}

// Parity-test access to the intermediates of the last dvp_prove call.  name in:
//  "a","b","c","i" (m each), "a2","b2","c2","i2", "r2", "q2", "ka","kb" (m), "kr" (2m),
//  "alpha","a0","b0","i0","r0" (1), "bar_wts","z_vals2inv","D","D2" (m, canonical)
extern "C" int dvp_prover_debug_read(dvp_prover* p, const char* name, uint64_t* out, size_t n_elems) {
  if (!p || !name || !out) return DVP_EINVAL;
  const size_t m = p->m;
  std::string s(name);
  const Fr* src = nullptr;
  size_t n = m;
  bool mont = false;
  if (s == "a") src = p->E; else if (s == "b") src = p->E + m; else if (s == "c") src = p->E + 2 * m; else if (s == "i") src = p->E + 3 * m;
  else if (s == "a2") src = p->E2; else if (s == "b2") src = p->E2 + m; else if (s == "c2") src = p->E2 + 2 * m; else if (s == "i2") src = p->E2 + 3 * m;
  else if (s == "r2") src = p->r2; else if (s == "q2") src = p->SA + p->n_wires;
  else if (s == "ka") src = p->SK; else if (s == "kb") src = p->SK + m; else if (s == "kr") { src = p->SK + 2 * m; n = 2 * m; }
  else if (s == "bar_wts") { src = p->barw; mont = true; } else if (s == "z_vals2inv") { src = p->z2inv; mont = true; }
  else if (s == "D") { src = p->dD; mont = true; } else if (s == "D2") { src = p->dD2; mont = true; }
  else if (s == "alpha") { if (n_elems != 1) return DVP_EINVAL; memcpy(out, p->alpha_canon.v, 32); return DVP_OK; }
  else if (s == "a0" || s == "b0" || s == "i0" || s == "r0") {
    if (n_elems != 1) return DVP_EINVAL;
    int k = s == "a0" ? 0 : s == "b0" ? 1 : s == "i0" ? 2 : 3;
    DVP_HIP(hipDeviceSynchronize());  // straight from the device: valid after the challenge phase, before finish
    DVP_HIP(hipMemcpy(out, p->abir0 + k, 32, hipMemcpyDeviceToHost));
    return DVP_OK;
  } else return DVP_EINVAL;
  if (n_elems != n) return DVP_EINVAL;
  if (mont) {
    DevBuf tmp;
    DVP_TRY(tmp.alloc(n * sizeof(Fr)));
    hipLaunchKernelGGL(k_from_mont_vec, dim3(cdiv(n, PT)), dim3(PT), 0, 0, src, tmp.as<Fr>(), n);
    DVP_HIP(hipMemcpy(out, tmp.p, n * sizeof(Fr), hipMemcpyDeviceToHost));
  } else {
    DVP_HIP(hipMemcpy(out, src, n * sizeof(Fr), hipMemcpyDeviceToHost));
  }
  return DVP_OK;
}

// Domain tables for setup-side callers (compute_barycentric_weights / evaluate_vanishing_poly_at_domain +
// batch_inversion, src/ec_fft.rs:284-335,407-419): which = 0 -> (1/Z_D'(D), 1/Z_D(D')), which = 1 -> mirrored.
// device flavour (setup.hip): canonical outputs, m = leaves / 2 entries each
int ecfft_domain_tables_dev(dvp_ecfft* t, int which, Fr* d_bar_weights, Fr* d_zinv_other, hipStream_t st) {
  if (!t || !d_bar_weights || !d_zinv_other || (which != 0 && which != 1) || t->log_n < 2) return DVP_EINVAL;
  const size_t m = t->n_leaves / 2;
  const int kk = t->log_n - 1;
  DVP_TRY(ecfft_device_consts(t));
  hipLaunchKernelGGL(k_domain_tables, dim3(cdiv(m, PT)), dim3(PT), 0, st, t->layer(0), (uint32_t)m, t->d_x0, t->d_t, kk, t->layer(kk),
                     which, d_bar_weights, d_zinv_other);
  hipLaunchKernelGGL(k_from_mont_vec, dim3(cdiv(m, PT)), dim3(PT), 0, st, d_bar_weights, d_bar_weights, m);
  hipLaunchKernelGGL(k_from_mont_vec, dim3(cdiv(m, PT)), dim3(PT), 0, st, d_zinv_other, d_zinv_other, m);
  DVP_HIP(hipGetLastError());
  return DVP_OK;
}
extern "C" int dvp_ecfft_domain_tables(dvp_ecfft* t, int which, uint64_t* bar_weights, uint64_t* zinv_other) {
  if (!t || !bar_weights || !zinv_other || (which != 0 && which != 1) || t->log_n < 2) return DVP_EINVAL;
  const size_t m = t->n_leaves / 2;
  DevBuf bw, zi;
  DVP_TRY(bw.alloc(m * sizeof(Fr)));
  DVP_TRY(zi.alloc(m * sizeof(Fr)));
  DVP_TRY(ecfft_domain_tables_dev(t, which, bw.as<Fr>(), zi.as<Fr>(), 0));
  DVP_HIP(hipMemcpy(bar_weights, bw.p, m * sizeof(Fr), hipMemcpyDeviceToHost));
  DVP_HIP(hipMemcpy(zinv_other, zi.p, m * sizeof(Fr), hipMemcpyDeviceToHost));
  return DVP_OK;
}
extern "C" int dvp_prover_domain_tables(dvp_prover* p, int which, uint64_t* bar_weights, uint64_t* zinv_other) {
  if (!p) return DVP_EINVAL;
  return dvp_ecfft_domain_tables(p->tree, which, bar_weights, zinv_other);
}

extern "C" int dvp_prover_domains(dvp_prover* p, uint64_t* d, uint64_t* d2) {
  if (!p || !d || !d2) return DVP_EINVAL;
  DVP_TRY(dvp_prover_debug_read(p, "D", d, p->m));
  return dvp_prover_debug_read(p, "D2", d2, p->m);
}
