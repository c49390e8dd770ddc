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
__global__ void __launch_bounds__(256) k_build_mats(const Fr* __restrict__ Ld, int sl, uint32_t nd, Fr x0, int src, int dst,
                             Fr* __restrict__ dec, Fr* __restrict__ rec) {
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
    dec[4 * (size_t)i + 0] = fr_mul(s1, iv1);
    dec[4 * (size_t)i + 1] = fr_neg(fr_mul(s0, iv0));
    dec[4 * (size_t)i + 2] = fr_neg(iv1);
    dec[4 * (size_t)i + 3] = iv0;
  }
  {
    Fr s0 = Ld[(size_t)(2 * i + dst) << sl];
    Fr s1 = Ld[(size_t)(2 * i + dst + nd) << sl];
    Fr v0 = fr_pow_u64(fr_sub(s0, x0), e);
    Fr v1 = fr_pow_u64(fr_sub(s1, x0), e);
    rec[4 * (size_t)i + 0] = v0;
    rec[4 * (size_t)i + 1] = fr_mul(s0, v0);
    rec[4 * (size_t)i + 2] = v1;
    rec[4 * (size_t)i + 3] = fr_mul(s1, v1);
  }
}

// One butterfly pass over `batch` vectors of n values (in place).  Blocks of size 2h; pair (i, i+h)
// inside each block uses matrix i of this layer.  One thread = one pair for all batch vectors, so
// the 128-byte matrix is read once per pair and reused `batch` times.
template <int BATCH>
__global__ void __launch_bounds__(256) k_butterfly(Fr* __restrict__ data, const Fr* __restrict__ mats, int lh, uint32_t n) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (n >> 1)) return;
  uint32_t h = 1u << lh;
  uint32_t i = tid & (h - 1);
  uint32_t i0 = ((tid >> lh) << (lh + 1)) | i;
  uint32_t i1 = i0 + h;
  const Fr* m = mats + 4 * (size_t)i;
  Fr m00 = m[0], m01 = m[1], m10 = m[2], m11 = m[3];
#pragma unroll
  for (int b = 0; b < BATCH; ++b) {
    Fr* v = data + (size_t)b * n;
    Fr e0 = v[i0], e1 = v[i1];
    v[i0] = fr_add(fr_mul(m00, e0), fr_mul(m01, e1));
    v[i1] = fr_add(fr_mul(m10, e0), fr_mul(m11, e1));
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

// ---- context -----------------------------------------------------------------------------------
struct MatSet {
  Fr* dec = nullptr;  // (n-1) x 4 Fr, layer d at offset 4*(n - (n>>d))
  Fr* rec = nullptr;
};

struct dvp_ecfft {
  int log_n = 0;
  uint32_t n_leaves = 0;
  int device = 0;
  Fr* layers = nullptr;  // layer d at offset layer_off[d], N>>d entries, Montgomery
  std::vector<size_t> layer_off;
  std::vector<Fr> x0, t;        // per layer, Montgomery (host copies)
  std::map<int, MatSet> mats;   // key = sl*2 + to_even
  std::map<int, Fr*> xnn;       // key = sl ; (N>>sl) entries: leaf^((N>>sl)/2)
  Fr* scratch = nullptr;        // 2 x N Fr work space for enter/exit
  std::mutex mu;

  Fr* layer(int d) const { return layers + layer_off[d]; }
};

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
    size_t bytes = (size_t)(n - 1) * 4 * sizeof(Fr);
    DVP_HIP(hipMalloc((void**)&ms.dec, bytes));
    DVP_HIP(hipMalloc((void**)&ms.rec, bytes));
    int src = to_even ? 1 : 0, dst = to_even ? 0 : 1;
    int ln = 31 - __builtin_clz(n);
    for (int d = 0; d < ln; ++d) {
      uint32_t nd = n >> d;
      size_t off = 4 * (size_t)(n - nd);
      hipLaunchKernelGGL(k_build_mats, dim3(cdiv(nd >> 1, TPB)), dim3(TPB), 0, st, c->layer(d), sl, nd, c->x0[d],
                         src, dst, ms.dec + off, ms.rec + off);
    }
    DVP_HIP(hipGetLastError());
  }
  c->mats[key] = ms;
  *out = &c->mats[key];
  return DVP_OK;
}

template <int B>
static void launch_bfly(Fr* data, const Fr* mats, int lh, uint32_t n, hipStream_t st) {
  hipLaunchKernelGGL((k_butterfly<B>), dim3(cdiv(n >> 1, TPB)), dim3(TPB), 0, st, data, mats, lh, n);
}

// in-place extend of `batch` vectors of n = (N>>sl)/2 values
static int extend_inplace(dvp_ecfft* c, int sl, int to_even, Fr* data, uint32_t batch, hipStream_t st) {
  uint32_t n = (c->n_leaves >> sl) >> 1;
  if (n <= 1) return DVP_OK;
  MatSet* ms;
  DVP_TRY(build_matset(c, sl, to_even, &ms, st));
  int ln = 31 - __builtin_clz(n);
  // Blocks are contiguous, so `batch` vectors of n behave like one vector of batch*n for the block
  // structure; the BATCH template only buys matrix reuse when the batch is small and n is large.
  auto pass = [&](const Fr* mats, int lh) {
    if (batch == 4) launch_bfly<4>(data, mats, lh, n, st);
    else if (batch == 2) launch_bfly<2>(data, mats, lh, n, st);
    else launch_bfly<1>(data, mats, lh, (uint32_t)((size_t)batch * n), st);
  };
  for (int d = 0; d < ln; ++d) pass(ms->dec + 4 * (size_t)(n - (n >> d)), ln - d - 1);
  for (int d = ln - 1; d >= 0; --d) pass(ms->rec + 4 * (size_t)(n - (n >> d)), ln - d - 1);
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
extern "C" int dvp_ecfft_create(uint32_t log_n, int shifted, uint32_t base_log, dvp_ecfft** out) {
  if (!out || log_n < 1 || log_n > (uint32_t)ECFFT_LOG_ORDER) return DVP_EINVAL;
  if (base_log == 0) base_log = log_n;
  if (base_log < log_n || base_log > (uint32_t)ECFFT_LOG_ORDER) return DVP_EINVAL;
  dvp_ecfft* c = new dvp_ecfft();
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
  if (!fr_is_zero(tab[log_n - 1].y)) {  // order-2 point must have y == 0
    delete c;
    return DVP_EINVAL;
  }
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
  *out = c;
  return DVP_OK;
}

extern "C" void dvp_ecfft_destroy(dvp_ecfft* c) {
  if (!c) return;
  (void)hipFree(c->layers);
  for (auto& kv : c->mats) {
    (void)hipFree(kv.second.dec);
    (void)hipFree(kv.second.rec);
  }
  for (auto& kv : c->xnn) (void)hipFree(kv.second);
  if (c->scratch) (void)hipFree(c->scratch);
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
  if (d_in != d_out)
    DVP_HIP(hipMemcpyAsync(d_out, d_in, (size_t)batch * n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
  return extend_inplace(c, 0, 0, (Fr*)d_out, batch, st);
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

// enter: bottom-up over recursion depth k (sub-problem size sz = N>>k on the stride-2^k subtree).
// All 2^k sub-problems of a depth share the subtree, so each depth is ONE batched extend
// (batch 2^(k+1), vectors of sz/2) plus one combine kernel.
extern "C" int dvp_ecfft_enter_dev(dvp_ecfft* c, const void* d_coeffs, void* d_out, void* stream) {
  if (!c || !d_coeffs || !d_out) return DVP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const uint32_t N = c->n_leaves;
  if (!c->scratch) DVP_HIP(hipMalloc((void**)&c->scratch, (size_t)2 * N * sizeof(Fr)));
  Fr* even = c->scratch;      // evaluations on the even leaves (inputs of this depth)
  Fr* odd = c->scratch + N;   // their extension to the odd leaves
  // depth log_n: N sub-problems of size 1: evaluation == coefficient
  DVP_HIP(hipMemcpyAsync(even, d_coeffs, (size_t)N * sizeof(Fr), hipMemcpyDeviceToDevice, st));
  for (int k = c->log_n - 1; k >= 0; --k) {
    uint32_t sz = N >> k, h = sz >> 1, nsub = 1u << k;
    DVP_HIP(hipMemcpyAsync(odd, even, (size_t)N * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    DVP_TRY(extend_inplace(c, k, 0, odd, 2 * nsub, st));
    Fr* xnn;
    DVP_TRY(get_xnn(c, k, &xnn, st));
    Fr* dst = (k == 0) ? (Fr*)d_out : even;
    // combine reads even/odd and writes `dst`; when dst == even we go through d_out as a bounce
    Fr* tmp = (Fr*)d_out;
    hipLaunchKernelGGL(k_enter_combine, dim3(cdiv(nsub * h, TPB)), dim3(TPB), 0, st, even, odd, xnn, tmp, h, nsub * h);
    DVP_HIP(hipGetLastError());
    if (dst != tmp) DVP_HIP(hipMemcpyAsync(dst, tmp, (size_t)N * sizeof(Fr), hipMemcpyDeviceToDevice, st));
  }
  return DVP_OK;
}

extern "C" int dvp_ecfft_enter(dvp_ecfft* c, const uint64_t* coeffs, uint64_t* out) {
  if (!c || !coeffs || !out) return DVP_EINVAL;
  size_t bytes = (size_t)c->n_leaves * sizeof(Fr);
  DevBuf in, o;
  DVP_TRY(in.alloc(bytes));
  DVP_TRY(o.alloc(bytes));
  DVP_HIP(hipMemcpy(in.p, coeffs, bytes, hipMemcpyHostToDevice));
  DVP_TRY(check_canonical_dev(in.as<Fr>(), c->n_leaves, 0));
  DVP_TRY(dvp_ecfft_enter_dev(c, in.p, o.p, 0));
  DVP_HIP(hipMemcpy(out, o.p, bytes, hipMemcpyDeviceToHost));
  return DVP_OK;
}
