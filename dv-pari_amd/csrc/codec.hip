// xsk233 30-byte codec, batched fixed-base multiplication (the SRS commitment loop) and the
// wire-format MSM entry point.
//
//   CurvePoint::to_bytes / from_bytes       src/curve.rs:93-109      -> dvp_points_encode / _decode
//   read_point_vec_from_file's decode loop  src/io_utils.rs:217-226  -> dvp_points_decode
//   point_scalar_mul_gen sweeps             src/srs.rs:130-133,138-144,155-158 -> dvp_mulgen_batch*
//   multi_scalar_mul on file-format inputs  src/proving.rs:462-463,511-512,666-680 -> dvp_msm_xsk233
//
// CODEC RULE (candidate; byte-level parity with xs233 is UNPINNED -- the reference holds no
// known-answer bytes and the xs233 source is not available offline):
//   xsk233 = { P + N : P in E[r] }, N = (0,1).  With (x,s)-coordinates on the N=(0,0) model,
//   Pornin's encoding is w = sqrt(s/x) of the element P + N; expressed on the E[r] representative
//   P = (x,y) that is  w = sqrt(x + y/x + 1)  (lambda-coordinate of P plus one); neutral -> 0.
//   Decoding: e = w^2 + w, solve x^2 + e x + 1 = 0 (half-trace), roots are x(P) and x(P+N) = 1/x;
//   y = x (w^2 + 1 + x); keep the root whose point lies in E[r] = 4E (two trace tests).
#include <atomic>
#include <cstdlib>
#include <mutex>

#include "common.h"
#include "k233.cuh"
#include "tau.cuh"
#include "codec.cuh"

namespace dvp {

int msm_affine_dev(const void* d_scalars, const void* d_bases, const void* d_inf, size_t n, void* d_out_xy,
                   void* d_out_inf, hipStream_t st);
int gf_sqr_tables(GfSqrTables* out, hipStream_t st);

// NIST K-233 base point
__constant__ uint32_t K233_GX[8] = {0xefad6126u, 0x0a4c9d6eu, 0x19c26bf5u, 0x149563a4u,
                                    0x29f22ff4u, 0x7e731af1u, 0x32ba853au, 0x00000172u};
__constant__ uint32_t K233_GY[8] = {0x56fae6a3u, 0x56e0c110u, 0xf18aeb9bu, 0x27a8cd9bu,
                                    0x555a67c4u, 0x19b7f70fu, 0x537dece8u, 0x000001dbu};

// P in E[r] (affine, x != 0)?  E[r] = 4E: Tr(x) = 0 and a half of P has Tr(x_half) = 0.
// half-trace through the byte table (30 lookups instead of 232 squarings)
__device__ __forceinline__ bool k233_in_subgroup(const Aff& p, const GfSqrTables& T, const GfLdsK& L) {
  if (gf_is_zero(p.x)) return false;
  if (gf_trace(p.x)) return false;
  Gf lam = gf_sqr_tab(p.x, T.th);                   // lam^2 + lam = x
  Gf u2 = gf_add(p.y, gf_mul(gf_add(lam, gf_one()), p.x, L));  // x_half^2
  return gf_trace(u2) == 0;
}

__device__ __forceinline__ bool k233_on_curve(const Aff& p) {
  // y^2 + xy = x^3 + 1
  Gf lhs = gf_add(gf_sqr(p.y), gf_mul(p.x, p.y));
  Gf rhs = gf_add(gf_mul(gf_sqr(p.x), p.x), gf_one());
  return gf_eq(lhs, rhs);
}

__global__ void __launch_bounds__(256)
k_encode(const Aff* __restrict__ pts, const uint8_t* __restrict__ inf, size_t n, GfSqrTables T, uint8_t* __restrict__ out, int rule) {
  extern __shared__ char lds_raw[];
  GfLdsK L = gf_ldsk_init(lds_raw);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Gf w = gf_zero();
  if (!(inf && inf[i])) {
    Aff p = pts[i];
    Gf lam1 = gf_add(gf_add(p.x, gf_mul(p.y, gf_inv_fast(p.x, T, L), L)), gf_one());
    w = gf_sqr_tab(gf_sqr_tab(lam1, T.t116), T.t116);  // sqrt = 232 squarings = two 116-step table passes
    if (rule) w = codec_present(w, rule, T);
  }
  store30(out + i * 30, w, rule);
}

// one point (a proof's commit_p / kzg_k), latency only: one quad of lanes through the quad-cooperative multiplier
// (45 us against 90 for a single lane), infinity flag read as the u32 the MSM writes
__global__ void __launch_bounds__(64)
k_encode_point(const Aff* __restrict__ pt, const uint32_t* __restrict__ inf32, GfSqrTables T, uint8_t* __restrict__ out, int rule) {
  extern __shared__ char lds_raw[];
  GfLdsQ L = gf_ldsq_init(lds_raw);
  if (threadIdx.x >= 4) return;
  Gf w = gf_zero();
  if (!*inf32) {
    Aff p = *pt;
    Gf lam1 = gf_add(gf_add(p.x, gf_mul(p.y, gf_inv_fast(p.x, T, L), L)), gf_one());
    w = gf_sqr_tab(gf_sqr_tab(lam1, T.t116), T.t116);
    if (rule) w = codec_present(w, rule, T);
  }
  if (threadIdx.x == 0) store30(out, w, rule);
}

__global__ void __launch_bounds__(256, 2)
k_decode(const uint8_t* __restrict__ enc, size_t n, GfSqrTables T, Aff* __restrict__ out, uint8_t* __restrict__ inf,
         unsigned long long* __restrict__ err, int rule) {
  extern __shared__ char lds_raw[];
  GfLdsK L = gf_ldsk_init(lds_raw);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t top;
  Gf w = load30(enc + i * 30, &top, rule);
  Aff r;
  r.x = gf_zero();
  r.y = gf_zero();
  bool ok = (top == 0), is_inf = false;
  if (ok && gf_is_zero(w)) {
    is_inf = true;
  } else if (ok) {
    if (rule) w = codec_absorb(w, rule, T);
    Gf w2 = gf_sqr(w);
    Gf e = gf_add(w2, w);
    ok = !gf_is_zero(e);
    if (ok) {
      Gf einv = gf_inv_fast(e, T, L);
      Gf cst = gf_sqr(einv);  // 1/e^2
      ok = gf_trace(cst) == 0;
      if (ok) {
        Gf z = gf_sqr_tab(cst, T.th);   // half-trace: z^2 + z = 1/e^2
        Gf lam = gf_add(w2, gf_one());  // x + y/x
        Aff c0, c1;
        c0.x = gf_mul(e, z, L);
        c1.x = gf_add(c0.x, e);
        c0.y = gf_mul(c0.x, gf_add(lam, c0.x), L);
        c1.y = gf_mul(c1.x, gf_add(lam, c1.x), L);
        bool s0 = k233_in_subgroup(c0, T, L), s1 = k233_in_subgroup(c1, T, L);
        if (s0) r = c0; else if (s1) r = c1; else ok = false;
      }
    }
  }
  if (!ok) {
    atomicMin(err, (unsigned long long)i);
    is_inf = true;
  }
  out[i] = r;
  inf[i] = is_inf ? 1 : 0;
}

// ---- fixed-base tables: tab[w][d] = sum_t d_t tau^(c w + t)(G), affine ------------------------------
constexpr int GEN_C = 16;  // 15 windows, 63 MB table (L2 / Infinity-Cache resident): 15 mixed additions per scalar
constexpr int GEN_W = TAU_DIGITS / GEN_C;  // 15

__global__ void __launch_bounds__(256) k_gen_table(Aff* __restrict__ tab) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (uint32_t)(GEN_W << GEN_C)) return;
  uint32_t w = tid >> GEN_C, d = tid & ((1u << GEN_C) - 1);
  Aff g;
#pragma unroll
  for (int k = 0; k < 8; ++k) { g.x.w[k] = K233_GX[k]; g.y.w[k] = K233_GY[k]; }
  // g <- tau^(c w)(G)
  for (uint32_t s = 0; s < w * GEN_C; ++s) { g.x = gf_sqr(g.x); g.y = gf_sqr(g.y); }
  Ld acc = ld_infinity();
  for (int t = 0; t < GEN_C; ++t) {
    if ((d >> t) & 1) acc = ld_madd(acc, g);
    g.x = gf_sqr(g.x);
    g.y = gf_sqr(g.y);
  }
  Aff a;
  ld_to_aff(acc, &a);
  tab[tid] = a;  // d == 0 -> zeros, never read
}

__global__ void __launch_bounds__(256, 2)
k_mulgen(const uint32_t* __restrict__ scalars, size_t n, const Aff* __restrict__ tab, GfSqrTables T, Aff* __restrict__ out,
         uint8_t* __restrict__ out_inf, unsigned long long* __restrict__ err) {
  extern __shared__ char lds_raw[];
  GfLdsK L = gf_ldsk_init(lds_raw);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = scalars[i * 8 + k];
  uint32_t r0[5], r1[5];
  if (!tau_scalar_is_canonical(s)) {
    atomicMin(err, (unsigned long long)i);
#pragma unroll
    for (int k = 0; k < 5; ++k) r0[k] = r1[k] = 0;
  } else {
    tau_partial_reduce(s, r0, r1);
  }
  // digits sixteen per step (tau.cuh), windows of GEN_C cut from a 64-bit shift register; LDS-comb products
  Ld acc = ld_infinity();
  uint64_t buf = 0;
  int nb = 0, w = 0;
#pragma unroll 1
  for (int done = 0; done < GEN_W * GEN_C; done += 16) {
    buf |= (uint64_t)tau_step16(r0, r1) << nb;
    nb += 16;
    while (nb >= GEN_C && w < GEN_W) {
      uint32_t d = (uint32_t)buf & ((1u << GEN_C) - 1);
      buf >>= GEN_C;
      nb -= GEN_C;
      if (d) ld_madd_ip(acc, tab[((size_t)w << GEN_C) + d], L);
      ++w;
    }
  }
  Aff a;
  a.x = gf_zero();
  a.y = gf_zero();
  bool fin = !ld_is_inf(acc);
  if (fin) {
    Gf zi = gf_inv_fast(acc.Z, T, L);
    a.x = gf_mul(acc.X, zi, L);
    a.y = gf_mul(acc.Y, gf_sqr(zi), L);
  }
  out[i] = a;
  out_inf[i] = fin ? 0 : 1;
}

// the rule in force: DVP_CODEC_RULE from the environment at first use, dvp_codec_set_rule at run time
static std::atomic<int> g_codec_rule{-1};
static int codec_rule() {
  int r = g_codec_rule.load();
  if (r < 0) {
    const char* e = getenv("DVP_CODEC_RULE");
    r = e ? atoi(e) : 0;
    if (r < 0 || r >= CODEC_RULES) r = 0;
    g_codec_rule.store(r);
  }
  return r;
}

static std::mutex g_gen_mu;
static Aff* g_gen_tab[16] = {nullptr};

int gen_table(const Aff** out, hipStream_t st) {
  int dev;
  DVP_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return DVP_EINVAL;
  std::lock_guard<std::mutex> g(g_gen_mu);
  if (!g_gen_tab[dev]) {
    Aff* t;
    size_t cnt = (size_t)GEN_W << GEN_C;
    DVP_HIP(hipMalloc((void**)&t, cnt * sizeof(Aff)));
    hipLaunchKernelGGL(k_gen_table, dim3(cdiv(cnt, 256)), dim3(256), 0, st, t);
    DVP_HIP(hipGetLastError());
    DVP_HIP(hipStreamSynchronize(st));
    g_gen_tab[dev] = t;
  }
  *out = g_gen_tab[dev];
  return DVP_OK;
}

static int read_err(unsigned long long* d_err, hipStream_t st, int code) {
  unsigned long long e;
  DVP_HIP(hipMemcpyAsync(&e, d_err, 8, hipMemcpyDeviceToHost, st));
  DVP_HIP(hipStreamSynchronize(st));
  if (e != ~0ull) {
    g_last_error_index = (int64_t)e;
    return code;
  }
  return DVP_OK;
}

int mulgen_dev(const void* d_scalars, size_t n, Aff* d_out, uint8_t* d_inf, hipStream_t st) {
  const Aff* tab;
  DVP_TRY(gen_table(&tab, st));
  DevBuf err;
  DVP_TRY(err.alloc(8));
  DVP_HIP(hipMemsetAsync(err.p, 0xff, 8, st));
  GfSqrTables T;
  DVP_TRY(gf_sqr_tables(&T, st));
  hipLaunchKernelGGL(k_mulgen, dim3(cdiv(n, 256)), dim3(256), 4 * GF_LDSK_BYTES_PER_WAVE, st, (const uint32_t*)d_scalars, n, tab, T, d_out,
                     d_inf, err.as<unsigned long long>());
  DVP_HIP(hipGetLastError());
  return read_err(err.as<unsigned long long>(), st, DVP_EINVAL);
}

int encode_dev(const Aff* d_pts, const uint8_t* d_inf, size_t n, uint8_t* d_out, hipStream_t st) {
  GfSqrTables T;
  DVP_TRY(gf_sqr_tables(&T, st));
  hipLaunchKernelGGL(k_encode, dim3(cdiv(n, 256)), dim3(256), 4 * GF_LDSK_BYTES_PER_WAVE, st, d_pts, d_inf, n, T, d_out, codec_rule());
  DVP_HIP(hipGetLastError());
  return DVP_OK;
}

int encode_point_dev(const Aff* d_pt, const uint32_t* d_inf32, uint8_t* d_out, hipStream_t st) {
  GfSqrTables T;
  DVP_TRY(gf_sqr_tables(&T, st));
  hipLaunchKernelGGL(k_encode_point, dim3(1), dim3(64), GF_LDS_BYTES_PER_WAVE, st, d_pt, d_inf32, T, d_out, codec_rule());
  DVP_HIP(hipGetLastError());
  return DVP_OK;
}

int decode_dev(const uint8_t* d_enc, size_t n, Aff* d_out, uint8_t* d_inf, hipStream_t st) {
  DevBuf err;
  DVP_TRY(err.alloc(8));
  DVP_HIP(hipMemsetAsync(err.p, 0xff, 8, st));
  GfSqrTables T;
  DVP_TRY(gf_sqr_tables(&T, st));
  hipLaunchKernelGGL(k_decode, dim3(cdiv(n, 256)), dim3(256), 4 * GF_LDSK_BYTES_PER_WAVE, st, d_enc, n, T, d_out, d_inf,
                     err.as<unsigned long long>(), codec_rule());
  DVP_HIP(hipGetLastError());
  return read_err(err.as<unsigned long long>(), st, DVP_EDECODE);
}

// CurvePoint::add over two vectors (src/curve.rs:84-90): out[i] = a[i] + b[i] on the E[r] representative; complete
// (doubling, inverses and the neutral element are handled), register-only arithmetic -- a convenience entry, not a hot path.
__global__ void __launch_bounds__(128)
k_points_add(const Aff* __restrict__ a, const uint8_t* __restrict__ ainf, const Aff* __restrict__ b, const uint8_t* __restrict__ binf,
             size_t n, Aff* __restrict__ out, uint8_t* __restrict__ oinf) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Ld acc = (ainf && ainf[i]) ? ld_infinity() : ld_from_aff(a[i]);
  if (!(binf && binf[i])) acc = ld_madd(acc, b[i]);
  Aff r;
  bool fin = ld_to_aff(acc, &r);
  out[i] = r;
  oinf[i] = fin ? 0 : 1;
}

}  // namespace dvp

using namespace dvp;

extern "C" int dvp_mulgen_batch_affine(const uint64_t* scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf) {
  if (!n) return DVP_OK;
  if (!scalars || !out_xy || !out_inf) return DVP_EINVAL;
  DevBuf ds, dp, di;
  DVP_TRY(ds.alloc(n * 32));
  DVP_TRY(dp.alloc(n * 64));
  DVP_TRY(di.alloc(n));
  DVP_HIP(hipMemcpy(ds.p, scalars, n * 32, hipMemcpyHostToDevice));
  DVP_TRY(mulgen_dev(ds.p, n, dp.as<Aff>(), di.as<uint8_t>(), 0));
  DVP_HIP(hipMemcpy(out_xy, dp.p, n * 64, hipMemcpyDeviceToHost));
  DVP_HIP(hipMemcpy(out_inf, di.p, n, hipMemcpyDeviceToHost));
  return DVP_OK;
}

extern "C" int dvp_mulgen_batch(const uint64_t* scalars, size_t n, uint8_t* out_enc) {
  if (!n) return DVP_OK;
  if (!scalars || !out_enc) return DVP_EINVAL;
  DevBuf ds, dp, di, de;
  DVP_TRY(ds.alloc(n * 32));
  DVP_TRY(dp.alloc(n * 64));
  DVP_TRY(di.alloc(n));
  DVP_TRY(de.alloc(n * 30));
  DVP_HIP(hipMemcpy(ds.p, scalars, n * 32, hipMemcpyHostToDevice));
  DVP_TRY(mulgen_dev(ds.p, n, dp.as<Aff>(), di.as<uint8_t>(), 0));
  DVP_TRY(encode_dev(dp.as<Aff>(), di.as<uint8_t>(), n, de.as<uint8_t>(), 0));
  DVP_HIP(hipMemcpy(out_enc, de.p, n * 30, hipMemcpyDeviceToHost));
  return DVP_OK;
}

extern "C" int dvp_points_encode(const uint64_t* xy, const uint8_t* inf, size_t n, uint8_t* out_enc) {
  if (!n) return DVP_OK;
  if (!xy || !out_enc) return DVP_EINVAL;
  DevBuf dp, di, de;
  DVP_TRY(dp.alloc(n * 64));
  DVP_TRY(de.alloc(n * 30));
  DVP_HIP(hipMemcpy(dp.p, xy, n * 64, hipMemcpyHostToDevice));
  if (inf) {
    DVP_TRY(di.alloc(n));
    DVP_HIP(hipMemcpy(di.p, inf, n, hipMemcpyHostToDevice));
  }
  DVP_TRY(encode_dev(dp.as<Aff>(), di.as<uint8_t>(), n, de.as<uint8_t>(), 0));
  DVP_HIP(hipMemcpy(out_enc, de.p, n * 30, hipMemcpyDeviceToHost));
  return DVP_OK;
}

extern "C" int dvp_points_add(const uint64_t* a_xy, const uint8_t* a_inf, const uint64_t* b_xy, const uint8_t* b_inf, size_t n,
                              uint64_t* out_xy, uint8_t* out_inf) {
  if (!n) return DVP_OK;
  if (!a_xy || !b_xy || !out_xy || !out_inf) return DVP_EINVAL;
  DevBuf da, db, dai, dbi, dout, doi;
  DVP_TRY(da.alloc(n * 64));
  DVP_TRY(db.alloc(n * 64));
  DVP_TRY(dout.alloc(n * 64));
  DVP_TRY(doi.alloc(n));
  DVP_HIP(hipMemcpy(da.p, a_xy, n * 64, hipMemcpyHostToDevice));
  DVP_HIP(hipMemcpy(db.p, b_xy, n * 64, hipMemcpyHostToDevice));
  if (a_inf) {
    DVP_TRY(dai.alloc(n));
    DVP_HIP(hipMemcpy(dai.p, a_inf, n, hipMemcpyHostToDevice));
  }
  if (b_inf) {
    DVP_TRY(dbi.alloc(n));
    DVP_HIP(hipMemcpy(dbi.p, b_inf, n, hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(k_points_add, dim3(cdiv(n, 128)), dim3(128), 0, 0, da.as<Aff>(), dai.as<uint8_t>(), db.as<Aff>(), dbi.as<uint8_t>(), n,
                     dout.as<Aff>(), doi.as<uint8_t>());
  DVP_HIP(hipGetLastError());
  DVP_HIP(hipMemcpy(out_xy, dout.p, n * 64, hipMemcpyDeviceToHost));
  DVP_HIP(hipMemcpy(out_inf, doi.p, n, hipMemcpyDeviceToHost));
  return DVP_OK;
}

extern "C" int dvp_points_decode(const uint8_t* enc, size_t n, uint64_t* out_xy, uint8_t* out_inf) {
  if (!n) return DVP_OK;
  if (!enc || !out_xy || !out_inf) return DVP_EINVAL;
  DevBuf dp, di, de;
  DVP_TRY(dp.alloc(n * 64));
  DVP_TRY(di.alloc(n));
  DVP_TRY(de.alloc(n * 30));
  DVP_HIP(hipMemcpy(de.p, enc, n * 30, hipMemcpyHostToDevice));
  int rc = decode_dev(de.as<uint8_t>(), n, dp.as<Aff>(), di.as<uint8_t>(), 0);
  DVP_HIP(hipMemcpy(out_xy, dp.p, n * 64, hipMemcpyDeviceToHost));
  DVP_HIP(hipMemcpy(out_inf, di.p, n, hipMemcpyDeviceToHost));
  return rc;
}

// scalars: n x 32 B canonical LE; bases: n x 30 B xsk233 encodings; out: 30 B encoding of the sum
extern "C" int dvp_msm_xsk233(const uint8_t* scalars, const uint8_t* bases_enc, size_t n, uint8_t out_enc[30]) {
  if ((n && (!scalars || !bases_enc)) || !out_enc) return DVP_EINVAL;
  DevBuf ds, dp, di, de, dout;
  DVP_TRY(ds.alloc(n * 32));
  DVP_TRY(dp.alloc(n * 64));
  DVP_TRY(di.alloc(n));
  DVP_TRY(de.alloc(n * 30 + 32));
  DVP_TRY(dout.alloc(64 + 16));
  if (n) {
    DVP_HIP(hipMemcpy(ds.p, scalars, n * 32, hipMemcpyHostToDevice));
    DVP_HIP(hipMemcpy(de.p, bases_enc, n * 30, hipMemcpyHostToDevice));
    DVP_TRY(decode_dev(de.as<uint8_t>(), n, dp.as<Aff>(), di.as<uint8_t>(), 0));
  }
  DVP_TRY(msm_affine_dev(ds.p, dp.p, di.p, n, dout.p, (char*)dout.p + 64, 0));
  // the infinity flag of the result is a u32; k_encode wants a byte mask
  uint32_t inf32;
  DVP_HIP(hipMemcpy(&inf32, (char*)dout.p + 64, 4, hipMemcpyDeviceToHost));
  uint8_t inf8 = inf32 ? 1 : 0;
  DVP_HIP(hipMemcpy((char*)dout.p + 72, &inf8, 1, hipMemcpyHostToDevice));
  DVP_TRY(encode_dev(dout.as<Aff>(), (uint8_t*)dout.p + 72, 1, de.as<uint8_t>(), 0));
  DVP_HIP(hipMemcpy(out_enc, de.p, 30, hipMemcpyDeviceToHost));
  return DVP_OK;
}

// which presentation of the encoded field element the 30 bytes hold (see the table at the top of this file); 0 = the default
// candidate.  Affects every later encode / decode of the process (SRS files written under one rule must be read under it).
extern "C" int dvp_codec_set_rule(int rule) {
  if (rule < 0 || rule >= CODEC_RULES) return DVP_EINVAL;
  g_codec_rule.store(rule);
  return DVP_OK;
}
extern "C" int dvp_codec_get_rule(void) { return codec_rule(); }
