/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of oracle/pyref.py's FFTree (leaves, isogeny chain, extend, enter, exit) so that the ECFFT
 * kernels can be compared element for element at sizes the Python big-int oracle cannot reach in a test run
 * (2^12 .. 2^20 evaluations).  Same algorithms, same recursion, function for function:
 *
 *   dvo_fftree_new      <- pyref.FFTree.__init__      (build_ec_fftrees, src/ec_fft.rs:93-170; constants :205-229
 *                                                      are passed in by the caller from pyref.ECFFT_*)
 *   dvo_fftree_extend   <- pyref.FFTree._extend       (FFTree::extend(evals, Moiety::S1), call site src/proving.rs:412)
 *   dvo_fftree_enter    <- pyref.FFTree.enter         (FFTree::enter, call sites src/ec_fft.rs:317,411)
 *   dvo_fftree_exit     <- pyref.FFTree.exit          (FFTree::exit, call site src/ec_fft.rs:266), with the table
 *                                                      <Z_0^2 mod X^h> ("z0z0_rem_xnn_s", src/tree_io.rs:34-48) by brute force
 *   dvo_fftree_eval     <- Horner at the leaves: the DEFINITION of enter (independent of the recursion)
 *   dvo_fftree_matrices <- the 2x2 matrices of Lemma 3.2 as FFTree::{decompose,recombine}_matrices holds them
 *
 * The algorithm lives in the un-vendored crate alpenlabs/ecfft@9c6cac7 (Cargo.toml:39); the results are mathematically
 * unique given the domain, and tests/test_oracle_ecfft.py pins this file against pyref (itself pinned against O(n^2)
 * Lagrange interpolation, the shape of the reference's own test src/ec_fft.rs:883-907) and against dvo_fftree_eval.
 *
 * Fr arithmetic: 4 x 64-bit Montgomery, R = 2^256 (what ark-ff gives the reference's Fr, src/curve.rs:16-22).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Build: gcc -O3 -fopenmp -fPIC -shared dvp_oracle_ecfft.c -o _build/libdvp_oracle_ecfft.so
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef unsigned __int128 u128;
typedef struct { u64 v[4]; } fe; /* Montgomery form unless stated otherwise */

static const u64 FR_P[4] = {0x6efb1ad5f173abdfull, 0x00069d5bb915bcd4ull, 0x0000000000000000ull, 0x0000008000000000ull};
static const u64 FR_NINV = 0xa2918b898c382fe1ull; /* -p^-1 mod 2^64 */
static fe FE_R2, FE_ONE; /* 2^512 mod p and 2^256 mod p, derived at first use by modular doubling */
static int g_init = 0;

static inline int fe_is_zero(const fe* a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static inline int ge_p(const u64 t[4], u64 top) {
  if (top) return 1;
  for (int j = 3; j >= 0; --j) {
    if (t[j] > FR_P[j]) return 1;
    if (t[j] < FR_P[j]) return 0;
  }
  return 1;
}
static inline void sub_p(u64 t[4]) {
  u128 br = 0;
  for (int j = 0; j < 4; ++j) { u128 d = (u128)t[j] - FR_P[j] - (u64)br; t[j] = (u64)d; br = (d >> 64) & 1; }
}
static inline fe fe_add(fe a, fe b) {
  fe r;
  u128 c = 0;
  for (int j = 0; j < 4; ++j) { c += (u128)a.v[j] + b.v[j]; r.v[j] = (u64)c; c >>= 64; }
  if (ge_p(r.v, (u64)c)) sub_p(r.v);
  return r;
}
static inline fe fe_sub(fe a, fe b) {
  fe r;
  u128 br = 0;
  for (int j = 0; j < 4; ++j) { u128 d = (u128)a.v[j] - b.v[j] - (u64)br; r.v[j] = (u64)d; br = (d >> 64) & 1; }
  if (br) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) { c += (u128)r.v[j] + FR_P[j]; r.v[j] = (u64)c; c >>= 64; }
  }
  return r;
}
static inline fe fe_mul(fe a, fe b) {
  u64 t[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) { c += (u128)a.v[j] * b.v[i] + t[j]; t[j] = (u64)c; c >>= 64; }
    u64 t4 = t[4] + (u64)c;
    u64 mq = t[0] * FR_NINV;
    c = (u128)mq * FR_P[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; ++j) { c += (u128)mq * FR_P[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
    c += t4;
    t[3] = (u64)c;
    t[4] = (u64)(c >> 64);
  }
  fe r;
  memcpy(r.v, t, 32);
  if (ge_p(r.v, t[4])) sub_p(r.v);
  return r;
}
/* OpenMP sizes its team by the affinity mask, which on a shared host is far larger than the CPU quota of the container:
 * the caller passes the number of threads it may really use (oracle/c_oracle.py reads the cgroup quota) */
void dvo_ecfft_set_threads(int n) {
#ifdef _OPENMP
  if (n >= 1) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
static void fe_init(void) {
  if (g_init) return;
  fe x = {{1, 0, 0, 0}};
  for (int i = 0; i < 256; ++i) x = fe_add(x, x);
  FE_ONE = x; /* 2^256 mod p */
  for (int i = 0; i < 256; ++i) x = fe_add(x, x);
  FE_R2 = x; /* 2^512 mod p */
  g_init = 1;
}
static inline fe fe_to_mont(const u64 c[4]) { fe a; memcpy(a.v, c, 32); return fe_mul(a, FE_R2); }
static inline void fe_from_mont(fe a, u64 out[4]) { fe one = {{1, 0, 0, 0}}; fe r = fe_mul(a, one); memcpy(out, r.v, 32); }
static fe fe_pow_u64(fe a, u64 e) {
  fe r = FE_ONE;
  while (e) {
    if (e & 1) r = fe_mul(r, a);
    a = fe_mul(a, a);
    e >>= 1;
  }
  return r;
}
static fe fe_inv(fe a) { /* a^(p-2) */
  u64 e[4];
  memcpy(e, FR_P, 32);
  e[0] -= 2;
  fe r = FE_ONE;
  for (int i = 255; i >= 0; --i) {
    r = fe_mul(r, r);
    if ((e[i >> 6] >> (i & 63)) & 1) r = fe_mul(r, a);
  }
  return r;
}
/* in place; every element must be non-zero (the callers only invert differences of distinct domain points and powers of them) */
static void fe_batch_inv(fe* a, size_t n) {
  if (!n) return;
  fe* pre = (fe*)malloc(n * sizeof(fe));
  fe run = FE_ONE;
  for (size_t i = 0; i < n; ++i) { pre[i] = run; run = fe_mul(run, a[i]); }
  fe inv = fe_inv(run);
  for (size_t i = n; i-- > 0;) { fe t = fe_mul(inv, pre[i]); inv = fe_mul(inv, a[i]); a[i] = t; }
  free(pre);
}
/* the same split over OpenMP threads for long vectors */
static void fe_batch_inv_par(fe* a, size_t n) {
  const size_t chunk = 4096;
  if (n < 2 * chunk) { fe_batch_inv(a, n); return; }
  const size_t nch = (n + chunk - 1) / chunk;
#pragma omp parallel for schedule(static)
  for (size_t c = 0; c < nch; ++c) {
    size_t lo = c * chunk, hi = lo + chunk < n ? lo + chunk : n;
    fe_batch_inv(a + lo, hi - lo);
  }
}

/* ---- short-Weierstrass affine addition on y^2 = x^3 + a x + b (pyref.sw_add) ---------------------------------------- */
typedef struct { fe x, y; int inf; } swpt;
static swpt sw_add(swpt p, swpt q, fe a) {
  if (p.inf) return q;
  if (q.inf) return p;
  fe lam;
  fe dx = fe_sub(q.x, p.x);
  if (fe_is_zero(&dx)) {
    fe sy = fe_add(p.y, q.y);
    if (fe_is_zero(&sy)) { swpt r; memset(&r, 0, sizeof r); r.inf = 1; return r; }
    fe x2 = fe_mul(p.x, p.x);
    fe num = fe_add(fe_add(fe_add(x2, x2), x2), a);
    lam = fe_mul(num, fe_inv(fe_add(p.y, p.y)));
  } else {
    lam = fe_mul(fe_sub(q.y, p.y), fe_inv(dx));
  }
  swpt r;
  r.inf = 0;
  r.x = fe_sub(fe_sub(fe_mul(lam, lam), p.x), q.x);
  r.y = fe_sub(fe_mul(lam, fe_sub(p.x, r.x)), p.y);
  return r;
}

/* ---- the tree -------------------------------------------------------------------------------------------------------- */
typedef struct dvo_fftree {
  int log_n;
  size_t n;
  fe** layers; /* layers[d]: n >> d leaves, d = 0 .. log_n */
  fe* x0;      /* psi_d(x) = x + t_d / (x - x0_d) */
  fe* t;
  fe** zz;     /* zz[sl]: <Z_0^2 mod X^h> on the stride-2^sl tree, built on demand */
  fe** vpow;   /* vpow[d * (log_n + 1) + sl][j] = (L_d[j << sl] - x0_d)^(h - 1), h = (leaves of that strided layer) / 4: the
                  powers every extend on that (layer, stride) needs, computed once (enter / exit call extend many times) */
  fe** dinv;   /* dinv[(d * (log_n + 1) + sl) * 2 + src][3i .. 3i+2] = 1/v0, 1/v1, 1/(s1 - s0) of source pair i: the inverses of
                  the decompose step, shared by every sub-block of a recursion level */
} dvo_fftree;

void dvo_fftree_free(dvo_fftree* T) {
  if (!T) return;
  if (T->layers) for (int d = 0; d <= T->log_n; ++d) free(T->layers[d]);
  if (T->zz) for (int d = 0; d <= T->log_n; ++d) free(T->zz[d]);
  if (T->vpow) for (int k = 0; k < (T->log_n + 1) * (T->log_n + 1); ++k) free(T->vpow[k]);
  if (T->dinv) for (int k = 0; k < 2 * (T->log_n + 1) * (T->log_n + 1); ++k) free(T->dinv[k]);
  free(T->layers); free(T->zz); free(T->vpow); free(T->dinv); free(T->x0); free(T->t);
  free(T);
}

/* consts: a, gen.x, gen.y, coset.x, coset.y (canonical limbs; src/ec_fft.rs:209-229); gen has order 2^log_order */
dvo_fftree* dvo_fftree_new(int log_n, int shifted, int base_log_n, int log_order, const u64* consts) {
  fe_init();
  if (log_n < 1 || log_n > log_order || base_log_n < log_n || base_log_n > log_order) return NULL;
  dvo_fftree* T = (dvo_fftree*)calloc(1, sizeof(dvo_fftree));
  T->log_n = log_n;
  T->n = (size_t)1 << log_n;
  const size_t n = T->n;
  fe a = fe_to_mont(consts);
  swpt g = {fe_to_mont(consts + 4), fe_to_mont(consts + 8), 0};
  swpt coset = {fe_to_mont(consts + 12), fe_to_mont(consts + 16), 0};
  if (shifted) { /* src/ec_fft.rs:151-155 */
    swpt bg = g;
    for (int i = 0; i < log_order - base_log_n; ++i) bg = sw_add(bg, bg, a);
    coset = sw_add(coset, bg, a);
  }
  for (int i = 0; i < log_order - log_n; ++i) g = sw_add(g, g, a);
  /* tab[j] = 2^j g; tab[log_n - j] has order 2^j */
  swpt* tab = (swpt*)malloc((size_t)log_n * sizeof(swpt));
  tab[0] = g;
  for (int j = 1; j < log_n; ++j) tab[j] = sw_add(tab[j - 1], tab[j - 1], a);
  if (!fe_is_zero(&tab[log_n - 1].y)) { free(tab); dvo_fftree_free(T); return NULL; } /* the order-2 point has y = 0 */
  T->layers = (fe**)calloc((size_t)log_n + 1, sizeof(fe*));
  T->zz = (fe**)calloc((size_t)log_n + 1, sizeof(fe*));
  T->vpow = (fe**)calloc((size_t)(log_n + 1) * (log_n + 1), sizeof(fe*));
  T->dinv = (fe**)calloc((size_t)2 * (log_n + 1) * (log_n + 1), sizeof(fe*));
  T->x0 = (fe*)malloc((size_t)log_n * sizeof(fe));
  T->t = (fe*)malloc((size_t)log_n * sizeof(fe));
  for (int d = 0; d <= log_n; ++d) T->layers[d] = (fe*)malloc((n >> d) * sizeof(fe));
  /* leaves x(coset + i g) (src/ec_fft.rs:158-162): P_{i + 2^j} = P_i + 2^j g, one shared inversion per doubling of the
   * set (coset lies outside <g>, so no addition is exceptional) */
  {
    fe* px = (fe*)malloc(n * sizeof(fe));
    fe* py = (fe*)malloc(n * sizeof(fe));
    fe* den = (fe*)malloc(n * sizeof(fe));
    px[0] = coset.x; py[0] = coset.y;
    for (int j = 0; j < log_n; ++j) {
      const size_t cnt = (size_t)1 << j;
      const fe qx = tab[j].x, qy = tab[j].y;
#pragma omp parallel for schedule(static) if (cnt >= 4096)
      for (size_t i = 0; i < cnt; ++i) den[i] = fe_sub(qx, px[i]);
      fe_batch_inv_par(den, cnt);
#pragma omp parallel for schedule(static) if (cnt >= 4096)
      for (size_t i = 0; i < cnt; ++i) {
        fe lam = fe_mul(fe_sub(qy, py[i]), den[i]);
        fe x3 = fe_sub(fe_sub(fe_mul(lam, lam), px[i]), qx);
        px[cnt + i] = x3;
        py[cnt + i] = fe_sub(fe_mul(lam, fe_sub(px[i], x3)), py[i]);
      }
    }
    memcpy(T->layers[0], px, n * sizeof(fe));
    free(px); free(py); free(den);
  }
  /* isogeny chain: q[j-1] = x of the point of order 2^j, pushed through the maps as we go */
  fe* q = (fe*)malloc((size_t)log_n * sizeof(fe));
  for (int j = 1; j <= log_n; ++j) q[j - 1] = tab[log_n - j].x;
  fe acur = a;
  for (int d = 0; d < log_n; ++d) {
    const fe x0 = q[d];
    fe x0sq = fe_mul(x0, x0);
    const fe t = fe_add(fe_add(fe_add(x0sq, x0sq), x0sq), acur); /* 3 x0^2 + a_d */
    T->x0[d] = x0;
    T->t[d] = t;
    const size_t half = (n >> d) >> 1;
    fe* cur = T->layers[d];
    fe* nxt = T->layers[d + 1];
#pragma omp parallel for schedule(static) if (half >= 4096)
    for (size_t i = 0; i < half; ++i) nxt[i] = fe_sub(cur[i], x0);
    fe_batch_inv_par(nxt, half);
#pragma omp parallel for schedule(static) if (half >= 4096)
    for (size_t i = 0; i < half; ++i) nxt[i] = fe_add(cur[i], fe_mul(t, nxt[i]));
    for (int j = d + 1; j < log_n; ++j) q[j] = fe_add(q[j], fe_mul(t, fe_inv(fe_sub(q[j], x0))));
    fe t5 = fe_add(fe_add(fe_add(t, t), fe_add(t, t)), t);
    acur = fe_sub(acur, t5); /* a_{d+1} = a_d - 5 t */
  }
  free(q);
  free(tab);
  return T;
}

int dvo_fftree_log_n(const dvo_fftree* T) { return T ? T->log_n : 0; }
void dvo_fftree_layer(const dvo_fftree* T, int d, u64* out) { /* canonical */
  for (size_t i = 0; i < (T->n >> d); ++i) fe_from_mont(T->layers[d][i], out + 4 * i);
}

/* ---- extend (pyref.FFTree._extend) -------------------------------------------------------------------------------------
 * n evaluations on the even (to_even: odd) leaves of the depth-d layer of the stride-2^sl tree -> the other half */
#define LEAF(d, j) (T->layers[d][(size_t)(j) << sl])
/* (s - x0_d)^(h - 1) for every leaf s of the strided layer (2n leaves, h = n / 2), cached per (d, sl) */
static const fe* layer_powers(const dvo_fftree* T, int d, int sl, size_t n) {
  fe** slot = &((dvo_fftree*)T)->vpow[d * (T->log_n + 1) + sl];
  if (*slot) return *slot;
  const fe x0 = T->x0[d];
  const u64 e = (u64)(n / 2) - 1;
  fe* v = (fe*)malloc(2 * n * sizeof(fe));
#pragma omp parallel for schedule(static) if (n >= 256)
  for (size_t j = 0; j < 2 * n; ++j) v[j] = fe_pow_u64(fe_sub(LEAF(d, j), x0), e);
  *slot = v;
  return v;
}
static const fe* layer_inverses(const dvo_fftree* T, int d, int sl, int src, size_t n) {
  fe** slot = &((dvo_fftree*)T)->dinv[(d * (T->log_n + 1) + sl) * 2 + src];
  if (*slot) return *slot;
  const size_t h = n / 2;
  const fe* vp = layer_powers(T, d, sl, n);
  fe* inv = (fe*)malloc(3 * h * sizeof(fe));
#pragma omp parallel for schedule(static) if (h >= 512)
  for (size_t i = 0; i < h; ++i) {
    fe s0 = LEAF(d, 2 * i + src), s1 = LEAF(d, 2 * i + src + n);
    inv[3 * i] = vp[2 * i + src];
    inv[3 * i + 1] = vp[2 * i + src + n];
    inv[3 * i + 2] = fe_sub(s1, s0);
  }
  fe_batch_inv_par(inv, 3 * h);
  *slot = inv;
  return inv;
}
static void extend_rec(const dvo_fftree* T, const fe* ev, size_t n, int d, int sl, int to_even, fe* out) {
  if (n == 1) { out[0] = ev[0]; return; }
  const size_t h = n / 2;
  const int src = to_even ? 1 : 0, dst = to_even ? 0 : 1;
  const fe* vp = layer_powers(T, d, sl, n);
  const fe* inv = layer_inverses(T, d, sl, src, n);
  fe* p0 = (fe*)malloc(4 * h * sizeof(fe));
  fe* p1 = p0 + h;
  fe* f0 = p1 + h;
  fe* f1 = f0 + h;
#pragma omp parallel for schedule(static) if (h >= 512)
  for (size_t i = 0; i < h; ++i) { /* [e0; e1] = [[v0, s0 v0], [v1, s1 v1]] [p0; p1] solved for (p0, p1) */
    fe s0 = LEAF(d, 2 * i + src), s1 = LEAF(d, 2 * i + src + n);
    fe t0 = fe_mul(ev[i], inv[3 * i]), t1 = fe_mul(ev[i + h], inv[3 * i + 1]), dinv = inv[3 * i + 2];
    p1[i] = fe_mul(fe_sub(t1, t0), dinv);
    p0[i] = fe_mul(fe_sub(fe_mul(s1, t0), fe_mul(s0, t1)), dinv);
  }
  extend_rec(T, p0, h, d + 1, sl, to_even, f0);
  extend_rec(T, p1, h, d + 1, sl, to_even, f1);
#pragma omp parallel for schedule(static) if (h >= 512)
  for (size_t i = 0; i < h; ++i) {
    fe s0 = LEAF(d, 2 * i + dst), s1 = LEAF(d, 2 * i + dst + n);
    fe v0 = vp[2 * i + dst], v1 = vp[2 * i + dst + n];
    out[i] = fe_mul(v0, fe_add(f0[i], fe_mul(s0, f1[i])));
    out[i + h] = fe_mul(v1, fe_add(f0[i], fe_mul(s1, f1[i])));
  }
  free(p0);
}

/* ---- enter (pyref.FFTree.enter): coefficients -> evaluations on the leaves of the stride-2^sl tree ------------------- */
static void enter_rec(const dvo_fftree* T, const fe* c, size_t n, int sl, fe* out) {
  if (n == 1) { out[0] = c[0]; return; }
  const size_t h = n / 2;
  fe* u0 = (fe*)malloc(h * sizeof(fe));
  fe* v0 = (fe*)malloc(h * sizeof(fe));
  fe* u1 = (fe*)malloc(h * sizeof(fe));
  fe* v1 = (fe*)malloc(h * sizeof(fe));
  enter_rec(T, c, h, sl + 1, u0);
  enter_rec(T, c + h, h, sl + 1, v0);
  extend_rec(T, u0, h, 0, sl, 0, u1);
  extend_rec(T, v0, h, 0, sl, 0, v1);
#pragma omp parallel for schedule(static) if (h >= 512)
  for (size_t i = 0; i < h; ++i) {
    out[2 * i] = fe_add(u0[i], fe_mul(fe_pow_u64(LEAF(0, 2 * i), (u64)h), v0[i]));
    out[2 * i + 1] = fe_add(u1[i], fe_mul(fe_pow_u64(LEAF(0, 2 * i + 1), (u64)h), v1[i]));
  }
  free(u0); free(v0); free(u1); free(v1);
}

/* Z_0(x), the vanishing polynomial of the even leaves of the stride-2^sl tree, through the isogeny chain
 * (pyref.FFTree.vanish_even_at): the m even leaves all map to layers[k][0], m = 2^k */
static fe vanish_even_at(const dvo_fftree* T, fe x, int sl) {
  const size_t m = (T->n >> sl) >> 1;
  int k = 0;
  while (((size_t)1 << k) < m) ++k;
  fe u = x, v = FE_ONE;
  for (int d = 0; d < k; ++d) {
    fe x0 = T->x0[d], t = T->t[d];
    fe uv = fe_mul(u, v), vv = fe_mul(v, v);
    fe nu = fe_add(fe_sub(fe_mul(u, u), fe_mul(x0, uv)), fe_mul(t, vv));
    fe nv = fe_sub(uv, fe_mul(x0, vv));
    u = nu; v = nv;
  }
  return fe_sub(u, fe_mul(T->layers[k][0], v));
}

/* <Z_0^2 mod X^h> on the stride-2^sl tree by brute force from the roots (pyref.FFTree._z0z0_rem_xnn): O(n^2) */
static const fe* z0z0_rem_xnn(dvo_fftree* T, int sl) {
  if (T->zz[sl]) return T->zz[sl];
  const size_t n = T->n >> sl, h = n / 2;
  fe* z = (fe*)calloc(h + 1, sizeof(fe)); /* prod (X - even leaves), low to high */
  z[0] = FE_ONE;
  for (size_t r = 0; r < h; ++r) {
    const fe root = LEAF(0, 2 * r);
    for (size_t k = r + 1; k > 0; --k) z[k] = fe_sub(z[k - 1], fe_mul(root, z[k]));
    fe zero; memset(&zero, 0, sizeof zero);
    z[0] = fe_sub(zero, fe_mul(root, z[0]));
  }
  fe* zz = (fe*)calloc(h, sizeof(fe));
#pragma omp parallel for schedule(dynamic, 64) if (h >= 256)
  for (size_t k = 0; k < h; ++k) {
    fe acc; memset(&acc, 0, sizeof acc);
    for (size_t i = 0; i <= k; ++i) acc = fe_add(acc, fe_mul(z[i], z[k - i]));
    zz[k] = acc;
  }
  fe* out = (fe*)malloc(n * sizeof(fe));
#pragma omp parallel for schedule(static) if (n >= 256)
  for (size_t j = 0; j < n; ++j) {
    const fe x = LEAF(0, j);
    fe acc; memset(&acc, 0, sizeof acc);
    for (size_t k = h; k-- > 0;) acc = fe_add(fe_mul(acc, x), zz[k]);
    out[j] = acc;
  }
  free(z); free(zz);
  T->zz[sl] = out;
  return out;
}

/* <P Z_0^-1 mod A> on S from <P> on S, A given by its evaluations a_vals on S (pyref.FFTree._redc_z0) */
static void redc_z0(const dvo_fftree* T, const fe* ev, const fe* a_vals, size_t n, int sl, fe* out) {
  const size_t h = n / 2;
  fe* t0 = (fe*)malloc(h * sizeof(fe));
  fe* g1 = (fe*)malloc(h * sizeof(fe));
  fe* h1 = (fe*)malloc(h * sizeof(fe));
  fe* h0 = (fe*)malloc(h * sizeof(fe));
  fe* inv = (fe*)malloc(2 * h * sizeof(fe));
#pragma omp parallel for schedule(static) if (h >= 512)
  for (size_t i = 0; i < h; ++i) {
    inv[i] = a_vals[2 * i];
    inv[h + i] = vanish_even_at(T, LEAF(0, 2 * i + 1), sl);
  }
  fe_batch_inv_par(inv, 2 * h);
  for (size_t i = 0; i < h; ++i) t0[i] = fe_mul(ev[2 * i], inv[i]);
  extend_rec(T, t0, h, 0, sl, 0, g1);
  for (size_t i = 0; i < h; ++i) h1[i] = fe_mul(fe_sub(ev[2 * i + 1], fe_mul(g1[i], a_vals[2 * i + 1])), inv[h + i]);
  extend_rec(T, h1, h, 0, sl, 1, h0);
  for (size_t i = 0; i < h; ++i) { out[2 * i] = h0[i]; out[2 * i + 1] = h1[i]; }
  free(t0); free(g1); free(h1); free(h0); free(inv);
}

/* exit (pyref.FFTree.exit): evaluations on the leaves of the stride-2^sl tree -> coefficients */
static void exit_rec(dvo_fftree* T, const fe* ev, size_t n, int sl, fe* out) {
  if (n == 1) { out[0] = ev[0]; return; }
  const size_t h = n / 2;
  fe* xnn = (fe*)malloc(n * sizeof(fe));
  fe* r1 = (fe*)malloc(n * sizeof(fe));
  fe* u = (fe*)malloc(n * sizeof(fe));
  fe* half = (fe*)malloc(h * sizeof(fe));
#pragma omp parallel for schedule(static) if (n >= 512)
  for (size_t j = 0; j < n; ++j) xnn[j] = fe_pow_u64(LEAF(0, j), (u64)h);
  const fe* c = z0z0_rem_xnn(T, sl);
  redc_z0(T, ev, xnn, n, sl, r1);
  for (size_t j = 0; j < n; ++j) r1[j] = fe_mul(r1[j], c[j]);
  redc_z0(T, r1, xnn, n, sl, u); /* <P mod X^h> on S */
  for (size_t i = 0; i < h; ++i) half[i] = u[2 * i];
  exit_rec(T, half, h, sl + 1, out);
  {
    fe* inv = (fe*)malloc(h * sizeof(fe));
    for (size_t i = 0; i < h; ++i) inv[i] = xnn[2 * i];
    fe_batch_inv_par(inv, h);
    for (size_t i = 0; i < h; ++i) half[i] = fe_mul(fe_sub(ev[2 * i], u[2 * i]), inv[i]);
    free(inv);
  }
  exit_rec(T, half, h, sl + 1, out + h);
  free(xnn); free(r1); free(u); free(half);
}

/* ---- canonical-limb wrappers ------------------------------------------------------------------------------------------- */
static fe* load(const u64* in, size_t n) {
  fe* a = (fe*)malloc((n ? n : 1) * sizeof(fe));
#pragma omp parallel for schedule(static) if (n >= 4096)
  for (size_t i = 0; i < n; ++i) a[i] = fe_to_mont(in + 4 * i);
  return a;
}
static void store(const fe* a, size_t n, u64* out) {
#pragma omp parallel for schedule(static) if (n >= 4096)
  for (size_t i = 0; i < n; ++i) fe_from_mont(a[i], out + 4 * i);
}
/* n_evals = (n >> sl) / 2 values */
int dvo_fftree_extend(const dvo_fftree* T, const u64* ev, int sl, int to_even, u64* out) {
  if (!T || sl < 0 || sl >= T->log_n) return -1;
  const size_t n = (T->n >> sl) >> 1;
  fe* a = load(ev, n);
  fe* o = (fe*)malloc(n * sizeof(fe));
  extend_rec(T, a, n, 0, sl, to_even, o);
  store(o, n, out);
  free(a); free(o);
  return 0;
}
int dvo_fftree_enter(const dvo_fftree* T, const u64* coeffs, int sl, u64* out) {
  if (!T || sl < 0 || sl > T->log_n) return -1;
  const size_t n = T->n >> sl;
  fe* a = load(coeffs, n);
  fe* o = (fe*)malloc(n * sizeof(fe));
  enter_rec(T, a, n, sl, o);
  store(o, n, out);
  free(a); free(o);
  return 0;
}
int dvo_fftree_exit(dvo_fftree* T, const u64* ev, int sl, u64* out) {
  if (!T || sl < 0 || sl > T->log_n) return -1;
  const size_t n = T->n >> sl;
  fe* a = load(ev, n);
  fe* o = (fe*)malloc(n * sizeof(fe));
  exit_rec(T, a, n, sl, o);
  store(o, n, out);
  free(a); free(o);
  return 0;
}
/* the definition of enter: out[j] = sum_k coeffs[k] leaf_j^k (Horner), O(n^2) */
int dvo_fftree_eval(const dvo_fftree* T, const u64* coeffs, int sl, u64* out) {
  if (!T || sl < 0 || sl > T->log_n) return -1;
  const size_t n = T->n >> sl;
  fe* c = load(coeffs, n);
  fe* o = (fe*)malloc(n * sizeof(fe));
#pragma omp parallel for schedule(static) if (n >= 64)
  for (size_t j = 0; j < n; ++j) {
    const fe x = LEAF(0, j);
    fe acc; memset(&acc, 0, sizeof acc);
    for (size_t k = n; k-- > 0;) acc = fe_add(fe_mul(acc, x), c[k]);
    o[j] = acc;
  }
  store(o, n, out);
  free(c); free(o);
  return 0;
}
/* The 2x2 matrices of one direction of extend on the whole tree (sl = 0), layer by layer: out holds (n_evals - 1) matrices
 * of 4 canonical elements (m00 m01 m10 m11), n_evals = leaves / 2, layer d (n_evals >> (d+1) matrices, pair i built from
 * the layer-d leaves 2i+s and 2i+s+(n_evals >> d)) at matrix offset n_evals - (n_evals >> d).
 *   which = 1 (recombine): M = [[v0, s0 v0], [v1, s1 v1]], v = (s - x0_d)^(pairs - 1), from the DESTINATION leaves (s = 1 - to_even)
 *   which = 0 (decompose): M^-1 of the same form built from the SOURCE leaves (s = to_even)
 * -- FFTree::recombine_matrices / decompose_matrices (Lemma 3.2 of the ECFFT paper) as extend_impl indexes them. */
int dvo_fftree_matrices(const dvo_fftree* T, int to_even, int which, u64* out) {
  if (!T || T->log_n < 2) return -1;
  const int sl = 0;
  const size_t n = T->n >> 1;
  const int src = to_even ? 1 : 0, dst = to_even ? 0 : 1, s = which ? dst : src;
  for (int d = 0; ((size_t)n >> d) > 1; ++d) {
    const size_t nd = n >> d, h = nd >> 1, off = n - nd;
    const fe x0 = T->x0[d];
#pragma omp parallel for schedule(static) if (h >= 512)
    for (size_t i = 0; i < h; ++i) {
      fe s0 = LEAF(d, 2 * i + s), s1 = LEAF(d, 2 * i + s + nd);
      fe v0 = fe_pow_u64(fe_sub(s0, x0), (u64)h - 1), v1 = fe_pow_u64(fe_sub(s1, x0), (u64)h - 1);
      fe m[4] = {v0, fe_mul(s0, v0), v1, fe_mul(s1, v1)};
      if (!which) { /* inverse: 1/det [[m11, -m01], [-m10, m00]] */
        fe det = fe_sub(fe_mul(m[0], m[3]), fe_mul(m[1], m[2]));
        fe di = fe_inv(det);
        fe zero; memset(&zero, 0, sizeof zero);
        fe r[4] = {fe_mul(m[3], di), fe_mul(fe_sub(zero, m[1]), di), fe_mul(fe_sub(zero, m[2]), di), fe_mul(m[0], di)};
        memcpy(m, r, sizeof r);
      }
      for (int k = 0; k < 4; ++k) fe_from_mont(m[k], out + 4 * (4 * (off + i) + k));
    }
  }
  return 0;
}
/* Z_0(x) of the whole tree's even leaves (canonical in / out), for completeness of the python mirror */
void dvo_fftree_vanish_even_at(const dvo_fftree* T, const u64 x[4], int sl, u64 out[4]) {
  fe_from_mont(vanish_even_at(T, fe_to_mont(x), sl), out);
}
