/*
 * ORACLE (test infrastructure, NOT product code): the reference's MSM shape -- one scalar multiplication per point plus
 * an addition (src/curve.rs:141-158) -- on OpenSSL's sect233k1 (EC_POINT_mul), as an independent third-party CPU
 * datapoint next to the C port ("B2").  Loaded only by bench.py's cpu_baseline leg and the tests.
 *
 * Build: gcc -O2 -fPIC -shared -pthread dvp_oracle_ossl.c -lcrypto -o _build/libdvp_oracle_ossl.so
 */
#include <openssl/bn.h>
#include <openssl/ec.h>
#include <openssl/obj_mac.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>

typedef uint64_t u64;
typedef struct { const u64 *scalars, *bases; size_t lo, hi; unsigned char out[64]; int inf, ok; } job;

static void* worker(void* arg) {
  job* j = (job*)arg;
  j->ok = 0;
  EC_GROUP* g = EC_GROUP_new_by_curve_name(NID_sect233k1);
  BN_CTX* ctx = BN_CTX_new();
  if (!g || !ctx) return NULL;
  EC_POINT *acc = EC_POINT_new(g), *p = EC_POINT_new(g), *r = EC_POINT_new(g);
  BIGNUM *x = BN_new(), *y = BN_new(), *k = BN_new();
  EC_POINT_set_to_infinity(g, acc);
  int ok = 1;
  for (size_t i = j->lo; i < j->hi && ok; ++i) {
    BN_lebin2bn((const unsigned char*)(j->bases + 8 * i), 32, x);
    BN_lebin2bn((const unsigned char*)(j->bases + 8 * i + 4), 32, y);
    BN_lebin2bn((const unsigned char*)(j->scalars + 4 * i), 32, k);
    ok = EC_POINT_set_affine_coordinates(g, p, x, y, ctx) && EC_POINT_mul(g, r, NULL, p, k, ctx) && EC_POINT_add(g, acc, acc, r, ctx);
  }
  j->inf = EC_POINT_is_at_infinity(g, acc);
  memset(j->out, 0, 64);
  if (ok && !j->inf) {
    ok = EC_POINT_get_affine_coordinates(g, acc, x, y, ctx);
    BN_bn2lebinpad(x, j->out, 32);
    BN_bn2lebinpad(y, j->out + 32, 32);
  }
  j->ok = ok;
  BN_free(x); BN_free(y); BN_free(k);
  EC_POINT_free(acc); EC_POINT_free(p); EC_POINT_free(r);
  BN_CTX_free(ctx);
  EC_GROUP_free(g);
  return NULL;
}

/* sum_i scalars[i] * bases[i] on `threads` threads; the per-thread partial sums are added by thread 0's group.
 * returns 0 on success; out = x || y (little-endian 32 B each), *out_inf = 1 for the neutral element */
int dvo_openssl_msm(const u64* scalars, const u64* bases, size_t n, int threads, unsigned char out[64], int* out_inf) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  static job jobs[256];
  pthread_t th[256];
  size_t per = (n + (size_t)threads - 1) / (size_t)threads;
  for (int t = 0; t < threads; ++t) {
    jobs[t].scalars = scalars; jobs[t].bases = bases;
    jobs[t].lo = (size_t)t * per < n ? (size_t)t * per : n;
    jobs[t].hi = (size_t)(t + 1) * per < n ? (size_t)(t + 1) * per : n;
    pthread_create(&th[t], NULL, worker, &jobs[t]);
  }
  EC_GROUP* g = EC_GROUP_new_by_curve_name(NID_sect233k1);
  BN_CTX* ctx = BN_CTX_new();
  EC_POINT *acc = EC_POINT_new(g), *p = EC_POINT_new(g);
  BIGNUM *x = BN_new(), *y = BN_new();
  EC_POINT_set_to_infinity(g, acc);
  int ok = 1;
  for (int t = 0; t < threads; ++t) {
    pthread_join(th[t], NULL);
    ok = ok && jobs[t].ok;
    if (ok && !jobs[t].inf) {
      BN_lebin2bn(jobs[t].out, 32, x);
      BN_lebin2bn(jobs[t].out + 32, 32, y);
      ok = EC_POINT_set_affine_coordinates(g, p, x, y, ctx) && EC_POINT_add(g, acc, acc, p, ctx);
    }
  }
  *out_inf = EC_POINT_is_at_infinity(g, acc);
  memset(out, 0, 64);
  if (ok && !*out_inf) {
    ok = EC_POINT_get_affine_coordinates(g, acc, x, y, ctx);
    BN_bn2lebinpad(x, out, 32);
    BN_bn2lebinpad(y, out + 32, 32);
  }
  BN_free(x); BN_free(y);
  EC_POINT_free(acc); EC_POINT_free(p);
  BN_CTX_free(ctx);
  EC_GROUP_free(g);
  return ok ? 0 : -1;
}
