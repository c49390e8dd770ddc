// The reference's on-disk formats and the cache-dir flavour of Proof::prove (host code, no kernels).
//
//   Fr vector file      u64-LE n || n x 29 B canonical LE      src/io_utils.rs:42-70,122-179
//   point vector file   u64-LE n || n x 30 B xsk233 encoding   src/io_utils.rs:83-111,187-239
//   witness file        u32-BE n || n x 32 B big-endian        src/gnark_r1cs.rs:58-77,188-198
//   R1CS dump           u32-LE nCoeffs || nCoeffs x 32 B BE || u32-LE nRows || rows of
//                       (nL, nR, nO: u32-LE) + (wire_id, coeff_id: u32-LE) terms
//                                                              src/gnark_r1cs.rs:84-91,121-185
//   file names          src/artifacts.rs:18-83
//
// dvp_prover_open_cache_dir is the loading half of Proof::prove (src/proving.rs:435-470,509-511,666-672): the R1CS
// dump and the five SRS vectors are parsed / decoded ONCE and stay in HBM; dvp_prove_cache_dir keeps the opened
// contexts in a process-wide table so that repeated prove(cache_dir, ..) calls only pay for the proof.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <tuple>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/dvpari.h"
#include "../../include/dvpari_internal.h"  // dvp_tune_get (DVP_CACHE_REPLICAS)

namespace {

// p, little-endian 64-bit limbs (src/curve.rs:17)
const uint64_t P64[4] = {0x6efb1ad5f173abdfull, 0x00069d5bb915bcd4ull, 0x0000000000000000ull, 0x0000008000000000ull};

struct U256 {
  uint64_t w[4];
};
inline bool geq(const U256& a, const U256& b) {
  for (int i = 3; i >= 0; --i)
    if (a.w[i] != b.w[i]) return a.w[i] > b.w[i];
  return true;
}
inline void sub(U256& a, const U256& b) {
  uint64_t br = 0;
  for (int i = 0; i < 4; ++i) {
    uint64_t d = a.w[i] - b.w[i], d2 = d - br;
    br = (uint64_t)(a.w[i] < b.w[i]) | (uint64_t)(d < br);
    a.w[i] = d2;
  }
}
inline U256 shl(const U256& a, int k) {  // k < 64
  if (k == 0) return a;
  U256 r;
  for (int i = 3; i >= 0; --i) r.w[i] = (a.w[i] << k) | (i ? a.w[i - 1] >> (64 - k) : 0);
  return r;
}
// from_be_bytes_mod_order on 32 bytes: x < 2^256, p has 232 bits -> 25 conditional subtractions
inline void be32_mod_p(const uint8_t* be, uint64_t out[4]) {
  U256 x;
  for (int i = 0; i < 4; ++i) {
    uint64_t v = 0;
    for (int j = 0; j < 8; ++j) v = (v << 8) | be[8 * (3 - i) + j];
    x.w[i] = v;
  }
  U256 p;
  memcpy(p.w, P64, 32);
  for (int k = 24; k >= 0; --k) {
    U256 s = shl(p, k);
    if (geq(x, s)) sub(x, s);
  }
  memcpy(out, x.w, 32);
}
inline bool canonical(const uint64_t v[4]) {
  U256 a, p;
  memcpy(a.w, v, 32);
  memcpy(p.w, P64, 32);
  return !geq(a, p);
}

struct Mapped {
  const uint8_t* p = nullptr;
  size_t len = 0;
  int fd = -1;
  ~Mapped() {
    if (p && len) munmap((void*)p, len);
    if (fd >= 0) close(fd);
  }
  int open_ro(const char* path) {
    fd = ::open(path, O_RDONLY);
    if (fd < 0) return DVP_EIO;
    struct stat st;
    if (fstat(fd, &st) != 0) return DVP_EIO;
    len = (size_t)st.st_size;
    if (len == 0) return DVP_OK;
    void* q = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    if (q == MAP_FAILED) {
      len = 0;
      return DVP_EIO;
    }
    p = (const uint8_t*)q;
    return DVP_OK;
  }
};

inline uint64_t rd64le(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}
inline uint32_t rd32le(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}

// u64-LE count || count x elt bytes: returns the payload
int counted_payload(const Mapped& f, size_t elt, const uint8_t** payload, size_t* n) {
  if (f.len < 8) return DVP_EIO;  // "File too short for length prefix"
  uint64_t cnt = rd64le(f.p);
  if (cnt > (f.len - 8) / elt) return DVP_EIO;  // "File too short for expected point data"
  *payload = f.p + 8;
  *n = (size_t)cnt;
  return DVP_OK;
}

int write_counted(const char* path, const uint8_t* payload, size_t n, size_t elt, const std::vector<uint8_t>* packed) {
  FILE* f = fopen(path, "wb");
  if (!f) return DVP_EIO;
  uint64_t cnt = n;
  bool ok = fwrite(&cnt, 8, 1, f) == 1;
  const uint8_t* src = packed ? packed->data() : payload;
  if (ok && n) ok = fwrite(src, elt, n, f) == n;
  ok = (fclose(f) == 0) && ok;
  return ok ? DVP_OK : DVP_EIO;
}

struct DumpView {
  uint32_t n_coeffs = 0, n_rows = 0, n_wires = 0;
  uint64_t nnz[3] = {0, 0, 0};
  const uint8_t* coeffs = nullptr;
  const uint8_t* rows = nullptr;  // first row header
};

// one validating pass over the dump: sizes, and max wire id + 1 (accumulate_m_values, src/srs.rs:56-62)
int dump_scan(const uint8_t* buf, size_t len, DumpView* v) {
  if (!buf || len < 8) return DVP_EIO;
  size_t off = 0;
  v->n_coeffs = rd32le(buf);
  off = 4;
  if ((len - off) / 32 < v->n_coeffs) return DVP_EIO;
  v->coeffs = buf + off;
  off += (size_t)32 * v->n_coeffs;
  if (len - off < 4) return DVP_EIO;
  v->n_rows = rd32le(buf + off);
  off += 4;
  v->rows = buf + off;
  uint32_t maxw = 0;
  bool any = false;
  for (uint32_t r = 0; r < v->n_rows; ++r) {
    if (len - off < 12) return DVP_EIO;
    uint32_t cnt[3] = {rd32le(buf + off), rd32le(buf + off + 4), rd32le(buf + off + 8)};
    off += 12;
    for (int k = 0; k < 3; ++k) {
      if ((len - off) / 8 < cnt[k]) return DVP_EIO;
      for (uint32_t t = 0; t < cnt[k]; ++t) {
        uint32_t w = rd32le(buf + off + 8 * (size_t)t), c = rd32le(buf + off + 8 * (size_t)t + 4);
        if (c >= v->n_coeffs) return DVP_EINVAL;  // the reference would index out of bounds and panic
        if (w > maxw) maxw = w;
        any = true;
      }
      v->nnz[k] += cnt[k];
      off += (size_t)8 * cnt[k];
    }
  }
  v->n_wires = any ? maxw + 1 : 1;
  return DVP_OK;
}

void dump_fill(const DumpView& v, uint64_t* coeffs, uint32_t* const row_ptr[3], uint32_t* const wire[3], uint32_t* const coeff[3]) {
  for (uint32_t i = 0; i < v.n_coeffs; ++i) be32_mod_p(v.coeffs + 32 * (size_t)i, coeffs + 4 * (size_t)i);
  const uint8_t* q = v.rows;
  uint32_t pos[3] = {0, 0, 0};
  for (int k = 0; k < 3; ++k) row_ptr[k][0] = 0;
  for (uint32_t r = 0; r < v.n_rows; ++r) {
    uint32_t cnt[3] = {rd32le(q), rd32le(q + 4), rd32le(q + 8)};
    q += 12;
    for (int k = 0; k < 3; ++k) {
      for (uint32_t t = 0; t < cnt[k]; ++t) {
        wire[k][pos[k]] = rd32le(q);
        coeff[k][pos[k]] = rd32le(q + 4);
        ++pos[k];
        q += 8;
      }
      row_ptr[k][r + 1] = pos[k];
    }
  }
}

std::string join(const char* dir, const char* name) {
  std::string s(dir);
  if (!s.empty() && s.back() != '/') s.push_back('/');
  return s + name;
}

// Provers opened by dvp_prove_cache_dir, keyed by (cache_dir, n_public, HIP device).  An entry is shared: a caller holds
// a reference for the whole of its prove, which runs under the entry's own mutex (a dvp_prover is one set of device
// buffers: two proofs on it must not overlap), and dvp_cache_dir_release only drops the table's reference, so the
// prover is destroyed when the last prove in flight returns.
struct OpenEntry {
  dvp_prover* p = nullptr;
  std::mutex mu;                    // guards the opening of `p`
  std::atomic<bool> opened{false};  // the files have been read (under mu); rc then holds the result
  int rc = DVP_OK;
  // A second prover over the same files, opened the first time a proof arrives while the first prover is busy: two host
  // threads calling dvp_prove_cache_dir on one cache_dir then have two proofs in flight on the GPU (+11 % constraints/s at
  // 2^20) instead of taking turns.  Costs one more decode of the SRS files and one more set of tables,
  // so it is opened only when (i) the first prover has finished a proof (its tables and MSM workspaces exist: what is in use
  // on the device is then a fair measure of one prover) and (ii) free HBM exceeds 1.25 x what is in use -- a replica must never
  // make the first prover's own allocations fail.  Callers take whichever of the two provers frees first (slot_cv).
  dvp_prover* p2 = nullptr;
  bool opened2 = false, opening2 = false;
  int rc2 = DVP_OK;
  std::mutex slot_mu;
  std::condition_variable slot_cv;
  bool busy[2] = {false, false};
  uint64_t proofs_done = 0;
  uint64_t room_retry_at = 1;  // a second prover is considered once proofs_done reaches this (first: after one completed proof)
  ~OpenEntry() {
    if (p) dvp_prover_destroy(p);
    if (p2) dvp_prover_destroy(p2);
  }
};
typedef std::tuple<std::string, uint32_t, int> OpenKey;
std::mutex g_open_mu;
std::map<OpenKey, std::shared_ptr<OpenEntry>> g_open;

int current_device() {
  int d = -1;
  (void)hipGetDevice(&d);
  return d;
}
// The table lock only covers the lookup / insertion of a placeholder; reading the files, decoding the points and building
// the tables (seconds at full size) happen under the ENTRY's mutex, so opening one cache_dir neither blocks proofs on
// another one (or on another device) nor dvp_cache_dir_release.  A failed open leaves no entry behind.
int open_entry(const char* cache_dir, uint32_t n_public, std::shared_ptr<OpenEntry>* out) {
  OpenKey key(std::string(cache_dir), n_public, current_device());
  std::shared_ptr<OpenEntry> e;
  {
    std::lock_guard<std::mutex> g(g_open_mu);
    auto it = g_open.find(key);
    if (it == g_open.end()) it = g_open.emplace(key, std::make_shared<OpenEntry>()).first;
    e = it->second;
  }
  if (!e->opened.load(std::memory_order_acquire)) {  // (an opened entry is not locked here: its mutex is held for whole proofs)
    std::lock_guard<std::mutex> ge(e->mu);
    if (!e->opened.load(std::memory_order_relaxed)) {
      e->rc = dvp_prover_open_cache_dir(cache_dir, n_public, &e->p);
      e->opened.store(true, std::memory_order_release);
    }
  }
  if (e->rc != DVP_OK) {
    std::lock_guard<std::mutex> g(g_open_mu);
    auto it = g_open.find(key);
    if (it != g_open.end() && it->second == e) g_open.erase(it);
    return e->rc;
  }
  *out = e;
  return DVP_OK;
}

}  // namespace

extern "C" int dvp_file_fr_vec_write(const char* path, const uint64_t* limbs, size_t n) {
  if (!path || (!limbs && n)) return DVP_EINVAL;
  std::vector<uint8_t> packed(n * 29);
  for (size_t i = 0; i < n; ++i) {
    if (!canonical(limbs + 4 * i)) return DVP_EINVAL;
    memcpy(packed.data() + 29 * i, limbs + 4 * i, 29);  // little-endian host: the low 29 bytes of the 32
  }
  return write_counted(path, nullptr, n, 29, &packed);
}

extern "C" int dvp_file_fr_vec_read(const char* path, uint64_t* out, size_t cap, size_t* n) {
  if (!path || !n) return DVP_EINVAL;
  Mapped f;
  int rc = f.open_ro(path);
  if (rc) return rc;
  const uint8_t* pl;
  rc = counted_payload(f, 29, &pl, n);
  if (rc) return rc;
  if (!out) return DVP_OK;
  if (cap < *n) return DVP_EINVAL;
  for (size_t i = 0; i < *n; ++i) {
    uint64_t v[4] = {0, 0, 0, 0};
    memcpy(v, pl + 29 * i, 29);
    // deserialize_uncompressed_unchecked (src/io_utils.rs:160) does not range-check; this boundary carries
    // canonical values only, so a value >= p is rejected instead of being passed on
    if (!canonical(v)) return DVP_EINVAL;
    memcpy(out + 4 * i, v, 32);
  }
  return DVP_OK;
}

extern "C" int dvp_file_point_vec_write(const char* path, const uint8_t* enc, size_t n) {
  if (!path || (!enc && n)) return DVP_EINVAL;
  return write_counted(path, enc, n, 30, nullptr);
}

extern "C" int dvp_file_point_vec_read(const char* path, uint8_t* out, size_t cap, size_t* n) {
  if (!path || !n) return DVP_EINVAL;
  Mapped f;
  int rc = f.open_ro(path);
  if (rc) return rc;
  const uint8_t* pl;
  rc = counted_payload(f, 30, &pl, n);
  if (rc) return rc;
  if (!out) return DVP_OK;
  if (cap < *n) return DVP_EINVAL;
  memcpy(out, pl, 30 * *n);
  return DVP_OK;
}

extern "C" int dvp_file_witness_read(const char* path, uint64_t* out, size_t cap, size_t* n) {
  if (!path || !n) return DVP_EINVAL;
  Mapped f;
  int rc = f.open_ro(path);
  if (rc) return rc;
  if (f.len < 4) return DVP_EIO;
  uint32_t cnt = ((uint32_t)f.p[0] << 24) | ((uint32_t)f.p[1] << 16) | ((uint32_t)f.p[2] << 8) | f.p[3];
  if ((f.len - 4) / 32 < cnt) return DVP_EIO;
  *n = cnt;
  if (!out) return DVP_OK;
  if (cap < cnt) return DVP_EINVAL;
  for (uint32_t i = 0; i < cnt; ++i) be32_mod_p(f.p + 4 + 32 * (size_t)i, out + 4 * (size_t)i);  // gnark_element_to_fr
  return DVP_OK;
}

extern "C" int dvp_file_witness_write(const char* path, const uint64_t* limbs, size_t n) {
  if (!path || (!limbs && n) || n > 0xffffffffull) return DVP_EINVAL;
  FILE* f = fopen(path, "wb");
  if (!f) return DVP_EIO;
  uint8_t hdr[4] = {(uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n};
  bool ok = fwrite(hdr, 4, 1, f) == 1;
  for (size_t i = 0; ok && i < n; ++i) {
    uint8_t be[32];
    for (int j = 0; j < 32; ++j) be[j] = (uint8_t)(limbs[4 * i + (31 - j) / 8] >> (8 * ((31 - j) % 8)));
    ok = fwrite(be, 32, 1, f) == 1;
  }
  ok = (fclose(f) == 0) && ok;
  return ok ? DVP_OK : DVP_EIO;
}

extern "C" int dvp_r1cs_dump_sizes(const uint8_t* buf, size_t len, uint32_t* n_coeffs, uint32_t* n_rows, uint64_t nnz[3],
                                   uint32_t* n_wires) {
  if (!buf || !n_coeffs || !n_rows || !nnz || !n_wires) return DVP_EINVAL;
  DumpView v;
  int rc = dump_scan(buf, len, &v);
  if (rc) return rc;
  *n_coeffs = v.n_coeffs;
  *n_rows = v.n_rows;
  *n_wires = v.n_wires;
  for (int k = 0; k < 3; ++k) nnz[k] = v.nnz[k];
  return DVP_OK;
}

extern "C" int dvp_r1cs_dump_fill(const uint8_t* buf, size_t len, uint64_t* coeffs, uint32_t* const row_ptr[3], uint32_t* const wire[3],
                                  uint32_t* const coeff_id[3]) {
  if (!buf || !coeffs || !row_ptr || !wire || !coeff_id) return DVP_EINVAL;
  for (int k = 0; k < 3; ++k)
    if (!row_ptr[k] || !wire[k] || !coeff_id[k]) return DVP_EINVAL;
  DumpView v;
  int rc = dump_scan(buf, len, &v);
  if (rc) return rc;
  for (int k = 0; k < 3; ++k)
    if (v.nnz[k] > 0xffffffffull) return DVP_EINVAL;
  dump_fill(v, coeffs, row_ptr, wire, coeff_id);
  return DVP_OK;
}

extern "C" int dvp_prover_open_cache_dir(const char* cache_dir, uint32_t n_public, dvp_prover** out) {
  if (!cache_dir || !out) return DVP_EINVAL;
  *out = nullptr;
  Mapped dump;
  int rc = dump.open_ro(join(cache_dir, "r1cs_to_dvsnark").c_str());  // R1CS_CONSTRAINTS_FILE, src/artifacts.rs:76
  if (rc) return rc;
  DumpView v;
  rc = dump_scan(dump.p, dump.len, &v);
  if (rc) return rc;
  if (v.n_rows == 0) return DVP_EINVAL;
  for (int k = 0; k < 3; ++k)
    if (v.nnz[k] > 0xffffffffull) return DVP_EINVAL;
  // the witness length is fixed by the commitment key: multi_scalar_mul asserts |assignment| == |g_m| (src/curve.rs:142)
  static const char* const names[5] = {"g_m", "g_q", "g_k_0", "g_k_1", "g_k_2"};  // src/artifacts.rs:18-27
  Mapped srs[5];
  const uint8_t* payload[5];
  size_t cnt[5];
  for (int i = 0; i < 5; ++i) {
    rc = srs[i].open_ro(join(cache_dir, names[i]).c_str());
    if (rc) return rc;
    rc = counted_payload(srs[i], 30, &payload[i], &cnt[i]);
    if (rc) return rc;
  }
  uint32_t log_m = 0;
  while ((1ull << log_m) < v.n_rows) ++log_m;  // next_power_of_two, src/gnark_r1cs.rs:291
  if (log_m < 1) log_m = 1;
  const size_t m = (size_t)1 << log_m;
  if (cnt[0] < v.n_wires || cnt[0] < 1 + (size_t)n_public || cnt[0] > 0xffffffffull) return DVP_EINVAL;
  if (cnt[1] != m || cnt[2] != m || cnt[3] != m || cnt[4] != 2 * m) return DVP_EINVAL;
  std::vector<uint64_t> coeffs((size_t)4 * v.n_coeffs);
  std::vector<uint32_t> rp[3], wi[3], ci[3];
  uint32_t *rpp[3], *wip[3], *cip[3];
  for (int k = 0; k < 3; ++k) {
    rp[k].resize((size_t)v.n_rows + 1);
    wi[k].resize(v.nnz[k] ? v.nnz[k] : 1);
    ci[k].resize(v.nnz[k] ? v.nnz[k] : 1);
    rpp[k] = rp[k].data();
    wip[k] = wi[k].data();
    cip[k] = ci[k].data();
  }
  dump_fill(v, coeffs.data(), rpp, wip, cip);
  dvp_prover* p = nullptr;
  rc = dvp_prover_create(log_m, n_public, (uint32_t)cnt[0], &p);
  if (rc) return rc;
  rc = dvp_prover_set_coeffs(p, coeffs.data(), v.n_coeffs);
  for (int k = 0; k < 3 && !rc; ++k) rc = dvp_prover_set_matrix(p, k, v.n_rows, rpp[k], wip[k], cip[k]);
  for (int i = 0; i < 5 && !rc; ++i) rc = dvp_prover_set_srs_encoded(p, i, payload[i], cnt[i]);
  if (rc) {
    dvp_prover_destroy(p);
    return rc;
  }
  *out = p;
  return DVP_OK;
}

extern "C" int dvp_prove_cache_dir(const char* cache_dir, const uint64_t* public_inputs, uint32_t n_public,
                                   const uint64_t* private_inputs, uint32_t n_private, uint8_t proof[118]) {
  if (!cache_dir || !proof) return DVP_EINVAL;
  std::shared_ptr<OpenEntry> e;
  int rc = open_entry(cache_dir, n_public, &e);
  if (rc) return rc;
  long long replicas = 1;
  (void)dvp_tune_get("DVP_CACHE_REPLICAS", &replicas);
  int slot = -1;
  {
    std::unique_lock<std::mutex> g(e->slot_mu);
    for (;;) {
      if (!e->busy[0]) { slot = 0; break; }
      if (replicas >= 2 && e->opened2 && e->rc2 == DVP_OK && !e->busy[1]) { slot = 1; break; }
      if (replicas >= 2 && !e->opened2 && !e->opening2 && e->proofs_done >= e->room_retry_at) {
        // the room check (device-wide, other tenants included) and the open itself run OUTSIDE slot_mu, which every release needs;
        // a failed check is not latched: it is asked again after another 16 completed proofs (memory may have come back)
        e->opening2 = true;
        g.unlock();
        size_t free_b = 0, total_b = 0;
        const bool room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)free_b > 1.25 * (double)(total_b - free_b);
        dvp_prover* q = nullptr;
        const int r2 = room ? dvp_prover_open_cache_dir(cache_dir, n_public, &q) : DVP_ENOMEM;
        g.lock();
        e->opening2 = false;
        if (room) {
          e->p2 = q;
          e->rc2 = r2;
          e->opened2 = true;
        } else {
          e->room_retry_at = e->proofs_done + 16;
        }
        e->slot_cv.notify_all();
        continue;
      }
      e->slot_cv.wait(g);
    }
    e->busy[slot] = true;
  }
  rc = dvp_prove(slot == 0 ? e->p : e->p2, public_inputs, n_public, private_inputs, n_private, proof);
  {
    std::lock_guard<std::mutex> g(e->slot_mu);
    e->busy[slot] = false;
    if (rc == DVP_OK) ++e->proofs_done;
  }
  e->slot_cv.notify_one();
  return rc;
}

// the prover dvp_prove_cache_dir uses for (cache_dir, n_public) on the current device, opened if need be: BORROWED --
// valid until dvp_cache_dir_release; for inspection (dvp_prover_debug_read, dvp_prover_msm_plan), not for concurrent proving
extern "C" int dvp_cache_dir_prover(const char* cache_dir, uint32_t n_public, dvp_prover** out) {
  if (!cache_dir || !out) return DVP_EINVAL;
  std::shared_ptr<OpenEntry> e;
  int rc = open_entry(cache_dir, n_public, &e);
  if (rc) return rc;
  *out = e->p;
  return DVP_OK;
}

extern "C" void dvp_cache_dir_release(const char* cache_dir) {
  std::lock_guard<std::mutex> g(g_open_mu);
  for (auto it = g_open.begin(); it != g_open.end();) {
    if (!cache_dir || std::get<0>(it->first) == cache_dir)
      it = g_open.erase(it);  // the prover goes when the last prove holding the entry returns
    else
      ++it;
  }
}
