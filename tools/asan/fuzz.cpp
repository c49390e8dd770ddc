// Fuzz driver of the host-side parsers under AddressSanitizer + UBSan (make asan; CPU only, never on the GPU box):
//   * dvp_r1cs_dump_sizes / dvp_r1cs_dump_fill on truncated and byte-flipped SP1 dumps (src/gnark_r1cs.rs:121-185)
//   * dvp_fftr_sections / dvp_fftr_read_fr on damaged FFTR tree files (src/tree_io.rs:217-350), incl. nested subtrees
//   * dvp_file_fr_vec_read / dvp_file_point_vec_read / dvp_file_witness_read on damaged vector files (src/io_utils.rs:42-239)
//   * dvp_prover_open_cache_dir on a directory of damaged files (stops at the first GPU entry: tools/asan/stubs.cpp)
// Every call must return a status; the sanitizers abort on any out-of-bounds access, overflowing index computation or leak.
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dvpari.h"
#define FAIL() do { fprintf(stderr, "fuzz check failed at line %d\n", __LINE__); fflush(stdout); abort(); } while (0)

static uint64_t rng_s = 0x5eed0008ull;
static uint64_t rnd() { rng_s += 0x9E3779B97F4A7C15ull; uint64_t z = rng_s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static void put32(std::vector<uint8_t>& b, uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
static void put64(std::vector<uint8_t>& b, uint64_t v) { for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
static void write_file(const std::string& p, const std::vector<uint8_t>& b) { FILE* f = fopen(p.c_str(), "wb"); if (!f) FAIL(); if (!b.empty()) fwrite(b.data(), 1, b.size(), f); fclose(f); }
static std::vector<uint8_t> damage(const std::vector<uint8_t>& good, int it) {
  std::vector<uint8_t> b = good;
  if (it % 3 == 0) b.resize(rnd() % (good.size() + 1));
  else { int k = 1 + (int)(rnd() % 4); while (k-- && !b.empty()) b[rnd() % b.size()] = (uint8_t)rnd(); }
  if (it % 7 == 0 && b.size() >= 8) { size_t o = rnd() % (b.size() - 7); for (int i = 0; i < 8; ++i) b[o + i] = 0xff; }  // huge counts
  return b;
}

static std::vector<uint8_t> good_dump(uint32_t n_coeffs, uint32_t n_rows) {
  std::vector<uint8_t> b;
  put32(b, n_coeffs);
  for (uint32_t i = 0; i < n_coeffs; ++i) for (int j = 0; j < 32; ++j) b.push_back(j < 4 ? 0 : (uint8_t)rnd());
  put32(b, n_rows);
  for (uint32_t r = 0; r < n_rows; ++r) {
    uint32_t c[3] = {(uint32_t)(rnd() % 4), (uint32_t)(rnd() % 3), (uint32_t)(rnd() % 3)};
    for (int k = 0; k < 3; ++k) put32(b, c[k]);
    for (int k = 0; k < 3; ++k) for (uint32_t t = 0; t < c[k]; ++t) { put32(b, (uint32_t)(rnd() % 50)); put32(b, (uint32_t)(rnd() % n_coeffs)); }
  }
  return b;
}
static void fuzz_dump(int iters) {
  const std::vector<uint8_t> good = good_dump(6, 40);
  int ok = 0, bad = 0;
  for (int it = 0; it < iters; ++it) {
    std::vector<uint8_t> b = it ? damage(good, it) : good;
    uint32_t nc = 0, nr = 0, nw = 0;
    uint64_t nnz[3] = {0, 0, 0};
    int rc = dvp_r1cs_dump_sizes(b.data(), b.size(), &nc, &nr, nnz, &nw);
    if (rc != DVP_OK) { ++bad; continue; }
    ++ok;
    std::vector<uint64_t> coeffs((size_t)4 * nc + 4);
    std::vector<uint32_t> rp[3], wi[3], ci[3];
    uint32_t *rpp[3], *wip[3], *cip[3];
    for (int k = 0; k < 3; ++k) { rp[k].resize((size_t)nr + 1); wi[k].resize(nnz[k] + 1); ci[k].resize(nnz[k] + 1); rpp[k] = rp[k].data(); wip[k] = wi[k].data(); cip[k] = ci[k].data(); }
    if (dvp_r1cs_dump_fill(b.data(), b.size(), coeffs.data(), rpp, wip, cip) != DVP_OK) FAIL();  // sizes accepted it: fill must too
    for (int k = 0; k < 3; ++k) if (rp[k][nr] != nnz[k]) FAIL();
  }
  printf("r1cs dump: %d cases, %d accepted, %d rejected\n", iters, ok, bad);
  if (!ok || !bad) FAIL();
}

static std::vector<uint8_t> fr_blob(size_t n) { std::vector<uint8_t> b; put64(b, n); for (size_t i = 0; i < n; ++i) for (int j = 0; j < 29; ++j) b.push_back(j == 28 ? (uint8_t)(rnd() & 0x7f) : (uint8_t)rnd()); return b; }
static std::vector<uint8_t> node(const std::vector<std::pair<uint8_t, std::vector<uint8_t>>>& secs) {  // src/tree_io.rs:144-214: header, metas, payloads
  std::vector<uint8_t> b;
  put32(b, (uint32_t)secs.size());
  put32(b, 0);
  uint64_t off = 8 + 24 * secs.size();
  for (auto& s : secs) { b.push_back(s.first); for (int i = 0; i < 7; ++i) b.push_back(0); put64(b, off); put64(b, s.second.size()); off += s.second.size(); }
  for (auto& s : secs) b.insert(b.end(), s.second.begin(), s.second.end());
  return b;
}
static std::vector<uint8_t> fftr_file(const std::vector<uint8_t>& top) { std::vector<uint8_t> b = {'F', 'F', 'T', 'R', 0, 0, 0, 0}; put64(b, top.size()); b.insert(b.end(), top.begin(), top.end()); return b; }
static void fuzz_fftr(const std::string& dir, int iters) {
  std::vector<uint8_t> inner = node({{0, fr_blob(4)}});
  std::vector<uint8_t> top = node({{0, fr_blob(16)}, {1, fr_blob(28)}, {2, fr_blob(28)}, {12, inner}});
  const std::vector<uint8_t> good = fftr_file(top);
  const std::string path = dir + "/tree";
  int ok = 0, bad = 0;
  for (int it = 0; it < iters; ++it) {
    write_file(path, it ? damage(good, it) : good);
    for (uint32_t depth = 0; depth < 3; ++depth) {
      uint8_t ids[13];
      uint64_t lens[13];
      uint32_t ns = 0;
      int rc = dvp_fftr_sections(path.c_str(), depth, ids, lens, &ns);
      if (rc != DVP_OK) { ++bad; continue; }
      ++ok;
      if (ns > 13) FAIL();
      for (uint32_t s = 0; s < ns; ++s) {
        size_t n = 0;
        if (dvp_fftr_read_fr(path.c_str(), depth, ids[s], nullptr, 0, &n) != DVP_OK) continue;
        if (n > (1u << 20)) continue;  // a damaged count that still fits the file: nothing to read into
        std::vector<uint64_t> out(4 * n + 4);
        (void)dvp_fftr_read_fr(path.c_str(), depth, ids[s], out.data(), n, &n);
      }
    }
  }
  printf("fftr: %d cases x 3 depths, %d accepted, %d rejected\n", iters, ok, bad);
  if (!ok || !bad) FAIL();
}

static void fuzz_vec_files(const std::string& dir, int iters) {
  std::vector<uint8_t> fr = fr_blob(33), pts, wit;
  put64(pts, 21);
  for (int i = 0; i < 21 * 30; ++i) pts.push_back((uint8_t)rnd());
  wit = {0, 0, 0, 9};
  for (int i = 0; i < 9 * 32; ++i) wit.push_back((uint8_t)rnd());
  const std::string pf = dir + "/vec";
  int ok = 0, bad = 0;
  for (int it = 0; it < iters; ++it) {
    const int kind = it % 3;
    const std::vector<uint8_t>& good = kind == 0 ? fr : kind == 1 ? pts : wit;
    write_file(pf, it < 3 ? good : damage(good, it));
    size_t n = 0;
    int rc = kind == 0 ? dvp_file_fr_vec_read(pf.c_str(), nullptr, 0, &n) : kind == 1 ? dvp_file_point_vec_read(pf.c_str(), nullptr, 0, &n)
                                                                                     : dvp_file_witness_read(pf.c_str(), nullptr, 0, &n);
    if (rc != DVP_OK) { ++bad; continue; }
    if (n > (1u << 20)) FAIL();  // a count the file cannot hold must have been refused
    std::vector<uint64_t> out(4 * n + 4);
    std::vector<uint8_t> o8(30 * n + 4);
    rc = kind == 0 ? dvp_file_fr_vec_read(pf.c_str(), out.data(), n, &n) : kind == 1 ? dvp_file_point_vec_read(pf.c_str(), o8.data(), n, &n)
                                                                                    : dvp_file_witness_read(pf.c_str(), out.data(), n, &n);
    if (rc == DVP_OK) ++ok; else ++bad;
  }
  printf("vector files: %d cases, %d accepted, %d rejected\n", iters, ok, bad);
  if (!ok || !bad) FAIL();
}

static void fuzz_cache_dir(const std::string& dir, int iters) {
  const std::vector<uint8_t> good = good_dump(6, 8);
  std::vector<uint8_t> pts;
  put64(pts, 8);
  for (int i = 0; i < 8 * 30; ++i) pts.push_back((uint8_t)rnd());
  static const char* const names[5] = {"g_m", "g_q", "g_k_0", "g_k_1", "g_k_2"};
  int rcs[8] = {0};
  for (int it = 0; it < iters; ++it) {
    write_file(dir + "/r1cs_to_dvsnark", it % 2 ? damage(good, it) : good);
    for (int i = 0; i < 5; ++i) {
      std::vector<uint8_t> p = pts;
      if (i == 4) { p.clear(); put64(p, 16); for (int k = 0; k < 16 * 30; ++k) p.push_back((uint8_t)rnd()); }
      if (i == 0) { p.clear(); put64(p, 50); for (int k = 0; k < 50 * 30; ++k) p.push_back((uint8_t)rnd()); }
      write_file(dir + "/" + names[i], it % 5 == i ? damage(p, it + i) : p);
    }
    dvp_prover* h = nullptr;
    int rc = dvp_prover_open_cache_dir(dir.c_str(), 2, &h);
    if (rc == DVP_OK || h) FAIL();  // no GPU here: the best case ends at the first device entry (DVP_EHIP)
    ++rcs[(-rc) & 7];
  }
  printf("open_cache_dir: %d cases; statuses EINVAL %d EHIP(reached the device entry) %d EIO %d\n", iters, rcs[1], rcs[4], rcs[6]);
  if (!rcs[4]) FAIL();
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  char tmpl[] = "/tmp/dvp_asan_XXXXXX";
  const char* d = mkdtemp(tmpl);
  if (!d) return 1;
  const std::string dir(d);
  fuzz_dump(iters);
  fuzz_fftr(dir, iters / 4);
  fuzz_vec_files(dir, iters / 2);
  fuzz_cache_dir(dir, iters / 20);
  std::string cmd = "rm -rf " + dir;
  (void)!system(cmd.c_str());
  printf("asan fuzz: clean\n");
  return 0;
}
