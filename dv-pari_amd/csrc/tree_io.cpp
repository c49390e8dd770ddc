// FFTR tree files -- the sectioned on-disk container of src/tree_io.rs (host code, no kernels).
//
//   file        b"FFTR\0\0\0\0" || u64-LE total || node                          src/tree_io.rs:1-15,144-165,217-231
//   node        u32-LE section_count || u32 pad || section_count x meta || blobs  src/tree_io.rs:167-214
//   meta        u8 id || 7 pad || u64-LE off || u64-LE len   (off from the start of the node)   :97-118
//   section id  0 f (leaves + layers), 1 recombine_matrices, 2 decompose_matrices, 3 rational_maps, 4..11 the
//               enter/exit tables, 12 = one complete child node (the subtree)      :32-48
//
// The container is pinned by the reference's own source.  The BLOBS are `serialize_compressed` output of types of the
// third-party ecfft / ark-serialize crates (src/tree_io.rs:120-133), which are not in the reference tree; what this
// reader assumes about them is ark-serialize's published layout for a Vec of prime-field elements,
//   u64-LE count || count x 29 bytes little-endian canonical (232-bit modulus -> 29 bytes; the same element format the
//   reference's own Fr-vector files use, src/io_utils.rs:53-54,127),
// with BinaryTree<T> = newtype around Vec<T> (so `f` of an n-leaf tree holds 2n elements, the leaves in its second
// half: FFTree::f.leaves(), src/ec_fft.rs:179-189) and Mat2x2<F> = 4 consecutive elements.  A section whose length
// does not fit that layout is reported as DVP_EIO, never guessed at.
//
// Use: the GPU prover regenerates its twiddles from the curve constants (src/ec_fft.rs:205-229) and reads no tree
// file; this reader exists to CHECK a reference-built tree2n / treen against the regenerated domain
// (dv-pari_amd/tree_io.py: check_tree_file) and to write files in the same container.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/dvpari.h"

namespace {

const uint8_t MAGIC[8] = {'F', 'F', 'T', 'R', 0, 0, 0, 0};
const uint64_t P64[4] = {0x6efb1ad5f173abdfull, 0x00069d5bb915bcd4ull, 0x0000000000000000ull, 0x0000008000000000ull};

struct Map {
  const uint8_t* p = nullptr;
  size_t len = 0;
  int fd = -1;
  ~Map() {
    if (p && len) munmap((void*)p, len);
    if (fd >= 0) close(fd);
  }
  int open_ro(const char* path) {
    fd = ::open(path, O_RDONLY);
    if (fd < 0) return DVP_EIO;
    struct stat st;
    if (fstat(fd, &st) != 0) return DVP_EIO;
    len = (size_t)st.st_size;
    if (!len) return DVP_EIO;
    void* q = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    if (q == MAP_FAILED) { len = 0; return DVP_EIO; }
    p = (const uint8_t*)q;
    return DVP_OK;
  }
};
uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

struct Sec { uint8_t id; uint64_t off, len; };
// the node that starts at `node` (node_len bytes): its section table, every range checked against the node
int parse_node(const uint8_t* node, size_t node_len, std::vector<Sec>* out) {
  if (node_len < 8) return DVP_EIO;
  const uint32_t cnt = rd32(node);
  if (cnt > 13 || (node_len - 8) / 24 < cnt) return DVP_EIO;
  const size_t table_end = 8 + (size_t)24 * cnt;
  out->clear();
  for (uint32_t i = 0; i < cnt; ++i) {
    const uint8_t* m = node + 8 + 24 * (size_t)i;
    Sec s{m[0], rd64(m + 8), rd64(m + 16)};
    if (s.id > 12) return DVP_EIO;                                          // "unknown section id", :50-70
    if (s.off < table_end || s.off > node_len || s.len > node_len - s.off) return DVP_EIO;
    for (const Sec& t : *out)
      if (t.id == s.id) return DVP_EIO;
    out->push_back(s);
  }
  return DVP_OK;
}
// walks `depth` subtree links down from the root node
int find_node(const Map& f, uint32_t depth, const uint8_t** node, size_t* node_len, std::vector<Sec>* secs) {
  if (f.len < 16 || memcmp(f.p, MAGIC, 8) != 0) return DVP_EIO;  // "not an FFTR file", :225
  const uint64_t total = rd64(f.p + 8);
  if (total > f.len - 16) return DVP_EIO;
  *node = f.p + 16;
  *node_len = (size_t)total;
  for (uint32_t d = 0;; ++d) {
    int rc = parse_node(*node, *node_len, secs);
    if (rc) return rc;
    if (d == depth) return DVP_OK;
    const Sec* sub = nullptr;
    for (const Sec& s : *secs)
      if (s.id == 12) sub = &s;
    if (!sub) return DVP_EINVAL;  // the tree is not that deep
    *node += sub->off;
    *node_len = (size_t)sub->len;
  }
}
bool canonical29(const uint8_t* e) {  // value < p ?
  uint64_t v[4] = {0, 0, 0, 0};
  memcpy(v, e, 29);
  for (int i = 3; i >= 0; --i)
    if (v[i] != P64[i]) return v[i] < P64[i];
  return false;
}

}  // namespace

// Sections of the node `depth` subtree links below the root: ids[k], lens[k] (blob bytes), k < *n_sections (<= 13).
extern "C" int dvp_fftr_sections(const char* path, uint32_t depth, uint8_t ids[13], uint64_t lens[13], uint32_t* n_sections) {
  if (!path || !ids || !lens || !n_sections) return DVP_EINVAL;
  Map f;
  int rc = f.open_ro(path);
  if (rc) return rc;
  const uint8_t* node;
  size_t node_len;
  std::vector<Sec> secs;
  rc = find_node(f, depth, &node, &node_len, &secs);
  if (rc) return rc;
  *n_sections = (uint32_t)secs.size();
  for (size_t k = 0; k < secs.size(); ++k) { ids[k] = secs[k].id; lens[k] = secs[k].len; }
  return DVP_OK;
}

// Field elements of section `id` (0, 1, 2, 4..11) of that node as canonical 4 x u64 limbs.  out == NULL: only *n (elements;
// a matrix section holds 4 per matrix).  A blob that is not `u64 count || count x elt x 29 B` or holds a value >= p
// -> DVP_EIO.
extern "C" int dvp_fftr_read_fr(const char* path, uint32_t depth, uint8_t id, uint64_t* out, size_t cap, size_t* n) {
  if (!path || !n || id == 3 || id > 11) return DVP_EINVAL;
  Map f;
  int rc = f.open_ro(path);
  if (rc) return rc;
  const uint8_t* node;
  size_t node_len;
  std::vector<Sec> secs;
  rc = find_node(f, depth, &node, &node_len, &secs);
  if (rc) return rc;
  const Sec* s = nullptr;
  for (const Sec& t : secs)
    if (t.id == id) s = &t;
  if (!s) return DVP_EINVAL;  // "missing section"
  if (s->len < 8) return DVP_EIO;
  const uint8_t* blob = node + s->off;
  const uint64_t cnt = rd64(blob);
  const uint64_t per = (id == 1 || id == 2) ? 4 : 1;
  if (cnt > (s->len - 8) / (29 * per) || 8 + cnt * 29 * per != s->len) return DVP_EIO;
  *n = (size_t)(cnt * per);
  if (!out) return DVP_OK;
  if (cap < *n) return DVP_EINVAL;
  for (size_t i = 0; i < *n; ++i) {
    const uint8_t* e = blob + 8 + 29 * i;
    if (!canonical29(e)) return DVP_EIO;
    uint64_t* o = out + 4 * i;
    o[0] = o[1] = o[2] = o[3] = 0;
    memcpy(o, e, 29);
  }
  return DVP_OK;
}

// Writes an FFTR file with ONE node (what read_minimal_fftree_from_file needs is sections 0, 1, 2, src/tree_io.rs:353-433):
// section k has id ids[k] and elems[k] canonical field elements at data[k] (a matrix section must hold a multiple of 4).
extern "C" int dvp_fftr_write(const char* path, uint32_t n_sections, const uint8_t* ids, const uint64_t* const* data, const uint64_t* elems) {
  if (!path || !n_sections || n_sections > 12 || !ids || !data || !elems) return DVP_EINVAL;
  std::vector<uint64_t> lens(n_sections);
  uint64_t cur = 8 + 24 * (uint64_t)n_sections, total = cur;
  for (uint32_t k = 0; k < n_sections; ++k) {
    if (ids[k] == 3 || ids[k] > 11 || (elems[k] && !data[k])) return DVP_EINVAL;
    const uint64_t per = (ids[k] == 1 || ids[k] == 2) ? 4 : 1;
    if (elems[k] % per) return DVP_EINVAL;
    for (uint32_t j = 0; j < k; ++j)
      if (ids[j] == ids[k]) return DVP_EINVAL;
    lens[k] = 8 + 29 * elems[k];
    total += lens[k];
  }
  FILE* f = fopen(path, "wb");
  if (!f) return DVP_EIO;
  bool ok = fwrite(MAGIC, 8, 1, f) == 1 && fwrite(&total, 8, 1, f) == 1;
  uint32_t hdr[2] = {n_sections, 0};
  ok = ok && fwrite(hdr, 8, 1, f) == 1;
  for (uint32_t k = 0; ok && k < n_sections; ++k) {
    uint8_t m[24] = {0};
    m[0] = ids[k];
    memcpy(m + 8, &cur, 8);
    memcpy(m + 16, &lens[k], 8);
    ok = fwrite(m, 24, 1, f) == 1;
    cur += lens[k];
  }
  std::vector<uint8_t> buf;
  for (uint32_t k = 0; ok && k < n_sections; ++k) {
    const uint64_t per = (ids[k] == 1 || ids[k] == 2) ? 4 : 1;
    const uint64_t cnt = elems[k] / per;
    buf.resize(8 + 29 * (size_t)elems[k]);
    memcpy(buf.data(), &cnt, 8);
    for (uint64_t i = 0; i < elems[k]; ++i) {
      const uint64_t* v = data[k] + 4 * i;
      uint8_t tmp[32];
      memcpy(tmp, v, 32);
      if (!canonical29(tmp) || tmp[29] || tmp[30] || tmp[31]) { ok = false; break; }
      memcpy(buf.data() + 8 + 29 * i, tmp, 29);
    }
    ok = ok && fwrite(buf.data(), buf.size(), 1, f) == 1;
  }
  ok = (fclose(f) == 0) && ok;
  return ok ? DVP_OK : DVP_EIO;
}
