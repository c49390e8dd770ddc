// BLAKE3 (hash mode) for the Fiat-Shamir transcript (src/proving.rs:79-198).  Host-only glue,
// written from the BLAKE3 specification; checked against the official test vectors in tests/.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <vector>

namespace dvp {
namespace b3 {

static const uint32_t IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
static const int PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };

static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static inline void g(uint32_t* s, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
  s[a] = s[a] + s[b] + mx; s[d] = rotr(s[d] ^ s[a], 16);
  s[c] = s[c] + s[d];      s[b] = rotr(s[b] ^ s[c], 12);
  s[a] = s[a] + s[b] + my; s[d] = rotr(s[d] ^ s[a], 8);
  s[c] = s[c] + s[d];      s[b] = rotr(s[b] ^ s[c], 7);
}
static inline void compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t block_len,
                            uint32_t flags, uint32_t out[16]) {
  uint32_t s[16], m[16], t[16];
  for (int i = 0; i < 8; ++i) s[i] = cv[i];
  for (int i = 0; i < 4; ++i) s[8 + i] = IV[i];
  s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = block_len; s[15] = flags;
  memcpy(m, block, 64);
  for (int r = 0; r < 7; ++r) {
    g(s, 0, 4, 8, 12, m[0], m[1]); g(s, 1, 5, 9, 13, m[2], m[3]); g(s, 2, 6, 10, 14, m[4], m[5]); g(s, 3, 7, 11, 15, m[6], m[7]);
    g(s, 0, 5, 10, 15, m[8], m[9]); g(s, 1, 6, 11, 12, m[10], m[11]); g(s, 2, 7, 8, 13, m[12], m[13]); g(s, 3, 4, 9, 14, m[14], m[15]);
    if (r < 6) { for (int i = 0; i < 16; ++i) t[i] = m[PERM[i]]; memcpy(m, t, 64); }
  }
  for (int i = 0; i < 8; ++i) { out[i] = s[i] ^ s[i + 8]; out[i + 8] = s[i + 8] ^ cv[i]; }
}
struct Output { uint32_t cv[8]; uint32_t block[16]; uint64_t counter; uint32_t block_len, flags; };
static inline void words_of(const uint8_t* p, size_t len, uint32_t w[16]) {
  uint8_t buf[64] = {0};
  memcpy(buf, p, len);
  for (int i = 0; i < 16; ++i) w[i] = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) | ((uint32_t)buf[4 * i + 2] << 16) | ((uint32_t)buf[4 * i + 3] << 24);
}
static inline Output chunk_output(const uint8_t* chunk, size_t len, uint64_t counter) {
  Output o;
  memcpy(o.cv, IV, 32);
  size_t nblocks = len ? (len + 63) / 64 : 1;
  for (size_t i = 0; i < nblocks; ++i) {
    size_t bl = (i == nblocks - 1) ? len - 64 * i : 64;
    uint32_t flags = (i == 0 ? CHUNK_START : 0) | (i == nblocks - 1 ? CHUNK_END : 0);
    uint32_t w[16];
    words_of(chunk + 64 * i, bl, w);
    if (i == nblocks - 1) { memcpy(o.block, w, 64); o.counter = counter; o.block_len = (uint32_t)bl; o.flags = flags; return o; }
    uint32_t out[16];
    compress(o.cv, w, counter, 64, flags, out);
    memcpy(o.cv, out, 32);
  }
  return o;
}
static inline void hash(const uint8_t* data, size_t len, uint8_t out32[32]) {
  size_t nchunks = len ? (len + 1023) / 1024 : 1;
  std::vector<std::vector<uint32_t>> stack;
  Output o;
  for (size_t c = 0; c < nchunks; ++c) {
    size_t cl = (c == nchunks - 1) ? len - 1024 * c : 1024;
    o = chunk_output(data + 1024 * c, cl, c);
    if (c == nchunks - 1) break;
    uint32_t out[16];
    compress(o.cv, o.block, o.counter, o.block_len, o.flags, out);
    std::vector<uint32_t> cv(out, out + 8);
    size_t t = c + 1;
    while ((t & 1) == 0) {
      uint32_t blk[16];
      memcpy(blk, stack.back().data(), 32);
      memcpy(blk + 8, cv.data(), 32);
      stack.pop_back();
      compress(IV, blk, 0, 64, PARENT, out);
      cv.assign(out, out + 8);
      t >>= 1;
    }
    stack.push_back(cv);
  }
  while (!stack.empty()) {
    uint32_t out[16];
    compress(o.cv, o.block, o.counter, o.block_len, o.flags, out);
    Output p;
    memcpy(p.cv, IV, 32);
    memcpy(p.block, stack.back().data(), 32);
    memcpy(p.block + 8, out, 32);
    stack.pop_back();
    p.counter = 0; p.block_len = 64; p.flags = PARENT;
    o = p;
  }
  uint32_t out[16];
  compress(o.cv, o.block, o.counter, o.block_len, o.flags | ROOT, out);
  for (int i = 0; i < 8; ++i) { out32[4 * i] = (uint8_t)out[i]; out32[4 * i + 1] = (uint8_t)(out[i] >> 8); out32[4 * i + 2] = (uint8_t)(out[i] >> 16); out32[4 * i + 3] = (uint8_t)(out[i] >> 24); }
}

}  // namespace b3
}  // namespace dvp
