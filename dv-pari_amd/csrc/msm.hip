// Multi-scalar multiplication on sect233k1 for gfx950 -- replaces multi_scalar_mul
// (src/curve.rs:141-158; call sites src/proving.rs:463,512,680), which in the reference is n
// independent xsk233_mul_frob calls plus an add tree.  Here: a Koblitz-aware bucket method.
//
//  1. recode   each scalar s (mod r) is reduced modulo delta = (tau^233-1)/(tau-1) in Z[tau]
//              (Solinas partial reduction) and expanded in base tau with digits {0,1}
//              (tau^2 + tau + 2 = 0 is a canonical number system), <= 240 digits.  A window is c
//              consecutive digits; window w, pattern d  <->  bucket key w*2^c + d.
//              Because the window multipliers are powers of tau, the final combination uses the
//              Frobenius (3 squarings) where an integer-window Pippenger needs doublings -- on a
//              GPU that turns a ~230-step serial doubling chain into a log-depth tree.
//              Fixed-base contexts (the prover's SRS vectors) pre-rotate the bases so that all windows share ONE bucket
//              set (MsmFixedCtx; the default flavour cuts SIGNED windows from the scalar's binary digits instead,
//              k_recode_signed).
//  2. sort     counting sort of (key, point index): histogram (atomics) -> scan -> scatter.
//  3. reduce   segmented sum per key by fixed fan-in K: level 1 gathers affine bases and does mixed
//              Lopez-Dahab additions; further levels add projective partials.  Tasks never span
//              keys, so skewed scalars (many equal digits) stay load-balanced.
//  4. merge    per window, D_t = sum of buckets whose pattern has bit t, by a pruned
//              sum-over-subsets tree (2*2^c additions per window, depth c).
//  5. tail     result = sum_j tau^j(D_j): per-point Frobenius powers, then a pairwise add tree.
//
// All of it runs on the VALU (no MFMA: there is no dense contraction in this algorithm, and
// gfx950 has no carry-less multiplier -- see gf233.cuh).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <mutex>
#include <type_traits>
#include <vector>

#include "common.h"
#include "k233.cuh"
#include "tau.cuh"
#include "codec.cuh"

namespace dvp {

// Entry word of the fixed-base mode (one u32 per (slot, scalar)): bit 31 = valid, bits 20..27 = row of the pre-rotated
// table (the power of tau the base is taken at), bits 0..19 = bucket key.  0 = empty slot.
// Bit 28 = the entry SUBTRACTS its point (signed-digit flavours): carried into bit 31 of the sorted item (ITEM_NEG) and applied
// where the point is loaded (-(x, y) = (x, x + y): free).
constexpr uint32_t FXW_VALID = 0x80000000u, FXW_KEY_MASK = 0xfffffu, FXW_NEG = 0x10000000u, ITEM_NEG = 0x80000000u;
constexpr int FXW_ROW_SHIFT = 20;
__host__ __device__ __forceinline__ uint32_t fxw_key(uint32_t d) { return d & FXW_KEY_MASK; }
__host__ __device__ __forceinline__ uint32_t fxw_row(uint32_t d) { return (d >> FXW_ROW_SHIFT) & 0xffu; }

// One thread per scalar: digits[w][i] (c-bit patterns) + bucket histogram.
// Scalars >= r are rejected (flag); points flagged infinite contribute nothing.
template <class DIGIT>
__global__ void __launch_bounds__(256)
k_recode(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf, uint32_t n, int c, int W, int n_narrow,
         DIGIT* __restrict__ digits, unsigned long long* __restrict__ err) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = scalars[(size_t)i * 8 + k];
  bool skip = inf && inf[i];
  if (!tau_scalar_is_canonical(s)) {
    atomicMin(err, (unsigned long long)i);
    skip = true;
  }
  uint32_t r0[5], r1[5];
  tau_partial_reduce(s, r0, r1);
  if (skip) {
#pragma unroll
    for (int k = 0; k < 5; ++k) r0[k] = r1[k] = 0;
  }
  // expansion: u = r0 & 1; r0 -= u; h = r0 >> 1 (arithmetic); (r0, r1) = (r1 - h, -h)
  // windows 0 .. n_narrow-1 are c-1 digits wide, the rest c (fixed-base mode evens the windows out, see MsmFixedCtx).
  // Digits arrive sixteen at a time (tau_step16) into a 64-bit shift register; windows are cut from its low end.
  const int total = W * c - n_narrow;
  uint64_t buf = 0;
  int nb = 0, w = 0, width = n_narrow > 0 ? c - 1 : c;
  for (int done = 0; done < total; done += 16) {
    buf |= (uint64_t)tau_step16(r0, r1) << nb;
    nb += 16;
    while (nb >= width && w < W) {
      const uint32_t dv = (uint32_t)(buf & ((1ull << width) - 1));
      if (sizeof(DIGIT) == 4)  // fixed-base mode: entry word = valid | table row | bucket key (see FXW_*)
        digits[(size_t)w * n + i] = (DIGIT)(dv ? (FXW_VALID | ((uint32_t)w << FXW_ROW_SHIFT) | dv) : 0u);
      else
        digits[(size_t)w * n + i] = (DIGIT)dv;
      buf >>= width;
      nb -= width;
      ++w;
      width = w < n_narrow ? c - 1 : c;
    }
  }
  uint32_t rest = (uint32_t)buf | (uint32_t)(buf >> 32);  // digits beyond the last window
#pragma unroll
  for (int k = 0; k < 5; ++k) rest |= r0[k] | r1[k];
  if (rest) atomicMin(err, (unsigned long long)i | (1ull << 62));  // expansion longer than W*c digits
}

// Signed aligned windows over the BINARY digits (MsmFixedCtx::signed_digits; table row w = 2^(o_w) P): the
// textbook signed-digit bucket method.  Window w holds d_w = bits [c w, c w + c) plus the carry of the window below; a digit
// above 2^(c-1) becomes d_w - 2^c with a carry of one, so |d_w| <= 2^(c-1): HALF the buckets per window bit of the unsigned
// windows (key = |d_w|; |d_w| = 2^(c-1), probability 2^-c, shares key 0, whose bucket the tail weighs by 2^(c-1)), the sign
// travels with the entry (FXW_NEG) and is applied when the point is loaded.  W = ceil(234 / c) windows take a canonical scalar
// (< 2^232) including the last carry.  No tau-adic expansion: a few shifts per window.
// the entry words of one scalar, window by window: f(w, word), word == 0 where the window contributes nothing.  Returns the last carry
// (zero for a canonical scalar: the widths add up to >= 234).
template <class F>
__device__ __forceinline__ uint32_t fx_recode_signed_each(const uint32_t* s /* 9 words, s[8] = 0 */, int c, int W, int n_narrow, F&& f) {
  // the n_narrow LOW windows are c - 1 bits wide, the rest c (widths evened out so that the 234 bits fill every window: a short
  // top window would send every scalar into a handful of buckets)
  uint32_t carry = 0;
  int bit = 0;
#pragma unroll 1
  for (int w = 0; w < W; ++w) {
    const int width = w < n_narrow ? c - 1 : c;
    const uint32_t half = 1u << (width - 1), mask = (1u << width) - 1;
    const int wd = bit >> 5, sh = bit & 31;
    uint32_t lo = wd < 8 ? s[wd] : 0u, hi = wd + 1 < 9 ? s[wd + 1] : 0u;
    uint32_t d = (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) & mask;
    bit += width;
    d += carry;
    carry = 0;
    uint32_t word = 0;
    if (d > mask) {  // all ones + carry: digit 0, carry on
      carry = 1;
    } else if (d > half) {  // d - 2^width < 0
      carry = 1;
      d = (1u << width) - d;  // |d| in [1, 2^(width-1))
      word = FXW_VALID | FXW_NEG | ((uint32_t)w << FXW_ROW_SHIFT) | d;
    } else if (d) {
      word = FXW_VALID | ((uint32_t)w << FXW_ROW_SHIFT) | (d & ((1u << (c - 1)) - 1));  // a full-width d == 2^(c-1) -> key 0
    }
    f(w, word);
  }
  return carry;
}
// loads scalar i; false = it contributes no entries (neutral base, or not canonical -- `report` then records its index)
__device__ __forceinline__ bool fx_load_scalar(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf, uint32_t i, uint32_t* s,
                                               unsigned long long* __restrict__ err, bool report) {
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = scalars[(size_t)i * 8 + k];
  s[8] = 0;
  bool skip = inf && inf[i];
  if (!tau_scalar_is_canonical(s)) {
    if (report) atomicMin(err, (unsigned long long)i);
    skip = true;
  }
  return !skip;
}
__global__ void __launch_bounds__(256)
k_recode_signed(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf, uint32_t n, int c, int W, int n_narrow,
                uint32_t* __restrict__ words, unsigned long long* __restrict__ err) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[9];
  const bool live = fx_load_scalar(scalars, inf, i, s, err, true);
  const uint32_t carry = fx_recode_signed_each(s, c, W, n_narrow, [&](int w, uint32_t word) { words[(size_t)w * n + i] = live ? word : 0u; });
  if (carry && live) atomicMin(err, (unsigned long long)i | (1ull << 62));  // cannot happen: the widths add up to >= 234
}

// ---- exclusive scan of u32 (3 kernels; up to 4096*1024 elements) ---------------------------------
constexpr int SCAN_TPB = 256, SCAN_EPT = 4, SCAN_BLK = SCAN_TPB * SCAN_EPT;

// Exclusive scan of one value per thread over a block of NW waves (<= 16): wave-level inclusive scans by shuffles, the wave totals
// scanned by the first wave, two barriers (+ one so that `sh` can be reused at once).  The first version was Hillis-Steele over the
// whole block in LDS -- two barriers per doubling step, 16-20 barriers of 16 waves for a 1024-thread block -- and those barriers
// were most of the 19 us a staged-scatter block took.  sh: >= 32 words.
template <int NW>
__device__ __forceinline__ uint32_t block_scan_waves(uint32_t v, uint32_t* total, uint32_t* sh) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  uint32_t x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = (uint32_t)__shfl_up((int)x, o);
    if (lane >= o) x += y;
  }
  if (lane == 63) sh[w] = x;
  __syncthreads();
  if (w == 0) {
    uint32_t z = lane < NW ? sh[lane] : 0u;
#pragma unroll
    for (int o = 1; o < NW; o <<= 1) {
      const uint32_t y = (uint32_t)__shfl_up((int)z, o);
      if (lane >= o) z += y;
    }
    if (lane < NW) sh[16 + lane] = z;
  }
  __syncthreads();
  const uint32_t base = w ? sh[16 + w - 1] : 0u;
  *total = sh[16 + NW - 1];
  __syncthreads();
  return base + x - v;
}
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total, uint32_t* sh) {
  return block_scan_waves<SCAN_TPB / 64>(v, total, sh);  // sh: 256 entries
}

// PAD2: every count is rounded up to even first (fixed-base sort: buckets then start at even positions of the sorted
// item list, so the list read as uint2 pairs IS the first pair round's descriptor array, see k_post_sort)
template <bool PAD2>
__global__ void __launch_bounds__(SCAN_TPB) k_scan_local(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                         uint32_t* __restrict__ bsum, uint32_t m) {
  __shared__ uint32_t sh[SCAN_TPB];
  uint32_t base = blockIdx.x * SCAN_BLK + threadIdx.x * SCAN_EPT;
  uint32_t v[SCAN_EPT], s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_EPT; ++k) {
    v[k] = (base + k < m) ? in[base + k] : 0;
    if (PAD2) v[k] = (v[k] + 1u) & ~1u;
    s += v[k];
  }
  uint32_t tot;
  uint32_t ex = block_exclusive_scan(s, &tot, sh);
#pragma unroll
  for (int k = 0; k < SCAN_EPT; ++k) {
    if (base + k < m) out[base + k] = ex;
    ex += v[k];
  }
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// the same with a division fused in: v = ceil(in / K) (outputs per bucket of a pair round, tasks per bucket of the
// fan-in-K reducer) is written to vout and scanned
__global__ void __launch_bounds__(SCAN_TPB) k_scan_local_div(const uint32_t* __restrict__ in, uint32_t K, uint32_t* __restrict__ vout,
                                                             uint32_t* __restrict__ out, uint32_t* __restrict__ bsum, uint32_t m) {
  __shared__ uint32_t sh[SCAN_TPB];
  uint32_t base = blockIdx.x * SCAN_BLK + threadIdx.x * SCAN_EPT;
  uint32_t v[SCAN_EPT], s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_EPT; ++k) {
    v[k] = (base + k < m) ? (in[base + k] + K - 1) / K : 0;
    if (base + k < m) vout[base + k] = v[k];
    s += v[k];
  }
  uint32_t tot;
  uint32_t ex = block_exclusive_scan(s, &tot, sh);
#pragma unroll
  for (int k = 0; k < SCAN_EPT; ++k) {
    if (base + k < m) out[base + k] = ex;
    ex += v[k];
  }
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_TPB) k_scan_bsums(uint32_t* __restrict__ bsum, uint32_t nb, uint32_t* total_out) {
  __shared__ uint32_t sh[SCAN_TPB];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nb; base += SCAN_TPB) {
    uint32_t idx = base + threadIdx.x;
    uint32_t v = idx < nb ? bsum[idx] : 0;
    uint32_t tot;
    uint32_t ex = block_exclusive_scan(v, &tot, sh);
    if (idx < nb) bsum[idx] = ex + carry;
    carry += tot;
  }
  if (threadIdx.x == 0) *total_out = carry;
}

__global__ void __launch_bounds__(SCAN_TPB) k_scan_add(uint32_t* __restrict__ out, const uint32_t* __restrict__ bsum, uint32_t m) {
  uint32_t base = blockIdx.x * SCAN_BLK + threadIdx.x * SCAN_EPT;
  uint32_t add = bsum[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_EPT; ++k)
    if (base + k < m) out[base + k] += add;
}

// Short inputs (<= SCAN_FOLD_NB blocks of 1 024 values: the bucket arrays of a small MSM) fold the scan of the block sums into the last
// launch: every block adds up the raw sums of the blocks in front of it itself (a uniform loop over <= 64 words) -- two dependent
// launches per scan instead of three (round 6: a one-shot MSM of 2^16 points scans its 24 576 counts four times)
constexpr uint32_t SCAN_FOLD_NB = 64;
__global__ void __launch_bounds__(SCAN_TPB) k_scan_add_fold(uint32_t* __restrict__ out, const uint32_t* __restrict__ bsum_raw, uint32_t m,
                                                           uint32_t* __restrict__ total_out) {
  uint32_t add = 0;
  for (uint32_t b = 0; b < blockIdx.x; ++b) add += bsum_raw[b];
  const uint32_t base = blockIdx.x * SCAN_BLK + threadIdx.x * SCAN_EPT;
#pragma unroll
  for (int k = 0; k < SCAN_EPT; ++k)
    if (base + k < m) out[base + k] += add;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = add + bsum_raw[blockIdx.x];
}

// out[0..m) = exclusive scan of in[0..m); out[m] = total
static int scan_exclusive(const uint32_t* in, uint32_t* out, uint32_t m, uint32_t* bsum, hipStream_t st, bool pad2 = false) {
  uint32_t nb = cdiv(m, SCAN_BLK);
  if (pad2)
    hipLaunchKernelGGL(k_scan_local<true>, dim3(nb), dim3(SCAN_TPB), 0, st, in, out, bsum, m);
  else
    hipLaunchKernelGGL(k_scan_local<false>, dim3(nb), dim3(SCAN_TPB), 0, st, in, out, bsum, m);
  if (nb <= SCAN_FOLD_NB)
    hipLaunchKernelGGL(k_scan_add_fold, dim3(nb), dim3(SCAN_TPB), 0, st, out, bsum, m, out + m);
  else {
    hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(SCAN_TPB), 0, st, bsum, nb, out + m);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(SCAN_TPB), 0, st, out, bsum, m);
  }
  DVP_HIP(hipGetLastError());
  return DVP_OK;
}

// vout[i] = ceil(in[i] / K); out = exclusive scan of vout (out[m] = total): the per-round bookkeeping of the bucket reducers
static int scan_exclusive_div(const uint32_t* in, uint32_t K, uint32_t* vout, uint32_t* out, uint32_t m, uint32_t* bsum, hipStream_t st) {
  uint32_t nb = cdiv(m, SCAN_BLK);
  hipLaunchKernelGGL(k_scan_local_div, dim3(nb), dim3(SCAN_TPB), 0, st, in, K, vout, out, bsum, m);
  if (nb <= SCAN_FOLD_NB)
    hipLaunchKernelGGL(k_scan_add_fold, dim3(nb), dim3(SCAN_TPB), 0, st, out, bsum, m, out + m);
  else {
    hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(SCAN_TPB), 0, st, bsum, nb, out + m);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(SCAN_TPB), 0, st, out, bsum, m);
  }
  DVP_HIP(hipGetLastError());
  return DVP_OK;
}

// ---- counting sort by (window, pattern) without global atomics ---------------------------------------
// Scattered global atomics run at ~16 G/s on MI355X (they execute at the memory side), which made
// histogram + scatter cost as much as a third of the MSM.  Instead each block owns a chunk of
// SORT_CHUNK scalars of ONE window and keeps the 2^c counters in LDS (c <= 15: 2^15 x 4 B = 128 KB of
// the CU's 160 KB):
//   k_hist_local  : LDS histogram (two u16 counters per word) -> hist[w][chunk][2^c] (u16)
//   k_hist_scan   : per (w, pattern): exclusive prefix over chunks -> chunk_off (u32) and cnt[w][pattern]
//   (scan of cnt -> off, as before)
//   k_scatter_local: LDS cursors = off + chunk_off, ds_add_rtn per digit -> items[]
constexpr uint32_t SORT_CHUNK = 32768;  // counts fit in u16
constexpr int SORT_TPB = 1024;

__global__ void __launch_bounds__(SORT_TPB)
k_hist_local(const uint16_t* __restrict__ digits, uint32_t n, int c, uint16_t* __restrict__ hist) {
  extern __shared__ uint32_t lds_cnt[];  // 2^(c-1) words
  const uint32_t chunk = blockIdx.x, w = blockIdx.y, nb = 1u << c, nchunks = gridDim.x;
  for (uint32_t k = threadIdx.x; k < (nb >> 1); k += SORT_TPB) lds_cnt[k] = 0;
  __syncthreads();
  uint32_t lo = chunk * SORT_CHUNK, hi = min(n, lo + SORT_CHUNK);
  const uint16_t* dg = digits + (size_t)w * n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += SORT_TPB) {
    uint32_t d = dg[i];
    if (d) atomicAdd(&lds_cnt[d >> 1], 1u << (16 * (d & 1)));
  }
  __syncthreads();
  uint32_t* out = (uint32_t*)(hist + ((size_t)w * nchunks + chunk) * nb);
  for (uint32_t k = threadIdx.x; k < (nb >> 1); k += SORT_TPB) out[k] = lds_cnt[k];
}

__global__ void __launch_bounds__(256)
k_hist_scan(const uint16_t* __restrict__ hist, uint32_t nchunks, int c, int W, uint32_t* __restrict__ chunk_off,
            uint32_t* __restrict__ cnt) {
  uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t nb = 1u << c;
  if (key >= ((uint32_t)W << c)) return;
  uint32_t w = key >> c, b = key & (nb - 1);
  uint32_t run = 0;
  for (uint32_t ch = 0; ch < nchunks; ++ch) {
    size_t idx = ((size_t)w * nchunks + ch) * nb + b;
    uint32_t v = hist[idx];
    chunk_off[idx] = run;
    run += v;
  }
  cnt[key] = run;
}

__global__ void __launch_bounds__(SORT_TPB)
k_scatter_local(const uint16_t* __restrict__ digits, uint32_t n, int c, const uint32_t* __restrict__ off,
                const uint32_t* __restrict__ chunk_off, uint32_t* __restrict__ items) {
  extern __shared__ uint32_t lds_cur[];  // 2^c words
  const uint32_t chunk = blockIdx.x, w = blockIdx.y, nb = 1u << c, nchunks = gridDim.x;
  const uint32_t* co = chunk_off + ((size_t)w * nchunks + chunk) * nb;
  const uint32_t* of = off + ((size_t)w << c);
  for (uint32_t k = threadIdx.x; k < nb; k += SORT_TPB) lds_cur[k] = of[k] + co[k];
  __syncthreads();
  uint32_t lo = chunk * SORT_CHUNK, hi = min(n, lo + SORT_CHUNK);
  const uint16_t* dg = digits + (size_t)w * n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += SORT_TPB) {
    uint32_t d = dg[i];
    if (d) items[atomicAdd(&lds_cur[d], 1u)] = i;
  }
}

// ---- two-level counting sort for the fixed-base mode (keys of up to 20 bits) ----------------------------
// Level 1 partitions the entries by the top `hi` key bits (LDS cursors per block, up to 2^10 partitions); level 2 is the
// LDS counting sort above applied inside each partition on the low `lo` <= 15 bits.  Both levels scatter single entries,
// so the more bins a level has the fewer entries of a chunk share a 32-byte sector: with 2^15 level-2 bins each 4-byte
// entry costs a whole sector (1.3 GB of HBM writes for 160 MB of entries), with 2^9 level-1 bins the two level-1
// streams thrash the L2 instead.  The split is a per-context knob (MsmFixedCtx::hi_bits); staging each chunk in LDS
// and writing whole runs would remove the trade-off and is the known next step for this stage.
// An entry is the pair (table row, scalar i) (FXW_* entry words); its id e = row * n_total + i0 + i indexes the
// pre-rotated base table.
constexpr int FX_C_MAX = 20, FX_NP_MAX = 1 << (FX_C_MAX / 2);  // c = lo + hi bits, chosen per context
constexpr uint32_t FX_CHUNK = 8192;  // entries per block at both levels: the staged scatters keep a whole chunk in LDS (78 KB / 61 KB: two blocks per CU, so one block's copy-out overlaps the other's staging; 16384 = one block per CU: sort 1.27 -> 1.17 ms per proof; 4096: 1.54)
struct FxBits {
  int lo, hi;  // low bits sorted in LDS, high bits partitioned first
  __host__ __device__ uint32_t np() const { return 1u << hi; }
};

__global__ void __launch_bounds__(SORT_TPB)
k_part_hist(const uint32_t* __restrict__ digits, size_t total, FxBits fb, uint32_t* __restrict__ phist) {
  __shared__ uint32_t h[FX_NP_MAX];
  const uint32_t FX_NP = fb.np();
  const int FX_LO = fb.lo;
  for (uint32_t k = threadIdx.x; k < FX_NP; k += SORT_TPB) h[k] = 0;
  __syncthreads();
  size_t lo = (size_t)blockIdx.x * FX_CHUNK, hi = min(total, lo + FX_CHUNK);
  for (size_t e = lo + threadIdx.x; e < hi; e += SORT_TPB) {
    uint32_t d = digits[e];
    if (d) atomicAdd(&h[fxw_key(d) >> FX_LO], 1u);
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < FX_NP; k += SORT_TPB) phist[(size_t)blockIdx.x * FX_NP + k] = h[k];
}
// exclusive prefix of every partition's counts over the level-1 blocks, in two coalesced launches (round 5; the first version gave a
// block a partition and walked phist down a column -- a 128-byte line per 4-byte count, twice: 52 us per MSM).
// pass A: a block takes FX_SCAN_CB consecutive level-1 blocks; a thread owns partition columns (consecutive threads, consecutive
// partitions: whole lines in, whole lines out) and leaves the prefix WITHIN the chunk in pbase and the chunk's totals in ctot[part][chunk];
// pass B: a wave per partition scans its chunk totals in place (exclusive) and leaves the partition's count.
// A level-1 block b then starts its run of partition p at pstart[p] + ctot[p][b / FX_SCAN_CB] + pbase[b][p].
constexpr uint32_t FX_SCAN_CB = 32;
__global__ void __launch_bounds__(256)
k_part_scan_chunks(const uint32_t* __restrict__ phist, uint32_t nblk, FxBits fb, uint32_t* __restrict__ pbase, uint32_t* __restrict__ ctot, uint32_t nch) {
  const uint32_t FX_NP = fb.np();
  const uint32_t ch = blockIdx.x, b0 = ch * FX_SCAN_CB, b1 = min(nblk, b0 + FX_SCAN_CB);
  for (uint32_t part = threadIdx.x; part < FX_NP; part += 256) {
    uint32_t run = 0;
#pragma unroll 8
    for (uint32_t b = b0; b < b1; ++b) {
      const uint32_t v = phist[(size_t)b * FX_NP + part];
      pbase[(size_t)b * FX_NP + part] = run;
      run += v;
    }
    ctot[(size_t)part * nch + ch] = run;
  }
}
__global__ void __launch_bounds__(256)
k_part_scan_totals(uint32_t* __restrict__ ctot, uint32_t nch, FxBits fb, uint32_t* __restrict__ pcount) {
  const uint32_t part = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
  if (part >= fb.np()) return;  // whole waves
  uint32_t* row = ctot + (size_t)part * nch;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nch; base += 64) {
    const uint32_t idx = base + lane;
    const uint32_t v = idx < nch ? row[idx] : 0;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(x, o);
      if ((int)lane >= o) x += y;
    }
    if (idx < nch) row[idx] = carry + x - v;
    carry += __shfl(x, 63);
  }
  if (lane == 0) pcount[part] = carry;
}
// partition starts and the global level-2 chunk index of each partition (one block of FX_NP_MAX threads)
__global__ void __launch_bounds__(FX_NP_MAX)
k_part_starts(const uint32_t* __restrict__ pcount, FxBits fb, uint32_t* __restrict__ pstart, uint32_t* __restrict__ cstart) {
  __shared__ uint32_t sp[FX_NP_MAX], sc[FX_NP_MAX];
  const uint32_t FX_NP = fb.np(), t = threadIdx.x;
  uint32_t pc = t < FX_NP ? pcount[t] : 0, cc = (pc + FX_CHUNK - 1) / FX_CHUNK;
  sp[t] = pc;
  sc[t] = cc;
  __syncthreads();
  for (uint32_t o = 1; o < FX_NP_MAX; o <<= 1) {
    uint32_t a = t >= o ? sp[t - o] : 0, b = t >= o ? sc[t - o] : 0;
    __syncthreads();
    sp[t] += a;
    sc[t] += b;
    __syncthreads();
  }
  if (t < FX_NP) {
    pstart[t] = sp[t] - pc;
    cstart[t] = sc[t] - cc;
  }
  if (t == FX_NP - 1) {
    pstart[FX_NP] = sp[t];
    cstart[FX_NP] = sc[t];
  }
}
__global__ void __launch_bounds__(SORT_TPB)
k_part_scatter(const uint32_t* __restrict__ digits, size_t total, uint32_t n, uint32_t n_total, uint32_t i0, FxBits fb,
               const uint32_t* __restrict__ pbase, const uint32_t* __restrict__ cbase, uint32_t nch, const uint32_t* __restrict__ pstart,
               uint16_t* __restrict__ plo, uint32_t* __restrict__ pid) {
  __shared__ uint32_t cur[FX_NP_MAX];
  const uint32_t FX_NP = fb.np();
  const int FX_LO = fb.lo;
  const uint32_t ch = blockIdx.x / FX_SCAN_CB;
  for (uint32_t k = threadIdx.x; k < FX_NP; k += SORT_TPB) cur[k] = pstart[k] + cbase[(size_t)k * nch + ch] + pbase[(size_t)blockIdx.x * FX_NP + k];
  __syncthreads();
  size_t lo = (size_t)blockIdx.x * FX_CHUNK, hi = min(total, lo + FX_CHUNK);
  for (size_t e = lo + threadIdx.x; e < hi; e += SORT_TPB) {
    uint32_t d = digits[e];
    if (!d) continue;
    const uint32_t i = (uint32_t)(e % n), key = fxw_key(d);
    uint32_t pos = atomicAdd(&cur[key >> FX_LO], 1u);
    plo[pos] = (uint16_t)(key & ((1u << FX_LO) - 1));
    pid[pos] = (fxw_row(d) * n_total + i0 + i) | ((d & FXW_NEG) ? ITEM_NEG : 0u);
  }
}
// the partition whose chunk range [cstart[k], cstart[k+1]) holds chunk g (empty partitions share their start with the
// next one, so take the LAST k with cstart[k] <= g)
__device__ __forceinline__ uint32_t fx_chunk_partition(const uint32_t* __restrict__ cstart, uint32_t FX_NP, uint32_t g) {
  uint32_t lo = 0, hi = FX_NP;  // invariant: cstart[lo] <= g, and cstart[hi] > g or hi == FX_NP
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (cstart[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}
// level 2, per global chunk g (partition hi, local chunk g - cstart[hi])
__global__ void __launch_bounds__(SORT_TPB)
k_hist_local2(const uint16_t* __restrict__ plo, const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ cstart, FxBits fb,
              uint16_t* __restrict__ hist) {
  extern __shared__ uint32_t lds_cnt[];  // 2^(lo-1) words (two u16 counters per word)
  const uint32_t FX_NP = fb.np();
  const uint32_t g = blockIdx.x, nb = 1u << fb.lo;
  if (g >= cstart[FX_NP]) return;
  for (uint32_t k = threadIdx.x; k < (nb >> 1); k += SORT_TPB) lds_cnt[k] = 0;
  __syncthreads();
  uint32_t hi = fx_chunk_partition(cstart, FX_NP, g);
  uint32_t lo_e = pstart[hi] + (g - cstart[hi]) * FX_CHUNK, hi_e = min(pstart[hi + 1], lo_e + FX_CHUNK);
  // Round 5: eight entries per load (16 bytes) over the 16-byte-aligned body of the chunk, single entries for its ragged ends -- one
  // load per thread and chunk instead of eight 2-byte ones (100 MB at 1.3 TB/s: the kernel waited for its loads)
  auto count = [&](uint32_t d) { atomicAdd(&lds_cnt[d >> 1], 1u << (16 * (d & 1))); };
  const uint32_t body_lo = min(hi_e, (lo_e + 7u) & ~7u), body_hi = max(body_lo, hi_e & ~7u);
  for (uint32_t j = lo_e + threadIdx.x; j < body_lo; j += SORT_TPB) count(plo[j]);
  for (uint32_t j = body_lo + 8u * threadIdx.x; j < body_hi; j += 8u * SORT_TPB) {
    const uint4 v = *(const uint4*)(plo + j);
    count(v.x & 0xffffu); count(v.x >> 16); count(v.y & 0xffffu); count(v.y >> 16);
    count(v.z & 0xffffu); count(v.z >> 16); count(v.w & 0xffffu); count(v.w >> 16);
  }
  for (uint32_t j = body_hi + threadIdx.x; j < hi_e; j += SORT_TPB) count(plo[j]);
  __syncthreads();
  uint32_t* out = (uint32_t*)(hist + (size_t)g * nb);
  for (uint32_t k = threadIdx.x; k < (nb >> 1); k += SORT_TPB) out[k] = lds_cnt[k];
}
__global__ void __launch_bounds__(256)
k_hist_scan2(const uint16_t* __restrict__ hist, const uint32_t* __restrict__ cstart, FxBits fb, uint32_t* __restrict__ chunk_off,
             uint32_t* __restrict__ cnt) {
  uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;  // < 2^c
  if (key >= (1u << (fb.lo + fb.hi))) return;
  const uint32_t nb = 1u << fb.lo;
  uint32_t hi = key >> fb.lo, b = key & (nb - 1);
  uint32_t run = 0;
  for (uint32_t g = cstart[hi]; g < cstart[hi + 1]; ++g) {
    size_t idx = (size_t)g * nb + b;
    uint32_t v = hist[idx];
    chunk_off[idx] = run;
    run += v;
  }
  cnt[key] = run;
}
__global__ void __launch_bounds__(SORT_TPB)
k_scatter_local2(const uint16_t* __restrict__ plo, const uint32_t* __restrict__ pid, const uint32_t* __restrict__ pstart,
                 const uint32_t* __restrict__ cstart, FxBits fb, const uint32_t* __restrict__ off, const uint32_t* __restrict__ chunk_off,
                 uint32_t* __restrict__ items) {
  extern __shared__ uint32_t lds_cur[];  // 2^lo words
  const uint32_t FX_NP = fb.np();
  const int FX_LO = fb.lo;
  const uint32_t g = blockIdx.x, nb = 1u << FX_LO;
  if (g >= cstart[FX_NP]) return;
  uint32_t hi = fx_chunk_partition(cstart, FX_NP, g);
  const uint32_t* co = chunk_off + (size_t)g * nb;
  const uint32_t* of = off + ((size_t)hi << FX_LO);
  for (uint32_t k = threadIdx.x; k < nb; k += SORT_TPB) lds_cur[k] = of[k] + co[k];
  __syncthreads();
  uint32_t lo_e = pstart[hi] + (g - cstart[hi]) * FX_CHUNK, hi_e = min(pstart[hi + 1], lo_e + FX_CHUNK);
  for (uint32_t j = lo_e + threadIdx.x; j < hi_e; j += SORT_TPB) items[atomicAdd(&lds_cur[plo[j]], 1u)] = pid[j];
}

// ---- staged scatters (used when a level has <= SORT_TPB bins) ---------------------------------------------------------
// A block first places its chunk into LDS grouped by bin (LDS cursors start at the exclusive scan of the chunk's own
// histogram), then copies the staged chunk out in order: consecutive lanes hold consecutive entries of one bin, whose
// destinations are consecutive, so every bin leaves as one coalesced run instead of as single-entry sector writes.
__device__ __forceinline__ uint32_t block_scan_tpb(uint32_t v, uint32_t* sh, uint32_t* total) {
  return block_scan_waves<SORT_TPB / 64>(v, total, sh);
}
constexpr unsigned FX_STAGE1_LDS = (2 * FX_NP_MAX + SORT_TPB + 2 * FX_CHUNK) * 4;
constexpr unsigned FX_STAGE2_LDS = (2 * FX_NP_MAX + SORT_TPB + FX_CHUNK) * 4 + FX_CHUNK * 2;

__global__ void __launch_bounds__(SORT_TPB)
k_part_scatter_staged(const uint32_t* __restrict__ digits, size_t total, uint32_t n, uint32_t n_total, uint32_t i0, FxBits fb,
                      const uint32_t* __restrict__ phist, const uint32_t* __restrict__ pbase, const uint32_t* __restrict__ cbase, uint32_t nch,
                      const uint32_t* __restrict__ pstart, uint16_t* __restrict__ plo, uint32_t* __restrict__ pid) {
  extern __shared__ uint32_t lds_st[];
  uint32_t* cur = lds_st;             // [FX_NP_MAX]
  uint32_t* gdst = cur + FX_NP_MAX;   // [FX_NP_MAX] destination of staged position 0 of the bin
  uint32_t* sh = gdst + FX_NP_MAX;    // [SORT_TPB]
  uint32_t* st_d = sh + SORT_TPB;     // [FX_CHUNK]
  uint32_t* st_id = st_d + FX_CHUNK;  // [FX_CHUNK]
  const uint32_t FX_NP = fb.np(), t = threadIdx.x;
  const int FX_LO = fb.lo;
  const size_t hb = (size_t)blockIdx.x * FX_NP;
  uint32_t h = t < FX_NP ? phist[hb + t] : 0, tot;
  uint32_t ex = block_scan_tpb(h, sh, &tot);
  if (t < FX_NP) {
    cur[t] = ex;
    gdst[t] = pstart[t] + cbase[(size_t)t * nch + blockIdx.x / FX_SCAN_CB] + pbase[hb + t] - ex;
  }
  __syncthreads();
  const size_t lo = (size_t)blockIdx.x * FX_CHUNK, hi = min(total, lo + FX_CHUNK);
  const uint32_t w0 = (uint32_t)(lo / n);
  for (size_t e = lo + t; e < hi; e += SORT_TPB) {
    uint32_t d = digits[e];
    if (!d) continue;
    size_t i = e - (size_t)w0 * n;  // e mod n: a chunk spans few slots
    while (i >= n) i -= n;
    const uint32_t key = fxw_key(d);
    uint32_t pos = atomicAdd(&cur[key >> FX_LO], 1u);
    st_d[pos] = key;
    st_id[pos] = (fxw_row(d) * n_total + i0 + (uint32_t)i) | ((d & FXW_NEG) ? ITEM_NEG : 0u);
  }
  __syncthreads();
  const uint32_t mask = (1u << FX_LO) - 1;
  for (uint32_t pos = t; pos < tot; pos += SORT_TPB) {
    uint32_t d = st_d[pos], dst = gdst[d >> FX_LO] + pos;
    plo[dst] = (uint16_t)(d & mask);
    pid[dst] = st_id[pos];
  }
}

// ---- level 1 of the signed flavour straight from the scalars (round 4) -------------------------------------------------------
// k_recode_signed wrote W entry words per scalar (4 B each) that k_part_hist and the level-1 scatter then read back: 12 B of HBM
// traffic per entry for a few shifts' worth of work.  Here a block owns S = FX_CHUNK / W consecutive SCALARS (all their windows:
// <= FX_CHUNK entries) and both level-1 kernels recompute the words from the 32-byte scalar.  Which entries share a chunk changes,
// the sorted order inside a bucket with it -- the bucket's SUM does not.
__global__ void __launch_bounds__(SORT_TPB)
k_part_hist_signed(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf, uint32_t n, int c, int W, int n_narrow, uint32_t S,
                   FxBits fb, uint32_t* __restrict__ phist, unsigned long long* __restrict__ err) {
  __shared__ uint32_t h[FX_NP_MAX];
  const uint32_t FX_NP = fb.np();
  const int FX_LO = fb.lo;
  for (uint32_t k = threadIdx.x; k < FX_NP; k += SORT_TPB) h[k] = 0;
  __syncthreads();
  const uint32_t lo = blockIdx.x * S, hi = min(n, lo + S);
  for (uint32_t i = lo + threadIdx.x; i < hi; i += SORT_TPB) {
    uint32_t s[9];
    if (!fx_load_scalar(scalars, inf, i, s, err, true)) continue;
    const uint32_t carry = fx_recode_signed_each(s, c, W, n_narrow, [&](int, uint32_t word) {
      if (word) atomicAdd(&h[fxw_key(word) >> FX_LO], 1u);
    });
    if (carry) atomicMin(err, (unsigned long long)i | (1ull << 62));  // cannot happen
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < FX_NP; k += SORT_TPB) phist[(size_t)blockIdx.x * FX_NP + k] = h[k];
}
template <bool STAGED>
__global__ void __launch_bounds__(SORT_TPB)
k_part_scatter_signed(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf, uint32_t n, int c, int W, int n_narrow, uint32_t S,
                      uint32_t n_total, uint32_t i0, FxBits fb, const uint32_t* __restrict__ phist, const uint32_t* __restrict__ pbase,
                      const uint32_t* __restrict__ cbase, uint32_t nch, const uint32_t* __restrict__ pstart, uint16_t* __restrict__ plo,
                      uint32_t* __restrict__ pid) {
  extern __shared__ uint32_t lds_st[];
  uint32_t* cur = lds_st;             // [FX_NP_MAX]
  uint32_t* gdst = cur + FX_NP_MAX;   // [FX_NP_MAX] STAGED: destination of staged position 0 of the bin
  uint32_t* sh = gdst + FX_NP_MAX;    // [SORT_TPB]
  uint32_t* st_d = sh + SORT_TPB;     // [FX_CHUNK]
  uint32_t* st_id = st_d + FX_CHUNK;  // [FX_CHUNK]
  const uint32_t FX_NP = fb.np(), t = threadIdx.x;
  const int FX_LO = fb.lo;
  const size_t hb = (size_t)blockIdx.x * FX_NP;
  const uint32_t mask = (1u << FX_LO) - 1;
  uint32_t tot = 0;
  const uint32_t ch = blockIdx.x / FX_SCAN_CB;
  if (STAGED) {
    uint32_t h = t < FX_NP ? phist[hb + t] : 0;
    uint32_t ex = block_scan_tpb(h, sh, &tot);
    if (t < FX_NP) {
      cur[t] = ex;
      gdst[t] = pstart[t] + cbase[(size_t)t * nch + ch] + pbase[hb + t] - ex;
    }
  } else {
    for (uint32_t k = t; k < FX_NP; k += SORT_TPB) cur[k] = pstart[k] + cbase[(size_t)k * nch + ch] + pbase[hb + k];
  }
  __syncthreads();
  const uint32_t lo = blockIdx.x * S, hi = min(n, lo + S);
  for (uint32_t i = lo + t; i < hi; i += SORT_TPB) {
    uint32_t s[9];
    if (!fx_load_scalar(scalars, inf, i, s, nullptr, false)) continue;
    fx_recode_signed_each(s, c, W, n_narrow, [&](int w, uint32_t word) {
      if (!word) return;
      const uint32_t key = fxw_key(word);
      const uint32_t id = ((uint32_t)w * n_total + i0 + i) | ((word & FXW_NEG) ? ITEM_NEG : 0u);
      const uint32_t pos = atomicAdd(&cur[key >> FX_LO], 1u);
      if (STAGED) {
        st_d[pos] = key;
        st_id[pos] = id;
      } else {
        plo[pos] = (uint16_t)(key & mask);
        pid[pos] = id;
      }
    });
  }
  if (!STAGED) return;
  __syncthreads();
  for (uint32_t pos = t; pos < tot; pos += SORT_TPB) {
    uint32_t d = st_d[pos], dst = gdst[d >> FX_LO] + pos;
    plo[dst] = (uint16_t)(d & mask);
    pid[dst] = st_id[pos];
  }
}

__global__ void __launch_bounds__(SORT_TPB)
k_scatter_local2_staged(const uint16_t* __restrict__ plo, const uint32_t* __restrict__ pid, const uint32_t* __restrict__ pstart,
                        const uint32_t* __restrict__ cstart, FxBits fb, const uint32_t* __restrict__ off,
                        const uint32_t* __restrict__ chunk_off, const uint16_t* __restrict__ hist, uint32_t* __restrict__ items) {
  extern __shared__ uint32_t lds_st[];
  uint32_t* cur = lds_st;
  uint32_t* gdst = cur + FX_NP_MAX;
  uint32_t* sh = gdst + FX_NP_MAX;
  uint32_t* st_id = sh + SORT_TPB;                 // [FX_CHUNK]
  uint16_t* st_b = (uint16_t*)(st_id + FX_CHUNK);  // [FX_CHUNK]
  const uint32_t FX_NP = fb.np(), t = threadIdx.x;
  const int FX_LO = fb.lo;
  const uint32_t g = blockIdx.x, nb = 1u << FX_LO;  // nb <= SORT_TPB
  if (g >= cstart[FX_NP]) return;
  const uint32_t hi = fx_chunk_partition(cstart, FX_NP, g);
  uint32_t h = t < nb ? hist[(size_t)g * nb + t] : 0, tot;
  uint32_t ex = block_scan_tpb(h, sh, &tot);
  if (t < nb) {
    cur[t] = ex;
    gdst[t] = off[((size_t)hi << FX_LO) + t] + chunk_off[(size_t)g * nb + t] - ex;
  }
  __syncthreads();
  const uint32_t lo_e = pstart[hi] + (g - cstart[hi]) * FX_CHUNK, hi_e = min(pstart[hi + 1], lo_e + FX_CHUNK);
  auto place = [&](uint32_t b, uint32_t id) {
    const uint32_t pos = atomicAdd(&cur[b], 1u);
    st_id[pos] = id;
    st_b[pos] = (uint16_t)b;
  };
  // Round 5: eight entries per thread and trip over the aligned body of the chunk (one 16-byte load of keys, two of ids) instead of
  // eight trips of a 2-byte and a 4-byte load; which entry of a bin lands where inside the bin changes, the bin's content does not
  const uint32_t body_lo = min(hi_e, (lo_e + 7u) & ~7u), body_hi = max(body_lo, hi_e & ~7u);
  for (uint32_t j = lo_e + t; j < body_lo; j += SORT_TPB) place(plo[j], pid[j]);
  for (uint32_t j = body_lo + 8u * t; j < body_hi; j += 8u * SORT_TPB) {
    const uint4 k = *(const uint4*)(plo + j), a = *(const uint4*)(pid + j), c = *(const uint4*)(pid + j + 4);
    place(k.x & 0xffffu, a.x); place(k.x >> 16, a.y); place(k.y & 0xffffu, a.z); place(k.y >> 16, a.w);
    place(k.z & 0xffffu, c.x); place(k.z >> 16, c.y); place(k.w & 0xffffu, c.z); place(k.w >> 16, c.w);
  }
  for (uint32_t j = body_hi + t; j < hi_e; j += SORT_TPB) place(plo[j], pid[j]);
  __syncthreads();
  for (uint32_t pos = t; pos < tot; pos += SORT_TPB) items[gdst[st_b[pos]] + pos] = st_id[pos];
}

// pre-rotated base table for the fixed-base mode: T[w][i] = tau^(o_w)(P_i), o_w = first digit of window w
// (windows 0 .. n_narrow-1 are c-1 digits wide, the rest c)
__global__ void __launch_bounds__(256)
k_frob_table(const Aff* __restrict__ bases, uint32_t n, int FX_C, int FX_W, int n_narrow, Aff* __restrict__ table) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff p = bases[i];
  table[i] = p;
#pragma unroll 1
  for (int w = 1; w < FX_W; ++w) {
    int width = (w - 1) < n_narrow ? FX_C - 1 : FX_C;
    p.x = gf_sqr_n(p.x, width);
    p.y = gf_sqr_n(p.y, width);
    table[(size_t)w * n + i] = p;
  }
}

// signed aligned windows: T[w][i] = 2^(o_w) P_i, o_w = first bit of window w (the n_narrow low windows are c - 1 bits wide).
// The ~222 doublings of a base run in Lopez-Dahab coordinates (3M + 5S each) and the W - 1 row snapshots of the base share ONE
// inversion (Montgomery's trick over their Z's): ~0.9 k field products per base.  The first version doubled in affine
// coordinates, one table-driven inversion per doubling: 4.2 k products per base, 0.30 s per table at 2^20 constraints and 75 % of
// the profiled GPU time of a short run (prover open).  zs / pre: thread-private scratch in HBM, (W - 1) x n field elements each
// ([row][base]: coalesced), freed after the build.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_dbl_table(const Aff* __restrict__ bases, uint32_t n, int c, int W, int n_narrow, GfSqrTables T, Aff* __restrict__ table, Gf* __restrict__ zs,
            Gf* __restrict__ pre) {
  extern __shared__ char lds_raw[];
  GfLdsK L = gf_ldsk_init(lds_raw);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Aff p0 = bases[i];
  table[i] = p0;
  Ld p = ld_from_aff(p0);
  Gf run = gf_one();
#pragma unroll 1
  for (int w = 1; w < W; ++w) {
    const int width = (w - 1) < n_narrow ? c - 1 : c;  // row w starts where window w - 1 ends
#pragma unroll 1
    for (int k = 0; k < width; ++k) p = ld_dbl(p, L);
    Aff raw;
    raw.x = p.X;
    raw.y = p.Y;
    table[(size_t)w * n + i] = raw;  // normalised below
    zs[(size_t)(w - 1) * n + i] = p.Z;
    pre[(size_t)(w - 1) * n + i] = run;
    run = gf_mul(run, p.Z, L);
  }
  Gf inv = gf_inv_fast(run, T, L);  // a base of E[r] never doubles to infinity: every Z is non-zero
#pragma unroll 1
  for (int w = W - 1; w >= 1; --w) {
    const Gf z = zs[(size_t)(w - 1) * n + i];
    Gf zi, inv_next;
    gf_mul2(pre[(size_t)(w - 1) * n + i], z, inv, L, zi, inv_next);  // 1 / Z_w, and the running inverse stripped of Z_w
    inv = inv_next;
    Aff a = table[(size_t)w * n + i];
    a.x = gf_mul(a.x, zi, L);
    a.y = gf_mul(a.y, gf_sqr(zi), L);
    table[(size_t)w * n + i] = a;
  }
}

// EC kernels on the Karatsuba LDS multiplier: 256-thread blocks, 4 x 8 KB of half tables; the quad-cooperative
// flavours (latency-bound stages) keep the 16 KB comb tables
constexpr int EC_TPB = 256;
constexpr unsigned EC_LDS = (EC_TPB / 64) * GF_LDSK_BYTES_PER_WAVE;
constexpr unsigned EC_LDS_Q = (EC_TPB / 64) * GF_LDS_BYTES_PER_WAVE;
constexpr uint32_t MERGE_HEX_MAX = 8192;    // additions per level up to which 16 lanes per addition win: one chip-full of rows at two waves per SIMD (measured: 0 / 4096 / 8192 -> 21.01 / 20.94 / 20.87 ms per proof)
constexpr uint32_t MERGE_QUAD_MAX = 16384;  // additions per level up to which 4 lanes per addition win (measured: 35 us vs 41 us at 16384)

// the multiplier flavours behind one name: init, and which lane of a group stores
template <class LT> struct MergeMul;
template <> struct MergeMul<GfLdsK> { static __device__ __forceinline__ GfLdsK init(char* l) { return gf_ldsk_init(l); } static __device__ __forceinline__ bool lead(const GfLdsK&) { return true; } };
template <> struct MergeMul<GfLdsQ> { static __device__ __forceinline__ GfLdsQ init(char* l) { return gf_ldsq_init(l); } static __device__ __forceinline__ bool lead(const GfLdsQ& c) { return c.r == 0; } };
template <> struct MergeMul<GfLdsH> { static __device__ __forceinline__ GfLdsH init(char* l) { return gf_ldsh_init(l); } static __device__ __forceinline__ bool lead(const GfLdsH& c) { return c.r == 0; } };
// ---- segmented reduction by fan-in K ----------------------------------------------------------------
// largest key with toff[key] <= tid (toff has nkeys+1 entries, toff[nkeys] = total > tid)
__device__ __forceinline__ uint32_t find_key(const uint32_t* __restrict__ toff, uint32_t nkeys, uint32_t tid) {
  uint32_t lo = 0, hi = nkeys;  // invariant: toff[lo] <= tid < toff[hi]
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (toff[mid] <= tid) lo = mid; else hi = mid;
  }
  return lo;
}

// QUAD: one task per quad of lanes (quad-cooperative products, gf233.cuh).  A small MSM -- config #2's 2^16 points, the shard
// of one rank of an 8-way split -- has fewer reducer tasks than the chip has lanes and its fan-in-K chains are pure
// latency: 2.2x shorter per addition this way (the same trade k_merge<true> makes for the deep merge levels).
constexpr uint32_t ACCUM_QUAD_MAX = 49152;  // tasks: 4 lanes each = one chip-full of 3 blocks per CU
constexpr uint32_t ACCUM_HEX_MAX = 32768;   // tasks of the reducer's last level up to which 16 lanes per task win
template <bool INDIRECT, bool QUAD>
__global__ void __launch_bounds__(EC_TPB) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_accum_affine(const Aff* __restrict__ bases, const uint32_t* __restrict__ items, const uint32_t* __restrict__ cnt,
               const uint32_t* __restrict__ off, const uint32_t* __restrict__ toff, uint32_t nkeys, uint32_t K,
               Ld* __restrict__ out, uint32_t sign_mask) {
  extern __shared__ char lds_raw[];
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (QUAD) tid >>= 2;
  if (tid >= toff[nkeys]) return;
  uint32_t key = find_key(toff, nkeys, tid);
  uint32_t j = tid - toff[key];
  uint32_t start = off[key] + j * K;
  uint32_t len = min(K, cnt[key] - j * K);
  // INDIRECT: bases gathered through the sorted index list; otherwise `bases` is the compacted output of
  // the affine rounds, where x == 0 marks infinity
  auto item_point = [&](uint32_t v) -> Aff {  // table entry named by a sorted item; bit 31 (signed flavours) = subtract it
    Aff q = bases[v & ~sign_mask];
    if (v & sign_mask) q.y = gf_add(q.y, q.x);
    return q;
  };
  Aff first = INDIRECT ? item_point(items[start]) : bases[start];
  Ld acc = (!INDIRECT && gf_is_zero(first.x)) ? ld_infinity() : ld_from_aff(first);
  if (QUAD) {
    GfLdsQ L = gf_ldsq_init(lds_raw);
#pragma unroll 1
    for (uint32_t t = 1; t < len; ++t) {
      Aff q = INDIRECT ? item_point(items[start + t]) : bases[start + t];
      if (!INDIRECT && gf_is_zero(q.x)) continue;
      ld_madd_ip(acc, q, L);
    }
    if (L.r == 0) out[tid] = acc;
  } else {
    GfLdsK L = gf_ldsk_init(lds_raw);
#pragma unroll 1
    for (uint32_t t = 1; t < len; ++t) {
      Aff q = INDIRECT ? item_point(items[start + t]) : bases[start + t];
      if (!INDIRECT && gf_is_zero(q.x)) continue;
      ld_madd_ip(acc, q, L);
    }
    out[tid] = acc;
  }
}

// The same tasks without the exceptional branches (round 6; what k_bucket_pairs does for the fixed-base sizes): the first two entries
// are two AFFINE points (ld_add_aff_aff: Z1 = 1, 5M + 3S instead of 8M + 5S), the others go through ld_madd_fast; a task that meets
// an exceptional pair -- an infinity marker first, equal x -- goes on a list that k_accum_affine_rest redoes with the general
// formulas.  The general kernel kept the accumulator's exceptional paths alive on every lane: 253 VGPRs and 312-444 B of scratch
// per lane at two waves per SIMD.
template <bool INDIRECT, bool QUAD>
__global__ void __launch_bounds__(EC_TPB) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_accum_affine_fast(const Aff* __restrict__ bases, const uint32_t* __restrict__ items, const uint32_t* __restrict__ cnt,
                    const uint32_t* __restrict__ off, const uint32_t* __restrict__ toff, uint32_t nkeys, uint32_t K,
                    Ld* __restrict__ out, uint32_t sign_mask, uint32_t* __restrict__ rest_n, uint32_t* __restrict__ rest) {
  extern __shared__ char lds_raw[];
  using LT = typename std::conditional<QUAD, GfLdsQ, GfLdsK>::type;
  LT L = MergeMul<LT>::init(lds_raw);
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (QUAD) tid >>= 2;
  if (tid >= toff[nkeys]) return;
  const uint32_t key = find_key(toff, nkeys, tid);
  const uint32_t j = tid - toff[key];
  const uint32_t start = off[key] + j * K;
  const uint32_t len = min(K, cnt[key] - j * K);
  auto point = [&](uint32_t t) -> Aff {
    if (!INDIRECT) return bases[start + t];
    const uint32_t v = items[start + t];
    Aff q = bases[v & ~sign_mask];
    if (v & sign_mask) q.y = gf_add(q.y, q.x);
    return q;
  };
  const Aff first = point(0);
  bool ok = INDIRECT || !gf_is_zero(first.x);
  Ld acc = ld_from_aff(first);
  if (ok && len > 1) {
    const Aff q = point(1);
    ok = (INDIRECT || !gf_is_zero(q.x)) && !gf_eq(first.x, q.x);
    if (ok) ld_add_aff_aff(first, q, acc, L);
  }
#pragma unroll 1
  for (uint32_t t = 2; t < len && ok; ++t) {
    const Aff q = point(t);
    if (!INDIRECT && gf_is_zero(q.x)) continue;
    ok = ld_madd_fast(acc, q, L);
  }
  if (!MergeMul<LT>::lead(L)) return;
  if (ok)
    out[tid] = acc;
  else
    rest[atomicAdd(rest_n, 1u)] = tid;
}
// the listed tasks again, general formulas (one thread per task)
template <bool INDIRECT>
__global__ void __launch_bounds__(EC_TPB) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_accum_affine_rest(const Aff* __restrict__ bases, const uint32_t* __restrict__ items, const uint32_t* __restrict__ cnt,
                    const uint32_t* __restrict__ off, const uint32_t* __restrict__ toff, uint32_t nkeys, uint32_t K,
                    Ld* __restrict__ out, uint32_t sign_mask, const uint32_t* __restrict__ rest_n, const uint32_t* __restrict__ rest) {
  extern __shared__ char lds_raw[];
  GfLdsK L = gf_ldsk_init(lds_raw);
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *rest_n) return;
  const uint32_t tid = rest[r];
  const uint32_t key = find_key(toff, nkeys, tid);
  const uint32_t j = tid - toff[key];
  const uint32_t start = off[key] + j * K;
  const uint32_t len = min(K, cnt[key] - j * K);
  auto point = [&](uint32_t t) -> Aff {
    if (!INDIRECT) return bases[start + t];
    const uint32_t v = items[start + t];
    Aff q = bases[v & ~sign_mask];
    if (v & sign_mask) q.y = gf_add(q.y, q.x);
    return q;
  };
  const Aff first = point(0);
  Ld acc = (!INDIRECT && gf_is_zero(first.x)) ? ld_infinity() : ld_from_aff(first);
#pragma unroll 1
  for (uint32_t t = 1; t < len; ++t) {
    const Aff q = point(t);
    if (!INDIRECT && gf_is_zero(q.x)) continue;
    ld_madd_ip(acc, q, L);
  }
  out[tid] = acc;
}

// GROUP = lanes per task: 1, 4 (a quad) or 16 (a DPP row: the last level of a small MSM -- one or two additions per bucket, fewer
// tasks than the chip has rows -- is pure latency, and the sixteen-lane product is ~290 instructions against the quad's ~450)
template <int GROUP>
__global__ void __launch_bounds__(EC_TPB) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_accum_proj(const Ld* __restrict__ in, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off,
             const uint32_t* __restrict__ toff, uint32_t nkeys, uint32_t K, Ld* __restrict__ out) {
  extern __shared__ char lds_raw[];
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  tid /= (uint32_t)GROUP;
  if (tid >= toff[nkeys]) return;
  uint32_t key = find_key(toff, nkeys, tid);
  uint32_t j = tid - toff[key];
  uint32_t start = off[key] + j * K;
  uint32_t len = min(K, cnt[key] - j * K);
  Ld acc = in[start];
  using LT = typename std::conditional<GROUP == 16, GfLdsH, typename std::conditional<GROUP == 4, GfLdsQ, GfLdsK>::type>::type;
  LT L = MergeMul<LT>::init(lds_raw);
#pragma unroll 1
  for (uint32_t t = 1; t < len; ++t) {
    if (!ld_add_nodbl(acc, in[start + t], L)) {  // acc == the entry: double a copy read back
      acc = in[start + t];
      acc = ld_dbl(acc, L);
    }
  }
  if (MergeMul<LT>::lead(L)) out[tid] = acc;
}

// ---- what the pair rounds leave, one thread per BUCKET (fixed-base sizes: the rounds stop with 1-4 points per bucket) ----
// k_accum_affine gives every chunk of K entries a task and runs it as a serial chain of general mixed additions at two waves
// per SIMD (253 VGPRs: the exceptional branches of ld_madd_ip), behind a scan and a binary search per task and in front of a
// gather into the bucket array.  With the signed windows the load is bimodal -- the narrow windows reach only the lower half of
// the keys, which end the rounds with three points per bucket where the upper half has one -- so contiguous keys do the same
// work and a thread per bucket diverges no more than a thread per task did.  Here: the first two entries are two AFFINE
// points, so the mixed addition's Z1 = 1 and three of its eight products vanish (ld_add_aff_aff, 5M + 3S); further entries go
// through the mixed addition stripped of its exceptional branches (ld_madd_fast), straight into the bucket array (two waves
// per SIMD: 222 VGPRs; at three the loop spills 240 B per lane and the heavy half of the keys -- 1 024 workgroups -- is 1.33
// chip-fulls instead of exactly two: measured equal).  A bucket that meets an exceptional pair (an infinity marker first, equal x) goes on a list that k_bucket_rest
// redoes from its first entry with the general formulas.  No task scan, no search, no gather.
__global__ void __launch_bounds__(EC_TPB) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_bucket_pairs(const Aff* __restrict__ pts, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off, uint32_t nkeys,
               Ld* __restrict__ A, uint32_t* __restrict__ rest_n, uint32_t* __restrict__ rest) {
  extern __shared__ char lds_raw[];
  GfLdsK L = gf_ldsk_init(lds_raw);
  const uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
  if (key >= nkeys) return;
  const uint32_t c = cnt[key];
  if (c == 0) {
    A[key] = ld_infinity();
    return;
  }
  const uint32_t o = off[key];
  const Aff p = pts[o];
  if (c == 1) {
    A[key] = gf_is_zero(p.x) ? ld_infinity() : ld_from_aff(p);  // x == 0 marks infinity
    return;
  }
  Aff q = pts[o + 1];
  bool ok = !(gf_is_zero(p.x) || gf_is_zero(q.x) || gf_eq(p.x, q.x));
  Ld r;
  if (ok) {
    ld_add_aff_aff(p, q, r, L);
#pragma unroll 1
    for (uint32_t t = 2; t < c && ok; ++t) {
      q = pts[o + t];
      if (gf_is_zero(q.x)) continue;
      ok = ld_madd_fast(r, q, L);
    }
  }
  if (ok)
    A[key] = r;
  else
    rest[atomicAdd(rest_n, 1u)] = key;
}
__global__ void __launch_bounds__(EC_TPB) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_bucket_rest(const Aff* __restrict__ pts, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off,
              const uint32_t* __restrict__ rest_n, const uint32_t* __restrict__ rest, Ld* __restrict__ A) {
  extern __shared__ char lds_raw[];
  GfLdsK L = gf_ldsk_init(lds_raw);
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= *rest_n) return;
  const uint32_t key = rest[tid];
  const uint32_t c = cnt[key], o = off[key];
  const Aff first = pts[o];
  Ld acc = gf_is_zero(first.x) ? ld_infinity() : ld_from_aff(first);
#pragma unroll 1
  for (uint32_t t = 1; t < c; ++t) {
    const Aff q = pts[o + t];
    if (gf_is_zero(q.x)) continue;
    ld_madd_ip(acc, q, L);
  }
  A[key] = acc;
}

// ---- multi-squaring tables for the fast inversion (gf233.cuh) ----------------------------------------
__global__ void __launch_bounds__(256)
k_build_sqr_tables(Gf* __restrict__ tabs /* t29 | t58 | t116 | th | t14 | t7, 30 x 256 entries each */) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= 6 * 30 * 256) return;
  uint32_t which = tid / (30 * 256), e = tid - which * 30 * 256, pos = e >> 8, byte = e & 255;
  Gf v = gf_zero();
  v.w[pos >> 2] = byte << (8 * (pos & 3));
  if (pos == 29) v.w[7] &= 0x1FFu;  // bits >= 233 never occur in a reduced element
  if (which == 3) {
    tabs[tid] = gf_halftrace(v);
    return;
  }
  const int k = which == 0 ? 29 : which == 1 ? 58 : which == 2 ? 116 : which == 4 ? 14 : 7;
  tabs[tid] = gf_sqr_n(v, k);
}
static std::mutex g_sqr_mu;
static Gf* g_sqr_tab[16] = {nullptr};
int gf_sqr_tables(GfSqrTables* out, hipStream_t st) {
  int dev;
  DVP_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return DVP_EINVAL;
  std::lock_guard<std::mutex> g(g_sqr_mu);
  if (!g_sqr_tab[dev]) {
    Gf* t;
    DVP_HIP(hipMalloc((void**)&t, (size_t)6 * 30 * 256 * sizeof(Gf)));
    hipLaunchKernelGGL(k_build_sqr_tables, dim3(cdiv(6 * 30 * 256, 256)), dim3(256), 0, st, t);
    DVP_HIP(hipGetLastError());
    DVP_HIP(hipStreamSynchronize(st));
    g_sqr_tab[dev] = t;
  }
  out->t29 = g_sqr_tab[dev];
  out->t58 = g_sqr_tab[dev] + 30 * 256;
  out->t116 = g_sqr_tab[dev] + 2 * 30 * 256;
  out->th = g_sqr_tab[dev] + 3 * 30 * 256;
  const long long tabs = tune().gf_inv_tabs;  // 0: the three long runs only (rounds 1-5), 1: + the run of 14, 2: + the run of 7
  out->t14 = tabs >= 1 ? g_sqr_tab[dev] + 4 * 30 * 256 : nullptr;
  out->t7 = tabs >= 2 ? g_sqr_tab[dev] + 5 * 30 * 256 : nullptr;
  return DVP_OK;
}

// ---- batched-affine pair rounds --------------------------------------------------------------------
// One round halves every bucket: output slot (key, j) = in[2j] + in[2j+1] (or in[2j] alone when the
// count is odd), all in AFFINE coordinates.  An affine addition needs one field inversion; a thread owns
// B output slots and shares ONE inversion among them (Montgomery's trick): per addition 3 products for
// the trick + 2 products + 1 squaring for the chord/tangent formulas, against 8M + 5S for a mixed
// projective addition.  Prefix products are parked in HBM ([slot-in-thread][thread] layout, so the
// traffic is coalesced).  (0,0) -- not a curve point -- marks infinity.
//
// A round after the first is two launches.  k_round_desc resolves every output slot to its two input points once --
// (a, b) = indices into `pts`, b = NONE for the odd leftover of a bucket -- with a group of lanes per bucket, so the
// hot kernel does not search the bucket offsets (18 dependent L2 round trips per slot in the first version); the first
// round reads the sorted item list itself as its descriptors (below).  k_affine_round then runs two software-pipelined passes over its slots: the descriptor of
// slot k+2 and the operands of slot k+1 are in flight while slot k multiplies, so the random 64-byte gathers of the
// bases (HBM misses in the first round: the pre-rotated table is 5 GB) are off the critical path.
constexpr uint32_t AFF_NONE = 0xffffffffu;
// slot descriptor of the rounds after the first: the slot adds points i and i + 1 of the previous round's output (DESC_PAIR set) or
// passes point i on (a bucket's odd leftover); i < 2^31.  The first round reads the sorted item list as (a, b) pairs instead.
constexpr uint32_t DESC_PAIR = 0x80000000u;

// Both sorts lay every bucket out from an EVEN position of the item list (scan of the counts rounded up to
// even); an odd bucket's spare slot gets AFF_NONE.  The list read as uint2 pairs is then exactly the descriptor array of
// the first pair round -- (a, b) table indices, b = NONE for the odd leftover -- so that round needs no descriptor
// kernel and no scan: its output offsets are the item offsets halved (k_round0_offsets).
// the three per-key passes between the sort and the first pair round of an MSM whose bookkeeping is not done by the all-rounds scan (small
// one-shot MSMs) as ONE launch (round 6): the NONE behind every odd bucket, the first round's counts and offsets, the largest bucket
__global__ void __launch_bounds__(256)
k_post_sort(uint32_t* __restrict__ items, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off /* nkeys + 1, even */, uint32_t nkeys,
            uint32_t* __restrict__ ocnt, uint32_t* __restrict__ ooff /* nkeys + 1 */, uint32_t* __restrict__ d_max) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t m = 0;
  if (k <= nkeys) {
    const uint32_t o = off[k];
    ooff[k] = o >> 1;
    if (k < nkeys) {
      const uint32_t c = cnt[k];
      if (c & 1) items[o + c] = AFF_NONE;
      ocnt[k] = (c + 1) >> 1;
      m = c;
    }
  }
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(d_max, m);
}
__global__ void __launch_bounds__(256)
k_round0_offsets(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off /* nkeys + 1, even */, uint32_t nkeys,
                 uint32_t* __restrict__ ocnt, uint32_t* __restrict__ ooff /* nkeys + 1 */) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > nkeys) return;
  ooff[k] = off[k] >> 1;
  if (k < nkeys) ocnt[k] = (cnt[k] + 1) >> 1;
}

// ---- the bookkeeping of ALL pair rounds at once (round 4) --------------------------------------------------------------------
// Round r's input counts are c_r = ceil(c_{r-1} / 2) of the sort's counts c_0, its offsets their exclusive scan: R scans that depend
// on nothing but c_0, so one three-launch scan carries all of them (a column of R running sums per key) where the rounds used to
// pay three launches each.  rp: (c_r, o_r) for r = 1 .. R, each nkeys + 1 words (o_r[nkeys] = total), at rp + 2 (r - 1) (nkeys + 1).
__global__ void __launch_bounds__(SCAN_TPB) k_mscan_local(const uint32_t* __restrict__ cnt0, uint32_t m, int R, uint32_t* __restrict__ rp,
                                                          uint32_t* __restrict__ bs /* [R][nb], then nb block maxima of cnt0 */) {
  __shared__ uint32_t sh[SCAN_TPB];
  const uint32_t base = blockIdx.x * SCAN_BLK + threadIdx.x * SCAN_EPT;
  uint32_t v[SCAN_EPT];
#pragma unroll
  for (int k = 0; k < SCAN_EPT; ++k) v[k] = (base + k < m) ? cnt0[base + k] : 0;
  {  // the largest count of this block (k_mscan_bsums reduces the blocks' maxima: no atomics -- 2 048 waves on one word cost 20 us)
    uint32_t mx = 0;
#pragma unroll
    for (int k = 0; k < SCAN_EPT; ++k) mx = max(mx, v[k]);
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < SCAN_TPB / 64; ++w) mx = max(mx, sh[w]);
      bs[(size_t)R * gridDim.x + blockIdx.x] = mx;
    }
    __syncthreads();
  }
  for (int r = 1; r <= R; ++r) {
    uint32_t* c = rp + (size_t)(2 * (r - 1)) * ((size_t)m + 1);
    uint32_t* o = c + ((size_t)m + 1);
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_EPT; ++k) {
      v[k] = (v[k] + 1u) >> 1;
      s += v[k];
    }
    uint32_t tot;
    uint32_t ex = block_exclusive_scan(s, &tot, sh);
#pragma unroll
    for (int k = 0; k < SCAN_EPT; ++k) {
      if (base + k < m) {
        c[base + k] = v[k];
        o[base + k] = ex;
      }
      ex += v[k];
    }
    if (threadIdx.x == 0) bs[(size_t)(r - 1) * gridDim.x + blockIdx.x] = tot;
  }
}
__global__ void __launch_bounds__(SCAN_TPB) k_mscan_bsums(uint32_t* __restrict__ bs, uint32_t nb, uint32_t m, uint32_t* __restrict__ rp,
                                                          uint32_t* __restrict__ off0 /* item offsets (or nullptr): off0[m] = the padded total */,
                                                          uint32_t* __restrict__ dmax /* the largest count (or nullptr) */) {
  __shared__ uint32_t sh[SCAN_TPB];
  const int r = blockIdx.x + 1;
  uint32_t* row = bs + (size_t)(r - 1) * nb;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nb; base += SCAN_TPB) {
    uint32_t idx = base + threadIdx.x;
    uint32_t v = idx < nb ? row[idx] : 0;
    uint32_t tot;
    uint32_t ex = block_exclusive_scan(v, &tot, sh);
    if (idx < nb) row[idx] = ex + carry;
    carry += tot;
  }
  if (threadIdx.x == 0) {
    rp[(size_t)(2 * (r - 1) + 1) * ((size_t)m + 1) + m] = carry;  // o_r[nkeys]
    if (r == 1 && off0) off0[m] = 2u * carry;
  }
  if (r == 1 && dmax) {  // the largest bucket, from the block maxima k_mscan_local left behind the R rows of block sums
    const uint32_t* bm = bs + (size_t)gridDim.x * nb;
    uint32_t mx = 0;
    for (uint32_t i = threadIdx.x; i < nb; i += SCAN_TPB) mx = max(mx, bm[i]);
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < SCAN_TPB / 64; ++w) mx = max(mx, sh[w]);
      *dmax = mx;
    }
  }
}
// Round 5: the last launch of the scan also finishes what the SORT needs from the counts, which used to be six launches of their own
// (a second three-launch scan, k_pad_odd_buckets, a memset and k_max_u32): the item offsets off0 -- the scan of the counts rounded up
// to even is exactly twice o_1, the scan of ceil(c_0 / 2) --, the NONE in the spare slot of every odd bucket (a position the
// scatter never writes, so it may be filled before the scatter runs); the largest bucket comes out of k_mscan_local's block maxima
// (k_mscan_bsums).
__global__ void __launch_bounds__(SCAN_TPB) k_mscan_add(uint32_t* __restrict__ rp, const uint32_t* __restrict__ bs, uint32_t m, int R,
                                                        const uint32_t* __restrict__ cnt0, uint32_t* __restrict__ off0, uint32_t* __restrict__ items) {
  const uint32_t base = blockIdx.x * SCAN_BLK + threadIdx.x * SCAN_EPT;
  for (int r = 1; r <= R; ++r) {
    uint32_t* o = rp + (size_t)(2 * (r - 1) + 1) * ((size_t)m + 1);
    const uint32_t add = bs[(size_t)(r - 1) * gridDim.x + blockIdx.x];
    if (r == 1 && off0) {
#pragma unroll
      for (int k = 0; k < SCAN_EPT; ++k)
        if (base + k < m) {
          const uint32_t v = o[base + k] + add, c = cnt0[base + k];
          o[base + k] = v;
          off0[base + k] = 2u * v;
          if (c & 1u) items[2u * v + c] = AFF_NONE;
        }
      continue;
    }
#pragma unroll
    for (int k = 0; k < SCAN_EPT; ++k)
      if (base + k < m) o[base + k] += add;
  }
}
// slot descriptors of rounds 1 .. R - 1 in one launch: LPK lanes per bucket walk its outputs round by round
// (round r pairs the outputs of round r - 1: inputs at o_r[key] + 2 j, + 1; outputs at o_(r+1)[key] + j)
struct DescPlan {
  unsigned long long at[40];  // round r's descriptors start at desc + at[r]
};
template <int LPK>
__global__ void __launch_bounds__(1024)
k_round_desc_all(const uint32_t* __restrict__ rp, uint32_t nkeys, int R, DescPlan plan, uint32_t* __restrict__ desc) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t key = gid / LPK, lane = gid % LPK;
  if (key >= nkeys) return;
  const size_t stride = (size_t)nkeys + 1;
  // the counts and offsets of up to eight rounds are asked for at once and the descriptors written afterwards: a round's three loads
  // used to sit in front of its stores, round after round
  for (int r0 = 1; r0 < R; r0 += 8) {
    uint32_t cr[8], o[8], oo[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = r0 + q;
      if (r < R) {
        const uint32_t* c = rp + (size_t)(2 * (r - 1)) * stride;
        cr[q] = c[key];
        o[q] = c[stride + key];
        oo[q] = c[3 * stride + key];
      } else {
        cr[q] = 0; o[q] = 0; oo[q] = 0;
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = r0 + q;
      if (r >= R) break;
      const uint32_t nout = (cr[q] + 1) >> 1;
      uint32_t* d = desc + plan.at[r];
      for (uint32_t j = lane; j < nout; j += LPK) d[oo[q] + j] = (o[q] + 2 * j) | ((2 * j + 1 < cr[q]) ? DESC_PAIR : 0u);
    }
  }
}

template <int LPK>
__global__ void __launch_bounds__(256)
k_round_desc(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off, const uint32_t* __restrict__ ooff /* scan of ceil(cnt/2), nkeys+1 */,
             uint32_t nkeys, uint32_t* __restrict__ desc) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t key = gid / LPK, lane = gid % LPK;
  if (key >= nkeys) return;
  const uint32_t c = cnt[key], o = off[key], oo = ooff[key], nout = (c + 1) >> 1;
  for (uint32_t j = lane; j < nout; j += LPK) desc[oo + j] = (o + 2 * j) | ((2 * j + 1 < c) ? DESC_PAIR : 0u);
}

// Thread t of T = ceil(total / B) owns output slots s = k*T + t, k < B (lane-consecutive slots: coalesced
// descriptors, outputs, prefix products and -- after the first round -- inputs).
// B (slots per thread = additions per shared inversion) is chosen ON THE DEVICE from the round's size so that the
// grid is a whole number of chip-fulls: all threads of a round take the same time, so with a fixed B the last
// partial wave of blocks ran on a mostly idle chip (2.2 chip-fulls cost 3: the first version averaged 64 % of its
// 12 waves per CU).  With `cap` = resident threads of the chip, R = ceil(total / (cap * AFF_BMAX)) chip-fulls of
// B = ceil(total / (cap * R)) slots each: B is 33..136 in the big rounds (25..48 through round 4) and shrinks to 1..8 in the last ones, where a
// round is pure latency and short threads are what is wanted.
// Round 4 (tools/wave_trace.py, profiles/r04_wave_trace_*): the instruction arbiter of a SIMD serves its OLDEST wave first.  Of
// three resident waves with identical work the first finished after 1.32 ms, the second after 1.55, the third after 1.93; every
// later chip-full inherited the stagger and the end of each launch drained through SIMDs with two and then one wave left --
// that, not gaps between workgroups (a freed slot was refilled in 10-15 us) nor the launch ramp, was the whole of the missing
// resident-wave fraction (0.83-0.88).  Rotating the user priority with the slot index hands the issue slots round: the three
// waves of a SIMD now finish within 2 % of each other, a one-chip-full round ends 10 % sooner (1.58 -> 1.41 ms).
#ifndef AFF_PRIO
#define AFF_PRIO 1
#endif
#ifndef AFF_XY
#define AFF_XY 1
#endif
__device__ __forceinline__ void aff_set_prio(uint32_t v) {
  switch (v % 3u) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    default: __builtin_amdgcn_s_setprio(2); break;
  }
}
// default (Tune::msm_aff_bmax).  48 through round 4 (swept flat 32 .. 128 then); re-swept at the end of round 5, with the product 15 %
// shorter: 48 / 64 / 68 / 72 / 136 / 255 -> 19.14 / 19.15 / 18.99 / 18.99 / 18.98 / 19.00 ms per 2^20 proof and 66.8 / - / - / 66.0 / 64.7-65.0 /
// 65.0 ms at 2^22 -- from 68 the first round of the commitment MSM and the second of the K MSM are ONE chip-full of 65 slots per thread
// instead of two of 33, from 130 the K MSM's first round is one of 129: fewer drains and a smaller inversion share
constexpr uint32_t AFF_BMAX = 136;
// bmax_bmin: bits 0..7 = most slots per thread, bits 8..15 = fewest (a small round then runs on fewer threads, each sharing its
// inversion among more additions: Tune::msm_aff_bmin)
__device__ __forceinline__ uint32_t aff_slots_per_thread(uint32_t total, uint32_t cap, uint32_t bmax_bmin) {
  const uint32_t bmax = bmax_bmin & 0xffu, bmin = (bmax_bmin >> 8) & 0xffu;
  const uint32_t units = (total + cap - 1) / cap;  // slots per resident thread if the round were one chip-full
  const uint32_t R = (units + bmax - 1) / bmax;
  const uint32_t B = R ? (units + R - 1) / R : 1;
  return B < bmin ? bmin : B;
}
// FIRST only names the launch: k_affine_round<true> is the first pair round of an MSM (random gathers out of the
// pre-rotated table -- the dominant kernel bench.py's roofline block is about), <false> the later rounds (coalesced
// inputs); the code is the same, the two symbols keep them apart in rocprofv3's per-kernel statistics.
// Wave-level trace of a pair round (diagnostics only: dvp_debug_wave_trace, tools/wave_trace.py).  TRACE instantiations stamp
// s_memrealtime (100 MHz, constant) and s_memtime (shader clock) at the start of a wave, after its first pass, after the shared
// inversion and at its end, with the hardware slot the wave ran in (HW_ID: wave, SIMD, CU, SH, SE; XCC_ID): one 64-byte record
// per wave behind a 64-byte header whose first word is the record counter.  The production launches use TRACE = false.
struct WaveTrace {
  unsigned long long* buf;  // [0] = records written, records from word 8
  uint32_t cap, tag;
};
template <bool FIRST, bool TRACE = false>
__global__ void __launch_bounds__(EC_TPB) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_affine_round(const Aff* __restrict__ pts, const uint2* __restrict__ desc, const uint32_t* __restrict__ total_ptr /* ooff[nkeys] */,
               uint32_t cap, uint32_t bmax, GfSqrTables T, Gf* __restrict__ prefix, Aff* __restrict__ out, uint32_t sign_mask, WaveTrace wt) {
  extern __shared__ char lds_raw[];
  unsigned long long tr_t0 = 0, tr_c0 = 0, tr_t1 = 0, tr_t2 = 0;
  if (TRACE) { tr_t0 = wall_clock64(); tr_c0 = clock64(); }
  GfLdsK L = gf_ldsk_init(lds_raw);
#if AFF_PRIO
  const uint32_t wslot = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 4);  // HW_ID.wave_id: this wave's slot in its SIMD
#endif
  // signed-digit flavours (first round only: the operands are table entries named by sorted items): bit 31 of an item says
  // "subtract"; -(x, y) = (x, x + y), applied as the y-coordinate is loaded
  const uint32_t sm = FIRST ? sign_mask : 0u;
  auto ldx = [&](uint32_t v) -> Gf { return pts[v & ~sm].x; };
  auto ldy = [&](uint32_t v, const Gf& x) -> Gf {
    Gf y = pts[v & ~sm].y;
    if (v & sm) y = gf_add(y, x);
    return y;
  };
  const uint32_t total = *total_ptr;
  const int B = (int)aff_slots_per_thread(total, cap, bmax);
  const uint32_t nthr = (total + B - 1) / B;
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= nthr) return;
  const uint2 none = make_uint2(AFF_NONE, AFF_NONE);
  // (round 6: the load is unconditional from a clamped index and the VALUE is selected.  Returning `none` from the early exits made
  // hipcc select between two ADDRESSES -- &desc[sk] and a private copy of `none`: 16 B of scratch per lane and, worse, a FLAT load per
  // descriptor, which counts on lgkmcnt beside the multiplier's LDS reads)
  auto ld_desc = [&](int k) -> uint2 {
    const uint32_t sk = (uint32_t)k * nthr + tid;
    const bool ok = k >= 0 && k < B && sk < total;
    if (FIRST) {
      uint2 v = none;
      if (ok) v = desc[sk];
      return v;
    }
    const uint32_t d = ((const uint32_t*)desc)[ok ? sk : 0u], a = d & ~DESC_PAIR;  // later rounds: one word per slot
    return make_uint2(ok ? a : AFF_NONE, (ok && (d & DESC_PAIR)) ? a + 1 : AFF_NONE);
  };
  const Gf one = gf_one();
  // pass 1: denominators and running product (x-coordinates only; y is touched when x1 == x2)
  Gf run = one;
  {
    uint2 d0 = ld_desc(0), d1 = ld_desc(1);
    Gf xa = gf_zero(), xb = gf_zero();
    if (d0.y != AFF_NONE) { xa = ldx(d0.x); xb = ldx(d0.y); }
#pragma unroll 1
    for (int k = 0; k < B; ++k) {
#if AFF_PRIO
      aff_set_prio(wslot + (uint32_t)k);
#endif
      const uint2 d2 = ld_desc(k + 2);
      Gf nxa = gf_zero(), nxb = gf_zero();
      if (d1.y != AFF_NONE) { nxa = ldx(d1.x); nxb = ldx(d1.y); }
      Gf den = one;
      if (d0.y != AFF_NONE && !gf_is_zero(xa) && !gf_is_zero(xb)) {
        Gf dd = gf_add(xa, xb);
        if (!gf_is_zero(dd)) den = dd;
        else if (gf_eq(ldy(d0.x, xa), ldy(d0.y, xb))) den = xa;  // doubling: lambda = x + y/x
      }
      prefix[(size_t)k * nthr + tid] = run;
      run = gf_mul(run, den, L);
      d0 = d1; d1 = d2; xa = nxa; xb = nxb;
    }
  }
  if (TRACE) tr_t1 = wall_clock64();
  Gf inv = gf_inv_fast(run, T, L);
  if (TRACE) tr_t2 = wall_clock64();
  // pass 2 (backwards): recover each inverse and finish the addition
  {
    uint2 e0 = ld_desc(B - 1), e1 = ld_desc(B - 2);
    Gf px = gf_zero(), qx = gf_zero(), pre = one;
    if (e0.x != AFF_NONE) px = ldx(e0.x);
    if (e0.y != AFF_NONE) { qx = ldx(e0.y); pre = prefix[(size_t)(B - 1) * nthr + tid]; }
#if AFF_XY
    Gf py = gf_zero(), qy = gf_zero();
    if (e0.x != AFF_NONE) py = ldy(e0.x, px);
    if (e0.y != AFF_NONE) qy = ldy(e0.y, qx);
#endif
#pragma unroll 1
    for (int k = B - 1; k >= 0; --k) {
#if AFF_PRIO
      aff_set_prio(wslot + (uint32_t)k);
#endif
      const uint2 e2 = ld_desc(k - 2);
#if AFF_XY
      // the next slot's WHOLE points and prefix product: x and y of a point share a 128-byte line, and a y fetched one slot
      // after its x (the round-3 pipeline) found the line evicted from the L2 again: 6.3 line requests per addition, 4.3 now
      Gf npx = gf_zero(), nqx = gf_zero(), npy = gf_zero(), nqy = gf_zero(), npre = one;
      if (e1.x != AFF_NONE) { npx = ldx(e1.x); npy = ldy(e1.x, npx); }
      if (e1.y != AFF_NONE) { nqx = ldx(e1.y); nqy = ldy(e1.y, nqx); npre = prefix[(size_t)(k - 1) * nthr + tid]; }
#else
      // this slot's y-coordinates (needed after the two products of the inverse recovery) ...
      Gf py = gf_zero(), qy = gf_zero();
      if (e0.x != AFF_NONE) py = ldy(e0.x, px);
      if (e0.y != AFF_NONE) qy = ldy(e0.y, qx);
      // ... and the next slot's x-coordinates and prefix product
      Gf npx = gf_zero(), nqx = gf_zero(), npre = one;
      if (e1.x != AFF_NONE) npx = ldx(e1.x);
      if (e1.y != AFF_NONE) { nqx = ldx(e1.y); npre = prefix[(size_t)(k - 1) * nthr + tid]; }
#endif
      if (e0.x != AFF_NONE) {
        const uint32_t sidx = (uint32_t)k * nthr + tid;
        Gf ox = px, oy = py;  // odd leftover, or q == infinity: pass p through
        if (e0.y != AFF_NONE) {
          if (gf_is_zero(px)) { ox = qx; oy = qy; }
          else if (!gf_is_zero(qx)) {
            Gf dd = gf_add(px, qx);
            const bool same_x = gf_is_zero(dd);
            const bool dbl = same_x && gf_eq(py, qy);
            if (same_x && !dbl) {  // p == -q
              ox = gf_zero(); oy = gf_zero();
            } else {
              Gf den = dbl ? px : dd;
              Gf dinv, inv_next;
              gf_mul2(pre, den, inv, L, dinv, inv_next);  // 1/den, and the running inverse stripped of this slot's factor
              inv = inv_next;
              Gf num = dbl ? py : gf_add(py, qy);
              Gf lam = gf_mul(num, dinv, L);
              if (dbl) lam = gf_add(lam, px);
              ox = gf_add(gf_add(gf_sqr(lam), lam), dd);  // dd == 0 when doubling (curve a = 0)
              // y3 = lam (x1 + x3) + x3 + y1   (the same expression covers the doubling)
              oy = gf_add(gf_add(gf_mul(gf_add(px, ox), lam, L), ox), py);
            }
          }
        }
        out[sidx].x = ox; out[sidx].y = oy;
      }
      e0 = e1; e1 = e2; px = npx; qx = nqx; pre = npre;
#if AFF_XY
      py = npy; qy = nqy;
#endif
    }
  }
#if AFF_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  if (TRACE) {
    const unsigned long long t3 = wall_clock64(), c3 = clock64();
    if ((threadIdx.x & 63u) == 0) {
      const unsigned long long i = atomicAdd(wt.buf, 1ull);
      if (i < wt.cap) {
        unsigned long long* r = wt.buf + 8 + 8 * i;
        r[0] = tr_t0; r[1] = tr_t1; r[2] = tr_t2; r[3] = t3;
        r[4] = tr_c0; r[5] = c3;
        r[6] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);  // HW_ID | XCC_ID
        r[7] = (unsigned long long)blockIdx.x | ((unsigned long long)wt.tag << 32) | ((unsigned long long)(uint32_t)B << 48);
      }
    }
  }
}

// dense bucket array from the last affine round: A[key] = (x, y, 1) or infinity
__global__ void __launch_bounds__(256)
k_bucket_gather_aff(const Aff* __restrict__ in, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off,
                    uint32_t nkeys, Ld* __restrict__ A) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nkeys) return;
  Ld r = ld_infinity();
  if (cnt[k]) {
    Aff p = in[off[k]];
    if (!gf_is_zero(p.x)) r = ld_from_aff(p);  // x == 0 marks infinity (no point of E[r] has x = 0)
  }
  A[k] = r;
}
// same, straight from the sorted items (keys that never had more than one point)
__global__ void __launch_bounds__(256)
k_bucket_gather_items(const Aff* __restrict__ bases, const uint32_t* __restrict__ items, const uint32_t* __restrict__ cnt,
                      const uint32_t* __restrict__ off, uint32_t nkeys, Ld* __restrict__ A, uint32_t sign_mask) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nkeys) return;
  Ld r = ld_infinity();
  if (cnt[k]) {
    const uint32_t v = items[off[k]];
    Aff q = bases[v & ~sign_mask];
    if (v & sign_mask) q.y = gf_add(q.y, q.x);
    r = ld_from_aff(q);
  }
  A[k] = r;
}

// dense bucket array: A[key] = the single remaining item of key, or infinity
__global__ void __launch_bounds__(256)
k_bucket_gather(const Ld* __restrict__ in, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off,
                uint32_t nkeys, Ld* __restrict__ A) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nkeys) return;
  A[k] = cnt[k] ? in[off[k]] : ld_infinity();
}

// ---- pruned sum-over-subsets merge (see file header, step 4) ---------------------------------------
// level j: blocks of 2^(j+1) buckets; slot s <= j: A[base+s] += A[base+2^j+s]; slot j+1 <- T_right
// QUAD: the deep levels have far fewer additions than the chip has lanes and are pure latency; there the four lanes
// of a quad share one addition (gf233.cuh, quad-cooperative product), 2.2x shorter per level.
// Round 4: the tree runs in lambda-projective coordinates (k233.cuh: 11M + 2S per full addition against 13M + 5S).  Level 0
// reads every bucket exactly once (as the left or the right operand of its pair) and converts it on the way in -- (X, Y, Z) ->
// (X^2, X^2 + Y, X Z), 1M + 1S -- so the right operand, which stays in place as D_0 of its block, is written back converted;
// k_tail converts what it reads back to Lopez-Dahab (1M) for its doublings.
// GROUP = lanes per addition: 1 (wide levels, Karatsuba multiplier), 4 (a quad, <= MERGE_QUAD_MAX additions) or 16 (a DPP row,
// <= MERGE_HEX_MAX additions: the deepest levels are pure latency and a lone wave pays per instruction, gf233.cuh)
template <int GROUP, bool FIRST>
__global__ void __launch_bounds__(EC_TPB) __attribute__((amdgpu_waves_per_eu(GROUP > 1 ? 1 : 2, 2))) k_merge(Ld* __restrict__ A, int j, uint32_t total /* nblocks*(j+1) */, Ld* __restrict__ save0 /* FIRST: where bucket 0 is kept as it was (or nullptr) */) {
  using LT = typename std::conditional<GROUP == 1, GfLdsK, typename std::conditional<GROUP == 4, GfLdsQ, GfLdsH>::type>::type;
  extern __shared__ char lds_raw[];
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  tid /= (uint32_t)GROUP;
  if (tid >= total) return;
  uint32_t blk = tid / (uint32_t)(j + 1), s = tid - blk * (uint32_t)(j + 1);
  size_t base = (size_t)blk << (j + 1);
  Ld l = A[base + s], r = A[base + ((size_t)1 << j) + s];
  LT L = MergeMul<LT>::init(lds_raw);
  const bool lead = MergeMul<LT>::lead(L);  // the lane of the group that stores
  if (FIRST) {  // j == 0
    // signed windows: bucket 0 (the digits of magnitude 2^(c-1)) is weighed separately by k_tail; the merge turns slot 0 into the total
    if (save0 && tid == 0 && lead) *save0 = l;
    lam_from_ld(l, L);
    lam_from_ld(r, L);
  }
  if (lead && s == 0 && j != 1) A[base + 1 + j] = r;  // j = 0: the converted right operand; j = 1: T_right is in place already
  if (!lam_add_ip(l, r, L)) {  // l == r (equal bucket sums: equal bases and scalars): double a copy read back
    l = A[base + s];
    if (FIRST) lam_from_ld(l, L);
    lam_dbl_ip(l, L);
  }
  if (lead) A[base + s] = l;
}

// Round 6, one-shot MSMs (W windows x c points = 240 at 2^16 points): the Frobenius powers and the first four levels of the add tree
// on MANY workgroups -- a workgroup takes 16 consecutive points, one row of 16 lanes each (the sixteen-lane product, ~290 instructions
// against the quad's ~450), and leaves their sum in part[blockIdx.x]; k_tail then adds the <= 16 group sums (n_narrow == -4) and
// converts.  The single-workgroup tail walked 120 + 60 + 30 + 15 additions in passes of 64 quads: 218 us of its own at 2^16 points.
// The sum is the same group element in another order: the affine result and its encoding are unchanged bit for bit.
__global__ void __launch_bounds__(EC_TPB) k_tail_groups(const Ld* __restrict__ A, int c, int W, int n_narrow, GfSqrTables T, Ld* __restrict__ buf /* 2 W c */,
                                                        Ld* __restrict__ part) {
  extern __shared__ char lds_raw[];
  GfLdsH H = gf_ldsh_init(lds_raw);
  const uint32_t cnt0 = (uint32_t)(W * c), g0 = blockIdx.x * 16u;
  const uint32_t cntl = min(16u, cnt0 - g0);
  Ld* in = buf + g0;
  Ld* out = buf + cnt0 + g0;
  {
    const uint32_t row = threadIdx.x >> 4;
    if (row < cntl) {  // whole rows together
      const uint32_t tid = g0 + row;
      const uint32_t w = tid / (uint32_t)c, t = tid - w * (uint32_t)c;
      const int k = (int)(w * (uint32_t)c) - min((int)w, n_narrow) + (int)t;
      Ld p = A[((size_t)w << c) + 1 + t];
      lam_to_ld(p, H);  // lambda-projective (the merge tree's output) -> Lopez-Dahab: Y = X (L + X)
      // the three coordinates' Frobenius powers side by side: the 16 lanes of the row hold the same point, lanes 0-4 / 5-9 / 10-15 take
      // X / Y / Z through the table passes (a serial chain of L2 round trips each) and the row gathers them again
      const uint32_t sel = H.r < 5u ? 0u : (H.r < 10u ? 1u : 2u);
      Gf v = sel == 0u ? p.X : (sel == 1u ? p.Y : p.Z);
      v = gf_sqr_n_fast(v, k, T);
      const int lane0 = (int)(threadIdx.x & 63u) - (int)H.r;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        p.X.w[q] = (uint32_t)__shfl((int)v.w[q], lane0);
        p.Y.w[q] = (uint32_t)__shfl((int)v.w[q], lane0 + 5);
        p.Z.w[q] = (uint32_t)__shfl((int)v.w[q], lane0 + 10);
      }
      if (H.r == 0) in[row] = p;
    }
  }
  __syncthreads();
  uint32_t cnt = cntl;
  while (cnt > 1) {
    const uint32_t half = (cnt + 1) / 2;
    const uint32_t i = threadIdx.x >> 4;
    if (i < half) {
      Ld a = in[2 * i];
      if (2 * i + 1 < cnt) ld_add_ip(a, in[2 * i + 1], H);
      if (H.r == 0) out[i] = a;
    }
    __syncthreads();
    Ld* t = in; in = out; out = t;
    cnt = half;
  }
  if (threadIdx.x == 0) part[blockIdx.x] = in[0];
}

// The whole Frobenius tail of an MSM in ONE single-workgroup launch: E[w*c+t] = tau^k(A[w*2^c + 1 + t]) with k = the first
// digit of window w plus t (window w starts at digit w*c - min(w, n_narrow): the first n_narrow windows are c-1 digits wide;
// up to 239 squarings per coordinate, done by table passes, not a serial chain), then the pairwise add tree (in[count]
// treated as infinity when count is odd), then the projective -> affine conversion.  Every step is pure latency -- 18..240
// points, log-depth -- so seven launches (each a ~5 us boundary plus a grid ramp) buy nothing over __syncthreads between
// the levels of one 256-thread block (64 quads, one addition per quad and pass).  buf: 2 * cnt Ld of scratch.
// n_narrow == -3: signed aligned windows over the binary digits (the default fixed-base flavour, W = 1): key = |digit|, so the
// result is sum_t 2^t A[1 + t], t < c - 1, plus 2^(c-1) x bucket 0 (doublings instead of Frobenius powers).
// out_enc (optional): the result's 30-byte encoding under `rule` as well (what k_encode_point would make of out_xy): the affine
// conversion and the encoding's 1 / x share ONE inversion, 1 / (X Z) -- a proof's two commitments are encoded right here instead
// of after a host round trip, a launch and a second 45 us inversion chain.
// HEX (round 4, the signed flavour's c points): a DPP row of 16 lanes per point instead of a quad -- the doublings, the add tree and the
// final inversion are one dependent chain of ~140 products, and a lone wave pays per instruction (gf233.cuh: ~290 against ~450 per
// product).  16 rows per workgroup: the points go out longest chain first, the few left over to the rows with the shortest ones.
template <bool HEX>
__global__ void __launch_bounds__(EC_TPB) k_tail(const Ld* __restrict__ A, int c, int W, int n_narrow, GfSqrTables T, Ld* __restrict__ buf,
                                                 uint32_t* __restrict__ out_xy, uint32_t* __restrict__ out_inf, uint8_t* __restrict__ out_enc, int rule,
                                                 unsigned long long* __restrict__ ctrl /* msm_core's control words: [0] err, [1] d_max | rest_n, [2] err_out */, unsigned long long* __restrict__ err_dst) {
  using LT = typename std::conditional<HEX, GfLdsH, GfLdsQ>::type;
  constexpr int GS = HEX ? 4 : 2;             // log2 lanes per point
  constexpr uint32_t NG = EC_TPB >> GS;       // points in flight per pass
  extern __shared__ char lds_raw[];
  LT L = MergeMul<LT>::init(lds_raw);
  const uint32_t grp = threadIdx.x >> GS;
  const uint32_t cnt0 = (uint32_t)(W * c);
  Ld* in = buf;
  Ld* out = buf + cnt0;
  if (n_narrow == -4) {
    // the cnt0 points are in buf already (the group sums of k_tail_groups, Frobenius powers applied): only the tree and the conversion
  } else if (n_narrow == -3) {
    // signed aligned windows: t doublings of A[1 + t], one point per quad of lanes (a serial chain of <= c - 1 doublings at
    // 3 products + 5 squarings each); the last point is bucket 0 (the digits of magnitude 2^(c-1); saved at buf[2 cnt0] before
    // the merge turned slot 0 into the total)
    for (uint32_t q = 0; q * NG < cnt0; ++q) {
      // point t needs t doublings: pass 0 hands the longest chains to groups 0, 1, ..; the next pass runs the other way round
      const uint32_t idx = q * NG + ((q & 1u) ? NG - 1 - grp : grp);
      if (idx >= cnt0) continue;  // whole groups skip together
      const uint32_t pt = cnt0 - 1 - idx;
      Ld p = pt + 1 < cnt0 ? A[1 + pt] : buf[2 * cnt0];
      if (pt + 1 < cnt0) lam_to_ld(p, L);  // the merge tree's lambda-projective output; bucket 0 was saved before the tree (Lopez-Dahab)
#pragma unroll 1
      for (uint32_t k = 0; k < pt; ++k) p = ld_dbl(p, L);
      if (L.r == 0) in[pt] = p;
    }
  } else {
    for (uint32_t tid = threadIdx.x; tid < cnt0; tid += EC_TPB) {
      uint32_t w = tid / (uint32_t)c, t = tid - w * (uint32_t)c;
      int k = (int)(w * (uint32_t)c) - min((int)w, n_narrow) + (int)t;
      Ld p = A[((size_t)w << c) + 1 + t];
      p.Y = gf_mul(p.X, gf_add(p.Y, p.X));  // lambda-projective (the merge tree's output) -> Lopez-Dahab: Y = X (L + X)
      p.X = gf_sqr_n_fast(p.X, k, T);
      p.Y = gf_sqr_n_fast(p.Y, k, T);
      p.Z = gf_sqr_n_fast(p.Z, k, T);
      in[tid] = p;
    }
  }
  __syncthreads();
  uint32_t cnt = cnt0;
  while (cnt > 1 && (HEX || (cnt + 1) / 2 > EC_TPB / 16)) {  // quad variant: while a level has more additions than the block has rows
    const uint32_t half = (cnt + 1) / 2;
    for (uint32_t i = grp; i < half; i += NG) {  // whole groups take the same trips
      Ld a = in[2 * i];
      if (2 * i + 1 < cnt) ld_add_ip(a, in[2 * i + 1], L);
      if (L.r == 0) out[i] = a;
    }
    __syncthreads();
    Ld* t = in; in = out; out = t;
    cnt = half;
  }
  // the last levels (<= 16 additions) and the conversion run on rows of 16 lanes in both variants (same LDS tables)
  GfLdsH H;
  H.l = L.l;
  H.r = threadIdx.x & 15u;
  while (cnt > 1) {
    const uint32_t half = (cnt + 1) / 2;
    for (uint32_t i = threadIdx.x >> 4; i < half; i += EC_TPB / 16) {
      Ld a = in[2 * i];
      if (2 * i + 1 < cnt) ld_add_ip(a, in[2 * i + 1], H);
      if (H.r == 0) out[i] = a;
    }
    __syncthreads();
    Ld* t = in; in = out; out = t;
    cnt = half;
  }
  if (threadIdx.x >= 16) return;  // one row: the inversion's products are a serial chain
  Aff a;
  Ld p = in[0];
  const bool fin = !ld_is_inf(p);
  a.x = gf_zero();
  a.y = gf_zero();
  Gf w = gf_zero();
  if (fin) {
    if (out_enc && !gf_is_zero(p.X)) {
      // x = X / Z, y = Y / Z^2, y / x = Y / (X Z): everything from inv = 1 / (X Z)
      const Gf inv = gf_inv_fast(gf_mul(p.X, p.Z, H), T, H);
      const Gf zi = gf_mul(inv, p.X, H);
      a.x = gf_mul(p.X, zi, H);
      a.y = gf_mul(p.Y, gf_sqr(zi), H);
      const Gf lam1 = gf_add(gf_add(a.x, gf_mul(p.Y, inv, H)), gf_one());
      w = gf_sqr_tab_wide(gf_sqr_tab_wide(lam1, T.t116), T.t116);
      if (rule) w = codec_present(w, rule, T);
    } else {
      Gf zi = gf_inv_fast(p.Z, T, H);
      a.x = gf_mul(p.X, zi, H);
      a.y = gf_mul(p.Y, gf_sqr(zi), H);
      if (out_enc) {  // x = 0 (the point of order two): the same formula as k_encode_point, where 1 / 0 reads 0
        const Gf lam1 = gf_add(gf_add(a.x, gf_mul(a.y, gf_inv_fast(a.x, T, H), H)), gf_one());
        w = gf_sqr_tab_wide(gf_sqr_tab_wide(lam1, T.t116), T.t116);
        if (rule) w = codec_present(w, rule, T);
      }
    }
  }
  if (threadIdx.x != 0) return;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    out_xy[k] = a.x.w[k];
    out_xy[8 + k] = a.y.w[k];
  }
  *out_inf = fin ? 0u : 1u;
  if (out_enc) store30(out_enc, w, rule);
  // the scalar-range word: to the caller's block (a deferred MSM, msm_core's d_err_defer) and to err_out, which the host copy of a
  // synchronous call reads; then the control words are left as the next MSM on this workspace expects them
  const unsigned long long e = ctrl[0];
  if (err_dst) *err_dst = e;
  ctrl[2] = e;
  ctrl[0] = ~0ull;
  ctrl[1] = 0ull;
}

// sum of n affine points (the partial MSM results of n GPUs or ranks) -> affine: one quad of lanes, n - 1 mixed
// additions + one inversion, all through the quad-cooperative multiplier (~15 us per point + ~50 us)
__global__ void __launch_bounds__(64) k_sum_points(const char* __restrict__ pts, uint32_t pt_stride_bytes, const uint32_t* __restrict__ inf, uint32_t n,
                                                  uint32_t inf_stride, GfSqrTables T, uint32_t* __restrict__ out_xy, uint32_t* __restrict__ out_inf) {
  extern __shared__ char lds_raw[];
  GfLdsQ L = gf_ldsq_init(lds_raw);
  if (threadIdx.x >= 4 || blockIdx.x != 0) return;
  Ld acc = ld_infinity();
#pragma unroll 1
  for (uint32_t i = 0; i < n; ++i) {
    if (inf[(size_t)i * inf_stride]) continue;
    Aff q = *(const Aff*)(pts + (size_t)i * pt_stride_bytes);
    ld_madd_ip(acc, q, L);
  }
  Aff a;
  a.x = gf_zero();
  a.y = gf_zero();
  const bool fin = !ld_is_inf(acc);
  if (fin) {
    Gf zi = gf_inv_fast(acc.Z, T, L);
    a.x = gf_mul(acc.X, zi, L);
    a.y = gf_mul(acc.Y, gf_sqr(zi), L);
  }
  if (threadIdx.x != 0) return;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    out_xy[k] = a.x.w[k];
    out_xy[8 + k] = a.y.w[k];
  }
  *out_inf = fin ? 0u : 1u;
}

// multiplier microbenchmark (bench.py's roofline leg runs it OUTSIDE the timed loop): a dependent chain of products per
// lane through the Karatsuba LDS multiplier at the occupancy of k_affine_round (256-thread blocks, 3 waves per SIMD)
__global__ void __launch_bounds__(EC_TPB) __attribute__((amdgpu_waves_per_eu(3, 3))) k_ubench_mul(Gf* __restrict__ out, int reps) {
  extern __shared__ char lds_raw[];
  GfLdsK L = gf_ldsk_init(lds_raw);
  const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  Gf x, y;
#pragma unroll
  for (int i = 0; i < 8; ++i) { x.w[i] = t * 2654435761u + i; y.w[i] = t * 40503u + 7 * i; }
  x.w[7] &= 0x1ff; y.w[7] &= 0x1ff;
#pragma unroll 1
  for (int r = 0; r < reps; ++r) { x = gf_mul(x, y, L); y.w[0] ^= x.w[3]; }
  out[t] = x;
}

// wave trace of the pair rounds (dvp_debug_wave_trace): device buffer, record capacity, launch tag (counts traced launches)
// (buffer, capacity) are published together under a mutex: an MSM in flight on another thread takes one consistent snapshot per launch
static std::mutex g_wave_trace_mu;
static unsigned long long* g_wave_trace_buf = nullptr;
static uint32_t g_wave_trace_cap = 0;
static std::atomic<uint32_t> g_wave_trace_tag{0};
static void wave_trace_snapshot(unsigned long long** buf, uint32_t* cap) {
  std::lock_guard<std::mutex> g(g_wave_trace_mu);
  *buf = g_wave_trace_buf;
  *cap = g_wave_trace_cap;
}

// ---- workspace -----------------------------------------------------------------------------------
struct MsmWorkspace {
  std::mutex mu;
  void* p = nullptr;
  size_t bytes = 0;
  int device = -1;
  // side channel for the one number the host needs mid-MSM (largest bucket): pinned word + its own stream, so the copy
  // overtakes the first pair round that is already enqueued on the caller's stream
  hipStream_t aux = nullptr;
  hipEvent_t ev = nullptr;
  hipEvent_t ev_heavy = nullptr;  // end of this workspace's last run of pair rounds (HeavyGate)
  // a DEFERRED MSM (msm_core, d_err_defer) returns with its kernels still in flight: ev_busy marks their end on busy_stream.  A later
  // call on the same stream is ordered behind them by the stream itself; a call on another stream waits for the event ON THE GPU
  // (and the slot choice prefers a workspace that is idle, so that two provers in flight keep to a workspace each)
  hipEvent_t ev_busy = nullptr;
  hipStream_t busy_stream = nullptr;
  bool busy_valid = false;
  bool ctrl_clean = false;  // the control words at the head of the workspace are as an MSM expects them (msm_core)
  bool busy_for(hipStream_t st) { return busy_valid && busy_stream != st && hipEventQuery(ev_busy) == hipErrorNotReady; }
  uint32_t* pinned = nullptr;
  // round bookkeeping (counts, offsets, descriptors of the later pair rounds) runs on a side stream while the first round fills
  // the chip: ev_pre = the first round's offsets are in place, ev_side[r] = round r may start
  hipStream_t side = nullptr;
  hipEvent_t ev_pre = nullptr;
  std::vector<hipEvent_t> ev_side;
  int ensure_side(size_t rounds) {
    if (!side) {
      DVP_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
      DVP_HIP(hipEventCreateWithFlags(&ev_pre, hipEventDisableTiming));
    }
    while (ev_side.size() < rounds) {
      hipEvent_t b = nullptr;
      DVP_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
      ev_side.push_back(b);
    }
    return DVP_OK;
  }
  int ensure_aux() {
    if (aux) return DVP_OK;
    DVP_HIP(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
    DVP_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    DVP_HIP(hipEventCreateWithFlags(&ev_heavy, hipEventDisableTiming));
    DVP_HIP(hipEventCreateWithFlags(&ev_busy, hipEventDisableTiming));
    DVP_HIP(hipHostMalloc((void**)&pinned, 64, hipHostMallocDefault));
    return DVP_OK;
  }
  // `need` is raised to the largest size any workspace of this device was ever asked for: which of the two slots a call gets
  // depends on timing, and a slot that had only seen the smaller MSM of a prover would otherwise be freed and re-allocated
  // (a device-wide synchronisation plus a multi-GB hipMalloc) in the middle of a later proof
  int ensure(size_t need, std::atomic<size_t>& dev_max) {
    size_t seen = dev_max.load();
    while (seen < need && !dev_max.compare_exchange_weak(seen, need)) {}
    if (seen > need) need = seen;
    int dev;
    DVP_HIP(hipGetDevice(&dev));
    if (p && (dev != device || bytes < need)) {
      // a deferred MSM's kernels may still be running in this allocation (the first proof: the K MSM is larger than the commitment
      // MSM in flight): wait for them explicitly rather than lean on hipFree's implicit device-wide synchronisation
      // (busy_valid was already consumed by the caller's hipStreamWaitEvent: the event itself is what says whether they ended)
      if (ev_busy) (void)hipEventSynchronize(ev_busy);
      (void)hipFree(p);
      p = nullptr;
      bytes = 0;
    }
    if (!p) {
      DVP_HIP(hipMalloc(&p, need));
      ctrl_clean = false;
      bytes = need;
      device = dev;
    }
    return DVP_OK;
  }
};
// grow-only workspaces, MSM_WS_SLOTS PER DEVICE: MSMs on different GPUs (in-library multi-GPU: one host thread per device,
// prove.hip) run concurrently, and so do up to MSM_WS_SLOTS MSMs on the SAME GPU (two provers on two host threads and two
// streams: the latency-bound stretches of one proof -- merge levels, tail, small rounds, host round trips -- are filled by the
// other's kernels); a further caller waits for slot 0.  The second slot allocates only when two calls do overlap.
constexpr int MSM_MAX_DEVICES = 16;
constexpr int MSM_WS_SLOTS = 2;
static MsmWorkspace g_ws_dev[MSM_MAX_DEVICES][MSM_WS_SLOTS];
static std::atomic<size_t> g_ws_need[MSM_MAX_DEVICES];
// Two MSMs on one device overlap everything EXCEPT their pair rounds: those fill the chip on their own (two of them side by side
// only contend: measured 4 % slower than taking turns), while everything around them -- sort, reducer, merge levels, tail, the
// prover's Fr stages and host round trips -- leaves lanes idle.  So the pair rounds of the second MSM wait ON THE GPU
// (hipStreamWaitEvent, no host block beyond the enqueue) for the end of the first one's, which staggers two provers that
// started in lockstep by themselves.
struct HeavyGate {
  std::mutex mu;
  hipEvent_t last = nullptr;
};
static HeavyGate g_heavy[MSM_MAX_DEVICES];

struct MsmPlan {
  uint32_t n;
  int c, W, n_narrow;
  uint32_t K, nkeys;
  size_t e_max, t1_max, t2_max;
};

static MsmPlan msm_plan(size_t n, const struct MsmFixedCtx* fx);
// Windows of the fixed-base mode.  tau-adic expansions of reduced scalars are at most 234 digits long (measured: 229-233
// typical) and their top 3-4 digits are mostly zero.  A uniform split leaves a short top window (234 mod c digits) whose
// entries pile into a few hundred buckets of the shared set -- n/2^7 entries per bucket for c = 16 -- and the reducer
// then runs 15-long serial chains.  So the 234 digits are split into W_main = ceil(234/c) windows whose widths differ by
// at most one (the n_narrow low windows are c-1 digits wide, the entropy-poor top windows c), plus one overflow window
// for digits >= 234 that exists only for completeness (the recode flags anything longer than it covers).
struct MsmFixedCtx {
  Aff* table = nullptr;  // [W][n_total]
  uint32_t n_total = 0;
  int c = FX_C_MAX, W = TAU_DIGITS / FX_C_MAX, n_narrow = 0;
  void set_c(int cc) {
    c = cc;
    int w_main = (234 + cc - 1) / cc;
    n_narrow = w_main * cc - 234;
    W = w_main + 1;
  }
  bool signed_digits = false;    // aligned windows with digits in [-2^(c-1), 2^(c-1)] over the scalar's BINARY digits (k_recode_signed,
                                 // table rows 2^(o_w) P from k_dbl_table): 2^(c-1) buckets.  false = the tau-adic aligned windows
                                 // (k_recode, table rows tau^(o_w) P from k_frob_table)
  // 234 bits (a canonical scalar < 2^232, the last carry, one spare so that the top digit stays positive) in W = ceil(234 / c)
  // windows whose widths differ by at most one: the n_narrow LOW windows are c - 1 wide.  A request whose windows would all be
  // narrow is the next smaller c.
  void set_c_signed(int cc) {
    for (;; --cc) {
      W = (234 + cc - 1) / cc;
      n_narrow = W * cc - 234;
      if (n_narrow < W || cc <= 2) break;
    }
    c = cc;
    signed_digits = true;
  }
  int rows() const { return W; }
  int key_bits() const { return signed_digits ? c - 1 : c; }
  int hi_bits = -1;  // level-1 partition bits of the sort (set at creation)
  FxBits bits() const { FxBits b; b.hi = hi_bits; b.lo = key_bits() - b.hi; return b; }
};
static MsmPlan msm_plan(size_t n, const MsmFixedCtx* fx) {
  const bool fixed = fx != nullptr;
  MsmPlan p;
  p.n = (uint32_t)n;
  // cost model: W(c) * (8.4 n + 28 * 2^c) field multiplications.  Windows are evened out as in the fixed-base mode
  // (see MsmFixedCtx): the 234 real digits go into ceil(234/c) windows of c or c-1 digits (narrow ones first), the
  // digits above them (never seen beyond 236) into overflow windows.  A uniform 240/c split left a short top window
  // whose entries all fell into a few dozen buckets and serialised the reducer.
  auto windows = [](int c, int* n_narrow) {
    int w_main = (234 + c - 1) / c;
    *n_narrow = w_main * c - 234;
    return w_main + (TAU_DIGITS - 234 + c - 1) / c;
  };
  double best = 1e300;
  p.c = 4;
  int nn;
  for (int c = 4; c <= 15; ++c) {  // <= 15: the sort keeps 2^c u32 cursors in LDS
    int W = windows(c, &nn);
    double cost = W * (8.4 * (double)n + 28.0 * (double)(1u << c));
    if (cost < best) { best = cost; p.c = c; }
  }
  // 2^15 .. 2^17 points are latency all the way (config #2: ~45 dependent launches): with the round-4 chains (lambda-projective merge,
  // 16 lanes per product in the deep levels and the tail) two merge levels fewer beat the model's window -- tools/small_msm_sweep.py,
  // c = 10 / K = 8 against the model's c = 12 / K = 4: 1.43 against 1.63 ms at 2^16, 1.82 against 2.02 ms at 2^17 (2^14 and 2^18: no difference)
  // Round 6 (tools/small_msm_ck.py, after the tail's group sums and the reducer's cheaper levels): the same holds at 2^18 (c = 10 / K = 8:
  // 2.04 against 2.21 ms), and 2^19 points want c = 13 where the model says 14 (3.17 against 3.43 ms); 2^14 and 2^20 are flat
  const bool small_latency = !fixed && n >= ((size_t)1 << 15) && n <= ((size_t)1 << 18);
  if (small_latency) p.c = 10;
  if (!fixed && n > ((size_t)1 << 18) && n <= ((size_t)1 << 19)) p.c = 13;
  if (tune().msm_c >= 2 && tune().msm_c <= 15) p.c = (int)tune().msm_c;
  p.W = windows(p.c, &p.n_narrow);
  p.nkeys = (uint32_t)p.W << p.c;
  if (fixed) {  // all windows share one bucket set (bases pre-rotated by tau^(c w))
    p.c = fx->c;
    p.W = fx->W;
    p.n_narrow = fx->n_narrow;
    p.nkeys = 1u << fx->key_bits();
  }
  p.e_max = n * (size_t)p.W;
  // fan-in: keep >= ~256k level-1 tasks in flight when the input allows it
  uint32_t K = (uint32_t)(p.e_max / 262144);
  if (K < 8) K = 8;
  if (K > 16) K = 16;
  // small inputs (config #2's 2^16 points: ~1.5 M entries) are chain latency in the reducer as well: 2^16 .. 2^18 points run
  // 5-10 % faster with three additions per task than with seven (tools/small_msm_sweep.py)
  if (!fixed && n <= ((size_t)1 << 18)) K = 4;
  if (small_latency) K = 8;
  if (!fixed && n > ((size_t)1 << 18) && n <= ((size_t)1 << 19)) K = 8;
  // fixed-base mode: the reducer only sees what the pair rounds leave (<= ~20 entries in the fullest buckets) and is pure
  // chain latency there: two levels of <= 7 additions beat one of <= 15 (2^20 prove: 24.26 against 24.58 ms)
  if (fixed) K = 8;
  if (tune().msm_k >= 2 && tune().msm_k <= 64) K = (uint32_t)tune().msm_k;
  p.K = K;
  p.t1_max = p.e_max / K + p.nkeys + 1;
  p.t2_max = p.t1_max / K + p.nkeys + 1;
  return p;
}

static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

// fx == nullptr: one-shot MSM over (d_scalars, d_bases).  fx != nullptr: fixed-base mode, the scalars
// d_scalars[0..n) belong to bases i0 .. i0+n of the pre-rotated table and d_inf is already offset by i0.
static int msm_core(const void* d_scalars, const void* d_bases, const void* d_inf, size_t n, const MsmFixedCtx* fx, uint32_t i0,
                    void* d_out_xy, void* d_out_inf, hipStream_t st, void* d_out_enc = nullptr /* + the result's 30-byte encoding */,
                    void* h_copy = nullptr, const void* d_copy = nullptr, size_t copy_bytes = 0 /* a device block the caller wants on the
                    host (pinned) when this call returns: it rides on the MSM's own final synchronisation */,
                    unsigned long long* d_err_defer = nullptr /* DEFERRED completion: the call returns once everything is enqueued -- no final
                    synchronisation, no h_copy; the scalar-range word (~0 = fine, else the index of the first scalar >= p) is written to
                    this device word by the tail kernel and the caller reads it with whatever it synchronises on later */) {
  if (n == 0 && d_err_defer) {
    DVP_HIP(hipMemsetAsync(d_out_xy, 0, 64, st));
    if (d_out_enc) DVP_HIP(hipMemsetAsync(d_out_enc, 0, 30, st));
    DVP_HIP(hipMemsetAsync(d_out_inf, 0, 4, st));
    DVP_HIP(hipMemsetAsync(d_out_inf, 1, 1, st));  // little-endian u32 1
    DVP_HIP(hipMemsetAsync(d_err_defer, 0xff, 8, st));
    return DVP_OK;
  }
  if (n == 0) {
    DVP_HIP(hipMemsetAsync(d_out_xy, 0, 64, st));
    if (d_out_enc) DVP_HIP(hipMemsetAsync(d_out_enc, 0, 30, st));
    if (h_copy) DVP_HIP(hipMemcpyAsync(h_copy, d_copy, copy_bytes, hipMemcpyDeviceToHost, st));
    uint32_t one = 1;
    DVP_HIP(hipMemcpyAsync(d_out_inf, &one, 4, hipMemcpyHostToDevice, st));
    DVP_HIP(hipStreamSynchronize(st));
    return DVP_OK;
  }
  if (n > (1u << 27)) return DVP_EINVAL;
  const uint32_t sign_mask = (fx && fx->signed_digits) ? ITEM_NEG : 0u;  // signed flavours keep the sign in bit 31 of an item
  {  // entry positions and pre-rotated table indices are 32-bit (0xffffffff = "no partner" in a slot descriptor)
    const MsmPlan pl = msm_plan(n, fx);
    const uint64_t tab = fx ? (uint64_t)fx->rows() * fx->n_total : (uint64_t)n;
    if ((uint64_t)pl.e_max >= 0xfffffff0ull || tab >= (sign_mask ? 0x7ffffff0ull : 0xfffffff0ull)) return DVP_EINVAL;
  }
  int cur_dev = 0;
  DVP_HIP(hipGetDevice(&cur_dev));
  if (cur_dev < 0 || cur_dev >= MSM_MAX_DEVICES) return DVP_EINVAL;
  {  // dynamic-LDS limits are a per-device property of a kernel: set them once on every device that runs an MSM
    static std::mutex attr_mu;
    static bool attr_done[MSM_MAX_DEVICES] = {false};
    std::lock_guard<std::mutex> ga(attr_mu);
    if (!attr_done[cur_dev]) {
      hipError_t attr_err = hipFuncSetAttribute((const void*)k_scatter_local, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      if (attr_err == hipSuccess)
        attr_err = hipFuncSetAttribute((const void*)k_hist_local, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
      if (attr_err == hipSuccess)
        attr_err = hipFuncSetAttribute((const void*)k_scatter_local2, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      if (attr_err == hipSuccess)
        attr_err = hipFuncSetAttribute((const void*)k_hist_local2, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
      if (attr_err == hipSuccess)
        attr_err = hipFuncSetAttribute((const void*)k_part_scatter_staged, hipFuncAttributeMaxDynamicSharedMemorySize, FX_STAGE1_LDS);
      if (attr_err == hipSuccess)
        attr_err = hipFuncSetAttribute((const void*)k_part_scatter_signed<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FX_STAGE1_LDS);
      if (attr_err == hipSuccess)
        attr_err = hipFuncSetAttribute((const void*)k_scatter_local2_staged, hipFuncAttributeMaxDynamicSharedMemorySize, FX_STAGE2_LDS);
      const void* ec[] = {(const void*)k_accum_affine<true, false>, (const void*)k_accum_affine<false, false>, (const void*)k_accum_affine<true, true>,
                          (const void*)k_accum_affine<false, true>, (const void*)k_accum_proj<1>, (const void*)k_accum_proj<4>, (const void*)k_accum_proj<16>,
                          (const void*)k_accum_affine_fast<true, false>, (const void*)k_accum_affine_fast<false, false>, (const void*)k_accum_affine_fast<true, true>,
                          (const void*)k_accum_affine_fast<false, true>, (const void*)k_accum_affine_rest<true>, (const void*)k_accum_affine_rest<false>,
                          (const void*)k_tail_groups, (const void*)k_merge<1, false>, (const void*)k_merge<4, false>, (const void*)k_merge<16, false>, (const void*)k_merge<1, true>, (const void*)k_merge<4, true>, (const void*)k_merge<16, true>,
                          (const void*)k_affine_round<true, false>, (const void*)k_affine_round<false, false>, (const void*)k_affine_round<true, true>,
                          (const void*)k_affine_round<false, true>, (const void*)k_sum_points, (const void*)k_tail<false>, (const void*)k_tail<true>,
                          (const void*)k_bucket_pairs, (const void*)k_bucket_rest};
      for (const void* f : ec)
        if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, EC_LDS_Q);
      DVP_HIP(attr_err);
      attr_done[cur_dev] = true;
    }
  }
  MsmPlan p = msm_plan(n, fx);
  const FxBits fb = fx ? fx->bits() : FxBits{0, 0};
  const uint32_t FX_NP = fb.np();
  std::unique_lock<std::mutex> g;
  int ws_slot = 0;
  const int ws_slots = tune().msm_ws_slots >= 1 && tune().msm_ws_slots <= MSM_WS_SLOTS ? (int)tune().msm_ws_slots : 1;
  // first choice: a workspace nobody holds AND whose last deferred MSM (if any) is finished or sits on this very stream
  for (int pass = 0; pass < 2 && !g.owns_lock(); ++pass)
    for (int sl = 0; sl < ws_slots && !g.owns_lock(); ++sl) {
      g = std::unique_lock<std::mutex>(g_ws_dev[cur_dev][sl].mu, std::try_to_lock);
      if (g.owns_lock() && pass == 0 && g_ws_dev[cur_dev][sl].busy_for(st)) g.unlock();
      if (g.owns_lock()) ws_slot = sl;
    }
  if (!g.owns_lock()) g = std::unique_lock<std::mutex>(g_ws_dev[cur_dev][0].mu);
  MsmWorkspace& g_ws = g_ws_dev[cur_dev][ws_slot];
  if (g_ws.busy_valid) {  // kernels of a deferred MSM may still be using this workspace
    if (g_ws.busy_stream != st) DVP_HIP(hipStreamWaitEvent(st, g_ws.ev_busy, 0));
    g_ws.busy_valid = false;
  }
  // carve the workspace
  size_t o = 0;
  auto carve = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes); return r; };
  size_t o_err = carve(32);  // err (u64) | d_max, rest_n (u32 each) | err_out (u64): the tail kernel's copy of err for the host
  size_t o_digits = carve(fx ? 16 : (size_t)p.W * n * 2);
  // signed flavour: level 1 of the sort reads the scalars themselves (k_part_hist_signed), a block per fx_S scalars
  const bool fused1 = fx && fx->signed_digits && tune().msm_sort_fused != 0;
  const uint32_t fx_S = fused1 ? (FX_CHUNK / (uint32_t)p.W > 0 ? FX_CHUNK / (uint32_t)p.W : 1u) : 0u;
  const uint32_t fx_nblk = fused1 ? cdiv(n, fx_S) : cdiv(p.e_max, FX_CHUNK);
  size_t o_digits32 = carve(fx && !fused1 ? p.e_max * 4 : 16);
  size_t o_plo = carve(fx ? p.e_max * 2 : 16);
  size_t o_pid = carve(fx ? p.e_max * 4 : 16);
  size_t o_phist = carve(fx ? (size_t)fx_nblk * FX_NP * 4 : 16);
  size_t o_pbase = carve(fx ? (size_t)fx_nblk * FX_NP * 4 : 16);
  const uint32_t fx_nch = cdiv(fx_nblk, FX_SCAN_CB);  // chunks of the level-1 scan (k_part_scan_chunks)
  size_t o_ctot = carve(fx ? (size_t)FX_NP * fx_nch * 4 : 16);
  size_t o_pstart = carve((FX_NP_MAX + 1) * 4 * 3);
  size_t o_cnt = carve(((size_t)p.nkeys + 1) * 4);
  size_t o_off = carve(((size_t)p.nkeys + 1) * 4);
  size_t o_ntask = carve(((size_t)p.nkeys + 1) * 4);
  size_t o_toff = carve(((size_t)p.nkeys + 1) * 4);
  size_t o_cnt2 = carve(((size_t)p.nkeys + 1) * 4);
  size_t o_off2 = carve(((size_t)p.nkeys + 1) * 4);
  size_t o_bsum = carve(((size_t)p.nkeys / SCAN_BLK + 8) * 4);
  size_t o_items = carve((p.e_max + (size_t)p.nkeys + 2) * 4);  // odd buckets are padded to even (k_post_sort)
  const size_t sort_cells = fx ? ((size_t)(fx_nblk + FX_NP + 1) << fb.lo) : ((size_t)p.W * cdiv(n, SORT_CHUNK) << p.c);
  size_t o_hist16 = carve(sort_cells * 2);
  size_t o_choff = carve(sort_cells * 4);
  size_t o_bufA = carve(p.t1_max * sizeof(Ld));
  size_t o_bufB = carve(p.t2_max * sizeof(Ld));
  const Tune tn = tune();
  const bool affine_mode = !tn.msm_proj;
  const size_t affA_n = p.e_max / 2 + p.nkeys + 1, affB_n = p.e_max / 4 + p.nkeys + 1;
  size_t o_affA = carve(affine_mode ? affA_n * sizeof(Aff) : 16);
  size_t o_affB = carve(affine_mode ? affB_n * sizeof(Aff) : 16);
  size_t o_prefix = carve(affine_mode ? (affA_n + 256) * sizeof(Gf) : 16);  // one prefix product per output slot (+ the B - 1 <= 254 slots the last thread's rows may overhang)
  // pair rounds the SIZE rule allows (the largest bucket, which only the device knows, may stop them sooner): rounds run while a
  // round still carries >= aff_min additions
  const size_t aff_min = (size_t)(tn.msm_aff_min > 0 ? tn.msm_aff_min : 1);
  // entries that really exist: the overflow windows are empty in practice
  const size_t e_est = (size_t)n * (size_t)((fx && fx->signed_digits) ? p.W : (234 + p.c - 1) / p.c);
  int ra_plan = 0;
  while (affine_mode && ra_plan < 40 && (e_est >> (ra_plan + 1)) >= aff_min) ++ra_plan;
  // bookkeeping of ALL later rounds behind the first one (side stream, see below): every round keeps its own counts / offsets /
  // descriptors, so nothing the side stream writes is ever read by a round still in flight
  const bool pipelined = affine_mode && tn.msm_round_pipeline != 0 && ra_plan >= 2 && e_est >= 4 * (size_t)p.nkeys;
  const bool side_stream = pipelined && tn.msm_round_pipeline == 1;
  std::vector<size_t> desc_at((size_t)ra_plan + 1, 0);  // round r's descriptors start at gdesc + desc_at[r] (r >= 1)
  size_t desc_n = affA_n + 64;
  if (pipelined) {
    size_t cap_r = p.e_max + p.nkeys, at = 0;
    for (int r = 0; r < ra_plan; ++r) {
      const size_t out_max = cap_r / 2 + p.nkeys + 1;
      if (r >= 1) { desc_at[r] = at; at += out_max + 64; }
      cap_r = out_max;
    }
    if (at > desc_n) desc_n = at;
  }
  size_t o_gdesc = carve(affine_mode ? desc_n * sizeof(uint32_t) : 16);  // one descriptor word per output slot
  size_t o_rp = carve(pipelined ? (size_t)ra_plan * 2 * ((size_t)p.nkeys + 1) * 4 : 16);  // (counts, offsets) of rounds 1 .. ra_plan
  size_t o_bsum2 = carve(((size_t)((ra_plan > 0 ? ra_plan : 1) + 1) * ((size_t)p.nkeys / SCAN_BLK + 1) + 8) * 4);  // scan scratch of the side stream: a row of block sums per round
  size_t o_bkt = carve((size_t)p.nkeys * sizeof(Ld));
  // k_bucket_pairs' list of buckets that met an exceptional pair (<= nkeys), or k_accum_affine_fast's list of such TASKS (chunks of K entries)
  const size_t rest_cap = (size_t)p.nkeys + 2 + p.e_max / (size_t)(p.K > 0 ? p.K : 1);
  size_t o_rest = carve(rest_cap * 4);
  const uint32_t tail_groups = ((uint32_t)(p.W * p.c) + 15u) / 16u;  // k_tail_groups: workgroups of 16 points
  size_t o_tail = carve(((size_t)2 * p.W * p.c + 1 + 2 * (size_t)tail_groups) * sizeof(Ld));
  DVP_TRY(g_ws.ensure(o, g_ws_need[cur_dev]));
  char* base = (char*)g_ws.p;
  auto* err = (unsigned long long*)(base + o_err);
  auto* digits = (uint16_t*)(base + o_digits);
  auto* digits32 = (uint32_t*)(base + o_digits32);
  auto* plo = (uint16_t*)(base + o_plo);
  auto* pid = (uint32_t*)(base + o_pid);
  auto* phist = (uint32_t*)(base + o_phist);
  auto* pbase = (uint32_t*)(base + o_pbase);
  auto* ctot = (uint32_t*)(base + o_ctot);
  auto* pstart = (uint32_t*)(base + o_pstart);
  auto* cstart = pstart + FX_NP_MAX + 1;
  auto* pcount = cstart + FX_NP_MAX + 1;
  auto* cnt = (uint32_t*)(base + o_cnt);
  auto* off = (uint32_t*)(base + o_off);
  auto* ntask = (uint32_t*)(base + o_ntask);
  auto* toff = (uint32_t*)(base + o_toff);
  auto* cnt2 = (uint32_t*)(base + o_cnt2);
  auto* off2 = (uint32_t*)(base + o_off2);
  auto* bsum = (uint32_t*)(base + o_bsum);
  auto* items = (uint32_t*)(base + o_items);
  auto* hist16 = (uint16_t*)(base + o_hist16);
  auto* chunk_off = (uint32_t*)(base + o_choff);
  Ld* bufA = (Ld*)(base + o_bufA);
  Ld* bufB = (Ld*)(base + o_bufB);
  Ld* bkt = (Ld*)(base + o_bkt);
  Aff* affA = (Aff*)(base + o_affA);
  Aff* affB = (Aff*)(base + o_affB);
  Gf* prefix = (Gf*)(base + o_prefix);
  uint32_t* gdesc = (uint32_t*)(base + o_gdesc);  // one word per output slot (rounds after the first)
  auto* rp = (uint32_t*)(base + o_rp);
  auto* bsum2 = (uint32_t*)(base + o_bsum2);
  Ld* tail = (Ld*)(base + o_tail);
  auto* rest_list = (uint32_t*)(base + o_rest);
  const uint32_t nk = p.nkeys;

  // counts / offsets of round r's outputs (r >= 1) from those of its inputs, and its slot descriptors, on stream s
  auto bookkeep = [&](int r, const uint32_t* ic, const uint32_t* io, uint32_t* oc, uint32_t* oo, uint32_t* dsc, hipStream_t s, uint32_t* bs) -> int {
    DVP_TRY(scan_exclusive_div(ic, 2u, oc, oo, nk, bs, s));
    // descriptors: lanes per bucket by the average bucket size of this round
    const size_t per_key = (e_est >> (r + 1)) / nk;
#define DVP_DESC_LAUNCH(LPK) hipLaunchKernelGGL((k_round_desc<LPK>), dim3(cdiv((size_t)nk * LPK, 256)), dim3(256), 0, s, ic, io, oo, nk, dsc)
    if (per_key >= 48) DVP_DESC_LAUNCH(64); else if (per_key >= 8) DVP_DESC_LAUNCH(16); else DVP_DESC_LAUNCH(4);
#undef DVP_DESC_LAUNCH
    return DVP_OK;
  };
  // Round bookkeeping in one piece (round 4).  A pair round's counts, offsets and slot descriptors depend on the bucket COUNTS only,
  // never on the points, so all rounds are prepared at once -- arrays per round, four launches (k_mscan_*, k_round_desc_all) -- as
  // soon as the sort has its counts, and the rounds then follow each other back to back (three 5 us scan launches and a descriptor
  // kernel per round used to sit between them).  Where the four launches run (Tune::msm_round_pipeline):
  //   2 (default)  on the caller's stream, before the sort's last scatter;
  //   1            on a side stream beside that scatter (HBM-bound).  One proof at a time the two are equal within noise (bench.py's
  //                configuration: 21.0-21.2 ms either way; only with every proof on the NULL stream did the side stream measure
  //                0.1-0.17 ms better), and with two proofs in flight the side stream costs 1 ms per proof (21.0 against 19.4):
  //                its workgroups land in the OTHER proof's pair rounds.  NEVER beside this MSM's own first round: a round is an
  //                exact number of chip-fulls, and a few foreign workgroups holding wave slots when it is dispatched push some of
  //                its workgroups into an extra pass (measured: +0.4 ms per MSM).
  // (rc(r), ro(r)) = counts / offsets of round r's INPUT; round 0's are the sort's.
  auto rc = [&](int r) -> uint32_t* { return r == 0 ? cnt : rp + (size_t)(2 * (r - 1)) * ((size_t)nk + 1); };
  auto ro = [&](int r) -> uint32_t* { return r == 0 ? off : rp + (size_t)(2 * (r - 1) + 1) * ((size_t)nk + 1); };
  int prepared = 0;  // pipelined: the bookkeeping of rounds < prepared is enqueued on the side stream (ev_side[0] = all of it is done)
  uint32_t* d_max = (uint32_t*)(err + 1);
  uint32_t* rest_n = d_max + 1;  // entries of k_bucket_pairs' list
  // the largest bucket goes to the host on the aux stream while the caller's stream carries on
  auto read_max = [&]() -> int {
    DVP_TRY(g_ws.ensure_aux());
    DVP_HIP(hipEventRecord(g_ws.ev, st));
    DVP_HIP(hipStreamWaitEvent(g_ws.aux, g_ws.ev, 0));
    DVP_HIP(hipMemcpyAsync(g_ws.pinned, d_max, 4, hipMemcpyDeviceToHost, g_ws.aux));
    return DVP_OK;
  };
  // pipelined bookkeeping on the caller's stream: the scan of all rounds also yields the item offsets, the odd buckets' padding and the
  // largest bucket (k_mscan_add)
  const bool sort_done_by_mscan = pipelined && !side_stream;
  auto prepare_rounds = [&]() -> int {  // call once the sort's counts (cnt) are final on `st`
    if (!pipelined) return DVP_OK;
    hipStream_t bk = st;  // Tune::msm_round_pipeline == 2: the same four launches on the caller's stream
    if (side_stream) {
      DVP_TRY(g_ws.ensure_side(1));
      DVP_HIP(hipEventRecord(g_ws.ev_pre, st));
      DVP_HIP(hipStreamWaitEvent(g_ws.side, g_ws.ev_pre, 0));
      bk = g_ws.side;
    }
    // (c_r, o_r), r = 1 .. ra_plan, in one three-launch scan (round 0's outputs included: its even-aligned buckets make the sorted
    // item list the descriptor array, and the scan of ceil(c_0 / 2) is that list's offsets halved), then every round's descriptors
    const uint32_t nb = cdiv(nk, SCAN_BLK);
    // (the caller's stream only: the side-stream flavour keeps the sort's own scan, padding and maximum)
    uint32_t* off0 = side_stream ? nullptr : off;
    hipLaunchKernelGGL(k_mscan_local, dim3(nb), dim3(SCAN_TPB), 0, bk, cnt, nk, ra_plan, rp, bsum2);
    hipLaunchKernelGGL(k_mscan_bsums, dim3(ra_plan), dim3(SCAN_TPB), 0, bk, bsum2, nb, nk, rp, off0, off0 ? d_max : (uint32_t*)nullptr);
    if (off0) DVP_TRY(read_max());  // the largest bucket is on its way to the host before the descriptors and the last scatter run
    hipLaunchKernelGGL(k_mscan_add, dim3(nb), dim3(SCAN_TPB), 0, bk, rp, bsum2, nk, ra_plan, (const uint32_t*)cnt, off0, items);
    DescPlan plan;
    for (int q = 0; q < 40; ++q) plan.at[q] = q < ra_plan ? desc_at[q] : 0;
    const size_t per_key = (e_est >> 2) / nk;  // round 1's outputs per bucket
    if (per_key >= 48)
      hipLaunchKernelGGL((k_round_desc_all<64>), dim3(cdiv((size_t)nk * 64, 1024)), dim3(1024), 0, bk, rp, nk, ra_plan, plan, gdesc);
    else if (per_key >= 8)
      hipLaunchKernelGGL((k_round_desc_all<16>), dim3(cdiv((size_t)nk * 16, 1024)), dim3(1024), 0, bk, rp, nk, ra_plan, plan, gdesc);
    else
      hipLaunchKernelGGL((k_round_desc_all<4>), dim3(cdiv((size_t)nk * 4, 1024)), dim3(1024), 0, bk, rp, nk, ra_plan, plan, gdesc);
    if (side_stream) DVP_HIP(hipEventRecord(g_ws.ev_side[0], g_ws.side));
    DVP_HIP(hipGetLastError());
    prepared = ra_plan;
    return DVP_OK;
  };
  ProfScope ps_total(PROF_MSM_TOTAL, st);
  ProfScope ps_sort(PROF_MSM_SORT, st);  // recode + counting sort
  // the control words [err | d_max, rest_n | err_out]: every MSM's tail kernel leaves err = ~0 and d_max = rest_n = 0 behind for the
  // next one (two memset nodes per MSM, ~6 us each plus their dispatch gaps, sat on the critical path); a fresh allocation, or a call
  // that returned before its tail kernel was enqueued, resets them here
  if (!g_ws.ctrl_clean) {
    DVP_HIP(hipMemsetAsync(err, 0xff, 8, st));
    DVP_HIP(hipMemsetAsync(err + 1, 0, 8, st));
  }
  g_ws.ctrl_clean = false;
  if (fused1)
    ;  // no entry words in HBM: both level-1 kernels recompute them
  else if (fx && fx->signed_digits)
    hipLaunchKernelGGL(k_recode_signed, dim3(cdiv(n, 256)), dim3(256), 0, st, (const uint32_t*)d_scalars, (const uint8_t*)d_inf, (uint32_t)n,
                       p.c, p.W, p.n_narrow, digits32, err);
  else if (fx)
    hipLaunchKernelGGL((k_recode<uint32_t>), dim3(cdiv(n, 256)), dim3(256), 0, st, (const uint32_t*)d_scalars, (const uint8_t*)d_inf,
                       (uint32_t)n, p.c, p.W, p.n_narrow, digits32, err);
  else
    hipLaunchKernelGGL((k_recode<uint16_t>), dim3(cdiv(n, 256)), dim3(256), 0, st, (const uint32_t*)d_scalars, (const uint8_t*)d_inf,
                       (uint32_t)n, p.c, p.W, p.n_narrow, digits, err);
  if (fx) {
    const uint32_t gmax = fx_nblk + FX_NP;  // upper bound on the number of level-2 chunks
    if (fused1)
      hipLaunchKernelGGL(k_part_hist_signed, dim3(fx_nblk), dim3(SORT_TPB), 0, st, (const uint32_t*)d_scalars, (const uint8_t*)d_inf, (uint32_t)n,
                         p.c, p.W, p.n_narrow, fx_S, fb, phist, err);
    else
      hipLaunchKernelGGL(k_part_hist, dim3(fx_nblk), dim3(SORT_TPB), 0, st, digits32, p.e_max, fb, phist);
    hipLaunchKernelGGL(k_part_scan_chunks, dim3(fx_nch), dim3(256), 0, st, phist, fx_nblk, fb, pbase, ctot, fx_nch);
    hipLaunchKernelGGL(k_part_scan_totals, dim3(cdiv((size_t)FX_NP * 64, 256)), dim3(256), 0, st, ctot, fx_nch, fb, pcount);
    hipLaunchKernelGGL(k_part_starts, dim3(1), dim3(FX_NP_MAX), 0, st, pcount, fb, pstart, cstart);
    const bool staged1 = FX_NP >= 64;           // few partitions: direct stores already coalesce
    const bool staged2 = fb.lo <= 10;           // one bin per thread in the block scan
    if (fused1 && staged1)
      hipLaunchKernelGGL(k_part_scatter_signed<true>, dim3(fx_nblk), dim3(SORT_TPB), FX_STAGE1_LDS, st, (const uint32_t*)d_scalars,
                         (const uint8_t*)d_inf, (uint32_t)n, p.c, p.W, p.n_narrow, fx_S, fx->n_total, i0, fb, phist, pbase, ctot, fx_nch, pstart, plo, pid);
    else if (fused1)
      hipLaunchKernelGGL(k_part_scatter_signed<false>, dim3(fx_nblk), dim3(SORT_TPB), (2 * FX_NP_MAX + SORT_TPB) * 4, st, (const uint32_t*)d_scalars,
                         (const uint8_t*)d_inf, (uint32_t)n, p.c, p.W, p.n_narrow, fx_S, fx->n_total, i0, fb, phist, pbase, ctot, fx_nch, pstart, plo, pid);
    else if (staged1)
      hipLaunchKernelGGL(k_part_scatter_staged, dim3(fx_nblk), dim3(SORT_TPB), FX_STAGE1_LDS, st, digits32, p.e_max, (uint32_t)n,
                         fx->n_total, i0, fb, phist, pbase, ctot, fx_nch, pstart, plo, pid);
    else
      hipLaunchKernelGGL(k_part_scatter, dim3(fx_nblk), dim3(SORT_TPB), 0, st, digits32, p.e_max, (uint32_t)n, fx->n_total, i0, fb, pbase,
                         ctot, fx_nch, pstart, plo, pid);
    hipLaunchKernelGGL(k_hist_local2, dim3(gmax), dim3(SORT_TPB), (1u << fb.lo) * 2, st, plo, pstart, cstart, fb, hist16);
    hipLaunchKernelGGL(k_hist_scan2, dim3(cdiv(nk, 256)), dim3(256), 0, st, hist16, cstart, fb, chunk_off, cnt);
    DVP_TRY(prepare_rounds());  // needs the counts only
    if (!sort_done_by_mscan) DVP_TRY(scan_exclusive(cnt, off, nk, bsum, st, /*pad2=*/true));
    if (staged2)
      hipLaunchKernelGGL(k_scatter_local2_staged, dim3(gmax), dim3(SORT_TPB), FX_STAGE2_LDS, st, plo, pid, pstart, cstart, fb, off,
                         chunk_off, hist16, items);
    else
      hipLaunchKernelGGL(k_scatter_local2, dim3(gmax), dim3(SORT_TPB), (1u << fb.lo) * 4, st, plo, pid, pstart, cstart, fb, off, chunk_off, items);
  } else {
    const uint32_t nchunks = cdiv(n, SORT_CHUNK), nb = 1u << p.c;
    hipLaunchKernelGGL(k_hist_local, dim3(nchunks, p.W), dim3(SORT_TPB), (nb >> 1) * 4, st, digits, (uint32_t)n, p.c, hist16);
    hipLaunchKernelGGL(k_hist_scan, dim3(cdiv(nk, 256)), dim3(256), 0, st, hist16, nchunks, p.c, p.W, chunk_off, cnt);
    DVP_TRY(prepare_rounds());  // needs the counts only
    if (!sort_done_by_mscan) DVP_TRY(scan_exclusive(cnt, off, nk, bsum, st, /*pad2=*/true));
    hipLaunchKernelGGL(k_scatter_local, dim3(nchunks, p.W), dim3(SORT_TPB), nb * 4, st, digits, (uint32_t)n, p.c, off, chunk_off, items);
  }
  ps_sort.stop();
  // ---- bucket accumulation: batched-affine pair rounds while a round still carries >= aff_min additions,
  // then the projective fan-in-K reducer on what is left (it has a short critical path and balances skew)
  uint32_t* pc[3] = {cnt, ntask, cnt2};
  uint32_t* po[3] = {off, toff, off2};
  bool r0_offsets_ready = false;
  if (!sort_done_by_mscan) {
    // odd buckets' NONE, the first round's counts / offsets (into the ring's next pair, where k_round0_offsets would put them) and the
    // largest bucket in one launch (d_max: zeroed by the previous MSM's tail kernel)
    hipLaunchKernelGGL(k_post_sort, dim3(cdiv(nk + 1, 256)), dim3(256), 0, st, items, cnt, off, nk, pc[1], po[1], d_max);
    r0_offsets_ready = true;
    DVP_TRY(read_max());
  }
  int cur = 0;  // index of the live (cnt, off) pair
  const Aff* bases0 = fx ? fx->table : (const Aff*)d_bases;
  const Aff* pts_in = bases0;
  size_t cap = p.e_max + nk;  // upper bound on the entries of the live array (incl. the even padding of the buckets)
  GfSqrTables Tsq;
  DVP_TRY(gf_sqr_tables(&Tsq, st));
  // resident threads of k_affine_round on this device (3 blocks of 256 per CU on MI355X: 196 608)
  int n_cu = 256, blk_per_cu = 3;
  DVP_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, cur_dev));
  // workgroup size of the pair rounds (Tune::msm_aff_tpb): the kernel has no block-level synchronisation (LDS tables are per wave),
  // so smaller workgroups only change how soon a finished wave's slot is handed to the next workgroup
  const uint32_t aff_tpb = (tn.msm_aff_tpb == 64 || tn.msm_aff_tpb == 128 || tn.msm_aff_tpb == 256) ? (uint32_t)tn.msm_aff_tpb : (uint32_t)EC_TPB;
  const uint32_t aff_lds = (aff_tpb / 64) * GF_LDSK_BYTES_PER_WAVE;
  DVP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blk_per_cu, (const void*)k_affine_round<false, false>, (int)aff_tpb, aff_lds));
  if (blk_per_cu < 1) blk_per_cu = 1;
  const uint32_t aff_cap = (uint32_t)n_cu * (uint32_t)blk_per_cu * aff_tpb;
  const uint32_t aff_bmax_only = tn.msm_aff_bmax >= 1 && tn.msm_aff_bmax <= 255 ? (uint32_t)tn.msm_aff_bmax : AFF_BMAX;
  const uint32_t aff_bmin = tn.msm_aff_bmin >= 1 && (uint32_t)tn.msm_aff_bmin <= aff_bmax_only ? (uint32_t)tn.msm_aff_bmin : 1u;
  const uint32_t aff_bmax = aff_bmax_only | (aff_bmin << 8);
  unsigned long long* wt_buf = nullptr;
  uint32_t wt_cap = 0;
  wave_trace_snapshot(&wt_buf, &wt_cap);  // one snapshot per MSM: every round of it traces into the same buffer or none does
  const bool wt_on = wt_buf != nullptr;
  // the round itself: d_total = the scanned output count (ooff[nkeys]), dsc = one (a, b) descriptor per output slot
  auto launch_round = [&](int r, const uint32_t* d_total, const uint2* dsc) -> int {
    size_t out_max = cap / 2 + nk + 1;
    Aff* outp = (r & 1) ? affB : affA;
    // grid: upper bound on the threads the device-side choice of B can ask for (R chip-fulls, see k_affine_round)
    const uint32_t r_max = cdiv(cdiv(out_max, aff_cap), aff_bmax_only);
    const uint32_t grid = r_max * (aff_cap / aff_tpb) + 1;
    {
      // r == 0 is the dominant kernel (it gathers the bases) and is timed launch by launch; the later rounds share ONE scope (ps_rest,
      // below): an event pair around every round put ~10 us of packet processing between two rounds that otherwise follow each
      // other back to back (kernel trace: 10-13 us gaps after every round, none between the merge levels)
      ProfScope ps0(r == 0 ? PROF_MSM_ACCUM_AFFINE : -1, st, (uint64_t)n);
      WaveTrace wt{wt_buf, wt_cap, g_wave_trace_tag.fetch_add(wt_on ? 1u : 0u)};
      if (r == 0 && wt.buf)
        hipLaunchKernelGGL((k_affine_round<true, true>), dim3(grid), dim3(aff_tpb), aff_lds, st, pts_in, dsc, d_total, aff_cap, aff_bmax, Tsq, prefix, outp, sign_mask, wt);
      else if (r == 0)
        hipLaunchKernelGGL((k_affine_round<true, false>), dim3(grid), dim3(aff_tpb), aff_lds, st, pts_in, dsc, d_total, aff_cap, aff_bmax, Tsq, prefix, outp, sign_mask, wt);
      else if (wt.buf)
        hipLaunchKernelGGL((k_affine_round<false, true>), dim3(grid), dim3(aff_tpb), aff_lds, st, pts_in, dsc, d_total, aff_cap, aff_bmax, Tsq, prefix, outp, 0u, wt);
      else
        hipLaunchKernelGGL((k_affine_round<false, false>), dim3(grid), dim3(aff_tpb), aff_lds, st, pts_in, dsc, d_total, aff_cap, aff_bmax, Tsq, prefix, outp, 0u, wt);
      ps0.stop();
    }
    pts_in = outp;
    cap = out_max;
    return DVP_OK;
  };
  auto run_round = [&](int r) -> int {
    if (pipelined) {  // r < ra_plan always (ra <= ra_plan); everything a round needs was prepared behind the sort
      if (r == 0 && side_stream) DVP_HIP(hipStreamWaitEvent(st, g_ws.ev_side[0], 0));
      DVP_TRY(launch_round(r, ro(r + 1) + nk, r == 0 ? (const uint2*)items : (const uint2*)(const void*)(gdesc + desc_at[r])));
      return DVP_OK;
    }
    const int nxt = (cur + 1) % 3;
    if (r == 0) {
      if (!(r0_offsets_ready && cur == 0))
        hipLaunchKernelGGL(k_round0_offsets, dim3(cdiv(nk + 1, 256)), dim3(256), 0, st, pc[cur], po[cur], nk, pc[nxt], po[nxt]);
    } else
      DVP_TRY(bookkeep(r, pc[cur], po[cur], pc[nxt], po[nxt], gdesc, st, bsum));
    DVP_TRY(launch_round(r, po[nxt] + nk, r == 0 ? (const uint2*)items : (const uint2*)(const void*)gdesc));
    cur = nxt;
    return DVP_OK;
  };
  // The number of rounds depends on the largest bucket, which only the device knows.  When the average bucket already
  // holds >= 4 entries the first round is certain to be needed, so it is enqueued BEFORE the host waits for that number:
  // the round trip (and the launch ramp after it) hides behind ~5 ms of round 0 instead of idling the GPU mid-MSM.
  int launched = 0;
  std::unique_lock<std::mutex> heavy;  // held from the first pair round's enqueue to the last one's (released on every return)
  if (ws_slots > 1 && affine_mode) {
    heavy = std::unique_lock<std::mutex>(g_heavy[cur_dev].mu);
    if (g_heavy[cur_dev].last && g_heavy[cur_dev].last != g_ws.ev_heavy) DVP_HIP(hipStreamWaitEvent(st, g_heavy[cur_dev].last, 0));
  }
  if (affine_mode && (e_est >> 1) >= aff_min && e_est >= 4 * (size_t)nk) {
    DVP_TRY(run_round(0));
    launched = 1;
  }
  DVP_HIP(hipStreamSynchronize(g_ws.aux));  // the caller's stream keeps running round 0 meanwhile
  count_host_wait(launched ? 1 : 0);
  const uint32_t max_cnt = *g_ws.pinned;
  int ra = 0;
  if (affine_mode)
    while (((uint64_t)1 << ra) < max_cnt && (e_est >> (ra + 1)) >= aff_min) ++ra;
  if (ra < launched) ra = launched;  // a first round over single-entry buckets only passes them through
  // the gate's event goes after the last round that still fills the chip several times over (>= Tune::msm_gate_min additions):
  // the late rounds are latency-bound and are exactly what the next MSM's first round should overlap
  int r_evt = -1;
  if (heavy.owns_lock() && ra > 0) {
    const size_t gate_min = tn.msm_gate_min > 0 ? (size_t)tn.msm_gate_min : 1;
    r_evt = 0;
    while (r_evt + 1 < ra && (e_est >> (r_evt + 2)) >= gate_min) ++r_evt;
  }
  auto gate_event = [&]() -> int {
    DVP_HIP(hipEventRecord(g_ws.ev_heavy, st));
    g_heavy[cur_dev].last = g_ws.ev_heavy;
    return DVP_OK;
  };
  if (r_evt >= 0 && r_evt < launched) DVP_TRY(gate_event());
  {
    int r = launched;
    if (r == 0 && ra > 0) {  // (small MSMs: the first round was not enqueued ahead of the largest-bucket read)
      DVP_TRY(run_round(0));
      if (r_evt == 0) DVP_TRY(gate_event());
      r = 1;
    }
    ProfScope ps_rest(ra > r ? PROF_MSM_AFFINE_REST : -1, st, (uint64_t)(ra > r ? ra - r : 0));
    for (; r < ra; ++r) {
      DVP_TRY(run_round(r));
      if (r == r_evt) DVP_TRY(gate_event());
    }
    ps_rest.stop();
  }
  if (pipelined && prepared) {
    // what the rounds left is described by the arrays of round `ra`: they head the ring of three the reducer works in
    if (!launched && side_stream) DVP_HIP(hipStreamWaitEvent(st, g_ws.ev_side[0], 0));  // (not reached: a pipelined MSM always starts its first round early)
    pc[0] = rc(ra); po[0] = ro(ra);
    cur = 0;
  }
  if (heavy.owns_lock()) heavy.unlock();
  uint64_t rem_max = ((uint64_t)max_cnt + ((uint64_t)1 << ra) - 1) >> ra;  // largest bucket after the affine rounds
  if (rem_max <= 1) {
    if (ra == 0)
      hipLaunchKernelGGL(k_bucket_gather_items, dim3(cdiv(nk, 256)), dim3(256), 0, st, bases0, items, cnt, off, nk, bkt, sign_mask);
    else
      hipLaunchKernelGGL(k_bucket_gather_aff, dim3(cdiv(nk, 256)), dim3(256), 0, st, pts_in, pc[cur], po[cur], nk, bkt);
  } else if (ra > 0 && rem_max <= (uint64_t)(tn.msm_bucket_pairs_max >= 0 ? tn.msm_bucket_pairs_max : 0)) {
    // a handful of points per bucket left: one thread per bucket; buckets that meet an exceptional pair are redone from a list
    hipLaunchKernelGGL(k_bucket_pairs, dim3(cdiv(nk, EC_TPB)), dim3(EC_TPB), EC_LDS, st, pts_in, pc[cur], po[cur], nk, bkt, rest_n, rest_list);
    hipLaunchKernelGGL(k_bucket_rest, dim3(cdiv(nk, EC_TPB)), dim3(EC_TPB), EC_LDS, st, pts_in, pc[cur], po[cur], rest_n, rest_list, bkt);
  } else {
    // level 1: mixed additions from affine inputs
    int nxt = (cur + 1) % 3;
    DVP_TRY(scan_exclusive_div(pc[cur], p.K, pc[nxt], po[nxt], nk, bsum, st));
    size_t tmax = cap / p.K + nk + 1;
    const size_t accum_quad_max = tn.msm_accum_quad_max > 0 ? (size_t)tn.msm_accum_quad_max : ACCUM_QUAD_MAX;
    const size_t accum_hex_max = tn.msm_accum_hex_max >= 0 ? (size_t)tn.msm_accum_hex_max : ACCUM_HEX_MAX;
    // tmax is an upper bound on the tasks (the real count sits on the device); quads when even the bound fits one chip-full
    // (the rest list holds task numbers: tmax of them at most; the bucket-pair list's room is nk entries and tmax can exceed it, so
    // the fast flavour runs only where the list is sure to fit.  Tune::msm_accum_fast = 0: the general kernel of rounds 1-5)
    const bool accum_fast = tn.msm_accum_fast != 0 && tmax <= rest_cap;
#define DVP_ACCUM_AFFINE(IND)                                                                                                                  \
  do {                                                                                                                                         \
    if (accum_fast) {                                                                                                                          \
      if (tmax <= accum_quad_max)                                                                                                              \
        hipLaunchKernelGGL((k_accum_affine_fast<IND, true>), dim3(cdiv(4 * tmax, EC_TPB)), dim3(EC_TPB), EC_LDS_Q, st, pts_in, items, pc[cur], \
                           po[cur], po[nxt], nk, p.K, bufA, sign_mask, rest_n, rest_list);                                                    \
      else                                                                                                                                     \
        hipLaunchKernelGGL((k_accum_affine_fast<IND, false>), dim3(cdiv(tmax, EC_TPB)), dim3(EC_TPB), EC_LDS, st, pts_in, items, pc[cur],      \
                           po[cur], po[nxt], nk, p.K, bufA, sign_mask, rest_n, rest_list);                                                    \
      hipLaunchKernelGGL((k_accum_affine_rest<IND>), dim3(cdiv(tmax, EC_TPB)), dim3(EC_TPB), EC_LDS, st, pts_in, items, pc[cur], po[cur],       \
                         po[nxt], nk, p.K, bufA, sign_mask, rest_n, rest_list);                                                                \
    } else if (tmax <= accum_quad_max)                                                                                                         \
      hipLaunchKernelGGL((k_accum_affine<IND, true>), dim3(cdiv(4 * tmax, EC_TPB)), dim3(EC_TPB), EC_LDS_Q, st, pts_in, items, pc[cur], po[cur], \
                         po[nxt], nk, p.K, bufA, sign_mask);                                                                                   \
    else                                                                                                                                       \
      hipLaunchKernelGGL((k_accum_affine<IND, false>), dim3(cdiv(tmax, EC_TPB)), dim3(EC_TPB), EC_LDS, st, pts_in, items, pc[cur], po[cur],     \
                         po[nxt], nk, p.K, bufA, sign_mask);                                                                                   \
  } while (0)
    if (ra == 0) {
      ProfScope ps0(PROF_MSM_ACCUM_AFFINE, st);  // small inputs: no pair rounds, this is the dominant kernel
      DVP_ACCUM_AFFINE(true);
      ps0.stop();
    } else
      DVP_ACCUM_AFFINE(false);
#undef DVP_ACCUM_AFFINE
    cur = nxt;
    cap = tmax;
    Ld *in = bufA, *outb = bufB;
    for (uint64_t left = (rem_max + p.K - 1) / p.K; left > 1; left = (left + p.K - 1) / p.K) {
      nxt = (cur + 1) % 3;
      DVP_TRY(scan_exclusive_div(pc[cur], p.K, pc[nxt], po[nxt], nk, bsum, st));
      tmax = cap / p.K + nk + 1;
      // the LAST level of a small MSM (every bucket ends with one task of at most K partial sums, typically one or two): rows of 16
      // lanes when the tasks fit ACCUM_HEX_MAX (measured at 2^16 points, round 6)
      if (left <= p.K && tmax <= accum_hex_max)
        hipLaunchKernelGGL(k_accum_proj<16>, dim3(cdiv(16 * tmax, EC_TPB)), dim3(EC_TPB), EC_LDS_Q, st, in, pc[cur], po[cur], po[nxt], nk, p.K, outb);
      else if (tmax <= accum_quad_max)
        hipLaunchKernelGGL(k_accum_proj<4>, dim3(cdiv(4 * tmax, EC_TPB)), dim3(EC_TPB), EC_LDS_Q, st, in, pc[cur], po[cur], po[nxt], nk, p.K, outb);
      else
        hipLaunchKernelGGL(k_accum_proj<1>, dim3(cdiv(tmax, EC_TPB)), dim3(EC_TPB), EC_LDS, st, in, pc[cur], po[cur], po[nxt], nk, p.K, outb);
      cap = tmax;
      cur = nxt;
      Ld* t = in; in = outb; outb = t;
    }
    hipLaunchKernelGGL(k_bucket_gather, dim3(cdiv(nk, 256)), dim3(256), 0, st, in, pc[cur], po[cur], nk, bkt);
  }
  ProfScope ps_tail(PROF_MSM_TAIL, st);  // merge tree, Frobenius tail, final add tree
  const int merge_levels = fx ? fx->key_bits() : p.c;
  // bucket 0 (digits of magnitude 2^(c-1)) is kept aside by the first merge level, before it turns slot 0 into the total: k_tail weighs
  // it by 2^(c-1) (a 96-byte device-to-device copy node of its own cost 14 us per MSM)
  Ld* save0 = sign_mask ? tail + 2 * (size_t)p.c : nullptr;
  for (int j = 0; j < merge_levels; ++j) {
    uint32_t total = (nk >> (j + 1)) * (uint32_t)(j + 1);
    const uint32_t quad_max = tn.msm_quad_max > 0 ? (uint32_t)tn.msm_quad_max : MERGE_QUAD_MAX;
    const uint32_t hex_max = tn.msm_hex_max >= 0 ? (uint32_t)tn.msm_hex_max : MERGE_HEX_MAX;
#define DVP_MERGE(G, F, LDSB) hipLaunchKernelGGL((k_merge<G, F>), dim3(cdiv((size_t)G * total, EC_TPB)), dim3(EC_TPB), LDSB, st, bkt, j, total, save0)
    if (total <= hex_max) { if (j == 0) DVP_MERGE(16, true, EC_LDS_Q); else DVP_MERGE(16, false, EC_LDS_Q); }
    else if (total <= quad_max) { if (j == 0) DVP_MERGE(4, true, EC_LDS_Q); else DVP_MERGE(4, false, EC_LDS_Q); }
    else { if (j == 0) DVP_MERGE(1, true, EC_LDS); else DVP_MERGE(1, false, EC_LDS); }
#undef DVP_MERGE
  }
  const int w_tail = fx ? 1 : p.W;  // fixed-base mode has a single bucket set
  uint32_t cntT = (uint32_t)(w_tail * p.c);
  Ld* ta = tail;  // 2 * cntT entries: the two halves of k_tail's ping-pong
  if (sign_mask && tn.msm_hex_max != 0)  // c points: one row of 16 lanes each
    hipLaunchKernelGGL(k_tail<true>, dim3(1), dim3(EC_TPB), EC_LDS_Q, st, bkt, p.c, w_tail, -3, Tsq, ta, (uint32_t*)d_out_xy, (uint32_t*)d_out_inf,
                       (uint8_t*)d_out_enc, d_out_enc ? dvp_codec_get_rule() : 0, err, d_err_defer);
  else if (!sign_mask && cntT > 32 && tn.msm_tail_groups != 0) {
    // many points (the one-shot MSM's W x c): group sums on cdiv(cntT, 16) workgroups first, then the single-workgroup tail on those
    Ld* part = tail + 2 * (size_t)cntT + 1;
    const int nparts = (int)cdiv(cntT, 16);
    hipLaunchKernelGGL(k_tail_groups, dim3(nparts), dim3(EC_TPB), EC_LDS_Q, st, bkt, p.c, w_tail, fx ? 0 : p.n_narrow, Tsq, ta, part);
    hipLaunchKernelGGL(k_tail<false>, dim3(1), dim3(EC_TPB), EC_LDS_Q, st, bkt, nparts, 1, -4, Tsq, part, (uint32_t*)d_out_xy,
                       (uint32_t*)d_out_inf, (uint8_t*)d_out_enc, d_out_enc ? dvp_codec_get_rule() : 0, err, d_err_defer);
  } else
    hipLaunchKernelGGL(k_tail<false>, dim3(1), dim3(EC_TPB), EC_LDS_Q, st, bkt, p.c, w_tail, sign_mask ? -3 : (fx ? 0 : p.n_narrow), Tsq, ta, (uint32_t*)d_out_xy,
                       (uint32_t*)d_out_inf, (uint8_t*)d_out_enc, d_out_enc ? dvp_codec_get_rule() : 0, err, d_err_defer);
  ps_tail.stop();
  ps_total.stop();
  DVP_HIP(hipGetLastError());
  g_ws.ctrl_clean = true;  // the tail kernel is enqueued: it resets the control words
  // the scalar-range flag is the only thing that needs the host
  // into the workspace's pinned words (a pageable destination makes the runtime stage the copy and take a host round trip of its own
  // between the two copies: 60 us per MSM)
  if (d_err_defer) {  // deferred: the workspace stays in use until this point of the stream
    DVP_HIP(hipEventRecord(g_ws.ev_busy, st));
    g_ws.busy_stream = st;
    g_ws.busy_valid = true;
    return DVP_OK;
  }
  volatile unsigned long long* h_err = (volatile unsigned long long*)(g_ws.pinned + 2);
  if (h_copy) DVP_HIP(hipMemcpyAsync(h_copy, d_copy, copy_bytes, hipMemcpyDeviceToHost, st));
  DVP_HIP(hipMemcpyAsync((void*)h_err, err + 2, 8, hipMemcpyDeviceToHost, st));  // err_out
  DVP_HIP(hipStreamSynchronize(st));
  count_host_wait(0);
  const unsigned long long e = *h_err;
  if (e != ~0ull) {
    g_last_error_index = (int64_t)(e & 0xffffffffull);
    return DVP_EINVAL;
  }
  return DVP_OK;
}

// d_pts: n affine points `pt_stride_bytes` apart (>= 64, a multiple of 16), d_inf32: n flags (u32, `inf_stride` words apart) -> their sum
int msm_sum_points_dev(const void* d_pts, uint32_t pt_stride_bytes, const void* d_inf32, uint32_t n, uint32_t inf_stride, void* d_out_xy,
                       void* d_out_inf, hipStream_t st) {
  GfSqrTables Tsq;
  DVP_TRY(gf_sqr_tables(&Tsq, st));
  hipLaunchKernelGGL(k_sum_points, dim3(1), dim3(64), GF_LDS_BYTES_PER_WAVE, st, (const char*)d_pts, pt_stride_bytes, (const uint32_t*)d_inf32, n,
                     inf_stride, Tsq, (uint32_t*)d_out_xy, (uint32_t*)d_out_inf);
  DVP_HIP(hipGetLastError());
  return DVP_OK;
}

int msm_affine_dev(const void* d_scalars, const void* d_bases, const void* d_inf, size_t n, void* d_out_xy,
                   void* d_out_inf, hipStream_t st) {
  return msm_core(d_scalars, d_bases, d_inf, n, nullptr, 0, d_out_xy, d_out_inf, st);
}

// ---- fixed-base contexts (the prover's SRS vectors) -----------------------------------------------
// range_hint = number of bases a typical call will cover (the per-GPU shard): the shared window size c
// minimises an empirical cost in field multiplications (pair additions + a per-bucket term, see below)
void msm_fixed_destroy(MsmFixedCtx* c);
// Table flavours: the default is aligned windows of SIGNED binary digits over W rows 2^(o_w) P (768 B per base at W = 12: 4.8 GB for
// both SRS vectors of a 2^20-constraint prover); DVP_MSM_ALIGNED_SIGNED = 0 selects round 1-2's aligned tau-adic windows over rows
// tau^(o_w) P instead (one window and twice the buckets more: 24.6 against 23.5 ms per proof in round 3) -- kept because it shares
// its recode, merge and Frobenius tail with the one-shot MSM and is the independent cross-check of the signed flavour in the
// tests.  (Rounds 2-3 also carried two SLIDING-window flavours over a multiple of every base for every digit position, 94-97 GB
// of tables at 2^20; the signed aligned windows matched their speed at a twentieth of the memory -- 23.54 against 23.57 / 23.98 ms
// -- and they were removed in round 4 together with their recoders, table builders and test matrix.)
static int msm_fixed_build(const Aff* d_bases, uint32_t n_total, size_t range_hint, MsmFixedCtx** out, hipError_t* alloc_err) {
  MsmFixedCtx* c = new MsmFixedCtx();
  c->n_total = n_total;
  double best = 1e300;
  int best_c = 8;
  const bool signed_aligned = tune().msm_aligned_signed != 0;
  // pair additions at ~5.6 product-equivalents + a per-bucket term for merge/reducer/sort; measured on MI355X at 2^20
  // constraints: 2^18 buckets beat 2^16, 2^17, 2^19 and 2^20 for every shard from 0.26 M to 4.2 M pairs of the tau-adic aligned
  // windows (2^20 wins from ~8 M pairs); the signed windows run the same model over their own entry and bucket counts
  // (ceil(234 / c) entries, 2^(c-1) buckets)
  if (signed_aligned) {
    for (int cc = 8; cc <= FX_C_MAX + 1; ++cc) {
      MsmFixedCtx probe;
      probe.set_c_signed(cc);
      if (probe.c != cc) continue;  // all windows narrow: the same plan as cc - 1
      const int kb = cc - 1;
      const double per_scalar = (double)probe.W;
      double cost = per_scalar * (double)range_hint * 5.6 + 10.0 * (double)(1u << kb) + (kb > 18 ? 25.0 * (double)((1u << kb) - (1u << 18)) : 0.0);
      if (cost < best) { best = cost; best_c = cc; }
    }
    if (tune().msm_fixed_c >= 8 && tune().msm_fixed_c <= FX_C_MAX + 1) best_c = (int)tune().msm_fixed_c;
    c->set_c_signed(best_c);
  } else {
    for (int cc = 8; cc <= FX_C_MAX; ++cc) {
      const double per_scalar = (double)((234 + cc - 1) / cc);
      double cost = per_scalar * (double)range_hint * 5.6 + 10.0 * (double)(1u << cc) + (cc > 18 ? 25.0 * (double)((1u << cc) - (1u << 18)) : 0.0);
      if (cost < best) { best = cost; best_c = cc; }
    }
    if (tune().msm_fixed_c >= 8 && tune().msm_fixed_c <= FX_C_MAX) best_c = (int)tune().msm_fixed_c;
    c->set_c(best_c);
  }
  const int kb = c->key_bits();
  c->hi_bits = kb / 2;  // even split: both levels have <= 2^10 bins and use the LDS-staged scatters
  if (int h = (int)tune().fx_hi; h >= 0 && h <= 10 && kb - h <= 15 && kb - h >= 1) c->hi_bits = h;
  hipError_t e = hipMalloc((void**)&c->table, (size_t)c->rows() * n_total * sizeof(Aff));
  *alloc_err = e;
  if (e == hipSuccess) {
    if (c->signed_digits) {
      GfSqrTables Tsq;
      int rc_t = gf_sqr_tables(&Tsq, 0);
      if (rc_t != DVP_OK) { msm_fixed_destroy(c); return rc_t; }
      e = hipFuncSetAttribute((const void*)k_dbl_table, hipFuncAttributeMaxDynamicSharedMemorySize, EC_LDS);
      Gf* scratch = nullptr;  // Z snapshots + prefix products of the shared inversion, (W - 1) x n each
      const size_t per = (size_t)(c->W > 1 ? c->W - 1 : 1) * n_total;
      if (e == hipSuccess) e = hipMalloc((void**)&scratch, 2 * per * sizeof(Gf));
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_dbl_table, dim3(cdiv(n_total, 256)), dim3(256), EC_LDS, 0, d_bases, n_total, c->c, c->W, c->n_narrow, Tsq, c->table,
                           scratch, scratch + per);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
      }
      if (scratch) (void)hipFree(scratch);
    } else
      hipLaunchKernelGGL(k_frob_table, dim3(cdiv(n_total, 256)), dim3(256), 0, 0, d_bases, n_total, c->c, c->W, c->n_narrow, c->table);
    if (e == hipSuccess) e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    c->table = *alloc_err == hipSuccess ? c->table : nullptr;
    msm_fixed_destroy(c);
    return hip_fail(e, "msm_fixed_create", __FILE__, __LINE__);
  }
  *out = c;
  return DVP_OK;
}
int msm_fixed_create(const Aff* d_bases, uint32_t n_total, size_t range_hint, MsmFixedCtx** out) {
  if ((uint64_t)n_total * 32 >= 0xfffffff0ull) return DVP_EINVAL;  // W <= 31 windows of pre-rotated copies, 32-bit indices
  hipError_t alloc_err = hipSuccess;
  return msm_fixed_build(d_bases, n_total, range_hint, out, &alloc_err);
}
int msm_fixed_info(const MsmFixedCtx* c, int* cbits, int* windows) {
  if (!c) return DVP_EINVAL;
  *cbits = c->c;
  *windows = c->W;
  return DVP_OK;
}
// bytes of the pre-rotated table; *signed_flavour = 1 for the signed binary windows (the default), 0 for the tau-adic aligned windows
const void* msm_fixed_table_ptr(const MsmFixedCtx* c) { return c ? (const void*)c->table : nullptr; }
uint64_t msm_fixed_table_bytes(const MsmFixedCtx* c, int* signed_flavour) {
  if (signed_flavour) *signed_flavour = c && c->signed_digits ? 1 : 0;
  return c ? (uint64_t)c->rows() * c->n_total * sizeof(Aff) : 0;
}
void msm_fixed_destroy(MsmFixedCtx* c) {
  if (!c) return;
  if (c->table) (void)hipFree(c->table);
  delete c;
}
// partial sum over bases [lo, hi) of the context; d_scalars / d_inf point at element lo
int msm_fixed_dev(const MsmFixedCtx* c, const void* d_scalars, const void* d_inf, uint32_t lo, uint32_t hi, void* d_out_xy,
                  void* d_out_inf, hipStream_t st) {
  if (!c || lo > hi || hi > c->n_total) return DVP_EINVAL;
  return msm_core(d_scalars, nullptr, d_inf, hi - lo, c, lo, d_out_xy, d_out_inf, st);
}
// the same with the result's 30-byte encoding written to d_out_enc by the tail kernel itself
// (h_copy, d_copy, copy_bytes): a device block copied to pinned host memory before the call's own final synchronisation
// d_err_defer != nullptr: deferred completion (see msm_core) -- returns once everything is enqueued
int msm_fixed_dev_enc(const MsmFixedCtx* c, const void* d_scalars, const void* d_inf, uint32_t lo, uint32_t hi, void* d_out_xy,
                      void* d_out_inf, void* d_out_enc, void* h_copy, const void* d_copy, size_t copy_bytes, hipStream_t st,
                      unsigned long long* d_err_defer) {
  if (!c || lo > hi || hi > c->n_total) return DVP_EINVAL;
  return msm_core(d_scalars, nullptr, d_inf, hi - lo, c, lo, d_out_xy, d_out_inf, st, d_out_enc, h_copy, d_copy, copy_bytes, d_err_defer);
}
int msm_affine_dev_enc(const void* d_scalars, const void* d_bases, const void* d_inf, size_t n, void* d_out_xy, void* d_out_inf,
                       void* d_out_enc, void* h_copy, const void* d_copy, size_t copy_bytes, hipStream_t st, unsigned long long* d_err_defer) {
  return msm_core(d_scalars, d_bases, d_inf, n, nullptr, 0, d_out_xy, d_out_inf, st, d_out_enc, h_copy, d_copy, copy_bytes, d_err_defer);
}

}  // namespace dvp

using namespace dvp;

extern "C" int dvp_msm_affine_dev(const void* d_scalars, const void* d_bases_xy, const void* d_bases_inf, size_t n,
                                  void* d_out_xy, void* d_out_inf, void* stream) {
  if ((n && (!d_scalars || !d_bases_xy)) || !d_out_xy || !d_out_inf) return DVP_EINVAL;
  return msm_affine_dev(d_scalars, d_bases_xy, d_bases_inf, n, d_out_xy, d_out_inf, (hipStream_t)stream);
}

// sum of n partial points laid out as n records of 80 bytes: x || y (64 B) + u32 infinity flag + pad -- the record a rank
// all-gathers (dv-pari_amd/distributed.py) and the one the in-library device threads hand back
extern "C" int dvp_points_sum_dev(const void* d_records, uint32_t n, void* d_out_xy, void* d_out_inf, void* stream) {
  if (!d_records || !n || !d_out_xy || !d_out_inf) return DVP_EINVAL;
  // the kernel walks the 80-byte records in place (point at +0, flag at +64): no staging buffer, nothing to wait for
  return msm_sum_points_dev(d_records, 80, (const char*)d_records + 64, n, 20, d_out_xy, d_out_inf, (hipStream_t)stream);
}

// GF(2^233) products per second of the hot kernels' multiplier alone (every CU busy, k_affine_round's occupancy): the
// ceiling bench.py's work model divides by, measured in the same process instead of quoted
// d_buf: device buffer of 64 + 64 * n_records bytes, zeroed by the caller (word 0 counts the records); NULL switches the trace off.
// While set, every pair round runs its TRACE instantiation and appends one record per wave (layout: WaveTrace above).
extern "C" int dvp_debug_wave_trace(void* d_buf, uint32_t n_records) {
  {
    std::lock_guard<std::mutex> g(g_wave_trace_mu);
    g_wave_trace_cap = d_buf ? n_records : 0;
    g_wave_trace_buf = (unsigned long long*)d_buf;
    g_wave_trace_tag.store(0);
  }
  // switching the trace off (or to another buffer): launches that took the old snapshot may still be writing into it -- the caller
  // is about to free it
  DVP_HIP(hipDeviceSynchronize());
  return DVP_OK;
}

extern "C" int dvp_ubench_gf_mul(int reps, double* products_per_s) {
  if (reps < 1 || !products_per_s) return DVP_EINVAL;
  int dev = 0, n_cu = 256;
  DVP_HIP(hipGetDevice(&dev));
  DVP_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  const int blocks = n_cu * 3 * 8;  // eight chip-fulls of 3 blocks per CU
  DevBuf out;
  DVP_TRY(out.alloc((size_t)blocks * EC_TPB * sizeof(Gf)));
  hipEvent_t e0, e1;
  DVP_HIP(hipEventCreate(&e0));
  DVP_HIP(hipEventCreate(&e1));
  float best = 1e30f;
  for (int it = 0; it < 3; ++it) {
    DVP_HIP(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_ubench_mul, dim3(blocks), dim3(EC_TPB), EC_LDS, 0, out.as<Gf>(), reps);
    DVP_HIP(hipEventRecord(e1, 0));
    DVP_HIP(hipEventSynchronize(e1));
    float ms = 0;
    DVP_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  DVP_HIP(hipGetLastError());
  *products_per_s = (double)blocks * EC_TPB * reps / (best * 1e-3);
  return DVP_OK;
}

// Random 64-byte gathers per second out of a device table (tools/ubench/gather.hip moved behind the ABI so that bench.py
// measures the ceiling of the first pair round's gathers in the same run, on the prover's own table): every lane reads
// whole 64-byte lines at hashed positions, four independent lines in flight per lane (the pair round keeps the two
// operands of the next slot in flight behind the current one's), at full occupancy -- the most the memory system gives.
__device__ __forceinline__ uint64_t ubench_mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
__global__ void __launch_bounds__(256) k_ubench_gather(const uint4* __restrict__ t, uint64_t nlines, int per, uint32_t seed, uint32_t* __restrict__ out) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll 4
  for (int k = 0; k < per; ++k) {
    const uint64_t idx = ubench_mix(tid * (uint64_t)per + k + ((uint64_t)seed << 40)) % nlines;
    const uint4* p = t + idx * 4;
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc.x ^= a.x ^ b.x ^ c.x ^ d.x; acc.y ^= a.y ^ b.y ^ c.y ^ d.y; acc.z ^= a.z ^ b.z ^ c.z ^ d.z; acc.w ^= a.w ^ b.w ^ c.w ^ d.w;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;  // keeps the loads alive
}
extern "C" int dvp_ubench_gather(const void* d_table, size_t table_bytes, int reps, double* gathers_per_s) {
  if (reps < 1 || reps > 64 || !gathers_per_s || table_bytes < (1u << 20)) return DVP_EINVAL;
  int dev = 0, n_cu = 256;
  DVP_HIP(hipGetDevice(&dev));
  DVP_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  DevBuf own, out;
  if (!d_table) {  // contents do not matter for a read-rate figure, but untouched pages may not be backed: write them once
    DVP_TRY(own.alloc(table_bytes));
    DVP_HIP(hipMemset(own.p, 0x5a, table_bytes));
    d_table = own.p;
  }
  DVP_TRY(out.alloc(16));
  const int per = 32;
  const uint32_t blocks = (uint32_t)n_cu * 3 * 8;
  const uint64_t nlines = table_bytes / 64;
  hipEvent_t e0, e1;
  DVP_HIP(hipEventCreate(&e0));
  DVP_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_ubench_gather, dim3(blocks), dim3(256), 0, 0, (const uint4*)d_table, nlines, per, 1u, out.as<uint32_t>());
  DVP_HIP(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(k_ubench_gather, dim3(blocks), dim3(256), 0, 0, (const uint4*)d_table, nlines, per, 2u + (uint32_t)r, out.as<uint32_t>());
  DVP_HIP(hipEventRecord(e1, 0));
  DVP_HIP(hipEventSynchronize(e1));
  float ms = 0;
  DVP_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  DVP_HIP(hipGetLastError());
  *gathers_per_s = (double)reps * blocks * 256.0 * per / (ms * 1e-3);
  return DVP_OK;
}


// the signed aligned windows (k_recode_signed): out_words[w * n + i] = 0 (digit 0) or 0x80000000 | 0x10000000 if the digit is
// negative | w << 20 | |digit| (|digit| = 2^(c-1) stored as key 0); *windows = ceil(234 / c)
extern "C" int dvp_debug_recode_signed(const uint64_t* scalars, size_t n, int c, uint32_t* out_words, int* windows) {
  if (!windows || c < 8 || c > FX_C_MAX + 1 || n > (1u << 24)) return DVP_EINVAL;
  MsmFixedCtx plan;
  plan.set_c_signed(c);
  if (plan.c != c) return DVP_EINVAL;  // a window size whose windows would all be one bit narrower: ask for c - 1
  *windows = plan.W;
  if (!out_words) return DVP_OK;
  if (!scalars || !n) return DVP_EINVAL;
  DevBuf ds, dw, de;
  DVP_TRY(ds.alloc(n * 32));
  DVP_TRY(dw.alloc((size_t)*windows * n * 4));
  DVP_TRY(de.alloc(8));
  DVP_HIP(hipMemcpy(ds.p, scalars, n * 32, hipMemcpyHostToDevice));
  DVP_HIP(hipMemset(de.p, 0xff, 8));
  hipLaunchKernelGGL(k_recode_signed, dim3(cdiv(n, 256)), dim3(256), 0, 0, ds.as<uint32_t>(), (const uint8_t*)nullptr, (uint32_t)n, c, *windows, plan.n_narrow,
                     dw.as<uint32_t>(), de.as<unsigned long long>());
  DVP_HIP(hipGetLastError());
  DVP_HIP(hipMemcpy(out_words, dw.p, (size_t)*windows * n * 4, hipMemcpyDeviceToHost));
  unsigned long long e = 0;
  DVP_HIP(hipMemcpy(&e, de.p, 8, hipMemcpyDeviceToHost));
  if (e != ~0ull) {
    g_last_error_index = (int64_t)(e & 0xffffffffull);
    return DVP_EINVAL;
  }
  return DVP_OK;
}

extern "C" int dvp_msm_affine(const uint64_t* scalars, const uint64_t* bases_xy, const uint8_t* bases_inf, size_t n,
                              uint64_t out_xy[8], int* out_is_infinity) {
  if ((n && (!scalars || !bases_xy)) || !out_xy || !out_is_infinity) return DVP_EINVAL;
  DevBuf ds, db, di, dout;
  DVP_TRY(ds.alloc(n * 32));
  DVP_TRY(db.alloc(n * 64));
  DVP_TRY(dout.alloc(64 + 16));
  if (n) {
    DVP_HIP(hipMemcpy(ds.p, scalars, n * 32, hipMemcpyHostToDevice));
    DVP_HIP(hipMemcpy(db.p, bases_xy, n * 64, hipMemcpyHostToDevice));
  }
  if (bases_inf && n) {
    DVP_TRY(di.alloc(n));
    DVP_HIP(hipMemcpy(di.p, bases_inf, n, hipMemcpyHostToDevice));
  }
  DVP_TRY(msm_affine_dev(ds.p, db.p, di.p, n, dout.p, (char*)dout.p + 64, 0));
  uint32_t inf;
  DVP_HIP(hipMemcpy(out_xy, dout.p, 64, hipMemcpyDeviceToHost));
  DVP_HIP(hipMemcpy(&inf, (char*)dout.p + 64, 4, hipMemcpyDeviceToHost));
  *out_is_infinity = (int)inf;
  return DVP_OK;
}

// ---- public fixed-base MSM context: many MSMs over one base vector (an SRS) -----------------------------
struct dvp_msm_ctx {
  MsmFixedCtx* fx = nullptr;
  uint8_t* d_inf = nullptr;
  size_t n = 0;
};

extern "C" void dvp_msm_ctx_destroy(dvp_msm_ctx* c);
extern "C" int dvp_msm_ctx_create(const uint64_t* bases_xy, const uint8_t* bases_inf, size_t n, size_t range_hint, dvp_msm_ctx** out) {
  if (!bases_xy || !out || !n || n >= ((size_t)1 << 27)) return DVP_EINVAL;
  if (!range_hint || range_hint > n) range_hint = n;
  DevBuf db;
  DVP_TRY(db.alloc(n * sizeof(Aff)));
  DVP_HIP(hipMemcpy(db.p, bases_xy, n * sizeof(Aff), hipMemcpyHostToDevice));
  dvp_msm_ctx* c = new dvp_msm_ctx();
  c->n = n;
  auto fail = [&](int rc) { dvp_msm_ctx_destroy(c); return rc; };
  hipError_t e = hipMalloc((void**)&c->d_inf, n);
  if (e == hipSuccess) e = bases_inf ? hipMemcpy(c->d_inf, bases_inf, n, hipMemcpyHostToDevice) : hipMemset(c->d_inf, 0, n);
  if (e != hipSuccess) return fail(hip_fail(e, "dvp_msm_ctx_create", __FILE__, __LINE__));
  int rc = msm_fixed_create(db.as<Aff>(), (uint32_t)n, range_hint, &c->fx);
  if (rc != DVP_OK) return fail(rc);
  *out = c;
  return DVP_OK;
}
extern "C" void dvp_msm_ctx_destroy(dvp_msm_ctx* c) {
  if (!c) return;
  msm_fixed_destroy(c->fx);
  if (c->d_inf) (void)hipFree(c->d_inf);
  delete c;
}
extern "C" int dvp_msm_ctx_plan(const dvp_msm_ctx* c, int* c_bits, int* windows) {
  if (!c || !c_bits || !windows) return DVP_EINVAL;
  return msm_fixed_info(c->fx, c_bits, windows);
}
extern "C" uint64_t dvp_msm_ctx_table_bytes(const dvp_msm_ctx* c, int* signed_windows) { return c ? msm_fixed_table_bytes(c->fx, signed_windows) : 0; }
// sum_{i in [lo,hi)} scalars[i - lo] * base[i]; d_scalars holds hi - lo canonical scalars (device)
extern "C" int dvp_msm_ctx_run_dev(dvp_msm_ctx* c, const void* d_scalars, size_t lo, size_t hi, void* d_out_xy, void* d_out_inf, void* stream) {
  if (!c || !d_scalars || !d_out_xy || !d_out_inf || lo > hi || hi > c->n) return DVP_EINVAL;
  return msm_fixed_dev(c->fx, d_scalars, c->d_inf + lo, (uint32_t)lo, (uint32_t)hi, d_out_xy, d_out_inf, (hipStream_t)stream);
}
extern "C" int dvp_msm_ctx_run(dvp_msm_ctx* c, const uint64_t* scalars, size_t lo, size_t hi, uint64_t out_xy[8], int* out_is_infinity) {
  if (!c || !scalars || !out_xy || !out_is_infinity || lo > hi || hi > c->n) return DVP_EINVAL;
  DevBuf ds, dout;
  DVP_TRY(ds.alloc((hi - lo) * 32));
  DVP_TRY(dout.alloc(80));
  if (hi > lo) DVP_HIP(hipMemcpy(ds.p, scalars, (hi - lo) * 32, hipMemcpyHostToDevice));
  DVP_TRY(msm_fixed_dev(c->fx, ds.p, c->d_inf + lo, (uint32_t)lo, (uint32_t)hi, dout.p, (char*)dout.p + 64, 0));
  uint32_t inf;
  DVP_HIP(hipMemcpy(out_xy, dout.p, 64, hipMemcpyDeviceToHost));
  DVP_HIP(hipMemcpy(&inf, (char*)dout.p + 64, 4, hipMemcpyDeviceToHost));
  *out_is_infinity = (int)inf;
  return DVP_OK;
}
