// Internal view of the ECFFT context shared by ecfft.hip and prove.hip.
#pragma once
#include <map>
#include <mutex>
#include <vector>

#include "common.h"
#include "fr.cuh"

using dvp::Fr;
using dvp::Fr29;
using dvp::Fr30;

struct MatSet {
  // twisted-butterfly constants (ecfft.hip): (n-1) x 2 entries each (Montgomery form, pre-sliced), layer d at offset 2*(n - (n>>d)):
  // decompose (1 / (s0 - s1), -s0), recombine (s0, s1)
  Fr30* dec = nullptr;
  Fr30* rec = nullptr;
  Fr30* win = nullptr;   // n entries: 1 / W^src(position): the input twist of an extend
  Fr30* wout = nullptr;  // n entries: W^dst(position): its output twist
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
  Fr* scratch = nullptr;        // 6 x N Fr work space for enter/exit
  Fr* d_x0 = nullptr;           // device copies of x0/t (Montgomery)
  Fr* d_t = nullptr;
  std::mutex mu;

  Fr* layer(int d) const { return layers + layer_off[d]; }
};


// in-place extend of `batch` vectors of (N >> sl)/2 values on the stride-2^sl subtree
int extend_inplace(dvp_ecfft* c, int sl, int to_even, dvp::Fr* data, uint32_t batch, hipStream_t st);
// out of place: `src` (batch vectors, read by the first pass only) -> `data`
int extend_from(dvp_ecfft* c, int sl, int to_even, const dvp::Fr* src, dvp::Fr* data, uint32_t batch, hipStream_t st);

namespace dvp {
// Z_0(x) = U - c0 V through the first kk isogenies (x Montgomery in, Montgomery out)
__device__ __forceinline__ Fr vanish_chain(Fr x, const Fr* __restrict__ x0s, const Fr* __restrict__ ts, int kk, Fr c0) {
  Fr u = x, v = fr_one_mont();
  for (int d = 0; d < kk; ++d) {
    Fr x0 = x0s[d], t = ts[d];
    Fr uv = fr_mul(u, v), vv = fr_sqr(v);
    Fr nu = fr_add(fr_sub(fr_sqr(u), fr_mul(x0, uv)), fr_mul(t, vv));
    Fr nv = fr_sub(uv, fr_mul(x0, vv));
    u = nu;
    v = nv;
  }
  return fr_sub(u, fr_mul(c0, v));
}

}  // namespace dvp
