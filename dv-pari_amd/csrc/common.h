// Shared host-side plumbing for libdvpari_hip.so: status codes, HIP error capture, scoped buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

#include "../../include/dvpari_internal.h"

namespace dvp {

extern thread_local int64_t g_last_error_index;
extern thread_local hipError_t g_last_hip_error;

inline int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  g_last_hip_error = e;
  fprintf(stderr, "[dvpari] HIP error %d (%s) at %s:%d: %s\n", (int)e, hipGetErrorString(e), file, line, what);
  return DVP_EHIP;
}

#define DVP_HIP(call)                                                        \
  do {                                                                       \
    hipError_t _e = (call);                                                  \
    if (_e != hipSuccess) return ::dvp::hip_fail(_e, #call, __FILE__, __LINE__); \
  } while (0)

#define DVP_TRY(call)        \
  do {                       \
    int _s = (call);         \
    if (_s != DVP_OK) return _s; \
  } while (0)

// RAII device buffer (host-side convenience for the host-pointer flavours of the ABI)
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  int alloc(size_t n) {
    release();
    if (n == 0) n = 16;
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) {
      p = nullptr;
      return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
    }
    bytes = n;
    return DVP_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <class T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

// Tuning knobs (tools/README.md).  Defaults are the measured optima; the environment (DVP_MSM_C, ...) is read ONCE, at
// the first use, and dvp_tune_set / dvp_tune_reset (include/dvpari.h) change a knob at run time -- that is how the tests
// force every window size / sort flavour / round shape, and how tools/ sweep.  Never read on a hot path.
struct Tune {
  long long msm_c = 0;          // DVP_MSM_C: one-shot window bits (0 = cost model)
  long long msm_k = 0;          // DVP_MSM_K: fan-in of the projective reducer (0 = default)
  long long msm_fixed_c = 0;    // DVP_MSM_FIXED_C: fixed-base window bits (0 = cost model)
  long long fx_hi = -1;         // DVP_FX_HI: level-1 partition bits of the fixed-base sort (-1 = c/2)
  long long msm_proj = 0;       // DVP_MSM_MODE=proj: skip the batched-affine rounds
  long long msm_aff_min = 1ll << 19;   // DVP_MSM_AFF_MIN: pair rounds run while a round has this many additions
  long long msm_gate_min = 1;   // DVP_MSM_GATE_MIN: pair rounds with at least this many additions take turns between concurrent MSMs (HeavyGate); smaller ones overlap
  long long msm_ws_slots = 2;   // DVP_MSM_WS_SLOTS: MSMs that may run at the same time on one device (1 or 2 workspaces; their pair rounds still take turns, HeavyGate)
  long long cache_replicas = 1;  // DVP_CACHE_REPLICAS: provers dvp_prove_cache_dir may hold per cache_dir (1 = callers take turns on one prover, the default since the end of round 4: with the witness in host memory a second prover's latency chains land in the first one's pair rounds and two callers got 23.3-23.8 ms per proof where taking turns gives 21.2-21.4; 2 = a second one is opened when two callers overlap)
  long long msm_aff_tpb = 256;  // DVP_MSM_AFF_TPB: workgroup size of the pair rounds (64, 128 or 256)
  long long msm_aff_bmin = 8;   // DVP_MSM_AFF_BMIN: fewest slots a round thread owns (small rounds then use fewer threads, each sharing its inversion among more additions)
  long long msm_aff_bmax = 136; // DVP_MSM_AFF_BMAX: most slots (additions per shared inversion) a round thread owns (48 through round 4; msm.hip: AFF_BMAX)
  long long ecfft_radix4 = 3;   // DVP_ECFFT_RADIX4: the unfused top of an extend: 3 = up to nine layers in ONE LDS-tiled launch (k_extend_top; what is above them as under 2), 2 = three layers per pass (k_butterfly8, then k_butterfly4 / k_butterfly for what is left), 1 = two, 0 = one
  long long ecfft_fold = 1;     // DVP_ECFFT_FOLD: enter / exit fold their pointwise stages into the first / last pass of their extends (0 = the separate launches of rounds 1-5: the parity tests run both)
  long long msm_accum_fast = 1;  // DVP_MSM_ACCUM_FAST: the fan-in-K reducer's first level without exceptional branches, exceptional tasks redone from a list (0 = the general kernel of rounds 1-5)
  long long msm_tail_groups = 1; // DVP_MSM_TAIL_GROUPS: a one-shot MSM's W x c tail points are first summed in groups of 16 on many workgroups (k_tail_groups; 0 = the single-workgroup tail alone)
  long long msm_accum_hex_max = -1;  // DVP_MSM_ACCUM_HEX_MAX: the fan-in reducer's LAST level uses a row of 16 lanes per task up to this many tasks (-1 = default, 0 = never)
  long long gf_inv_tabs = 1;     // DVP_GF_INV_TABS: squaring runs of the GF(2^233) inversion taken as table passes beyond the three long ones: 0 = none (rounds 1-5), 1 = the run of 14, 2 = the runs of 14 and 7
  long long msm_accum_quad_max = 0;    // DVP_MSM_ACCUM_QUAD_MAX: reducer launches with at most this many tasks (upper bound) use a quad of lanes per task (0 = default)
  long long msm_bucket_pairs_max = 12;  // DVP_MSM_BUCKET_PAIRS_MAX: what the pair rounds leave goes through k_bucket_pairs / k_bucket_rest (one thread per bucket) when no bucket holds more points than this; above it, and with 0, through the fan-in-K reducer
  long long msm_sort_fused = 1;  // DVP_MSM_SORT_FUSED: level 1 of the signed flavour's sort recomputes the entry words from the scalars (0 = k_recode_signed writes them to HBM first)
  long long msm_round_pipeline = 2;  // DVP_MSM_ROUND_PIPELINE: bookkeeping of the pair rounds (counts / offsets / descriptors): 2 = all rounds at once, four launches on the caller's stream before the first round (default); 1 = the same on a side stream behind the sort's last scatter (equal one proof at a time, 1 ms slower with two in flight); 0 = per round, between the rounds
  long long msm_hex_max = -1;   // DVP_MSM_HEX_MAX: merge levels up to this many additions use a row of 16 lanes each (-1 = default, 0 = never)
  long long msm_quad_max = 0;   // DVP_MSM_QUAD_MAX: merge levels up to this many additions use a quad of lanes each (0 = default)
  long long msm_fixed_min = 1ll << 16; // DVP_MSM_FIXED_MIN: smallest shard the prover sends through the fixed-base tables
  long long fr_bi_shape = 6;           // DVP_FR_BI_SHAPE: workgroup x elements per thread of k_batch_inverse: 0 = 512 x 16 (rounds 3-4), 1 = 256 x 8, 2 = 256 x 4, 3 = 512 x 8, 4 = 128 x 8, 5 = 128 x 4, 6 = 256 x 16 (default: 2^21 elements 128 us against 139), 7 = 1024 x 8
  long long horner_max_pub = -1;       // DVP_HORNER_MAX_PUB: public-input count up to which i(X) on D' is evaluated by Horner (-1 = default)
  long long prove_host_transcript = 0; // DVP_PROVE_HOST_TRANSCRIPT: 1 = dvp_prove_dev waits for the commitment MSM and hashes the transcript on the host (rounds 1-4); 0 = on the device, one stream wait per proof
  long long msm_aligned_signed = 1;    // DVP_MSM_ALIGNED_SIGNED: the aligned-window tables hold 2^(c w) P and the windows are signed binary digits (0 = the tau-adic aligned windows over rows tau^(o_w) P)
};
Tune& tune();

// devices of the in-library multi-GPU mode (dvp_set_devices); empty = single device (whatever is current)
std::vector<int> mgpu_devices();

inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// ---- per-kernel HIP-event timers (bench.py's roofline leg; off by default) ------------------------
// A timed launch records two events on the launch stream; prof_collect() (called where the host
// already synchronises) folds the elapsed times into named slots readable through dvp_profile_read.
enum ProfSlot { PROF_MSM_ACCUM_AFFINE = 0, PROF_MSM_TOTAL, PROF_EXTEND_TOTAL, PROF_PROVE_TOTAL, PROF_MSM_AFFINE_REST, PROF_MSM_SORT, PROF_MSM_TAIL, PROF_NSLOTS };
extern bool g_prof_enabled;
struct ProfScope {
  int slot;
  hipStream_t st;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  uint64_t key = 0;  // launch shape (first pair rounds: the MSM's pair count), see dvp_profile_round0_shapes
  ProfScope(int slot_, hipStream_t st_, uint64_t key_ = 0);
  void stop();
};
void prof_collect();
// host waits on the proof path, counted always (dvp_profile_read "host_waits_stream" / "host_waits_side"): kind 0 = a synchronisation
// of the caller's stream (the GPU idles until the host has reacted), kind 1 = the largest-bucket read of an MSM, taken on a side
// stream while the first pair round runs (the host waits, the GPU does not)
void count_host_wait(int kind);

}  // namespace dvp
