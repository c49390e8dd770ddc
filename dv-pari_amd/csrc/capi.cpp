// Library-wide C ABI pieces: status strings, device selection, last-error bookkeeping.
#include "common.h"

#include <atomic>
#include <map>
#include <mutex>
#include <vector>
#include <cstring>
#include <cstdlib>

namespace dvp {
thread_local int64_t g_last_error_index = -1;
thread_local hipError_t g_last_hip_error = hipSuccess;

bool g_prof_enabled = false;
static unsigned g_prof_mask = ~0u;  // which scopes record events (dvp_profile_enable: 1 = all, 2 = the dominant kernel's only)
static std::mutex g_prof_mu;
static double g_prof_ms[PROF_NSLOTS] = {0};
static uint64_t g_prof_n[PROF_NSLOTS] = {0};
struct Pending { int slot; hipEvent_t e0, e1; uint64_t key; };
struct Shape { double ms = 0; uint64_t n = 0; };
static std::map<uint64_t, Shape> g_prof_round0;  // first pair rounds by launch shape (key = (scalar, base) pairs of the MSM)
static std::vector<Pending> g_prof_pending;
static const char* kProfNames[PROF_NSLOTS] = {"msm_affine_round0", "msm_total", "extend_total", "prove_total", "msm_affine_rest", "msm_sort", "msm_tail"};

ProfScope::ProfScope(int slot_, hipStream_t st_, uint64_t key_) : slot(slot_), st(st_), key(key_) {
  if (!g_prof_enabled || slot_ < 0 || !((g_prof_mask >> slot_) & 1u)) return;  // slot -1: a scope that times nothing
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
  (void)hipEventRecord(e0, st);
}
void ProfScope::stop() {
  if (!e0) return;
  (void)hipEventRecord(e1, st);
  std::lock_guard<std::mutex> g(g_prof_mu);
  g_prof_pending.push_back({slot, e0, e1, key});
  e0 = e1 = nullptr;
}
void prof_collect() {
  std::lock_guard<std::mutex> g(g_prof_mu);
  for (auto& p : g_prof_pending) {
    float ms = 0;
    if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      g_prof_ms[p.slot] += ms;
      g_prof_n[p.slot] += (p.slot == PROF_MSM_AFFINE_REST && p.key) ? p.key : 1;  // the later pair rounds of an MSM share one scope: key = their number
      if (p.slot == PROF_MSM_ACCUM_AFFINE && p.key) { Shape& sh = g_prof_round0[p.key]; sh.ms += ms; sh.n += 1; }
    }
    (void)hipEventDestroy(p.e0);
    (void)hipEventDestroy(p.e1);
  }
  g_prof_pending.clear();
}

static std::atomic<uint64_t> g_host_waits[2];
void count_host_wait(int kind) { g_host_waits[kind & 1].fetch_add(1, std::memory_order_relaxed); }

static void tune_from_env(Tune& t) {
  t = Tune();
  auto geti = [](const char* name, long long dflt) { const char* e = getenv(name); return e ? atoll(e) : dflt; };
  t.msm_c = geti("DVP_MSM_C", t.msm_c);
  t.msm_k = geti("DVP_MSM_K", t.msm_k);
  t.msm_fixed_c = geti("DVP_MSM_FIXED_C", t.msm_fixed_c);
  t.fx_hi = geti("DVP_FX_HI", t.fx_hi);
  const char* m = getenv("DVP_MSM_MODE");
  t.msm_proj = (m && !strcmp(m, "proj")) ? 1 : 0;
  t.msm_aff_min = geti("DVP_MSM_AFF_MIN", t.msm_aff_min);
  t.msm_hex_max = geti("DVP_MSM_HEX_MAX", t.msm_hex_max);
  t.msm_round_pipeline = geti("DVP_MSM_ROUND_PIPELINE", t.msm_round_pipeline);
  t.msm_sort_fused = geti("DVP_MSM_SORT_FUSED", t.msm_sort_fused);
  t.msm_bucket_pairs_max = geti("DVP_MSM_BUCKET_PAIRS_MAX", t.msm_bucket_pairs_max);
  t.msm_aff_bmax = geti("DVP_MSM_AFF_BMAX", t.msm_aff_bmax);
  t.msm_aff_bmin = geti("DVP_MSM_AFF_BMIN", t.msm_aff_bmin);
  t.ecfft_radix4 = geti("DVP_ECFFT_RADIX4", t.ecfft_radix4);
  t.ecfft_fold = geti("DVP_ECFFT_FOLD", t.ecfft_fold);
  t.msm_accum_fast = geti("DVP_MSM_ACCUM_FAST", t.msm_accum_fast);
  t.msm_tail_groups = geti("DVP_MSM_TAIL_GROUPS", t.msm_tail_groups);
  t.msm_accum_hex_max = geti("DVP_MSM_ACCUM_HEX_MAX", t.msm_accum_hex_max);
  t.gf_inv_tabs = geti("DVP_GF_INV_TABS", t.gf_inv_tabs);
  t.msm_ws_slots = geti("DVP_MSM_WS_SLOTS", t.msm_ws_slots);
  t.msm_gate_min = geti("DVP_MSM_GATE_MIN", t.msm_gate_min);
  t.msm_aff_tpb = geti("DVP_MSM_AFF_TPB", t.msm_aff_tpb);
  t.cache_replicas = geti("DVP_CACHE_REPLICAS", t.cache_replicas);
  t.msm_quad_max = geti("DVP_MSM_QUAD_MAX", t.msm_quad_max);
  t.msm_accum_quad_max = geti("DVP_MSM_ACCUM_QUAD_MAX", t.msm_accum_quad_max);
  t.msm_fixed_min = geti("DVP_MSM_FIXED_MIN", t.msm_fixed_min);
  t.horner_max_pub = geti("DVP_HORNER_MAX_PUB", t.horner_max_pub);
  t.fr_bi_shape = geti("DVP_FR_BI_SHAPE", t.fr_bi_shape);
  t.msm_aligned_signed = geti("DVP_MSM_ALIGNED_SIGNED", t.msm_aligned_signed);
  t.prove_host_transcript = geti("DVP_PROVE_HOST_TRANSCRIPT", t.prove_host_transcript);
}
static std::mutex g_dev_mu;
static std::vector<int> g_devices;
std::vector<int> mgpu_devices() {
  std::lock_guard<std::mutex> g(g_dev_mu);
  return g_devices;
}
Tune& tune() {
  static Tune t = [] { Tune x; tune_from_env(x); return x; }();
  return t;
}
}  // namespace dvp

static long long* tune_slot(const char* name) {
  dvp::Tune& t = dvp::tune();
  struct { const char* n; long long* v; } tab[] = {
      {"DVP_MSM_C", &t.msm_c}, {"DVP_MSM_K", &t.msm_k}, {"DVP_MSM_FIXED_C", &t.msm_fixed_c}, {"DVP_FX_HI", &t.fx_hi},
      {"DVP_MSM_PROJ", &t.msm_proj}, {"DVP_MSM_AFF_MIN", &t.msm_aff_min}, {"DVP_MSM_HEX_MAX", &t.msm_hex_max}, {"DVP_MSM_ROUND_PIPELINE", &t.msm_round_pipeline}, {"DVP_MSM_SORT_FUSED", &t.msm_sort_fused}, {"DVP_MSM_BUCKET_PAIRS_MAX", &t.msm_bucket_pairs_max}, {"DVP_MSM_AFF_BMAX", &t.msm_aff_bmax}, {"DVP_MSM_AFF_BMIN", &t.msm_aff_bmin}, {"DVP_ECFFT_RADIX4", &t.ecfft_radix4}, {"DVP_ECFFT_FOLD", &t.ecfft_fold}, {"DVP_MSM_ACCUM_FAST", &t.msm_accum_fast}, {"DVP_MSM_TAIL_GROUPS", &t.msm_tail_groups}, {"DVP_MSM_ACCUM_HEX_MAX", &t.msm_accum_hex_max}, {"DVP_GF_INV_TABS", &t.gf_inv_tabs}, {"DVP_MSM_WS_SLOTS", &t.msm_ws_slots}, {"DVP_MSM_GATE_MIN", &t.msm_gate_min}, {"DVP_MSM_AFF_TPB", &t.msm_aff_tpb}, {"DVP_CACHE_REPLICAS", &t.cache_replicas},
      {"DVP_MSM_QUAD_MAX", &t.msm_quad_max}, {"DVP_MSM_ACCUM_QUAD_MAX", &t.msm_accum_quad_max}, {"DVP_MSM_FIXED_MIN", &t.msm_fixed_min}, {"DVP_HORNER_MAX_PUB", &t.horner_max_pub}, {"DVP_FR_BI_SHAPE", &t.fr_bi_shape},
      {"DVP_MSM_ALIGNED_SIGNED", &t.msm_aligned_signed}, {"DVP_PROVE_HOST_TRANSCRIPT", &t.prove_host_transcript}};
  for (auto& e : tab)
    if (!strcmp(name, e.n)) return e.v;
  return nullptr;
}
extern "C" int dvp_tune_set(const char* name, long long value) {
  long long* v = name ? tune_slot(name) : nullptr;
  if (!v) return DVP_EINVAL;
  *v = value;
  return DVP_OK;
}
extern "C" int dvp_tune_get(const char* name, long long* value) {
  long long* v = name ? tune_slot(name) : nullptr;
  if (!v || !value) return DVP_EINVAL;
  *value = *v;
  return DVP_OK;
}
extern "C" void dvp_tune_reset(void) { dvp::tune_from_env(dvp::tune()); }

// on = 1: every stage scope; on = 2: only the first pair rounds' (the dominant kernel: two events per MSM) -- an event pair is a packet of
// its own on the stream and puts ~10 us between two kernels that otherwise follow back to back, so the stage breakdown of a proof
// (seven scopes per MSM) is measured in a pass of its own, not in a timed loop
extern "C" void dvp_profile_enable(int on) {
  dvp::g_prof_mask = on == 2 ? (1u << dvp::PROF_MSM_ACCUM_AFFINE) : ~0u;
  dvp::g_prof_enabled = on != 0;
}
extern "C" void dvp_profile_reset(void) {
  dvp::prof_collect();
  std::lock_guard<std::mutex> g(dvp::g_prof_mu);
  for (int i = 0; i < dvp::PROF_NSLOTS; ++i) { dvp::g_prof_ms[i] = 0; dvp::g_prof_n[i] = 0; }
  dvp::g_prof_round0.clear();
  dvp::g_host_waits[0] = 0;
  dvp::g_host_waits[1] = 0;
}
// the "msm_affine_round0" slot split by launch shape: one entry per distinct MSM size seen since the last reset
extern "C" int dvp_profile_round0_shapes(uint64_t* pairs, double* total_ms, uint64_t* launches, int cap) {
  if (cap < 0 || (cap > 0 && (!pairs || !total_ms || !launches))) return DVP_EINVAL;
  dvp::prof_collect();
  std::lock_guard<std::mutex> g(dvp::g_prof_mu);
  int k = 0;
  for (auto& e : dvp::g_prof_round0) {
    if (k < cap) { pairs[k] = e.first; total_ms[k] = e.second.ms; launches[k] = e.second.n; }
    ++k;
  }
  return k;
}
extern "C" int dvp_profile_read(const char* name, double* total_ms, uint64_t* launches) {
  if (!name || !total_ms || !launches) return DVP_EINVAL;
  if (!strcmp(name, "host_waits_stream") || !strcmp(name, "host_waits_side")) {  // counts since the last dvp_profile_reset
    *total_ms = 0;
    *launches = dvp::g_host_waits[name[11] == 's' && name[12] == 'i' ? 1 : 0].load();
    return DVP_OK;
  }
  dvp::prof_collect();
  std::lock_guard<std::mutex> g(dvp::g_prof_mu);
  for (int i = 0; i < dvp::PROF_NSLOTS; ++i)
    if (!strcmp(name, dvp::kProfNames[i])) { *total_ms = dvp::g_prof_ms[i]; *launches = dvp::g_prof_n[i]; return DVP_OK; }
  return DVP_EINVAL;
}

extern "C" const char* dvp_strerror(int s) {
  switch (s) {
    case DVP_OK: return "ok";
    case DVP_EINVAL: return "invalid argument (size, null pointer or non-canonical field element)";
    case DVP_EDECODE: return "invalid xsk233 point encoding";
    case DVP_EUNSAT: return "R1CS constraint not satisfied by the witness";
    case DVP_EHIP: return "HIP runtime error";
    case DVP_EIO: return "I/O error";
    case DVP_ENOMEM: return "out of memory";
    case DVP_ECHALLENGE: return "Fiat-Shamir challenge lies in the evaluation domain";
    default: return "unknown status";
  }
}

extern "C" int dvp_version(void) { return 100; }

extern "C" int dvp_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return DVP_EHIP;
  return n;
}

// In-library multi-GPU: the devices dvp_prove / dvp_prove_dev / dvp_prove_cache_dir spread the two MSMs of a proof over
// (one host thread per entry).  ids[0] should be the device the prover was created on (its "home"); an id may repeat
// (that is how the path is tested on a one-GPU box).  n <= 1 (or ids == NULL) switches back to single-device proving.
extern "C" int dvp_set_devices(const int* ids, int n) {
  int cnt = 0;
  DVP_HIP(hipGetDeviceCount(&cnt));
  if (n < 0 || n > 64 || (n > 0 && !ids)) return DVP_EINVAL;
  for (int i = 0; i < n; ++i)
    if (ids[i] < 0 || ids[i] >= cnt) return DVP_EINVAL;
  std::lock_guard<std::mutex> g(dvp::g_dev_mu);
  dvp::g_devices.assign(ids, ids + (n > 1 ? n : 0));
  return DVP_OK;
}

extern "C" int dvp_set_device(int id) {
  DVP_HIP(hipSetDevice(id));
  return DVP_OK;
}

extern "C" int64_t dvp_last_error_index(void) { return dvp::g_last_error_index; }
