// Library-wide C ABI pieces: status strings, device selection, last-error bookkeeping.
#include "common.h"

#include <mutex>
#include <vector>
#include <cstring>

namespace dvp {
thread_local int64_t g_last_error_index = -1;
thread_local hipError_t g_last_hip_error = hipSuccess;

bool g_prof_enabled = false;
static std::mutex g_prof_mu;
static double g_prof_ms[PROF_NSLOTS] = {0};
static uint64_t g_prof_n[PROF_NSLOTS] = {0};
struct Pending { int slot; hipEvent_t e0, e1; };
static std::vector<Pending> g_prof_pending;
static const char* kProfNames[PROF_NSLOTS] = {"msm_affine_round0", "msm_total", "extend_total", "prove_total", "msm_affine_rest", "msm_sort", "msm_tail"};

ProfScope::ProfScope(int slot_, hipStream_t st_) : slot(slot_), st(st_) {
  if (!g_prof_enabled) return;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
  (void)hipEventRecord(e0, st);
}
void ProfScope::stop() {
  if (!e0) return;
  (void)hipEventRecord(e1, st);
  std::lock_guard<std::mutex> g(g_prof_mu);
  g_prof_pending.push_back({slot, e0, e1});
  e0 = e1 = nullptr;
}
void prof_collect() {
  std::lock_guard<std::mutex> g(g_prof_mu);
  for (auto& p : g_prof_pending) {
    float ms = 0;
    if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      g_prof_ms[p.slot] += ms;
      g_prof_n[p.slot] += 1;
    }
    (void)hipEventDestroy(p.e0);
    (void)hipEventDestroy(p.e1);
  }
  g_prof_pending.clear();
}
}  // namespace dvp

extern "C" void dvp_profile_enable(int on) { dvp::g_prof_enabled = on != 0; }
extern "C" void dvp_profile_reset(void) {
  dvp::prof_collect();
  std::lock_guard<std::mutex> g(dvp::g_prof_mu);
  for (int i = 0; i < dvp::PROF_NSLOTS; ++i) { dvp::g_prof_ms[i] = 0; dvp::g_prof_n[i] = 0; }
}
extern "C" int dvp_profile_read(const char* name, double* total_ms, uint64_t* launches) {
  if (!name || !total_ms || !launches) return DVP_EINVAL;
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
    case DVP_ERCCL: return "RCCL error";
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

extern "C" int dvp_set_device(int id) {
  DVP_HIP(hipSetDevice(id));
  return DVP_OK;
}

extern "C" int64_t dvp_last_error_index(void) { return dvp::g_last_error_index; }
