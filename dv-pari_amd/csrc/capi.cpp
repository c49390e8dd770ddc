// Library-wide C ABI pieces: status strings, device selection, last-error bookkeeping.
#include "common.h"

namespace dvp {
thread_local int64_t g_last_error_index = -1;
thread_local hipError_t g_last_hip_error = hipSuccess;
}  // namespace dvp

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
