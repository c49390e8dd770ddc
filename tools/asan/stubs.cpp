// Host-only sanitizer build (make asan): the parsers of untrusted files (csrc/cache.cpp, csrc/tree_io.cpp) compiled with
// -fsanitize=address,undefined and linked against these stand-ins for the entries that need a GPU.  Nothing here is product code.
#include <hip/hip_runtime_api.h>

#include "../../include/dvpari.h"
#include "../../include/dvpari_internal.h"

extern "C" {
int dvp_prover_create(uint32_t, uint32_t, uint32_t, dvp_prover**) { return DVP_EHIP; }
void dvp_prover_destroy(dvp_prover*) {}
int dvp_prover_set_coeffs(dvp_prover*, const uint64_t*, uint32_t) { return DVP_EHIP; }
int dvp_prover_set_matrix(dvp_prover*, int, uint32_t, const uint32_t*, const uint32_t*, const uint32_t*) { return DVP_EHIP; }
int dvp_prover_set_srs_encoded(dvp_prover*, int, const uint8_t*, size_t) { return DVP_EHIP; }
int dvp_prove(dvp_prover*, const uint64_t*, uint32_t, const uint64_t*, uint32_t, uint8_t*) { return DVP_EHIP; }
int dvp_tune_get(const char*, long long* v) { if (v) *v = 1; return DVP_OK; }
hipError_t hipGetDevice(int* d) { if (d) *d = 0; return hipSuccess; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) { if (f) *f = 0; if (t) *t = 0; return hipSuccess; }
}
