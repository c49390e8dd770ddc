// SRS::verifier_runs_setup behind the C ABI (src/srs.rs:177-361 -> compute_srs_matrices, src/srs.rs:112-167): reads
// cache_dir/r1cs_to_dvsnark, computes the five scalar vectors of the SRS on the device, turns them into points with the batched
// fixed-base multiplication (codec.hip: k_mulgen), writes g_m, g_q, g_k_0..2 in the reference's point-vector format and --
// optionally -- the domain files a reference prover / verifier would otherwise spend hours on (z_poly, z_polyd, bar_wts,
// bar_wtsd, z_vals2inv, z_vals2dinv; src/artifacts.rs:86-110).  Every vector stays on the device between the stages: the
// python orchestration this replaces (dv-pari_amd/srs.py, round 1-3) bounced each one through host memory per operation.
//
//   L_i(tau)  = Z_D(tau) / ((tau - d_i) Z_D'(d_i))                                    src/ec_fft.rs:340-390
//   L'_i(tau) likewise on D'; the unified-domain values interleaved [D_i, D'_i]        src/ec_fft.rs:424-450
//   m_j(tau, delta) = sum_i (A_ij + delta B_ij + delta^2 C'_ij) L_i(tau)               src/srs.rs:53-84 (accumulate_m_values)
//       with C' = C - D, D_ij = d_i^j on the public wires                              src/gnark_r1cs.rs:333-386
//   g_m[j] = eps m_j G,  g_q[i] = eps Z_D(tau) delta^2 L'_i(tau) G,
//   g_k[0][i] = L_i(tau) G,  g_k[1][i] = delta L_i(tau) G,  g_k[2] = delta^2 (unified-domain Lagrange values) G    src/srs.rs:126-160
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "ecfft_internal.h"
#include "k233.cuh"

extern "C" int dvp_ecfft_create(uint32_t log_n, int shifted, uint32_t base_log, dvp_ecfft** out);
extern "C" void dvp_ecfft_destroy(dvp_ecfft* c);
extern "C" int dvp_ecfft_exit_dev(dvp_ecfft* c, const void* d_evals, void* d_out, void* stream);
int ecfft_domain_tables_dev(dvp_ecfft* t, int which, Fr* d_bar_weights, Fr* d_zinv_other, hipStream_t st);

namespace dvp {
int batch_inverse_dev(Fr* d, size_t n, hipStream_t st);
int mulgen_dev(const void* d_scalars, size_t n, Aff* d_out, uint8_t* d_inf, hipStream_t st);
int encode_dev(const Aff* d_pts, const uint8_t* d_inf, size_t n, uint8_t* d_out, hipStream_t st);

namespace {
constexpr int TPB = 256;
// canonical copies of D (even leaves) and D' (odd leaves) of the 2m-leaf tree (layer 0 is stored in Montgomery form)
__global__ void __launch_bounds__(TPB) ks_domains(const Fr* __restrict__ L0, uint32_t m, Fr* __restrict__ d, Fr* __restrict__ d2) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  d[i] = fr_from_mont(L0[2 * (size_t)i]);
  d2[i] = fr_from_mont(L0[2 * (size_t)i + 1]);
}
__global__ void __launch_bounds__(TPB) ks_scalar_sub(Fr s, const Fr* __restrict__ a, Fr* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = fr_sub(s, a[i]);
}
// o = s * a * b   (a, b canonical; s in Montgomery form)
__global__ void __launch_bounds__(TPB) ks_mul_scale(const Fr* __restrict__ a, const Fr* __restrict__ b, Fr s_m, Fr* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = fr_mul(s_m, fr_mul(fr_to_mont(a[i]), b[i]));
}
// o[stride * i + off] = s * a[i] * b[i]
__global__ void __launch_bounds__(TPB) ks_mul_scale_strided(const Fr* __restrict__ a, const Fr* __restrict__ b, Fr s_m, Fr* __restrict__ o, size_t n,
                                                            uint32_t stride, uint32_t off) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[(size_t)stride * i + off] = fr_mul(s_m, fr_mul(fr_to_mont(a[i]), b[i]));
}
__global__ void __launch_bounds__(TPB) ks_scale(const Fr* __restrict__ a, Fr s_m, Fr* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = fr_mul(s_m, a[i]);
}
__global__ void __launch_bounds__(TPB) ks_mul(const Fr* __restrict__ a, const Fr* __restrict__ b, Fr* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = fr_mul(fr_to_mont(a[i]), b[i]);
}
__global__ void __launch_bounds__(TPB) ks_fill_one(Fr* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = fr_one_canon();
}
__global__ void __launch_bounds__(TPB) ks_dot_partial(const Fr* __restrict__ a, const Fr* __restrict__ b, size_t n, Fr* __restrict__ partial) {
  __shared__ Fr sh[TPB];
  Fr s = fr_zero();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    s = fr_add(s, fr_mul(fr_to_mont(a[i]), b[i]));
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = TPB / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fr_add(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
// accumulate_m_values on the TRANSPOSED matrices: thread j owns wire j and sums (A_ij + delta B_ij + delta^2 C_ij) L_i over the
// rows i that mention it (src/srs.rs:53-84 walks the rows and scatters; a gather per wire needs no atomics).  One CSR per
// matrix over the wires: col = row index, cid = coefficient id.
struct CsrT {
  const uint32_t* ptr;
  const uint32_t* row;
  const uint32_t* cid;
};
__global__ void __launch_bounds__(TPB) ks_m_values(CsrT A, CsrT B, CsrT C, const Fr* __restrict__ coeffs_m /* Montgomery */, const Fr* __restrict__ l_tau,
                                                   Fr delta_m, Fr delta2_m, Fr eps_m, uint32_t n_wires, Fr* __restrict__ out) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_wires) return;
  auto gather = [&](const CsrT& M) {
    Fr acc = fr_zero();
    for (uint32_t k = M.ptr[j]; k < M.ptr[j + 1]; ++k) acc = fr_add(acc, fr_mul(coeffs_m[M.cid[k]], l_tau[M.row[k]]));
    return acc;
  };
  Fr v = fr_add(fr_add(gather(A), fr_mul(delta_m, gather(B))), fr_mul(delta2_m, gather(C)));
  out[j] = fr_mul(eps_m, v);
}
// target -= s_m x (sum of the nb block partials of ks_dot_partial)   (canonical values, s_m Montgomery; one block)
__global__ void __launch_bounds__(TPB) ks_fold_apply(const Fr* __restrict__ partial, uint32_t nb, Fr s_m, Fr* __restrict__ target) {
  __shared__ Fr sh[TPB];
  Fr acc = fr_zero();
  for (uint32_t k = threadIdx.x; k < nb; k += TPB) acc = fr_add(acc, partial[k]);
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = TPB / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fr_add(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *target = fr_sub(*target, fr_mul(s_m, sh[0]));
}
__global__ void __launch_bounds__(TPB) ks_to_mont(const Fr* __restrict__ a, Fr* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = fr_to_mont(a[i]);
}
// evaluations of Z_S on the 2m leaves: 0 on S's own half, 1 / zinv_other on the other half
__global__ void __launch_bounds__(TPB) ks_vanish_evals(const Fr* __restrict__ z_other /* canonical Z_S on the other half */, uint32_t m, int which,
                                                       Fr* __restrict__ ev) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  ev[2 * (size_t)i + which] = fr_zero();
  ev[2 * (size_t)i + 1 - which] = z_other[i];
}
__global__ void __launch_bounds__(TPB) ks_check_monic(const Fr* __restrict__ co, uint32_t m, unsigned int* bad) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // coefficients m .. 2m-1
  if (i >= m) return;
  const Fr c = co[(size_t)m + i];
  const bool ok = i == 0 ? fr_eq(c, fr_one_canon()) : fr_is_zero(c);
  if (!ok) atomicExch(bad, 1u);
}

// evaluate_vanishing_poly_at_domain's verdict (src/proving.rs:290-295): the evaluations on D (even leaves) must all be zero
__global__ void __launch_bounds__(TPB) ks_first_nonzero_even(const Fr* __restrict__ ev, uint32_t m, unsigned int* __restrict__ first_bad) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  if (!fr_is_zero(ev[2 * (size_t)i])) atomicMin(first_bad, i);
}

struct File {
  const uint8_t* p = nullptr;
  size_t len = 0;
  int fd = -1;
  ~File() {
    if (p) munmap((void*)p, len);
    if (fd >= 0) close(fd);
  }
  int open_ro(const char* path) {
    fd = ::open(path, O_RDONLY);
    if (fd < 0) return DVP_EIO;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) return DVP_EIO;
    len = (size_t)st.st_size;
    void* m = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) return DVP_EIO;
    p = (const uint8_t*)m;
    return DVP_OK;
  }
};

// wire-major copy of a row-major CSR (counting sort by wire; stable, so rows ascend inside a wire)
void transpose(uint32_t n_rows, uint32_t n_wires, const uint32_t* rp, const uint32_t* wire, const uint32_t* cid, std::vector<uint32_t>& tp,
               std::vector<uint32_t>& trow, std::vector<uint32_t>& tcid) {
  const size_t nnz = rp[n_rows];
  tp.assign((size_t)n_wires + 1, 0);
  for (size_t k = 0; k < nnz; ++k) ++tp[wire[k] + 1];
  for (uint32_t j = 0; j < n_wires; ++j) tp[j + 1] += tp[j];
  trow.resize(nnz ? nnz : 1);
  tcid.resize(nnz ? nnz : 1);
  std::vector<uint32_t> cur(tp.begin(), tp.end() - 1);
  for (uint32_t r = 0; r < n_rows; ++r)
    for (uint32_t k = rp[r]; k < rp[r + 1]; ++k) {
      const uint32_t pos = cur[wire[k]]++;
      trow[pos] = r;
      tcid[pos] = cid[k];
    }
}

int up(DevBuf& b, const void* h, size_t bytes) {
  DVP_TRY(b.alloc(bytes));
  if (bytes) DVP_HIP(hipMemcpy(b.p, h, bytes, hipMemcpyHostToDevice));
  return DVP_OK;
}
Fr load_fr(const uint64_t v[4]) {
  Fr r;
  memcpy(r.v, v, 32);
  return r;
}
}  // namespace
}  // namespace dvp

using namespace dvp;

// The scalar pipeline of SRS::verifier_runs_setup on the device (src/srs.rs:177-361 -> compute_srs_matrices' inputs, :112-160): the
// domain tables from the isogeny chain, L_i(tau) / L'_i(tau) by the barycentric formula, the unified-domain values, accumulate_m_values
// as a gather per wire over the transposed matrices, the Vandermonde fold.  ONE implementation: dvp_setup_cache_dir feeds it the
// parsed dump, dvp_setup_scalars (the in-memory flavour srs.py uses for circuits that exist only in python) a caller's CSR.
namespace {
struct SetupCircuit {
  uint32_t n_coeffs = 0, n_rows = 0, n_wires = 0;
  const uint64_t* coeffs = nullptr;       // n_coeffs x 4, canonical
  const uint32_t* rp[3] = {nullptr, nullptr, nullptr};  // L, R, O: CSR over n_rows rows
  const uint32_t* wi[3] = {nullptr, nullptr, nullptr};
  const uint32_t* ci[3] = {nullptr, nullptr, nullptr};
};
struct SetupState {
  dvp_ecfft* tree = nullptr;
  uint32_t log_m = 0;
  size_t m = 0, n_sc = 0;
  DevBuf d, d2, bar, z2inv, bard, z2dinv, t1, l_tau, l_taud, sc;
  ~SetupState() {
    // the discrete logs of the SRS and the Lagrange values at tau are trapdoor material (src/srs.rs:41-50): zeroed before their memory
    // goes back to the allocator, on every exit
    for (DevBuf* b : {&sc, &l_tau, &l_taud, &t1})
      if (b->p) (void)hipMemset(b->p, 0, b->bytes);
    (void)hipDeviceSynchronize();
    if (tree) dvp_ecfft_destroy(tree);
  }
  Fr* s_gm() { return sc.as<Fr>(); }
};
int setup_scalars_core(const SetupCircuit& cc, const uint64_t tau[4], const Fr& tau_c, const Fr& delta_c, const Fr& eps_c, uint32_t n_public, SetupState& S,
                       hipStream_t st) {
  const uint32_t n_coeffs = cc.n_coeffs, n_rows = cc.n_rows, n_wires = cc.n_wires;
  const size_t m = S.m;
  DevBuf d_coeffs, tptr[3], trow[3], tcid[3];
  DVP_TRY(up(d_coeffs, cc.coeffs, (size_t)n_coeffs * 32));
  for (int k = 0; k < 3; ++k) {
    std::vector<uint32_t> a, b, c;
    transpose(n_rows, n_wires, cc.rp[k], cc.wi[k], cc.ci[k], a, b, c);
    DVP_TRY(up(tptr[k], a.data(), a.size() * 4));
    DVP_TRY(up(trow[k], b.data(), b.size() * 4));
    DVP_TRY(up(tcid[k], c.data(), c.size() * 4));
  }
  // ---- the domains and their tables (TREE_2N, src/srs.rs:216-346) ----
  DVP_TRY(dvp_ecfft_create(S.log_m + 1, 0, 0, &S.tree));
  dvp_ecfft* tree = S.tree;
  DevBuf &d = S.d, &d2 = S.d2, &bar = S.bar, &z2inv = S.z2inv, &bard = S.bard, &z2dinv = S.z2dinv, &t1 = S.t1, &l_tau = S.l_tau, &l_taud = S.l_taud, &sc = S.sc;
  for (DevBuf* b : {&d, &d2, &bar, &z2inv, &bard, &z2dinv, &t1, &l_tau, &l_taud}) DVP_TRY(b->alloc(m * sizeof(Fr)));
  S.n_sc = (size_t)n_wires + 5 * m;  // g_m | g_q | g_k_0 | g_k_1 | g_k_2
  DVP_TRY(sc.alloc(S.n_sc * sizeof(Fr)));
  Fr* s_gm = sc.as<Fr>();
  Fr* s_gq = s_gm + n_wires;
  Fr* s_k0 = s_gq + m;
  Fr* s_k1 = s_k0 + m;
  Fr* s_k2 = s_k1 + m;
  const dim3 gm(cdiv(m, TPB)), bt(TPB);
  hipLaunchKernelGGL(ks_domains, gm, bt, 0, st, tree->layer(0), (uint32_t)m, d.as<Fr>(), d2.as<Fr>());
  DVP_TRY(ecfft_domain_tables_dev(tree, 0, bar.as<Fr>(), z2inv.as<Fr>(), st));    // 1/Z_D'(D_i),  1/Z_D(D'_i)
  DVP_TRY(ecfft_domain_tables_dev(tree, 1, bard.as<Fr>(), z2dinv.as<Fr>(), st));  // 1/Z_D''(D'_i), 1/Z_D'(D_i)
  uint64_t zt[4], zdt[4];
  DVP_TRY(dvp_ecfft_vanish_at(tree, 0, tau, zt));
  DVP_TRY(dvp_ecfft_vanish_at(tree, 1, tau, zdt));
  const Fr z_tau = load_fr(zt), zd_tau = load_fr(zdt);
  if (fr_is_zero(z_tau) || fr_is_zero(zd_tau)) return DVP_EINVAL;  // tau lies in a domain
  const Fr z_tau_m = fr_to_mont(z_tau), zd_tau_m = fr_to_mont(zd_tau), delta_m = fr_to_mont(delta_c), eps_m = fr_to_mont(eps_c);
  const Fr delta2_m = fr_to_mont(fr_mul(delta_m, delta_c));  // fr_mul(Montgomery, canonical) = canonical product
  // L_i(tau), L'_i(tau)
  hipLaunchKernelGGL(ks_scalar_sub, gm, bt, 0, st, tau_c, d.as<Fr>(), t1.as<Fr>(), m);
  DVP_TRY(batch_inverse_dev(t1.as<Fr>(), m, st));
  hipLaunchKernelGGL(ks_mul_scale, gm, bt, 0, st, t1.as<Fr>(), bar.as<Fr>(), z_tau_m, l_tau.as<Fr>(), m);
  hipLaunchKernelGGL(ks_scalar_sub, gm, bt, 0, st, tau_c, d2.as<Fr>(), t1.as<Fr>(), m);
  DVP_TRY(batch_inverse_dev(t1.as<Fr>(), m, st));
  hipLaunchKernelGGL(ks_mul_scale, gm, bt, 0, st, t1.as<Fr>(), bard.as<Fr>(), zd_tau_m, l_taud.as<Fr>(), m);
  // g_k_0 = L(tau), g_k_1 = delta L(tau), g_k_2 = delta^2 x the unified-domain values, interleaved [D_i, D'_i] (src/ec_fft.rs:445-448)
  DVP_HIP(hipMemcpyAsync(s_k0, l_tau.p, m * sizeof(Fr), hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(ks_scale, gm, bt, 0, st, l_tau.as<Fr>(), delta_m, s_k1, m);
  const Fr zd_d2_m = fr_to_mont(fr_mul(delta2_m, zd_tau)), z_d2_m = fr_to_mont(fr_mul(delta2_m, z_tau));
  hipLaunchKernelGGL(ks_mul_scale_strided, gm, bt, 0, st, l_tau.as<Fr>(), z2dinv.as<Fr>(), zd_d2_m, s_k2, m, 2u, 0u);
  hipLaunchKernelGGL(ks_mul_scale_strided, gm, bt, 0, st, l_taud.as<Fr>(), z2inv.as<Fr>(), z_d2_m, s_k2, m, 2u, 1u);
  // g_q = eps Z_D(tau) delta^2 L'(tau)
  const Fr gq_m = fr_to_mont(fr_mul(eps_m, fr_mul(delta2_m, z_tau)));
  hipLaunchKernelGGL(ks_scale, gm, bt, 0, st, l_taud.as<Fr>(), gq_m, s_gq, m);
  // g_m = eps m(tau, delta): gather per wire over the transposed matrices
  {
    DevBuf coeffs_m;
    DVP_TRY(coeffs_m.alloc((size_t)(n_coeffs ? n_coeffs : 1) * sizeof(Fr)));
    if (n_coeffs) hipLaunchKernelGGL(ks_to_mont, dim3(cdiv(n_coeffs, TPB)), bt, 0, st, d_coeffs.as<Fr>(), coeffs_m.as<Fr>(), (size_t)n_coeffs);
    CsrT A{tptr[0].as<uint32_t>(), trow[0].as<uint32_t>(), tcid[0].as<uint32_t>()};
    CsrT B{tptr[1].as<uint32_t>(), trow[1].as<uint32_t>(), tcid[1].as<uint32_t>()};
    CsrT Cm{tptr[2].as<uint32_t>(), trow[2].as<uint32_t>(), tcid[2].as<uint32_t>()};
    hipLaunchKernelGGL(ks_m_values, dim3(cdiv(n_wires, TPB)), bt, 0, st, A, B, Cm, coeffs_m.as<Fr>(), l_tau.as<Fr>(), delta_m, delta2_m, eps_m,
                       n_wires, s_gm);
    DVP_HIP(hipGetLastError());
    DVP_HIP(hipStreamSynchronize(st));
  }
  // the Vandermonde fold C' = C - D: wire 1 + j of EVERY row carries -d_i^j (src/gnark_r1cs.rs:333-386)
  if (n_public) {
    DevBuf pw, part;
    const uint32_t nb = std::min<uint32_t>(cdiv(m, TPB), 1024u);
    DVP_TRY(pw.alloc(m * sizeof(Fr)));
    DVP_TRY(part.alloc((size_t)nb * sizeof(Fr)));
    hipLaunchKernelGGL(ks_fill_one, gm, bt, 0, st, pw.as<Fr>(), m);
    const Fr ed2_m = fr_to_mont(fr_mul(eps_m, fr_mul(delta2_m, fr_one_canon())));  // eps delta^2
    // per public input: block partials of <d^j, L(tau)>, then ONE block folds them and subtracts eps delta^2 x the sum from the wire's
    // scalar in place -- no copy to the host and no wait inside the loop (the first version took two host round trips per input)
    for (uint32_t j = 0; j < n_public; ++j) {
      hipLaunchKernelGGL(ks_dot_partial, dim3(nb), bt, 0, st, pw.as<Fr>(), l_tau.as<Fr>(), m, part.as<Fr>());
      hipLaunchKernelGGL(ks_fold_apply, dim3(1), bt, 0, st, part.as<Fr>(), nb, ed2_m, s_gm + 1 + j);
      if (j + 1 < n_public) hipLaunchKernelGGL(ks_mul, gm, bt, 0, st, pw.as<Fr>(), d.as<Fr>(), pw.as<Fr>(), m);
    }
    DVP_HIP(hipGetLastError());
    DVP_HIP(hipStreamSynchronize(st));  // pw / part go out of scope
  }
  DVP_HIP(hipGetLastError());
  return DVP_OK;
}
int setup_check_trapdoor(const uint64_t tau[4], const uint64_t delta[4], const uint64_t epsilon[4], Fr& tau_c, Fr& delta_c, Fr& eps_c) {
  if (!tau || !delta || !epsilon) return DVP_EINVAL;
  tau_c = load_fr(tau); delta_c = load_fr(delta); eps_c = load_fr(epsilon);
  if (!fr_is_canonical(tau_c) || !fr_is_canonical(delta_c) || !fr_is_canonical(eps_c)) return DVP_EINVAL;
  if (fr_is_zero(tau_c) || fr_is_zero(delta_c) || fr_is_zero(eps_c)) return DVP_EINVAL;  // src/srs.rs:199-201
  return DVP_OK;
}
}  // namespace

// TEST / HARNESS ONLY (include/dvpari_internal.h): the SRS scalars (discrete logs: trapdoor material) of an IN-MEMORY circuit -- CSR
// matrices L, R, O over n_rows <= 2^log2_m rows, coefficient table, n_wires -- in file order g_m | g_q | g_k_0 | g_k_1 | g_k_2
// ((n_wires + 5 m) x 4 u64): the same device pipeline dvp_setup_cache_dir runs on a parsed dump (srs.py: srs_scalars)
extern "C" int dvp_setup_scalars(const uint64_t tau[4], const uint64_t delta[4], const uint64_t epsilon[4], uint32_t log2_m, uint32_t n_public,
                                 uint32_t n_rows, uint32_t n_wires, const uint64_t* coeffs, uint32_t n_coeffs, const uint32_t* const row_ptr[3],
                                 const uint32_t* const wire_ids[3], const uint32_t* const coeff_ids[3], uint64_t* out_scalars, size_t out_cap) {
  Fr tau_c, delta_c, eps_c;
  DVP_TRY(setup_check_trapdoor(tau, delta, epsilon, tau_c, delta_c, eps_c));
  if (!row_ptr || !wire_ids || !coeff_ids || !out_scalars || (n_coeffs && !coeffs)) return DVP_EINVAL;
  if (log2_m < 1 || log2_m > DVP_MAX_LOG2_CONSTRAINTS || n_rows == 0 || n_rows > (1ull << log2_m) || n_wires < 1 + (uint64_t)n_public) return DVP_EINVAL;
  SetupCircuit cc;
  cc.n_coeffs = n_coeffs; cc.n_rows = n_rows; cc.n_wires = n_wires; cc.coeffs = coeffs;
  for (int k = 0; k < 3; ++k) {
    if (!row_ptr[k] || row_ptr[k][0] != 0) return DVP_EINVAL;
    for (uint32_t i = 0; i < n_rows; ++i)
      if (row_ptr[k][i + 1] < row_ptr[k][i]) { g_last_error_index = (int64_t)i; return DVP_EINVAL; }
    const size_t nnz = row_ptr[k][n_rows];
    if (nnz && (!wire_ids[k] || !coeff_ids[k])) return DVP_EINVAL;
    for (size_t e = 0; e < nnz; ++e)
      if (wire_ids[k][e] >= n_wires || coeff_ids[k][e] >= n_coeffs) { g_last_error_index = (int64_t)e; return DVP_EINVAL; }
    cc.rp[k] = row_ptr[k]; cc.wi[k] = wire_ids[k]; cc.ci[k] = coeff_ids[k];
  }
  SetupState S;
  S.log_m = log2_m;
  S.m = (size_t)1 << log2_m;
  if (out_cap < (size_t)n_wires + 5 * S.m) return DVP_EINVAL;
  hipStream_t st = nullptr;
  DVP_TRY(setup_scalars_core(cc, tau, tau_c, delta_c, eps_c, n_public, S, st));
  DVP_HIP(hipMemcpyAsync(out_scalars, S.sc.p, S.n_sc * sizeof(Fr), hipMemcpyDeviceToHost, st));
  DVP_HIP(hipStreamSynchronize(st));
  return DVP_OK;
}

// out_scalars (optional, host, (n_wires + 5 m) x 4 u64): the discrete logs of the bases written, in file order g_m | g_q |
// g_k_0 | g_k_1 | g_k_2 -- what a test pins the proof's commitments with.  *out_n_wires / *out_log2_m (optional) report the sizes.
extern "C" int dvp_setup_cache_dir_ex(const uint64_t tau[4], const uint64_t delta[4], const uint64_t epsilon[4], const char* cache_dir,
                                      uint32_t n_public, int write_precomputes, uint64_t* out_scalars, size_t out_cap, uint32_t* out_n_wires,
                                      uint32_t* out_log2_m) {
  if (!cache_dir) return DVP_EINVAL;
  Fr tau_c, delta_c, eps_c;
  DVP_TRY(setup_check_trapdoor(tau, delta, epsilon, tau_c, delta_c, eps_c));
  const std::string dir(cache_dir);
  auto path = [&](const char* name) { return dir + "/" + name; };
  // ---- the circuit (src/gnark_r1cs.rs:121-185) ----
  File dump;
  DVP_TRY(dump.open_ro(path("r1cs_to_dvsnark").c_str()));  // R1CS_CONSTRAINTS_FILE, src/artifacts.rs:76
  uint32_t n_coeffs = 0, n_rows = 0, n_wires = 0;
  uint64_t nnz[3] = {0, 0, 0};
  DVP_TRY(dvp_r1cs_dump_sizes(dump.p, dump.len, &n_coeffs, &n_rows, nnz, &n_wires));
  if (n_rows == 0 || n_wires < 1 + (uint64_t)n_public) return DVP_EINVAL;
  for (int k = 0; k < 3; ++k)
    if (nnz[k] > 0xffffffffull) return DVP_EINVAL;
  uint32_t log_m = 0;
  while ((1ull << log_m) < n_rows) ++log_m;  // next_power_of_two, src/gnark_r1cs.rs:291
  if (log_m < 1) log_m = 1;
  if (log_m > DVP_MAX_LOG2_CONSTRAINTS) return DVP_EINVAL;
  const size_t m = (size_t)1 << log_m;
  if (out_n_wires) *out_n_wires = n_wires;
  if (out_log2_m) *out_log2_m = log_m;
  if (out_scalars && out_cap < (size_t)n_wires + 5 * m) return DVP_EINVAL;
  std::vector<uint64_t> coeffs((size_t)4 * n_coeffs);
  std::vector<uint32_t> rp[3], wi[3], ci[3];
  uint32_t *rpp[3], *wip[3], *cip[3];
  for (int k = 0; k < 3; ++k) {
    rp[k].resize((size_t)n_rows + 1);
    wi[k].resize(nnz[k] ? nnz[k] : 1);
    ci[k].resize(nnz[k] ? nnz[k] : 1);
    rpp[k] = rp[k].data(); wip[k] = wi[k].data(); cip[k] = ci[k].data();
  }
  DVP_TRY(dvp_r1cs_dump_fill(dump.p, dump.len, coeffs.data(), rpp, wip, cip));
  SetupCircuit cc;
  cc.n_coeffs = n_coeffs; cc.n_rows = n_rows; cc.n_wires = n_wires; cc.coeffs = coeffs.data();
  for (int k = 0; k < 3; ++k) { cc.rp[k] = rpp[k]; cc.wi[k] = wip[k]; cc.ci[k] = cip[k]; }
  SetupState S;
  S.log_m = log_m;
  S.m = m;
  hipStream_t st = nullptr;
  DVP_TRY(setup_scalars_core(cc, tau, tau_c, delta_c, eps_c, n_public, S, st));
  dvp_ecfft* tree = S.tree;
  DevBuf &bar = S.bar, &z2inv = S.z2inv, &bard = S.bard, &z2dinv = S.z2dinv, &t1 = S.t1, &sc = S.sc;
  const size_t n_sc = S.n_sc;
  Fr* s_gm = sc.as<Fr>();
  Fr* s_gq = s_gm + n_wires;
  Fr* s_k0 = s_gq + m;
  Fr* s_k1 = s_k0 + m;
  Fr* s_k2 = s_k1 + m;
  const dim3 gm(cdiv(m, TPB)), bt(TPB);
  if (out_scalars) {
    DVP_HIP(hipMemcpyAsync(out_scalars, sc.p, n_sc * sizeof(Fr), hipMemcpyDeviceToHost, st));
    DVP_HIP(hipStreamSynchronize(st));
  }
  // ---- compute_srs_matrices: one batched fixed-base multiplication + encoding per vector, then the files (src/srs.rs:126-167) ----
  {
    static const char* const names[5] = {"g_m", "g_q", "g_k_0", "g_k_1", "g_k_2"};  // src/artifacts.rs:18-27
    const Fr* vec[5] = {s_gm, s_gq, s_k0, s_k1, s_k2};
    const size_t cnt[5] = {n_wires, m, m, m, 2 * m};
    DevBuf pts, inf, enc;
    DVP_TRY(pts.alloc(2 * m > n_wires ? 2 * m * sizeof(Aff) : (size_t)n_wires * sizeof(Aff)));
    DVP_TRY(inf.alloc(2 * m > n_wires ? 2 * m : (size_t)n_wires));
    DVP_TRY(enc.alloc((2 * m > n_wires ? 2 * m : (size_t)n_wires) * 30));
    std::vector<uint8_t> host;
    for (int i = 0; i < 5; ++i) {
      DVP_TRY(mulgen_dev(vec[i], cnt[i], pts.as<Aff>(), inf.as<uint8_t>(), st));
      DVP_TRY(encode_dev(pts.as<Aff>(), inf.as<uint8_t>(), cnt[i], enc.as<uint8_t>(), st));
      host.resize(cnt[i] * 30);
      DVP_HIP(hipMemcpyAsync(host.data(), enc.p, cnt[i] * 30, hipMemcpyDeviceToHost, st));
      DVP_HIP(hipStreamSynchronize(st));
      DVP_TRY(dvp_file_point_vec_write(path(names[i]).c_str(), host.data(), cnt[i]));
    }
  }
  // ---- the domain files (src/srs.rs:216-346; the reference quotes "2 hrs+" for z_poly at 2^23, src/artifacts.rs:92) ----
  if (write_precomputes) {
    static const char* const zp[2] = {"z_poly", "z_polyd"};
    static const char* const bw[2] = {"bar_wts", "bar_wtsd"};
    static const char* const zi[2] = {"z_vals2inv", "z_vals2dinv"};
    DevBuf ev, co, flag;
    DVP_TRY(ev.alloc(2 * m * sizeof(Fr)));
    DVP_TRY(co.alloc(2 * m * sizeof(Fr)));
    DVP_TRY(flag.alloc(4));
    std::vector<uint64_t> host(4 * (m + 1));
    for (int which = 0; which < 2; ++which) {
      Fr* b = which ? bard.as<Fr>() : bar.as<Fr>();
      Fr* z = which ? z2dinv.as<Fr>() : z2inv.as<Fr>();
      DVP_HIP(hipMemcpyAsync(host.data(), b, m * sizeof(Fr), hipMemcpyDeviceToHost, st));
      DVP_HIP(hipStreamSynchronize(st));
      DVP_TRY(dvp_file_fr_vec_write(path(bw[which]).c_str(), host.data(), m));
      DVP_HIP(hipMemcpyAsync(host.data(), z, m * sizeof(Fr), hipMemcpyDeviceToHost, st));
      DVP_HIP(hipStreamSynchronize(st));
      DVP_TRY(dvp_file_fr_vec_write(path(zi[which]).c_str(), host.data(), m));
      // compute_vanishing_polynomial (src/ec_fft.rs:241-282): Z_S on the 2m leaves (0 on S, 1 / zinv on the other half) -> exit
      DVP_HIP(hipMemcpyAsync(t1.p, z, m * sizeof(Fr), hipMemcpyDeviceToDevice, st));
      DVP_TRY(batch_inverse_dev(t1.as<Fr>(), m, st));
      hipLaunchKernelGGL(ks_vanish_evals, gm, bt, 0, st, t1.as<Fr>(), (uint32_t)m, which, ev.as<Fr>());
      DVP_TRY(dvp_ecfft_exit_dev(tree, ev.p, co.p, st));
      DVP_HIP(hipMemsetAsync(flag.p, 0, 4, st));
      hipLaunchKernelGGL(ks_check_monic, gm, bt, 0, st, co.as<Fr>(), (uint32_t)m, flag.as<unsigned int>());
      unsigned int bad = 0;
      DVP_HIP(hipMemcpyAsync(&bad, flag.p, 4, hipMemcpyDeviceToHost, st));
      DVP_HIP(hipMemcpyAsync(host.data(), co.p, (m + 1) * sizeof(Fr), hipMemcpyDeviceToHost, st));
      DVP_HIP(hipStreamSynchronize(st));
      if (bad) return DVP_EINVAL;  // not monic of degree m: the tree and the tables disagree
      DVP_TRY(dvp_file_fr_vec_write(path(zp[which]).c_str(), host.data(), m + 1));
    }
  }
  return DVP_OK;
}

extern "C" int dvp_setup_cache_dir(const uint64_t tau[4], const uint64_t delta[4], const uint64_t epsilon[4], const char* cache_dir,
                                   uint32_t n_public, int write_precomputes) {
  return dvp_setup_cache_dir_ex(tau, delta, epsilon, cache_dir, n_public, write_precomputes, nullptr, 0, nullptr, nullptr);
}

// ---- FFTR tree files from a regenerated tree, and prover_prepares_precomputes (src/proving.rs:225-325) ----------------------
extern "C" int dvp_debug_ecfft_layer(const dvp_ecfft* c, uint32_t d, uint64_t* out);
extern "C" int dvp_debug_ecfft_matrices(dvp_ecfft* c, int to_even, int which, uint64_t* out);
extern "C" int dvp_ecfft_enter_dev(dvp_ecfft* c, const void* d_coeffs, void* d_out, void* stream);

namespace {
// Sections 0-2 of an FFTree as the reference holds them (what read_minimal_fftree_from_file loads, src/tree_io.rs:353-433):
// BinaryTree<T> is a heap-ordered Vec -- entry 0 padding, a layer of k entries at [k, 2k).  For an N-leaf tree
//   f                   2N elements: layer d of the isogeny chain (N >> d values) at [N >> d, 2 (N >> d))
//   recombine/decompose N Mat2x2 each: layer d holds one matrix per pair at [(N >> d) / 2, N >> d); even positions are the pairs
//                       of even leaves (extend(.., Moiety::S1) decomposes with them, the mirrored direction recombines with
//                       them), odd positions the pairs of odd leaves; the last layer and entry 0 stay Mat2x2::identity.
struct TreeSections {
  std::vector<uint64_t> f, rec, dec;  // canonical limbs, 4 per element; a matrix = 4 elements row-major
};
int tree_sections_host(dvp_ecfft* t, TreeSections& s) {
  const size_t N = t->n_leaves;
  const int log_n = t->log_n;
  s.f.assign(2 * N * 4, 0);
  for (int d = 0; d <= log_n; ++d) DVP_TRY(dvp_debug_ecfft_layer(t, (uint32_t)d, s.f.data() + (N >> d) * 4));
  s.rec.assign(N * 16, 0);
  for (size_t i = 0; i < N; ++i) s.rec[i * 16] = s.rec[i * 16 + 12] = 1;  // identity
  s.dec = s.rec;
  if (log_n < 2) return DVP_OK;
  const size_t n = N / 2;
  std::vector<uint64_t> got((n - 1) * 16);
  // (to_even, which = 0 decompose / 1 recombine) -> (section, parity of the position it fills)
  static const struct { int to_even, which, parity; } plan[4] = {{0, 0, 0}, {1, 0, 1}, {1, 1, 0}, {0, 1, 1}};
  for (const auto& pl : plan) {
    DVP_TRY(dvp_debug_ecfft_matrices(t, pl.to_even, pl.which, got.data()));
    std::vector<uint64_t>& dst = pl.which ? s.rec : s.dec;
    for (int d = 0; d < log_n - 1; ++d) {
      const size_t nd = n >> d, off = n - nd;  // nd pairs in this layer, nd / 2 per parity
      for (size_t i = 0; i < nd / 2; ++i) memcpy(&dst[(nd + 2 * i + pl.parity) * 16], &got[(off + i) * 16], 128);
    }
  }
  return DVP_OK;
}
bool file_exists(const std::string& p) {
  struct stat sb;
  return stat(p.c_str(), &sb) == 0;
}
int mkdir_p(const std::string& dir) {
  std::string cur;
  for (size_t i = 0; i <= dir.size(); ++i) {
    if (i == dir.size() || dir[i] == '/') {
      if (!cur.empty() && mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST) return DVP_EIO;
    }
    if (i < dir.size()) cur.push_back(dir[i]);
  }
  return DVP_OK;
}
}  // namespace

// write_fftree_to_file of a minimal tree (sections f, recombine_matrices, decompose_matrices) for the regenerated `tree`
extern "C" int dvp_ecfft_write_tree_file(dvp_ecfft* tree, const char* path) {
  if (!tree || !path) return DVP_EINVAL;
  TreeSections s;
  DVP_TRY(tree_sections_host(tree, s));
  const uint8_t ids[3] = {0, 1, 2};
  const uint64_t* data[3] = {s.f.data(), s.rec.data(), s.dec.data()};
  const uint64_t N = tree->n_leaves;
  const uint64_t elems[3] = {2 * N, 4 * N, 4 * N};
  return dvp_fftr_write(path, 3, ids, data, elems);
}

// Compares a tree file with the regenerated `tree`: the leaves (second half of section f) must be identical; with matrices != 0
// also the inner layers of f and both matrix sections, entry for entry.  DVP_OK = identical; DVP_EINVAL = differs (*bad_section =
// 0/1/2, *bad_entry = first differing element of f / matrix of the section); DVP_EIO = unreadable or another shape.
extern "C" int dvp_ecfft_check_tree_file(dvp_ecfft* tree, const char* path, int matrices, int* bad_section, int64_t* bad_entry) {
  if (!tree || !path) return DVP_EINVAL;
  if (bad_section) *bad_section = -1;
  if (bad_entry) *bad_entry = -1;
  const size_t N = tree->n_leaves;
  TreeSections s;
  if (matrices)
    DVP_TRY(tree_sections_host(tree, s));
  else {
    s.f.assign(2 * N * 4, 0);
    DVP_TRY(dvp_debug_ecfft_layer(tree, 0, s.f.data() + N * 4));
  }
  std::vector<uint64_t> got;
  for (int sec = 0; sec < (matrices ? 3 : 1); ++sec) {
    const std::vector<uint64_t>& want = sec == 0 ? s.f : sec == 1 ? s.rec : s.dec;
    size_t cnt = 0;
    DVP_TRY(dvp_fftr_read_fr(path, 0, (uint8_t)sec, nullptr, 0, &cnt));
    if (cnt * 4 != want.size()) {
      if (bad_section) *bad_section = sec;
      return DVP_EIO;
    }
    got.resize(cnt * 4);
    DVP_TRY(dvp_fftr_read_fr(path, 0, (uint8_t)sec, got.data(), cnt, &cnt));
    const size_t lo = sec == 0 ? (matrices ? 1 : N) : 0;  // entry 0 of f is padding; leaves only without `matrices`
    for (size_t e = lo; e < cnt; ++e)
      if (memcmp(&got[e * 4], &want[e * 4], 32) != 0) {
        if (bad_section) *bad_section = sec;
        if (bad_entry) *bad_entry = (int64_t)(sec == 0 ? e : e / 4);
        return DVP_EINVAL;
      }
  }
  return DVP_OK;
}

// prover_prepares_precomputes(cache_dir, validate_precompute), src/proving.rs:225-325.
//   z_poly must exist (its length fixes m; DVP_EIO otherwise); TREE_2N is read when present, else generated (minimal) and written;
//   bar_wts / z_vals2inv are produced when missing -- here from the isogeny chain in milliseconds, where the reference builds
//   treen / treend and evaluates z_poly on them (it also leaves those two tree files behind; this entry does not).
//   validate_precompute: z_poly must not be all zero and must vanish on D (the reference's two asserts, :281-296 / :307-321);
//   beyond the reference, files that were FOUND (tree2n, bar_wts, z_vals2inv) are compared with the regenerated values.
// *report (optional): DVP_PREP_* bits of include/dvpari.h.  DVP_EINVAL = a validation failed (the bits say which).
extern "C" int dvp_prover_prepares_precomputes(const char* cache_dir, int validate_precompute, uint32_t* report) {
  if (!cache_dir) return DVP_EINVAL;
  uint32_t rep = 0;
  struct Rep { uint32_t* out; uint32_t* v; ~Rep() { if (out) *out = *v; } } rep_guard{report, &rep};
  const std::string dir(cache_dir);
  auto path = [&](const char* name) { return dir + "/" + name; };
  DVP_TRY(mkdir_p(dir));
  size_t nz = 0;
  DVP_TRY(dvp_file_fr_vec_read(path("z_poly").c_str(), nullptr, 0, &nz));
  if (nz < 3) return DVP_EINVAL;
  const size_t m = nz - 1;
  if (m & (m - 1)) return DVP_EINVAL;
  uint32_t log_m = 0;
  while (((size_t)1 << log_m) < m) ++log_m;
  if (log_m > DVP_MAX_LOG2_CONSTRAINTS) return DVP_EINVAL;
  std::vector<uint64_t> zpoly(nz * 4);
  DVP_TRY(dvp_file_fr_vec_read(path("z_poly").c_str(), zpoly.data(), nz, &nz));
  dvp_ecfft* tree = nullptr;
  DVP_TRY(dvp_ecfft_create(log_m + 1, 0, 0, &tree));
  struct TreeGuard { dvp_ecfft* t; ~TreeGuard() { if (t) dvp_ecfft_destroy(t); } } tree_guard{tree};
  // ---- TREE_2N (load_tree, :250-275) ----
  const std::string t2n = path("tree2n");
  if (!file_exists(t2n)) {
    DVP_TRY(dvp_ecfft_write_tree_file(tree, t2n.c_str()));
    rep |= DVP_PREP_WROTE_TREE2N;
  } else {
    int sec = -1;
    int64_t ent = -1;
    const int rc = dvp_ecfft_check_tree_file(tree, t2n.c_str(), validate_precompute ? 1 : 0, &sec, &ent);
    if (rc == DVP_EINVAL) {
      rep |= DVP_PREP_BAD_TREE2N;
      g_last_error_index = ent;
    }
    if (rc) return rc;
  }
  hipStream_t st = 0;
  const dim3 gm((unsigned)cdiv(m, TPB)), bt(TPB);
  // ---- bar_wts, z_vals2inv (compute_barycentric_weights / the inverted evaluations of z_poly on D', :284-304) ----
  DevBuf bar, zinv;
  DVP_TRY(bar.alloc(m * sizeof(Fr)));
  DVP_TRY(zinv.alloc(m * sizeof(Fr)));
  DVP_TRY(ecfft_domain_tables_dev(tree, 0, bar.as<Fr>(), zinv.as<Fr>(), st));
  // The regenerated tables belong to the MONIC Z_D; the reference derives both files from the z_poly it finds (1 / Z'(d_i) and
  // 1 / Z(d'_i), src/proving.rs:284-304).  For z_poly = c Z_D (which passes the reference's own checks) they are therefore the
  // monic tables times 1 / c: scaled here, so that what is written -- and what found files are compared with -- stays what a
  // reference prover would compute from this directory.  A z_poly whose leading coefficient is zero has no such c: nothing is
  // written or judged from it (DVP_PREP_BAD_Z_POLY).
  const bool monic = zpoly[m * 4] == 1 && !zpoly[m * 4 + 1] && !zpoly[m * 4 + 2] && !zpoly[m * 4 + 3];
  if (!monic) {
    Fr c;
    memcpy(c.v, &zpoly[m * 4], 32);
    if (fr_is_zero(c) || !fr_is_canonical(c)) {
      rep |= DVP_PREP_Z_POLY_NOT_MONIC | DVP_PREP_BAD_Z_POLY;
      return DVP_EINVAL;
    }
    const Fr cinv_m = fr_inv(fr_to_mont(c));  // Montgomery: ks_scale(canonical, Montgomery) -> canonical
    hipLaunchKernelGGL(ks_scale, gm, bt, 0, st, bar.as<Fr>(), cinv_m, bar.as<Fr>(), m);
    hipLaunchKernelGGL(ks_scale, gm, bt, 0, st, zinv.as<Fr>(), cinv_m, zinv.as<Fr>(), m);
    DVP_HIP(hipGetLastError());
  }
  std::vector<uint64_t> host(m * 4), found(m * 4);
  static const struct { const char* name; uint32_t wrote, bad; } files[2] = {{"bar_wts", DVP_PREP_WROTE_BAR_WTS, DVP_PREP_BAD_BAR_WTS},
                                                                            {"z_vals2inv", DVP_PREP_WROTE_Z_VALS2INV, DVP_PREP_BAD_Z_VALS2INV}};
  int verdict = DVP_OK;
  for (int k = 0; k < 2; ++k) {
    const std::string p = path(files[k].name);
    const bool have = file_exists(p);
    if (have && !validate_precompute) continue;
    DVP_HIP(hipMemcpyAsync(host.data(), k ? zinv.p : bar.p, m * sizeof(Fr), hipMemcpyDeviceToHost, st));
    DVP_HIP(hipStreamSynchronize(st));
    if (!have) {
      DVP_TRY(dvp_file_fr_vec_write(p.c_str(), host.data(), m));
      rep |= files[k].wrote;
      continue;
    }
    size_t cnt = 0;
    DVP_TRY(dvp_file_fr_vec_read(p.c_str(), nullptr, 0, &cnt));
    bool same = cnt == m;
    if (same) {
      DVP_TRY(dvp_file_fr_vec_read(p.c_str(), found.data(), m, &cnt));
      same = memcmp(found.data(), host.data(), m * 32) == 0;
    }
    if (!same) {
      rep |= files[k].bad;
      verdict = DVP_EINVAL;
    }
  }
  // ---- the vanishing polynomial (validate_precompute, :281-296 / :307-321) ----
  if (!monic) rep |= DVP_PREP_Z_POLY_NOT_MONIC;
  if (validate_precompute) {
    bool all_zero = true;
    for (uint64_t w : zpoly) all_zero = all_zero && w == 0;
    if (all_zero) {
      rep |= DVP_PREP_BAD_Z_POLY;
      return DVP_EINVAL;  // "all polynomial coefficients were zero"
    }
    DevBuf co, ev, flag;
    DVP_TRY(co.alloc(2 * m * sizeof(Fr)));
    DVP_TRY(ev.alloc(2 * m * sizeof(Fr)));
    DVP_TRY(flag.alloc(4));
    DVP_HIP(hipMemsetAsync(co.p, 0, 2 * m * sizeof(Fr), st));
    DVP_HIP(hipMemcpyAsync(co.p, zpoly.data(), nz * sizeof(Fr), hipMemcpyHostToDevice, st));
    DVP_TRY(dvp_ecfft_enter_dev(tree, co.p, ev.p, st));
    DVP_HIP(hipMemsetAsync(flag.p, 0xff, 4, st));
    hipLaunchKernelGGL(ks_first_nonzero_even, gm, bt, 0, st, ev.as<Fr>(), (uint32_t)m, flag.as<unsigned int>());
    unsigned int first_bad = 0;
    DVP_HIP(hipMemcpyAsync(&first_bad, flag.p, 4, hipMemcpyDeviceToHost, st));
    DVP_HIP(hipStreamSynchronize(st));
    if (first_bad != 0xffffffffu) {
      rep |= DVP_PREP_BAD_Z_POLY;  // "vanishing poly does not evaluate to zero at all points in domain"
      g_last_error_index = first_bad;
      return DVP_EINVAL;
    }
  }
  return verdict;
}
