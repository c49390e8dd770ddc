/*
 * dvpari.h -- C ABI of the MI355X-native DV-Pari proving backend (libdvpari_hip.so).
 *
 * Every entry point replaces one seam of the reference prover (alpenlabs/dv-pari); the seam is
 * cited next to it as <file>:<line> in the reference tree.  The reference has no plugin/operator
 * registry: its seams are Rust functions and its only FFI today is Rust -> C into xs233
 * (src/curve.rs:13).  A Rust maintainer binds these symbols with an `extern "C"` block
 * (INTEGRATION.md shows the stub) and the existing R1CS reader / setup / verifier keep working
 * unchanged.
 *
 * Conventions
 *  - plain pointers and sizes only; the caller allocates every buffer, outputs included;
 *  - Fr values cross as 4 x uint64 little-endian limbs of the CANONICAL value (< p), never in
 *    Montgomery form (the reference converts with into_bigint() before its own FFI,
 *    src/curve.rs:162-170);
 *  - K-233 points cross either as 30-byte xsk233 encodings (CompressedCurvePoint,
 *    src/curve.rs:66-67) or as affine (x,y) of the prime-order representative on
 *    y^2+xy=x^3+1, each coordinate 4 x uint64 LE (polynomial basis, bit i = z^i), plus an
 *    infinity flag -- never as the opaque xsk233_point struct;
 *  - `_dev` variants take DEVICE pointers (hipMalloc / torch CUDA tensors) and a hipStream_t
 *    passed as void*; they enqueue work and return without synchronising;
 *  - return value: 0 = ok, < 0 = dvp_status.  Nothing aborts or throws across the boundary
 *    (the reference's prove() unwrap()/assert!()s abort the process, src/proving.rs:437,462,548).
 */
#ifndef DVPARI_H
#define DVPARI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum dvp_status {
  DVP_OK = 0,
  DVP_EINVAL = -1,  /* bad size / null pointer / non-canonical scalar */
  DVP_EDECODE = -2, /* invalid 30-byte point (assert!(valid), src/io_utils.rs:223) */
  DVP_EUNSAT = -3,  /* R1CS row unsatisfied (assert_eq!, src/proving.rs:389-395) */
  DVP_EHIP = -4,    /* HIP runtime error */
  DVP_ERCCL = -5,
  DVP_EIO = -6,
  DVP_ENOMEM = -7,
  DVP_ECHALLENGE = -8 /* Fiat-Shamir challenge fell inside D u D' (assert!, src/proving.rs:548-556) */
} dvp_status;

const char* dvp_strerror(int status);
int dvp_version(void);
/* number of visible HIP devices; <0 on error.  The library never falls back to a CPU path. */
int dvp_device_count(void);
int dvp_set_device(int device_id);
/* last failing index for DVP_EDECODE / DVP_EUNSAT / DVP_EINVAL (thread-local), or -1 */
int64_t dvp_last_error_index(void);

/* ------------------------------------------------------------------------------------------ */
/* ECFFT over Fr -- replaces ecfft::FFTree as built by build_sect_ecfft_tree                    */
/* (src/ec_fft.rs:197-239) and used through extend/enter/exit.                                  */
/* ------------------------------------------------------------------------------------------ */
typedef struct dvp_ecfft dvp_ecfft;

/* domain_len = 2^log2_leaves leaves x(C' + i*G'), G' of order domain_len; shifted!=0 adds the
 * base generator (order 2^base_log2) to the coset: build_ec_fftrees(.., shift_by_one,
 * base_log_n, ..), src/ec_fft.rs:93-170.  Twiddles are regenerated from the constants at
 * src/ec_fft.rs:205-229 (the FFTR cache of src/tree_io.rs is not read). */
int dvp_ecfft_create(uint32_t log2_leaves, int shifted, uint32_t base_log2, dvp_ecfft** out);
void dvp_ecfft_destroy(dvp_ecfft* ctx);
uint32_t dvp_ecfft_log2_leaves(const dvp_ecfft* ctx);
/* FFTree::f.leaves() (src/ec_fft.rs:180): out[leaves][4], canonical */
int dvp_ecfft_leaves(const dvp_ecfft* ctx, uint64_t* out);
/* FFTree::extend(evals, Moiety::S1), call site src/proving.rs:412-415.  evals: batch vectors of
 * leaves/2 values on the even leaves; out: same shape, values on the odd leaves. */
int dvp_ecfft_extend(dvp_ecfft* ctx, const uint64_t* evals, uint32_t batch, uint64_t* out);
int dvp_ecfft_extend_dev(dvp_ecfft* ctx, const void* d_evals, uint32_t batch, void* d_out, void* stream);
/* FFTree::enter / FFTree::exit, call sites src/ec_fft.rs:266,317,411: `leaves` coefficients <->
 * `leaves` evaluations on all leaves. */
int dvp_ecfft_enter(dvp_ecfft* ctx, const uint64_t* coeffs, uint64_t* evals_out);
int dvp_ecfft_exit(dvp_ecfft* ctx, const uint64_t* evals, uint64_t* coeffs_out);
int dvp_ecfft_enter_dev(dvp_ecfft* ctx, const void* d_coeffs, void* d_out, void* stream);
int dvp_ecfft_exit_dev(dvp_ecfft* ctx, const void* d_evals, void* d_out, void* stream);
/* Z_D(x), Z_D'(x) for the even (which=0) / odd (which=1) half of the leaves, evaluated through
 * the isogeny chain (replaces DensePolynomial::evaluate on z_poly, src/ec_fft.rs:475) */
int dvp_ecfft_vanish_at(const dvp_ecfft* ctx, int which, const uint64_t x[4], uint64_t out[4]);

/* ------------------------------------------------------------------------------------------ */
/* Fr vector kernels -- ark_ff::batch_inversion (src/proving.rs:604,614; src/ec_fft.rs:482)     */
/* and evaluate_poly_at_alpha_using_barycentric_weights (src/ec_fft.rs:455-491).                */
/* ------------------------------------------------------------------------------------------ */
int dvp_fr_batch_inverse(uint64_t* vals, size_t n); /* in place; zeros stay zero */
int dvp_fr_batch_inverse_dev(void* d_vals, size_t n, void* stream);
int dvp_barycentric_eval(const uint64_t* domain, const uint64_t* bar_weights, const uint64_t z_at_alpha[4],
                         const uint64_t* evals, size_t n, const uint64_t alpha[4], uint64_t out[4]);

/* ------------------------------------------------------------------------------------------ */
/* sect233k1 / xsk233 group                                                                     */
/* ------------------------------------------------------------------------------------------ */
/* multi_scalar_mul(&[Fr], &[CurvePoint]) -> CurvePoint, src/curve.rs:141-158 (call sites
 * src/proving.rs:463,512,680).  Affine flavour: bases_xy[n][8] = x||y of the E[r]
 * representative; bases_inf[n] (may be NULL) marks neutral elements. */
int dvp_msm_affine(const uint64_t* scalars, const uint64_t* bases_xy, const uint8_t* bases_inf, size_t n,
                   uint64_t out_xy[8], int* out_is_infinity);
int dvp_msm_affine_dev(const void* d_scalars, const void* d_bases_xy, const void* d_bases_inf, size_t n,
                       void* d_out_xy /*8 x u64*/, void* d_out_inf /*u32*/, void* stream);
/* same seam with the reference's own wire formats: scalars n x 32 B canonical LE, bases n x 30 B
 * xsk233 encodings (read_point_vec_from_file payload, src/io_utils.rs:187-239). */
int dvp_msm_xsk233(const uint8_t* scalars, const uint8_t* bases_enc, size_t n, uint8_t out_enc[30]);

/* point_scalar_mul_gen over a vector + to_bytes: the SRS commitment loop of
 * compute_srs_matrices, src/srs.rs:126-160 (one fixed-base multiplication per scalar). */
int dvp_mulgen_batch(const uint64_t* scalars, size_t n, uint8_t* out_enc /* n x 30 */);
int dvp_mulgen_batch_affine(const uint64_t* scalars, size_t n, uint64_t* out_xy /* n x 8 */, uint8_t* out_inf);

/* CurvePoint::to_bytes / from_bytes over vectors (src/curve.rs:93-109, src/io_utils.rs:217-226) */
int dvp_points_encode(const uint64_t* xy, const uint8_t* inf, size_t n, uint8_t* out_enc);
int dvp_points_decode(const uint8_t* enc, size_t n, uint64_t* out_xy, uint8_t* out_inf);

#ifdef __cplusplus
}
#endif
#endif /* DVPARI_H */
