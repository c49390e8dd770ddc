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
  /* -5 is unused: no entry point of this library calls RCCL -- partial MSM results are combined by the host
   * (one process per GPU: torch.distributed all-gather + dvp_points_sum_dev) or by peer copies (dvp_set_devices) */
  DVP_EIO = -6,
  DVP_ENOMEM = -7,
  DVP_ECHALLENGE = -8 /* Fiat-Shamir challenge fell inside D u D' (assert!, src/proving.rs:548-556) */
} dvp_status;

const char* dvp_strerror(int status);
int dvp_version(void);
/* number of visible HIP devices; <0 on error.  The library never falls back to a CPU path. */
int dvp_device_count(void);
int dvp_set_device(int device_id);
/* In-library multi-GPU behind the unchanged prove signatures: after dvp_set_devices(ids, n), n > 1, every
 * dvp_prove / dvp_prove_dev / dvp_prove_cache_dir call shards the two MSMs of the proof (src/proving.rs:463,512,680) by
 * index range over the listed devices -- each device keeps its slice of the SRS bases and its fixed-base tables, one host
 * thread per device runs the partial MSM, scalars and partial points move by hipMemcpyPeer (xGMI), and the home device
 * (ids[0] = the device the prover was created on) adds the partial points.  The Fr stages (R1CS evaluation, ECFFT
 * extends, pointwise maps; < 15 % of a proof) stay on the home device.  An id may repeat (a one-GPU box can exercise
 * the path).  n <= 1 restores single-device proving.  Proof bytes are identical for every device list. */
int dvp_set_devices(const int* device_ids, int n);
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
/* the rayon pointwise maps of src/proving.rs:492-654 and the setup loops of src/srs.rs:53-84,138-160
 * as reusable vector ops (canonical in, canonical out): out = a o b ; out = s*a ; out = s - a ;
 * <a,b> ; sparse rows out[r] = sum coeffs[cid]*x[col] (eval_row, src/gnark_r1cs.rs:273-280) */
int dvp_fr_vec_mul(const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);
int dvp_fr_vec_scale(const uint64_t* a, const uint64_t s[4], size_t n, uint64_t* out);
/* out[i] = a[i] + s * b[i]: the (A + delta B + delta^2 C) combination of accumulate_m_values' three sums (src/srs.rs:73-80) */
int dvp_fr_vec_axpy(const uint64_t* a, const uint64_t s[4], const uint64_t* b, size_t n, uint64_t* out);
int dvp_fr_vec_scalar_sub(const uint64_t s[4], const uint64_t* a, size_t n, uint64_t* out);
int dvp_fr_vec_dot(const uint64_t* a, const uint64_t* b, size_t n, uint64_t out[4]);
int dvp_fr_spmv(const uint32_t* row_ptr, const uint32_t* col, const uint32_t* coeff_ids, uint32_t n_rows,
                const uint64_t* coeffs, uint32_t n_coeffs, const uint64_t* x, uint32_t n_cols, uint64_t* out);
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
/* sum of n partial points held as 80-byte records (x || y, u32 infinity flag, pad): what the ranks of a multi-process
 * prove all-gather, combined with n - 1 additions (RCCL has no reduction for curve points) */
int dvp_points_sum_dev(const void* d_records, uint32_t n, void* d_out_xy, void* d_out_inf, void* stream);
int dvp_msm_affine_dev(const void* d_scalars, const void* d_bases_xy, const void* d_bases_inf, size_t n,
                       void* d_out_xy /*8 x u64*/, void* d_out_inf /*u32*/, void* stream);
/* Fixed-base flavour of the same seam: the reference calls multi_scalar_mul with the SAME bases (the SRS
 * vectors g_m, g_q, g_k_*) in every proof, so a multiple 2^(o_w) P of every base is computed once for every window w
 * (W x the storage) and all windows then share one bucket set.  range_hint = bases a typical call covers
 * (the per-GPU shard; 0 = n).  run: sum over i in [lo, hi) of scalars[i - lo] * base[i]. */
typedef struct dvp_msm_ctx dvp_msm_ctx;
int dvp_msm_ctx_create(const uint64_t* bases_xy, const uint8_t* bases_inf, size_t n, size_t range_hint, dvp_msm_ctx** out);
void dvp_msm_ctx_destroy(dvp_msm_ctx* ctx);
/* window bits c and window count the context settled on (all W windows share one bucket set: 2^(c-1) buckets of |digit| for the
 * default signed windows) */
int dvp_msm_ctx_plan(const dvp_msm_ctx* ctx, int* c_bits, int* windows);
/* HBM held by the context's precomputed table.  Default flavour: aligned windows of signed binary digits over W ~ 12 multiples
 * 2^(o_w) P of every base (0.77 KB per base: 3.2 GB for the 4m bases of a 2^20-constraint prover); DVP_MSM_ALIGNED_SIGNED = 0
 * (environment, read once) selects aligned tau-adic windows over W + 1 Frobenius images instead.  *signed_windows (may be NULL)
 * reports which one this is (1 = the default).  Results do not depend on the flavour. */
uint64_t dvp_msm_ctx_table_bytes(const dvp_msm_ctx* ctx, int* signed_windows);
int dvp_msm_ctx_run(dvp_msm_ctx* ctx, const uint64_t* scalars, size_t lo, size_t hi, uint64_t out_xy[8], int* out_is_infinity);
int dvp_msm_ctx_run_dev(dvp_msm_ctx* ctx, const void* d_scalars, size_t lo, size_t hi, void* d_out_xy, void* d_out_inf, void* stream);
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
/* Which presentation of the encoded field element the 30 bytes of a point hold.  The reference reaches its codec through
 * xs233 (src/curve.rs:93-109), whose source is not available offline and for which the reference holds no known-answer
 * bytes, so the rule is selectable: rule = bit 0: the element is w + 1 | bit 1: big-endian bytes | bits 2..3: 0 w = sqrt(s/x),
 * 1 w^2 (= s/x = lambda), 2 sqrt(w) -- the classes every candidate of tools/pin_xsk233.py falls into; 0 (default) is the
 * candidate followed so far.  tools/pin_xsk233.py <k> <bytes of k G from xs233> names the number to set.  Process-wide; files
 * written under one rule must be read under it.  DVP_CODEC_RULE in the environment sets the initial value. */
int dvp_codec_set_rule(int rule);
int dvp_codec_get_rule(void);
/* CurvePoint::add over two vectors (src/curve.rs:84-90; complete: doubling, inverses, neutral).  Equality of two
 * CurvePoints (src/curve.rs:69-76) is equality of (x, y, infinity) on this representation. */
int dvp_points_add(const uint64_t* a_xy, const uint8_t* a_inf, const uint64_t* b_xy, const uint8_t* b_inf, size_t n, uint64_t* out_xy,
                   uint8_t* out_inf);

/* ------------------------------------------------------------------------------------------ */
/* Prover -- Proof::prove(cache_dir, public_inputs, private_inputs), src/proving.rs:426-688,      */
/* with the cache_dir artefacts handed over once and kept resident in HBM.                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct dvp_prover dvp_prover;
#define DVP_MAX_LOG2_CONSTRAINTS 24 /* largest validated / index-safe size; dvp_prover_create returns DVP_EINVAL above it */

/* m = 2^log2_m constraints (R1CSInstance::num_constraints, src/gnark_r1cs.rs:291); witness vector
 * = [1, public.., private..] of n_wires entries (src/proving.rs:355-359).  Builds TREE_2N and the
 * prover precomputes of prover_prepares_precomputes (src/proving.rs:225-325) on the device. */
int dvp_prover_create(uint32_t log2_m, uint32_t n_public, uint32_t n_wires, dvp_prover** out);
void dvp_prover_destroy(dvp_prover* p);
/* coefficient table of the R1CS dump (canonical; src/gnark_r1cs.rs:282-296) */
int dvp_prover_set_coeffs(dvp_prover* p, const uint64_t* coeffs, uint32_t n);
/* one of L/R/O (which = 0/1/2) as CSR over n_rows <= m rows of (wire_id, coeff_id) terms
 * (src/gnark_r1cs.rs:97-119).  The Vandermonde fold of update_to_include_vandermode_matrix_d
 * (:333-386) is applied inside the kernel, do not pre-apply it. */
int dvp_prover_set_matrix(dvp_prover* p, int which, uint32_t n_rows, const uint32_t* row_ptr, const uint32_t* wire_ids,
                          const uint32_t* coeff_ids);
/* SRS vectors, which = 0 g_m (n_wires), 1 g_q (m), 2 g_k_0 (m), 3 g_k_1 (m), 4 g_k_2 (2m)
 * (src/artifacts.rs:18-83, src/srs.rs:126-160): file payload (n x 30 B) or affine. */
int dvp_prover_set_srs_encoded(dvp_prover* p, int which, const uint8_t* enc, size_t n);
int dvp_prover_set_srs_affine(dvp_prover* p, int which, const uint64_t* xy, const uint8_t* inf, size_t n);
int dvp_prover_set_srs_affine_dev(dvp_prover* p, int which, const void* d_xy, const void* d_inf, size_t n);
/* proof = commit_p[30] | kzg_k[30] | a0[29] | b0[29]: the byte image of Proof::to_bits (:691-718) */
int dvp_prove(dvp_prover* p, const uint64_t* public_inputs, uint32_t n_public, const uint64_t* private_inputs,
              uint32_t n_private, uint8_t proof[118]);
/* the same with the assignment [1, public.., private..] (n_wires canonical Fr) already in HBM */
int dvp_prove_dev(dvp_prover* p, const void* d_assignment, uint8_t proof[118], void* stream);
/* prove in phases, so that the two MSMs -- the only stages that shard across GPUs -- can be split by
 * index range and their partial sums combined by the caller (all-gather + local add):
 *   begin -> msm_partial(0, lo, hi) -> challenge(commit point) -> msm_partial(1, lo, hi) -> finish */
int dvp_prove_begin(dvp_prover* p, const void* d_assignment, void* stream);
/* begin for a rank whose MSM shards need neither q2 nor r2 (they lie inside [w] and [k_a | k_b]): need_extend = 0
 * skips the extends and the quotient; need_extend != 0 is dvp_prove_begin */
int dvp_prove_begin_partial(dvp_prover* p, const void* d_assignment, int need_extend, void* stream);
/* The extends by VECTOR: the extends of a, b, c' (and i, unless it is evaluated by Horner: dvp_prover_extend_count
 * says 3 or 4) are independent (src/proving.rs:410-422).  After dvp_prove_begin_partial(need_extend = 0) a rank extends only
 * the vectors of `mask` (bit v = vector v of [a, b, c', i]), the ranks that need q2 / r2 exchange the extended vectors --
 * dvp_prover_extended_ptr(v) is the device address of vector v (m canonical Fr) to send from / receive into -- and
 * dvp_prove_quotient then forms r2 and q2 (src/proving.rs:492-508).  begin_partial(need_extend = 1) is exactly
 * begin_partial(0) + dvp_prove_extend_vectors(all) + dvp_prove_quotient. */
uint32_t dvp_prover_extend_count(const dvp_prover* p);
int dvp_prove_extend_vectors(dvp_prover* p, uint32_t mask, void* stream);
int dvp_prover_extended_ptr(dvp_prover* p, uint32_t v, void** d_ptr);
/* vector v of the current proof has been written into dvp_prover_extended_ptr(v) by the caller (received from the rank that
 * extended it).  dvp_prove_quotient returns DVP_EINVAL unless every vector was either extended here or marked since the
 * last begin: a missed exchange cannot turn into a silently wrong q2. */
int dvp_prove_mark_extended(dvp_prover* p, uint32_t v);
int dvp_prove_quotient(dvp_prover* p, void* stream);
size_t dvp_prover_msm_size(const dvp_prover* p, int which);
/* window bits / window count chosen for MSM `which` (0,0 until its fixed-base tables exist) */
int dvp_prover_msm_plan(const dvp_prover* p, int which, int* c_bits, int* windows);
/* HBM held by the fixed-base tables of MSM `which` on all devices (see dvp_msm_ctx_table_bytes) */
uint64_t dvp_prover_msm_table_bytes(const dvp_prover* p, int which, int* signed_windows);
int dvp_prover_msm_partial(dvp_prover* p, int which, size_t lo, size_t hi, void* d_out_xy, void* d_out_inf, void* stream);
int dvp_prove_challenge(dvp_prover* p, const void* d_commit_xy, const void* d_commit_inf, void* stream);
/* The same phase for provers that share one proof by INDEX (one process per GPU): the pointwise stages, the batch
 * inversions and the barycentric sums (src/proving.rs:561-654) are computed only where this rank needs them.
 * part 1: alpha from the commitment, 1/(d - alpha) on this rank's slice [d_lo, d_hi) of D (its share of the three
 *   barycentric sums) and on the part of D / D' that its K-scalar range [k_lo, k_hi) of [k_a | k_b | k_r] reads;
 *   d_record_out receives a 128-byte record (three partial sums + the rank's alpha-in-domain flag).
 * part 2: d_records = the n records of all ranks (all-gathered, any order; the slices must partition [0, m)):
 *   a0 b0 i0 r0, then the K scalars of [k_lo, k_hi) only -- the only range dvp_prover_msm_partial(1, ..) may then cover. */
int dvp_prove_challenge_partial(dvp_prover* p, const void* d_commit_xy, const void* d_commit_inf, size_t d_lo, size_t d_hi,
                                size_t k_lo, size_t k_hi, void* d_record_out, void* stream);
int dvp_prove_challenge_finish(dvp_prover* p, const void* d_records, uint32_t n_records, size_t k_lo, size_t k_hi, void* stream);
int dvp_prove_finish(dvp_prover* p, const void* d_kzg_xy, const void* d_kzg_inf, uint8_t proof[118], void* stream);
/* (D, D') = get_both_domains(tree2n), src/ec_fft.rs:179-189 */
int dvp_prover_domains(dvp_prover* p, uint64_t* d, uint64_t* d2);
/* which = 0: (1/Z_D'(D_i), 1/Z_D(D'_i)) = (bar_wts, z_vals2inv); which = 1: the D' mirrors
 * (bar_wtsd, z_vals2d_inv) -- compute_barycentric_weights / prepare_z_inv, src/srs.rs:267-311 */
int dvp_prover_domain_tables(dvp_prover* p, int which, uint64_t* bar_weights, uint64_t* zinv_other);

/* the same tables from a stand-alone 2m-leaf tree (TREE_2N): bar_weights[m], zinv_other[m], canonical */
int dvp_ecfft_domain_tables(dvp_ecfft* tree2n, int which, uint64_t* bar_weights, uint64_t* zinv_other);

/* Transcript::output, src/proving.rs:164-197, and the BLAKE3 hash it is built on */
int dvp_transcript_challenge(const uint8_t commit_p[30], const uint64_t* public_inputs, uint32_t n_public, uint64_t out[4]);
int dvp_blake3(const uint8_t* data, size_t len, uint8_t out[32]);

/* ------------------------------------------------------------------------------------------ */
/* cache_dir formats: the files the reference's setup / prover exchange.          */
/* Host-only code; the counted readers take out == NULL to query the element count.            */
/* ------------------------------------------------------------------------------------------ */
/* u64-LE n || n x 29 B canonical LE: write_fr_vec_to_file / read_fr_vec_from_file, src/io_utils.rs:42-70,122-179 */
int dvp_file_fr_vec_write(const char* path, const uint64_t* limbs, size_t n);
int dvp_file_fr_vec_read(const char* path, uint64_t* out, size_t cap, size_t* n);
/* u64-LE n || n x 30 B: write_point_vec_to_file / read_point_vec_from_file, src/io_utils.rs:83-111,187-239 */
int dvp_file_point_vec_write(const char* path, const uint8_t* enc, size_t n);
int dvp_file_point_vec_read(const char* path, uint8_t* out, size_t cap, size_t* n);
/* u32-BE n || n x 32 B big-endian, reduced mod p: load_witness_from_file, src/gnark_r1cs.rs:58-77,188-210 */
int dvp_file_witness_read(const char* path, uint64_t* out, size_t cap, size_t* n);
int dvp_file_witness_write(const char* path, const uint64_t* limbs, size_t n);
/* gnark/SP1 sparse R1CS dump (src/gnark_r1cs.rs:84-91,121-185) -> coefficient table (from_be_bytes_mod_order,
 * :282-289) + L/R/O as CSR.  n_wires = max wire id + 1 (accumulate_m_values, src/srs.rs:56-62).  Two calls: sizes,
 * then fill caller-allocated arrays (row_ptr[k]: n_rows+1, wire[k]/coeff_id[k]: nnz[k]). */
int dvp_r1cs_dump_sizes(const uint8_t* buf, size_t len, uint32_t* n_coeffs, uint32_t* n_rows, uint64_t nnz[3], uint32_t* n_wires);
int dvp_r1cs_dump_fill(const uint8_t* buf, size_t len, uint64_t* coeffs, uint32_t* const row_ptr[3], uint32_t* const wire[3],
                       uint32_t* const coeff_id[3]);
/* FFTR tree files (src/tree_io.rs:1-15,144-214,353-433): the sectioned container the reference stores its FFTrees in.
 * The prover never needs one (twiddles are regenerated from the curve constants); these entries let a host CHECK a
 * reference-built tree2n / treen against the regenerated domain and write files in the same container.  depth = number
 * of subtree links (section 12) to follow from the root node.  Blob layout assumed: ark-serialize Vec of field elements,
 * u64-LE count || count x 29-byte LE canonical (Mat2x2 = 4 elements); anything else is DVP_EIO (csrc/tree_io.cpp). */
int dvp_fftr_sections(const char* path, uint32_t depth, uint8_t ids[13], uint64_t lens[13], uint32_t* n_sections);
int dvp_fftr_read_fr(const char* path, uint32_t depth, uint8_t section_id, uint64_t* out, size_t cap, size_t* n_elems);
int dvp_fftr_write(const char* path, uint32_t n_sections, const uint8_t* ids, const uint64_t* const* data, const uint64_t* elems);

/* SRS::verifier_runs_setup(trapdoor, cache_dir, num_public_inputs, ..) (src/srs.rs:177-361 -> compute_srs_matrices,
 * src/srs.rs:112-167): reads cache_dir/r1cs_to_dvsnark, computes the five SRS scalar vectors on the device (Lagrange values
 * at tau on D, D' and the unified domain through the barycentric formula, accumulate_m_values over the transposed matrices,
 * the Vandermonde fold on the public wires), multiplies them onto the generator in batches and writes g_m, g_q, g_k_0,
 * g_k_1, g_k_2 as point-vector files (src/io_utils.rs:42-124).  write_precomputes != 0 also writes z_poly, z_polyd, bar_wts,
 * bar_wtsd, z_vals2inv, z_vals2dinv (Fr-vector files, src/artifacts.rs:86-110), so the directory is complete for a
 * reference prover / verifier.  tau, delta, epsilon: canonical, non-zero (src/srs.rs:199-201: DVP_EINVAL otherwise, and when
 * tau lies in an evaluation domain).  The is_fresh_setup / checksum bookkeeping of the reference is not part of this entry. */
int dvp_setup_cache_dir(const uint64_t tau[4], const uint64_t delta[4], const uint64_t epsilon[4], const char* cache_dir,
                        uint32_t n_public, int write_precomputes);

/* prover_prepares_precomputes(cache_dir, validate_precompute) (src/proving.rs:225-325): cache_dir/z_poly must exist (its length
 * fixes m; DVP_EIO when it does not, DVP_EINVAL when m is not a power of two); tree2n is read when present, else generated as a
 * minimal tree (the sections FFTree::extend needs, src/tree_io.rs:353-433) and written; bar_wts and z_vals2inv are produced when
 * missing (from the isogeny chain, in milliseconds -- the reference builds treen / treend and evaluates z_poly on them, and
 * leaves those two tree files behind; this entry does not).  validate_precompute != 0: z_poly must not be all zero and must
 * vanish on D (the reference's two asserts; DVP_EINVAL, dvp_last_error_index() = the first point of D where it does not) and,
 * beyond the reference, every file that was FOUND (tree2n with its matrices, bar_wts, z_vals2inv) is compared with the
 * regenerated values (DVP_EINVAL).  *report (optional) = DVP_PREP_* bits, set on every return. */
#define DVP_PREP_WROTE_TREE2N 0x1u
#define DVP_PREP_WROTE_BAR_WTS 0x2u
#define DVP_PREP_WROTE_Z_VALS2INV 0x4u
#define DVP_PREP_Z_POLY_NOT_MONIC 0x8u /* informational: c Z_D with c != 1 passes the reference's check too; bar_wts / z_vals2inv are then
                                        * written (and found files compared) as the monic tables times 1 / c, i.e. what the reference derives
                                        * from THAT z_poly; a zero leading coefficient is DVP_PREP_BAD_Z_POLY */
#define DVP_PREP_BAD_Z_POLY 0x100u
#define DVP_PREP_BAD_BAR_WTS 0x200u
#define DVP_PREP_BAD_Z_VALS2INV 0x400u
#define DVP_PREP_BAD_TREE2N 0x800u
int dvp_prover_prepares_precomputes(const char* cache_dir, int validate_precompute, uint32_t* report);
/* write_fftree_to_file (src/tree_io.rs:144-214) of the minimal tree of a regenerated `tree`: sections f, recombine_matrices,
 * decompose_matrices in the reference's BinaryTree order (layout restated in csrc/setup.hip; third-party, see dvp_fftr_* above) */
int dvp_ecfft_write_tree_file(dvp_ecfft* tree, const char* path);
/* a (reference-built) tree file against the regenerated `tree`: the leaves -- and with matrices != 0 the inner layers of f and
 * both matrix sections -- entry for entry.  DVP_OK identical; DVP_EINVAL differs (*bad_section 0/1/2, *bad_entry the first
 * differing element of f / matrix; both optional); DVP_EIO unreadable or of another size */
int dvp_ecfft_check_tree_file(dvp_ecfft* tree, const char* path, int matrices, int* bad_section, int64_t* bad_entry);

/* The loading half of Proof::prove (src/proving.rs:435-470,509-511,666-672): reads cache_dir/r1cs_to_dvsnark and the
 * SRS vectors g_m, g_q, g_k_0, g_k_1, g_k_2 (src/artifacts.rs:18-27,76), decodes the points on the GPU
 * (DVP_EDECODE = the reference's assert!(valid)) and returns a ready prover; the witness length is |g_m|. */
int dvp_prover_open_cache_dir(const char* cache_dir, uint32_t n_public, dvp_prover** out);
/* Proof::prove(cache_dir, public_inputs, private_inputs) itself: opens cache_dir on first use and keeps the prover in
 * a process-wide table keyed by (cache_dir, n_public, current HIP device); dvp_cache_dir_release(NULL) drops every entry.
 * Thread safety: concurrent calls are allowed; one prover = one set of device buffers, so concurrent callers of one entry take
 * turns on it (the default).  With DVP_CACHE_REPLICAS=2 in the environment an entry opens a SECOND prover from the same files
 * when two calls on it overlap after the first prover has completed a proof AND free device memory exceeds 1.25 x what is in
 * use (two proofs in flight on the GPU), callers taking whichever prover frees first; measured at the end of round 4 that is
 * SLOWER than taking turns when the witness comes from host memory, so it is opt-in.  A release during a
 * prove takes effect when that prove returns.  Files that change on disk after the first
 * call are not re-read: release the entry first.  Only SRS files whose 30-byte encodings follow this library's codec
 * rule are supported until that rule is pinned against xs233 (tools/pin_xsk233.py). */
int dvp_prove_cache_dir(const char* cache_dir, const uint64_t* public_inputs, uint32_t n_public, const uint64_t* private_inputs,
                        uint32_t n_private, uint8_t proof[118]);
void dvp_cache_dir_release(const char* cache_dir);
/* the cached prover of (cache_dir, n_public), opened if need be -- borrowed, for inspection only (debug reads, plans) */
int dvp_cache_dir_prover(const char* cache_dir, uint32_t n_public, dvp_prover** out);

#ifdef __cplusplus
}
#endif
#endif /* DVPARI_H */
