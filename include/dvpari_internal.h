/*
 * dvpari_internal.h -- entry points of libdvpari_hip.so that are NOT part of the drop-in boundary.
 *
 * include/dvpari.h is what a host of the reference prover binds (one entry per seam of alpenlabs/dv-pari).  The
 * symbols below exist for this repository's own tests, sweeps and measurement harness (tests/, tools/, bench.py):
 * tuning knobs, per-kernel timers, microbenchmarks and read-outs of intermediates.  They may change between builds;
 * a host program (examples/dvp_prove_cli.cpp) must build against dvpari.h alone.
 */
#ifndef DVPARI_INTERNAL_H
#define DVPARI_INTERNAL_H

#include "dvpari.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tuning knobs for tests, sweeps and A/B runs (tools/README.md lists them; the defaults are the measured optima and
 * the environment variables of the same names are read once, at first use).  dvp_tune_set returns DVP_EINVAL for an
 * unknown name; dvp_tune_reset goes back to defaults + environment; dvp_tune_get reads the current value.  Not
 * thread-safe against running calls. */
int dvp_tune_set(const char* name, long long value);
int dvp_tune_get(const char* name, long long* value);
void dvp_tune_reset(void);

/* Per-kernel HIP-event timers for the measurement harness (bench.py): off by default.  Names:
 * "msm_affine_round0" (first k_affine_round of an MSM, the dominant kernel: it gathers the bases), "msm_affine_rest"
 * (the later pair rounds), "msm_sort" (recode + counting sort), "msm_tail" (merge tree, Frobenius tail), "msm_total",
 * "extend_total", "prove_total"; and two counters that are always on (launches = count since dvp_profile_reset, total_ms = 0):
 * "host_waits_stream" (synchronisations of the proof's own stream: the GPU idles until the host has reacted) and
 * "host_waits_side" (an MSM's largest-bucket read, taken on a side stream while its first pair round runs). */
void dvp_profile_enable(int on);
void dvp_profile_reset(void);
int dvp_profile_read(const char* name, double* total_ms, uint64_t* launches);
/* "msm_affine_round0" split by launch shape: entry k = (pairs of the MSM, summed ms, launches) for every distinct MSM size
 * since the last reset; returns the number of shapes (fills at most `cap`), < 0 on error */
int dvp_profile_round0_shapes(uint64_t* pairs, double* total_ms, uint64_t* launches, int cap);

/* microbenchmark of the MSM kernels' GF(2^233) multiplier alone (products per second, whole chip, the pair rounds'
 * occupancy): the ceiling of bench.py's work model, measured in the same run */
int dvp_ubench_gf_mul(int reps, double* products_per_s);
/* the same for the ECFFT's multiplier: multiply-adds r = a b / R' + c per second of the lazy 30-bit-limb Fr multiplier (pairs of
 * independent chains, as the twisted butterflies issue them), whole chip: the ceiling of bench.py's work model for extend / enter /
 * exit (BASELINE configs #3 and #4) */
int dvp_ubench_fr_mul(int reps, double* muladds_per_s);
/* Wave-level trace of the batched-affine pair rounds (dvp::k_affine_round, tools/wave_trace.py).  d_buf = device buffer of
 * 64 + 64 * n_records bytes zeroed by the caller, NULL = off.  While set, every pair round appends one 64-byte record per
 * wave (8 u64: s_memrealtime at wave start / after pass 1 / after the shared inversion / at the end; s_memtime at start / end;
 * HW_ID | XCC_ID << 32; blockIdx | launch tag << 32 | slots per thread << 48); word 0 of the buffer counts them. */
int dvp_debug_wave_trace(void* d_buf, uint32_t n_records);
/* random 64-byte gathers per second (whole chip, two independent lines in flight per lane and step, as the first pair
 * round issues them) out of a device table of `table_bytes` bytes starting at d_table; d_table = NULL allocates a scratch
 * table of that size.  The ceiling of bench.py's gather model for dvp::k_affine_round<true>. */
int dvp_ubench_gather(const void* d_table, size_t table_bytes, int reps, double* gathers_per_s);
/* device address and size of the pre-rotated base table MSM `which` of a prover reads in its first pair round
 * (NULL / 0 before the first proof) -- what dvp_ubench_gather is pointed at */
int dvp_prover_msm_table_ptr(const dvp_prover* p, int which, const void** d_table, uint64_t* bytes);

/* parity-test access to the recode of the default fixed-base flavour alone (signed aligned windows): out_words[w * n + i] = 0 (digit 0) or 0x80000000 | 0x10000000 when the
 * digit is negative | w << 20 | |digit| (|digit| = 2^(c-1) is stored as key 0); *windows = ceil(234 / c_bits) */
int dvp_debug_recode_signed(const uint64_t* scalars, size_t n, int c_bits, uint32_t* out_words, int* windows);

/* TEST / HARNESS ONLY: the SRS scalars (discrete logs of the bases: trapdoor material) of an IN-MEMORY circuit -- CSR matrices L, R, O
 * over n_rows <= 2^log2_m rows (row_ptr[k]: n_rows + 1 entries), coefficient table (n_coeffs x 4 u64, canonical), n_wires -- in
 * file order g_m | g_q | g_k_0 | g_k_1 | g_k_2 ((n_wires + 5 m) x 4 u64, out_cap in elements).  The same device pipeline
 * dvp_setup_cache_dir runs on a parsed dump; dv-pari_amd/srs.py: srs_scalars (bench.py's synthetic circuit, the tests) calls it. */
int dvp_setup_scalars(const uint64_t tau[4], const uint64_t delta[4], const uint64_t epsilon[4], uint32_t log2_m, uint32_t n_public,
                      uint32_t n_rows, uint32_t n_wires, const uint64_t* coeffs, uint32_t n_coeffs, const uint32_t* const row_ptr[3],
                      const uint32_t* const wire_ids[3], const uint32_t* const coeff_ids[3], uint64_t* out_scalars, size_t out_cap);

/* TEST-ONLY (never bind this in a host): dvp_setup_cache_dir that also returns the discrete logs of the bases it wrote -- trapdoor
 * material, which dvp_setup_cache_dir itself zeroes on the device before it returns -- (host, (n_wires + 5 m) x 4 u64, file order
 * g_m | g_q | g_k_0 | g_k_1 | g_k_2; NULL = not wanted; out_cap in elements) and the circuit's sizes: parity tests pin the SRS
 * and the proof's commitments with them */
int dvp_setup_cache_dir_ex(const uint64_t tau[4], const uint64_t delta[4], const uint64_t epsilon[4], const char* cache_dir,
                           uint32_t n_public, int write_precomputes, uint64_t* out_scalars, size_t out_cap, uint32_t* out_n_wires,
                           uint32_t* out_log2_m);

/* the device flavour of the Fiat-Shamir transcript (what dvp_prove_dev runs between its two MSMs since round 5) on caller-supplied
 * inputs: alpha (canonical, to be compared with dvp_transcript_challenge) and -Z_D(alpha) (canonical) for this prover's domain;
 * n_public <= 35 (one BLAKE3 chunk of 29-byte inputs) */
int dvp_prover_debug_transcript_dev(dvp_prover* p, const uint8_t commit_p[30], const uint64_t* public_inputs, uint32_t n_public,
                                    uint64_t out_alpha[4], uint64_t out_neg_z_alpha[4]);

/* intermediates of the last proof, for parity tests (names: see prove.hip) */
int dvp_prover_debug_read(dvp_prover* p, const char* name, uint64_t* out, size_t n_elems);

/* the 2x2 butterfly matrices extend() runs on, for parity tests against the oracle and against FFTR tree files
 * (ecfft::FFTree::{decompose,recombine}_matrices, src/tree_io.rs:353-433): direction to_even = 0 is
 * FFTree::extend(.., Moiety::S1) (even leaves -> odd leaves), 1 the mirrored one; which = 0 decompose, 1 recombine.
 * out holds (n - 1) matrices of 4 canonical Fr (row-major m00 m01 m10 m11), n = leaves / 2, layer d (n >> (d+1) matrices
 * built from the pairs (L_d[2i+s], L_d[2i+s+n_d]) of the layer-d leaves) at matrix offset n - (n >> d). */
int dvp_debug_ecfft_matrices(dvp_ecfft* ctx, int to_even, int which, uint64_t* out);
/* layer d of the isogeny chain (FFTree::f.get_layers()[d]): leaves >> d canonical Fr, d = 0 .. log2_leaves */
int dvp_debug_ecfft_layer(const dvp_ecfft* ctx, uint32_t d, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* DVPARI_INTERNAL_H */
