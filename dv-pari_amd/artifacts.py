"""File names inside a cache_dir -- src/artifacts.rs:18-83."""
SRS_G_Q = "g_q"
SRS_G_M = "g_m"
SRS_G_K_0 = "g_k_0"
SRS_G_K_1 = "g_k_1"
SRS_G_K_2 = "g_k_2"
SRS_FILES = (SRS_G_M, SRS_G_Q, SRS_G_K_0, SRS_G_K_1, SRS_G_K_2)  # in dvp_prover_set_srs_* `which` order

# The FFTR tree files (src/tree_io.rs; names src/artifacts.rs:30-57).  The prover regenerates its twiddles on the device from
# the curve constants (src/ec_fft.rs:205-229) and reads none of them; prover_prepares_precomputes writes / checks TREE_2N.
TREE_2N = "tree2n"
TREE_2ND = "tree2nd"
TREE_N = "treen"
TREE_ND = "treend"
Z_POLY = "z_poly"
Z_POLYD = "z_polyd"
BAR_WTS = "bar_wts"
BAR_WTSD = "bar_wtsd"
Z_VALS2_INV = "z_vals2inv"
Z_VALS2D_INV = "z_vals2dinv"

R1CS_CONSTRAINTS_FILE = "r1cs_to_dvsnark"
R1CS_WITNESS_FILE = "witness_to_dvsnark"
