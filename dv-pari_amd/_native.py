"""ctypes binding of include/dvpari.h (the drop-in boundary) and include/dvpari_internal.h (test / measurement entries).  No torch types cross this boundary: pointers and sizes only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DVP_LIB") or os.path.join(_HERE, "libdvpari_hip.so")  # DVP_LIB: A/B-test another build

# PyTorch-ROCm wheels bundle their own libamdhip64; if this library pulls in /opt/rocm's copy first, torch
# later finds "No HIP GPUs".  Loading torch first makes both share one HIP runtime (torch is plumbing here:
# device memory, streams, torch.distributed -- never the compute path).
if not os.environ.get("DVP_NO_TORCH_PRELOAD"):
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional for the ctypes path
        pass

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950).  dv-pari_amd has no CPU fallback."
    )
lib = C.CDLL(LIB_PATH)

vp, u32, u64p, u8p, sz = C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t

_SIGS = {
    "dvp_strerror": (C.c_char_p, [C.c_int]),
    "dvp_version": (C.c_int, []),
    "dvp_device_count": (C.c_int, []),
    "dvp_set_device": (C.c_int, [C.c_int]),
    "dvp_set_devices": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "dvp_points_sum_dev": (C.c_int, [vp, u32, vp, vp, vp]),
    "dvp_last_error_index": (C.c_int64, []),
    "dvp_tune_set": (C.c_int, [C.c_char_p, C.c_longlong]),
    "dvp_tune_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_longlong)]),
    "dvp_tune_reset": (None, []),
    "dvp_ubench_gather": (C.c_int, [vp, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "dvp_prover_msm_table_ptr": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_uint64)]),
    "dvp_debug_ecfft_matrices": (C.c_int, [vp, C.c_int, C.c_int, u64p]),
    "dvp_debug_ecfft_layer": (C.c_int, [vp, u32, u64p]),
    "dvp_ubench_gf_mul": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "dvp_ubench_fr_mul": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "dvp_setup_scalars": (C.c_int, [u64p, u64p, u64p, u32, u32, u32, u32, u64p, u32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), u64p, sz]),
    "dvp_debug_wave_trace": (C.c_int, [vp, u32]),
    "dvp_profile_enable": (None, [C.c_int]),
    "dvp_profile_reset": (None, []),
    "dvp_profile_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "dvp_profile_round0_shapes": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]),
    "dvp_ecfft_create": (C.c_int, [u32, C.c_int, u32, C.POINTER(vp)]),
    "dvp_ecfft_destroy": (None, [vp]),
    "dvp_ecfft_log2_leaves": (u32, [vp]),
    "dvp_ecfft_leaves": (C.c_int, [vp, u64p]),
    "dvp_ecfft_extend": (C.c_int, [vp, u64p, u32, u64p]),
    "dvp_ecfft_extend_dev": (C.c_int, [vp, vp, u32, vp, vp]),
    "dvp_ecfft_enter": (C.c_int, [vp, u64p, u64p]),
    "dvp_ecfft_exit": (C.c_int, [vp, u64p, u64p]),
    "dvp_ecfft_enter_dev": (C.c_int, [vp, vp, vp, vp]),
    "dvp_ecfft_exit_dev": (C.c_int, [vp, vp, vp, vp]),
    "dvp_ecfft_vanish_at": (C.c_int, [vp, C.c_int, u64p, u64p]),
    "dvp_fr_batch_inverse": (C.c_int, [u64p, sz]),
    "dvp_fr_batch_inverse_dev": (C.c_int, [vp, sz, vp]),
    "dvp_fr_vec_mul": (C.c_int, [u64p, u64p, sz, u64p]),
    "dvp_fr_vec_scale": (C.c_int, [u64p, u64p, sz, u64p]),
    "dvp_fr_vec_scalar_sub": (C.c_int, [u64p, u64p, sz, u64p]),
    "dvp_fr_vec_axpy": (C.c_int, [u64p, u64p, u64p, sz, u64p]),
    "dvp_fr_vec_dot": (C.c_int, [u64p, u64p, sz, u64p]),
    "dvp_fr_spmv": (C.c_int, [vp, vp, vp, u32, u64p, u32, u64p, u32, u64p]),
    "dvp_barycentric_eval": (C.c_int, [u64p, u64p, u64p, u64p, sz, u64p, u64p]),
    "dvp_msm_affine": (C.c_int, [u64p, u64p, u8p, sz, u64p, C.POINTER(C.c_int)]),
    "dvp_msm_affine_dev": (C.c_int, [vp, vp, vp, sz, vp, vp, vp]),
    "dvp_msm_ctx_create": (C.c_int, [u64p, u8p, sz, sz, C.POINTER(vp)]),
    "dvp_msm_ctx_destroy": (None, [vp]),
    "dvp_msm_ctx_plan": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dvp_msm_ctx_table_bytes": (C.c_uint64, [vp, C.POINTER(C.c_int)]),
    "dvp_debug_recode_signed": (C.c_int, [vp, C.c_size_t, C.c_int, vp, C.POINTER(C.c_int)]),
    "dvp_msm_ctx_run": (C.c_int, [vp, u64p, sz, sz, u64p, C.POINTER(C.c_int)]),
    "dvp_msm_ctx_run_dev": (C.c_int, [vp, vp, sz, sz, vp, vp, vp]),
    "dvp_msm_xsk233": (C.c_int, [u8p, u8p, sz, u8p]),
    "dvp_mulgen_batch": (C.c_int, [u64p, sz, u8p]),
    "dvp_mulgen_batch_affine": (C.c_int, [u64p, sz, u64p, u8p]),
    "dvp_points_encode": (C.c_int, [u64p, u8p, sz, u8p]),
    "dvp_points_decode": (C.c_int, [u8p, sz, u64p, u8p]),
    "dvp_codec_set_rule": (C.c_int, [C.c_int]),
    "dvp_codec_get_rule": (C.c_int, []),
    "dvp_points_add": (C.c_int, [u64p, u8p, u64p, u8p, sz, u64p, u8p]),
    "dvp_prover_create": (C.c_int, [u32, u32, u32, C.POINTER(vp)]),
    "dvp_prover_destroy": (None, [vp]),
    "dvp_prover_set_coeffs": (C.c_int, [vp, u64p, u32]),
    "dvp_prover_set_matrix": (C.c_int, [vp, C.c_int, u32, vp, vp, vp]),
    "dvp_prover_set_srs_encoded": (C.c_int, [vp, C.c_int, u8p, sz]),
    "dvp_prover_set_srs_affine": (C.c_int, [vp, C.c_int, u64p, u8p, sz]),
    "dvp_prover_set_srs_affine_dev": (C.c_int, [vp, C.c_int, vp, vp, sz]),
    "dvp_prove": (C.c_int, [vp, u64p, u32, u64p, u32, u8p]),
    "dvp_prove_dev": (C.c_int, [vp, vp, u8p, vp]),
    "dvp_prove_begin": (C.c_int, [vp, vp, vp]),
    "dvp_prove_begin_partial": (C.c_int, [vp, vp, C.c_int, vp]),
    "dvp_prover_extend_count": (u32, [vp]),
    "dvp_prove_extend_vectors": (C.c_int, [vp, u32, vp]),
    "dvp_prover_extended_ptr": (C.c_int, [vp, u32, C.POINTER(vp)]),
    "dvp_prove_quotient": (C.c_int, [vp, vp]),
    "dvp_prove_mark_extended": (C.c_int, [vp, u32]),
    "dvp_prover_msm_size": (C.c_size_t, [vp, C.c_int]),
    "dvp_prover_msm_plan": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dvp_prover_msm_table_bytes": (C.c_uint64, [vp, C.c_int, C.POINTER(C.c_int)]),
    "dvp_prover_msm_partial": (C.c_int, [vp, C.c_int, sz, sz, vp, vp, vp]),
    "dvp_prove_challenge": (C.c_int, [vp, vp, vp, vp]),
    "dvp_prove_challenge_partial": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, vp, vp]),
    "dvp_prove_challenge_finish": (C.c_int, [vp, vp, C.c_uint32, C.c_size_t, C.c_size_t, vp]),
    "dvp_prove_finish": (C.c_int, [vp, vp, vp, u8p, vp]),
    "dvp_prover_debug_read": (C.c_int, [vp, C.c_char_p, u64p, sz]),
    "dvp_prover_debug_transcript_dev": (C.c_int, [vp, u8p, u64p, u32, u64p, u64p]),
    "dvp_prover_domains": (C.c_int, [vp, u64p, u64p]),
    "dvp_prover_domain_tables": (C.c_int, [vp, C.c_int, u64p, u64p]),
    "dvp_ecfft_domain_tables": (C.c_int, [vp, C.c_int, u64p, u64p]),
    "dvp_file_fr_vec_write": (C.c_int, [C.c_char_p, u64p, sz]),
    "dvp_file_fr_vec_read": (C.c_int, [C.c_char_p, u64p, sz, C.POINTER(C.c_size_t)]),
    "dvp_file_point_vec_write": (C.c_int, [C.c_char_p, u8p, sz]),
    "dvp_file_point_vec_read": (C.c_int, [C.c_char_p, u8p, sz, C.POINTER(C.c_size_t)]),
    "dvp_file_witness_read": (C.c_int, [C.c_char_p, u64p, sz, C.POINTER(C.c_size_t)]),
    "dvp_file_witness_write": (C.c_int, [C.c_char_p, u64p, sz]),
    "dvp_r1cs_dump_sizes": (C.c_int, [u8p, sz, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_uint64 * 3), C.POINTER(u32)]),
    "dvp_r1cs_dump_fill": (C.c_int, [u8p, sz, u64p, C.POINTER(vp * 3), C.POINTER(vp * 3), C.POINTER(vp * 3)]),
    "dvp_fftr_sections": (C.c_int, [C.c_char_p, u32, u8p, u64p, C.POINTER(u32)]),
    "dvp_fftr_read_fr": (C.c_int, [C.c_char_p, u32, C.c_uint8, u64p, sz, C.POINTER(C.c_size_t)]),
    "dvp_fftr_write": (C.c_int, [C.c_char_p, u32, u8p, vp, u64p]),
    "dvp_setup_cache_dir": (C.c_int, [u64p, u64p, u64p, C.c_char_p, u32, C.c_int]),
    "dvp_setup_cache_dir_ex": (C.c_int, [u64p, u64p, u64p, C.c_char_p, u32, C.c_int, u64p, sz, C.POINTER(u32), C.POINTER(u32)]),
    "dvp_prover_prepares_precomputes": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(u32)]),
    "dvp_ecfft_write_tree_file": (C.c_int, [vp, C.c_char_p]),
    "dvp_ecfft_check_tree_file": (C.c_int, [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "dvp_prover_open_cache_dir": (C.c_int, [C.c_char_p, u32, C.POINTER(vp)]),
    "dvp_prove_cache_dir": (C.c_int, [C.c_char_p, u64p, u32, u64p, u32, u8p]),
    "dvp_cache_dir_release": (None, [C.c_char_p]),
    "dvp_cache_dir_prover": (C.c_int, [C.c_char_p, u32, C.POINTER(vp)]),
    "dvp_transcript_challenge": (C.c_int, [u8p, u64p, u32, u64p]),
    "dvp_blake3": (C.c_int, [u8p, sz, u8p]),
}
EXPORTED = []
for _name, (_res, _args) in _SIGS.items():
    try:
        _f = getattr(lib, _name)
    except AttributeError:
        continue
    _f.restype, _f.argtypes = _res, _args
    EXPORTED.append(_name)


class DvpError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        self.index = lib.dvp_last_error_index()
        super().__init__(f"{where}: {lib.dvp_strerror(status).decode()} (status {status}, index {self.index})")


def check(status, where=""):
    if status != 0:
        raise DvpError(status, where)


def set_devices(ids):
    """dvp_set_devices: in-library multi-GPU over the listed devices (ids[0] = the prover's home device; an id may repeat);
    an empty list or a single id restores single-device proving"""
    ids = list(ids or [])
    arr = (C.c_int * max(len(ids), 1))(*ids)
    check(lib.dvp_set_devices(arr, len(ids)), "dvp_set_devices")


class tune:
    """with tune(DVP_MSM_FIXED_C=20, DVP_MSM_AFF_MIN=64): ...  -- run-time tuning knobs (dvp_tune_set).  On exit every
    knob this block changed goes back to the value it had on entry, so blocks nest."""

    def __init__(self, **knobs):
        self.knobs = knobs
        self.saved = {}

    def __enter__(self):
        for k, v in self.knobs.items():
            old = C.c_longlong(0)
            check(lib.dvp_tune_get(k.encode(), C.byref(old)), f"dvp_tune_get({k})")
            self.saved[k] = old.value
            check(lib.dvp_tune_set(k.encode(), int(v)), f"dvp_tune_set({k})")
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            lib.dvp_tune_set(k.encode(), v)
        self.saved = {}
        return False


# ---- int <-> limb helpers (canonical little-endian 4 x u64) ----------------------------------------
def ints_to_limbs(vals, words=4) -> np.ndarray:
    out = np.empty((len(vals), words), dtype=np.uint64)
    buf = b"".join(int(v).to_bytes(8 * words, "little") for v in vals)
    out[:] = np.frombuffer(buf, dtype="<u8").reshape(len(vals), words)
    return out


def limbs_to_ints(arr) -> list:
    a = np.ascontiguousarray(arr, dtype="<u8")
    a = a.reshape(-1, a.shape[-1])
    w = a.shape[1] * 8
    raw = a.tobytes()
    return [int.from_bytes(raw[i * w:(i + 1) * w], "little") for i in range(a.shape[0])]


def ptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)
