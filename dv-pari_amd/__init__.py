"""
dv-pari_amd -- host-side mirror of the DV-Pari prover hot path over the MI355X-native C ABI
(libdvpari_hip.so, declared in include/dvpari.h).

The module names follow the reference (alpenlabs/dv-pari):
    curve.multi_scalar_mul / point_scalar_mul_gen      <- src/curve.rs:129-158
    ec_fft.FFTree(.extend/.enter/.exit)                 <- ecfft crate as used by src/ec_fft.rs, src/proving.rs:410-422
There is no CPU fallback: importing the native library fails loudly when it is missing.

The directory name contains a hyphen, so import it with
    import importlib; dvp = importlib.import_module("dv-pari_amd")
"""
from . import _native  # noqa: F401  (raises if libdvpari_hip.so is missing)
from ._native import lib, DvpError, check, tune, set_devices  # noqa: F401
from . import artifacts, curve, ec_fft, fr, gnark_r1cs, io_utils, proving, srs, distributed, tree_io  # noqa: F401

P = 3450873173395281893717377931138512760570940988862252126328087024741343
