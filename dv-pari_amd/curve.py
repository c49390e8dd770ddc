"""Host mirror of src/curve.rs (multi_scalar_mul, point_scalar_mul_gen, to_bytes/from_bytes)."""
import ctypes as C

import numpy as np

from ._native import lib, check, ptr

FR_MODULUS = 3450873173395281893717377931138512760570940988862252126328087024741343
