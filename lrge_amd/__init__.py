"""lrge_amd -- MI355X-native overlap engine for liblrge's genome-size estimation hot path.

Host-side mirror of the reference interface (liblrge/src/lib.rs:128-145): `Estimate`, `EstimateResult`,
`twoset.Builder`/`TwoSetStrategy`, `ava.Builder`/`AvaStrategy`, on top of the C ABI in
include/lrge_hip.h (liblrge_hip.so: hand-written HIP kernels for gfx950).  No CPU fallback.
"""
from .estimate import Estimate, EstimateResult, LrgeError, LOWER_QUANTILE, UPPER_QUANTILE  # noqa: F401
from . import ava, twoset  # noqa: F401
from .ava import AvaStrategy  # noqa: F401
from .twoset import TwoSetStrategy  # noqa: F401

__version__ = "0.1.0"
