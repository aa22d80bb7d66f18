"""bess_amd -- MI355X-native PDAS best-subset solver behind the bess() / bess.linear API.

The package is a thin host layer over libbessx.so (HIP kernels for gfx950 + C++ path drivers,
C ABI in include/bessx.h).  It holds only what the hot path needs:
  capi    ctypes binding of the C ABI (Session, pywrap_bess, single-kernel ops)
  linear  the reference's estimator classes (PdasLm, PdasLogistic, ...; python/bess/linear.py)
  synth   synthetic inputs of the BASELINE configs
There is no CPU fallback: without the built library or without a GPU every solve raises.
"""
import os as _os

# The sequential path of an LM problem runs as several chunk chains side by side, a HIP stream each
# (csrc/bessx_kchunks.cpp).  The HIP runtime gives a process' streams 4 hardware queues unless GPU_MAX_HW_QUEUES says
# otherwise -- read once, when the runtime starts: ask for 8 before anything touches the GPU (no effect, and no harm,
# if the runtime is up already or the caller has chosen a value).  configs[1]: 12.7 ms per path against 16.2 ms.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import capi  # noqa: F,E402401
from .linear import (PdasLm, PdasLogistic, PdasPoisson, PdasCox, L0L2Lm, L0L2Logistic, L0L2Poisson,  # noqa: F401
                     L0L2Cox, GroupPdasLm, GroupPdasLogistic, GroupPdasPoisson, GroupPdasCox)

__all__ = ["capi", "PdasLm", "PdasLogistic", "PdasPoisson", "PdasCox", "L0L2Lm", "L0L2Logistic", "L0L2Poisson",
           "L0L2Cox", "GroupPdasLm", "GroupPdasLogistic", "GroupPdasPoisson", "GroupPdasCox"]
