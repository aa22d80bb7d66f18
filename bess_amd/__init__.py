"""bess_amd -- MI355X-native PDAS best-subset solver behind the bess() / bess.linear API.

The package is a thin host layer over libbessx.so (HIP kernels for gfx950 + C++ path drivers,
C ABI in include/bessx.h).  It holds only what the hot path needs:
  capi    ctypes binding of the C ABI (Session, pywrap_bess, single-kernel ops)
  linear  the reference's estimator classes (PdasLm, PdasLogistic, ...; python/bess/linear.py)
  synth   synthetic inputs of the BASELINE configs
There is no CPU fallback: without the built library or without a GPU every solve raises.
"""
# (Round 4 wrote GPU_MAX_HW_QUEUES=8 into the process environment here so that the chunk chains' streams would not
# share hardware queues.  The library now creates those streams with a hardware queue of their own
# (csrc/bessx_session.cpp: ctx_stream_create) and importing the package changes nothing in the caller's process.)
from . import capi  # noqa: F401
from .linear import (PdasLm, PdasLogistic, PdasPoisson, PdasCox, L0L2Lm, L0L2Logistic, L0L2Poisson,  # noqa: F401
                     L0L2Cox, GroupPdasLm, GroupPdasLogistic, GroupPdasPoisson, GroupPdasCox)

__all__ = ["capi", "PdasLm", "PdasLogistic", "PdasPoisson", "PdasCox", "L0L2Lm", "L0L2Logistic", "L0L2Poisson",
           "L0L2Cox", "GroupPdasLm", "GroupPdasLogistic", "GroupPdasPoisson", "GroupPdasCox"]
