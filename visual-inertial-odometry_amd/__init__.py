"""MI355X-native sliding-window VIO backend: host-side Python surface.

The product path is `csrc/libvio_hip.so` (hand-written HIP for gfx950 behind the C ABI of
include/vio_backend.h).  There is no CPU fallback: `load_hip()` raises if the library is missing.
"""
import os

from . import capi, sharded, stream, synth
from .capi import (CAM_DIM, LOSS_CAUCHY, LOSS_HUBER, LOSS_TRIVIAL, LOSS_TUKEY, MARG_OLD, MARG_SECOND_NEW,
                   NUM_FRAMES, POSE_DIM, PRIOR_DIM, WINDOW_SIZE, VioConfig, VioContext, VioError, VioLib,
                   VioPreint, VioSolveReport)

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.environ.get("VIO_HIP_LIB") or os.path.join(PKG_DIR, "csrc", "libvio_hip.so")     # (VIO_HIP_LIB: another build, for A/B measurements)

_hip = None


def load_hip():
    """Load the HIP product library.  Raises (never falls back) when it has not been built."""
    global _hip
    if _hip is None:
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64/libhsa-runtime64.  If this library were
        # loaded first it would pull in /opt/rocm's copies, and a later `import torch` would bring a second runtime
        # that finds no GPU (and RCCL would sit on the other one).  Importing torch first makes the dynamic loader
        # resolve our libamdhip64.so.7 to the copy torch already mapped.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _hip = VioLib(HIP_LIB, "vio_")
    return _hip


_hip_debug = None


def load_hip_debug():
    """The tests' build of the same sources with the diagnostic entry points (-DVIO_DEBUG_ENTRY_POINTS): csrc/diag/libvio_hip_debug.so."""
    global _hip_debug
    if _hip_debug is None:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _hip_debug = VioLib(os.environ.get("VIO_HIP_DEBUG_LIB") or os.path.join(PKG_DIR, "csrc", "diag", "libvio_hip_debug.so"), "vio_")
    return _hip_debug
