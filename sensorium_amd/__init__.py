"""sensorium_amd — MI355X-native DwiseNeuro training/inference hot path (drop-in for lRomul/sensorium's
``src/models/dwiseneuro.py`` + ``src/losses.py`` + the ``MouseModel`` step of ``src/argus_models.py``).

Importing the package loads ``csrc/libdwiseneuro_hip.so`` (hand-written gfx950 kernels behind a C-ABI,
include/dwn.h) and fails loudly if it is missing — there is no PyTorch/CPU fallback path.
"""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library is absent)
from .dwiseneuro import DwiseNeuro  # noqa: F401
from .losses import MicePoissonLoss  # noqa: F401

__all__ = ["DwiseNeuro", "MicePoissonLoss"]
