"""fullsubnet_amd - MI355X (gfx950) implementation of the FullSubNet enhancement path.

Host-side mirror of the reference's plugin surface (audio_zen ``Model.forward`` / ``Inferencer`` /
``stft`` / ``istft`` / cIRM helpers) over the C ABI of ``libfsn_hip.so`` (include/fsn_hip.h).
"""
from . import _lib  # noqa: F401
from .acoustics.feature import drop_band, istft, mag_phase, stft  # noqa: F401
from .acoustics.mask import (build_complex_ideal_ratio_mask, complex_mul, compress_cIRM,  # noqa: F401
                             decompress_cIRM)
from .graphs import GraphedCall  # noqa: F401
from .inferencer import Inferencer  # noqa: F401
from .model import Model  # noqa: F401
from .optim import ClipAdam  # noqa: F401
from .streaming import StreamingEnhancer  # noqa: F401

__version__ = "0.1.0"
