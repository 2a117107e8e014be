"""espnet_amd: MI355X-native (gfx950) hybrid CTC/attention ASR training + decode path.

Drop-in for the hot path of kan-bayashi/espnet (v0.9.5): same nn.Module surface
(`E2E`, `ESPnetASRModel`, encoders/decoders/CTC with identical state_dict keys), arithmetic
in hand-written HIP kernels behind the C ABI declared in include/espnet_amd.h.
"""
from .ops import get_precision, set_precision  # noqa: F401

__version__ = "0.1.0"
