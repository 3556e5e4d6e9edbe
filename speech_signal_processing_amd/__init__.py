"""speech_signal_processing_amd — MI355X (gfx950) native MFCC -> GMM-UBM / d-vector scoring.

Python host code keeps the call surface of kleinzcy/speech_signal_processing
(``utils.processing``, ``MFCC_DTW``, ``GMM_UBM``, ``d_vector``) and dispatches through
a ctypes C-ABI (include/ssp.h, libsspgpu.so) into hand-written HIP kernels.
There is no CPU fallback: without the built library and a gfx950 device every
compute entry point raises.
"""
from . import frontend  # noqa: F401  (host-side tables; numpy only)
from .frontend import MfccConfig, preset_inrepo, preset_sidekit, preset_sidekit_plp, preset_librosa  # noqa: F401

__all__ = ["frontend", "MfccConfig", "preset_inrepo", "preset_sidekit", "preset_sidekit_plp", "preset_librosa"]
