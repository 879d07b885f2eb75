"""MI355X-native Tacotron2 + WaveGlow + mel-STFT inference (HIP kernels behind a C ABI, ctypes host layer).

Importing the package never touches the GPU; creating a `HipEngine` / `HipRuntime` does and fails loudly
(`HipLibraryError`) when libtts_hip.so or the GPU is missing -- there is no CPU fallback in the product path.
"""
from .config import MelSTFTConfig, Tacotron2Config, WaveGlowConfig  # noqa: F401
from ._lib import HipLibraryError  # noqa: F401

__all__ = ['MelSTFTConfig', 'Tacotron2Config', 'WaveGlowConfig', 'HipLibraryError']
