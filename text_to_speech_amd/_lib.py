"""ctypes binding of libtts_hip.so (the C ABI declared in include/tts_hip.h).

There is NO fallback: if the shared library is missing or fails to load, every entry point raises `HipLibraryError`
(the reference's runtimes behave the same way -- e.g. `TensorRTRuntime` needs its engine file,
/root/reference/utils/keras/runtimes/tensorrt_runtime.py).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = 'libtts_hip.so'
LIB_PATH = os.path.join(_HERE, LIB_NAME)

MEM_HOST, MEM_DEVICE = 0, 1
ABI_VERSION = 11


class HipLibraryError(RuntimeError):
    """The HIP extension is missing, cannot be loaded, or reported an error."""


# name -> (restype, argtypes); must list exactly the functions include/tts_hip.h declares (checked by the CPU tests)
SIGNATURES = {
    'tts_hip_abi_version': (c_int, []),
    'tts_hip_create': (c_int, [c_int, POINTER(c_void_p)]),
    'tts_hip_destroy': (c_int, [c_void_p]),
    'tts_hip_last_error': (c_char_p, [c_void_p]),
    'tts_hip_set_tensor': (c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int]),
    'tts_hip_load_weights': (c_int, [c_void_p, c_char_p]),
    'tts_hip_finalize': (c_int, [c_void_p]),
    'tts_hip_check_weights_file': (c_int, [c_char_p, c_char_p, c_int]),
    'tts_hip_has_model': (c_int, [c_void_p, c_char_p]),
    'tts_hip_waveglow_infer': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_void_p, c_int]),
    'tts_hip_waveglow_infer_f16': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_void_p, c_int]),
    'tts_hip_waveglow_infer_f16x3': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_void_p, c_int]),
    'tts_hip_random_fill': (c_int, [c_void_p, c_int, c_uint64, c_uint64, c_void_p, c_int64, c_void_p]),
    'tts_hip_waveglow_infer_seeded': (c_int, [c_void_p, c_void_p, c_int, c_int, c_uint64, c_uint64, c_float, c_void_p, c_int, c_int]),
    'tts_hip_tacotron2_infer': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                        c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    'tts_hip_tacotron2_infer_f16': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                            c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    'tts_hip_set_decoder_mode': (c_int, [c_void_p, c_int]),
    'tts_hip_last_decoder_mode': (c_int, [c_void_p]),
    'tts_hip_set_waveglow_form': (c_int, [c_void_p, c_int]),
    'tts_hip_last_waveglow_form': (c_int, [c_void_p]),
    'tts_hip_waveglow_probe_acts': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_int, c_int, c_void_p, c_int]),
    'tts_hip_mel_stft': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int]),
    'tts_hip_waveglow_infer_async': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_void_p, c_int, c_void_p]),
    'tts_hip_mel_stft_async': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'tts_hip_tacotron2_encode': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, POINTER(c_void_p)]),
    'tts_hip_tacotron2_decode': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'tts_hip_tacotron2_decode_seeded': (c_int, [c_void_p, c_void_p, c_int, c_int, c_uint64, c_uint64, c_int, c_int, c_int, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'tts_hip_tacotron2_reencode': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    'tts_hip_encoded_free': (c_int, [c_void_p, c_void_p]),
    'tts_hip_kernel_timing': (c_int, [c_void_p, c_int]),
    'tts_hip_probe_mfma_f32': (c_int, [c_void_p, POINTER(c_double), POINTER(c_double)]),
    'tts_hip_kernel_time_us': (c_int, [c_void_p, c_int, POINTER(c_double), POINTER(c_int64)]),
    'tts_hip_synchronize': (c_int, [c_void_p]),
}

_lib = None


def load_library(path: str | None = None):
    """Loads (once) and returns the ctypes handle of libtts_hip.so with typed signatures."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get('TTS_HIP_LIBRARY') or LIB_PATH
    # PyTorch-ROCm bundles its own libamdhip64.so.7 and preloads it by path.  Two HIP runtimes in one process fight over
    # the device ("No HIP GPUs are available" in whichever initialises second), so make sure torch's copy is the one
    # already mapped when libtts_hip.so is opened: the dynamic linker then binds our DT_NEEDED soname to it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise HipLibraryError(
            f'{p} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'(or text_to_speech_amd/csrc/build.sh). There is no CPU fallback.')
    try:
        lib = ctypes.CDLL(p)
    except OSError as exc:
        raise HipLibraryError(f'cannot load {p}: {exc}') from exc
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise HipLibraryError(f'{p} does not export {name}') from exc
        fn.restype = res
        fn.argtypes = args
    if lib.tts_hip_abi_version() != ABI_VERSION:
        raise HipLibraryError(f'{p}: ABI version {lib.tts_hip_abi_version()} != expected {ABI_VERSION}')
    if path is None:
        _lib = lib
    return lib
